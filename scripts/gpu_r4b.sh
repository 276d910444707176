#!/bin/bash
# Round 4, call B: stride-2 work (merged sibling convs + conv_h2s): parity, then same-box A/B bench and the per-op table.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short --timeout 900 -x \
  -k "conv_layer or saturation or net_golden or net_vs_oracle or split_k or (benchmark_batch and 32-f16x2)" \
  > gpurun_out/r4b_tests.log 2>&1
echo "== tests exit $? :: $(tail -n 1 gpurun_out/r4b_tests.log)"
grep -E "FAILED|Error|error:|assert|default build" gpurun_out/r4b_tests.log | head -30
show() {
tail -n 1 $1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    print('value', d['value'], 'ms/step', d['ms_per_step'], 'roof', d['roofline']['kernel'], d['roofline']['achieved'], d['roofline']['frac'], 'lat', d.get('single_image_latency', {}).get('ms_per_frame'), 'maps', d['config'].get('maps_max_abs_vs_oracle'), d['config'].get('detections_equal'))
    tot = 0
    for k, v in sorted(d['kernel_classes'].items(), key=lambda kv: -kv[1]['ms']):
        tot += v['ms']
        if 's2' in k or 'fusesum' in k or 'h2s' in k: print('  %-40s n=%3d ms=%8.3f tflops=%7.2f gbs=%7.1f' % (k, v['launches'], v['ms'], v['tflops'], v['gbs']))
    print('  serial kernel sum', round(tot, 3))
except Exception as e:
    print('parse failed', e)
"
}
rm -f gpurun_out/tune_r4b.json
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-companion --tune-file gpurun_out/tune_r4b.json > gpurun_out/r4b_bench_new.log 2>&1
echo "== bench (merged s2 + h2s) exit $?"; show gpurun_out/r4b_bench_new.log
ROMP_MERGE_S2=0 ROMP_CONV_NO_H2S=1 timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-companion --no-latency --no-end-to-end --tune-file none > gpurun_out/r4b_bench_old.log 2>&1
echo "== bench (round-3 s2) exit $?"; show gpurun_out/r4b_bench_old.log
ROMP_CONV_NO_H2S=1 timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-companion --no-latency --no-end-to-end --tune-file none > gpurun_out/r4b_bench_merge_only.log 2>&1
echo "== bench (merged, old kernels) exit $?"; show gpurun_out/r4b_bench_merge_only.log
timeout 600 python scripts/op_table.py 32 f16x2 > gpurun_out/r4b_optable.log 2>&1
sed -n '/total serial/,$p' gpurun_out/r4b_optable.log | head -60
