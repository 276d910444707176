#!/bin/bash
# Round 4, call L: fuseup (fuse-layer output with its 1x1 up-convs inside): unit + network parity, same-box A/B bench.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short --timeout 900 -x -k "fuseup or fusesum or net_golden or net_vs_oracle or split_k or saturation or plan_file" > gpurun_out/r4l_tests.log 2>&1
echo "== tests exit $? :: $(tail -n 1 gpurun_out/r4l_tests.log)"; grep -E "FAILED|Error|assert|fuseup CO" gpurun_out/r4l_tests.log | head -20
show() {
tail -n 1 $1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('value', d['value'], 'ms/step', d['ms_per_step'], 'roof', d['roofline']['kernel'], d['roofline']['frac'], 'lat', d.get('single_image_latency', {}).get('ms_per_frame'), 'maps', d['config'].get('maps_max_abs_vs_oracle'), d['config'].get('detections_equal'))
tot = 0
for k, v in sorted(d['kernel_classes'].items(), key=lambda kv: -kv[1]['ms']):
    tot += v['ms']
    if 'fuse' in k or 'k1s1' in k: print('  %-40s n=%3d ms=%8.3f tflops=%7.2f gbs=%7.1f' % (k, v['launches'], v['ms'], v['tflops'], v['gbs']))
print('  serial kernel sum', round(tot, 3))
"
}
rm -f gpurun_out/tune_r4l.json
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-companion --tune-file gpurun_out/tune_r4l.json > gpurun_out/r4l_bench_new.log 2>&1
echo "== bench fuseup: exit $?"; show gpurun_out/r4l_bench_new.log
ROMP_FUSEUP=0 timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-companion --tune-file none > gpurun_out/r4l_bench_old.log 2>&1
echo "== bench ROMP_FUSEUP=0: exit $?"; show gpurun_out/r4l_bench_old.log
