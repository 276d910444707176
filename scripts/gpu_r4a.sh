#!/bin/bash
# Round 4, call A: the parity soft spots + saturation counter + dispatch-flag fix, then a bench line (no CPU leg).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short --timeout 600 \
  -k "stem_mfma or conv_math_all or saturation or committed_table or fused_basic_block or seam1x1 or range or fusesum or net_golden or net_vs_oracle or split_k or plan_file" \
  > gpurun_out/r4a_tests.log 2>&1
echo "== tests exit $? :: $(tail -n 1 gpurun_out/r4a_tests.log)"
grep -E "FAILED|Error|error:|assert|stem |committed|default build|conv_math=" gpurun_out/r4a_tests.log | head -40
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4a_bench.log 2>&1
echo "== bench exit $?"
tail -n 1 gpurun_out/r4a_bench.log | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    print('value', d['value'], 'ms/step', d['ms_per_step'], 'roof', d['roofline']['kernel'], d['roofline']['achieved'], d['roofline']['frac'], 'lat', d.get('single_image_latency', {}).get('ms_per_frame'))
    for k, v in sorted(d['kernel_classes'].items(), key=lambda kv: -kv[1]['ms']):
        print('  %-40s n=%3d ms=%8.3f tflops=%7.2f gbs=%7.1f' % (k, v['launches'], v['ms'], v['tflops'], v['gbs']))
except Exception as e:
    print('parse failed', e)
"
tail -n 5 gpurun_out/r4a_bench.log | cut -c1-300 | grep -v '^{'
