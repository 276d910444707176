#!/bin/bash
# Round 4, call H: after the h2_pack fix: debug flow once, then bench with a fresh table + per-op table.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 300 python scripts/debug_fault.py 32 > gpurun_out/r4h_debug.log 2>&1; echo "== debug flow exit $? :: $(tail -n 1 gpurun_out/r4h_debug.log)"
rm -f gpurun_out/tune_r4h.json
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-companion --tune-file gpurun_out/tune_r4h.json > gpurun_out/r4h_bench.log 2>&1
echo "== bench exit $?"
tail -n 1 gpurun_out/r4h_bench.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('value', d['value'], 'ms/step', d['ms_per_step'], 'roof', d['roofline']['kernel'], d['roofline']['achieved'], d['roofline']['frac'], 'lat', d.get('single_image_latency', {}).get('ms_per_frame'), 'maps', d['config'].get('maps_max_abs_vs_oracle'), d['config'].get('detections_equal'))
tot = 0
for k, v in sorted(d['kernel_classes'].items(), key=lambda kv: -kv[1]['ms']):
    tot += v['ms']
    print('  %-40s n=%3d ms=%8.3f tflops=%7.2f gbs=%7.1f' % (k, v['launches'], v['ms'], v['tflops'], v['gbs']))
print('  serial kernel sum', round(tot, 3))
"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short --timeout 900 -x -k "conv_layer and h2" > gpurun_out/r4h_tests.log 2>&1
echo "== conv_layer h2 tests exit $? :: $(tail -n 1 gpurun_out/r4h_tests.log)"
