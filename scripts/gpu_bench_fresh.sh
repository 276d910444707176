#!/bin/bash
# bench.py with a freshly measured variant table (saved to gpurun_out/tune_fresh.json for romp_amd/tune/), no CPU leg.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
rm -f gpurun_out/tune_fresh.json
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-companion --tune-file gpurun_out/tune_fresh.json $BENCH_ARGS > gpurun_out/bench_fresh.log 2>&1
echo "== bench exit $?"
tail -n 1 gpurun_out/bench_fresh.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('value', d['value'], 'ms/step', d['ms_per_step'], 'roof', d['roofline']['kernel'], d['roofline']['achieved'], d['roofline']['frac'], 'lat', d.get('single_image_latency', {}).get('ms_per_frame'), 'maps', d['config'].get('maps_max_abs_vs_oracle'), d['config'].get('detections_equal'))
tot = 0
for k, v in sorted(d['kernel_classes'].items(), key=lambda kv: -kv[1]['ms']):
    tot += v['ms']
    if v['ms'] > 0.1: print('  %-40s n=%3d ms=%8.3f tflops=%7.2f gbs=%7.1f' % (k, v['launches'], v['ms'], v['tflops'], v['gbs']))
print('  serial kernel sum', round(tot, 3))
"
