#!/bin/bash
# Round 4, call I: the direct H2 epilogue of conv_h2r / conv_h2s: parity, A/B sweeps (ROMP_CONV_DEBUG=512 = LDS-transposed), bench.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short --timeout 900 -x -k "(conv_layer and h2) or net_golden or net_vs_oracle" > gpurun_out/r4i_tests.log 2>&1
echo "== tests exit $? :: $(tail -n 1 gpurun_out/r4i_tests.log)"; grep -E "FAILED|Error|assert" gpurun_out/r4i_tests.log | head
for dbg in 0 512; do
  ROMP_CONV_DEBUG=$dbg SWEEP_CASES=s1 SWEEP_FILTER=h2r timeout 600 python scripts/conv_sweep.py > gpurun_out/r4i_sweep_s1_$dbg.log 2>&1
  ROMP_CONV_DEBUG=$dbg SWEEP_CASES=s2 SWEEP_FILTER=h2s timeout 600 python scripts/conv_sweep.py > gpurun_out/r4i_sweep_s2_$dbg.log 2>&1
  echo "---- ROMP_CONV_DEBUG=$dbg"
  cat gpurun_out/r4i_sweep_s1_$dbg.log gpurun_out/r4i_sweep_s2_$dbg.log | grep -v amdgpu.ids | awk '/^case/{n=0; print} !/^case/{if (n<2) print; n++}'
done
rm -f gpurun_out/tune_r4i.json
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-companion --tune-file gpurun_out/tune_r4i.json > gpurun_out/r4i_bench.log 2>&1
echo "== bench exit $?"
tail -n 1 gpurun_out/r4i_bench.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('value', d['value'], 'ms/step', d['ms_per_step'], 'roof', d['roofline']['kernel'], d['roofline']['achieved'], d['roofline']['frac'], 'lat', d.get('single_image_latency', {}).get('ms_per_frame'), 'maps', d['config'].get('maps_max_abs_vs_oracle'), d['config'].get('detections_equal'))
tot = 0
for k, v in sorted(d['kernel_classes'].items(), key=lambda kv: -kv[1]['ms']):
    tot += v['ms']
    if v['ms'] > 0.15: print('  %-40s n=%3d ms=%8.3f tflops=%7.2f gbs=%7.1f' % (k, v['launches'], v['ms'], v['tflops'], v['gbs']))
print('  serial kernel sum', round(tot, 3))
"
