#!/bin/bash
# Round 4, call D: P = 4 variants (h2r, h2s), packed h2_pack: parity of every variant, sweeps, bench with a fresh table.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short --timeout 900 -x -k "conv_layer or fusesum or net_golden" > gpurun_out/r4d_tests.log 2>&1
echo "== tests exit $? :: $(tail -n 1 gpurun_out/r4d_tests.log)"; grep -E "FAILED|Error|assert" gpurun_out/r4d_tests.log | head
SWEEP_CASES=s1 SWEEP_FILTER=h2r,h2_k3s1_mt1_nt1_tw16_ck32 timeout 900 python scripts/conv_sweep.py > gpurun_out/r4d_sweep_s1.log 2>&1
grep -v amdgpu.ids gpurun_out/r4d_sweep_s1.log | awk '/^case/{n=0; print} !/^case/{if (n<6) print; n++}'
SWEEP_CASES=s2 SWEEP_FILTER=h2s timeout 900 python scripts/conv_sweep.py > gpurun_out/r4d_sweep_s2.log 2>&1
grep -v amdgpu.ids gpurun_out/r4d_sweep_s2.log | awk '/^case/{n=0; print} !/^case/{if (n<4) print; n++}'
rm -f gpurun_out/tune_r4d.json
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-companion --tune-file gpurun_out/tune_r4d.json > gpurun_out/r4d_bench.log 2>&1
echo "== bench exit $?"
tail -n 1 gpurun_out/r4d_bench.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('value', d['value'], 'ms/step', d['ms_per_step'], 'roof', d['roofline']['kernel'], d['roofline']['achieved'], d['roofline']['frac'], 'lat', d.get('single_image_latency', {}).get('ms_per_frame'), 'maps', d['config'].get('maps_max_abs_vs_oracle'), d['config'].get('detections_equal'))
tot = 0
for k, v in sorted(d['kernel_classes'].items(), key=lambda kv: -kv[1]['ms']):
    tot += v['ms']
    print('  %-40s n=%3d ms=%8.3f tflops=%7.2f gbs=%7.1f' % (k, v['launches'], v['ms'], v['tflops'], v['gbs']))
print('  serial kernel sum', round(tot, 3))
"
