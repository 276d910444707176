#!/bin/bash
# Round 6, GPU call 2: the FP16_OVFL / TRAPSTS probe, conv_h2g (1x1 streamed-K GEMM) parity + sweep on ResNet-50's shapes, the ResNet-50
# line with a freshly measured table, the range-guard tests.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
O=gpurun_out/r06c2
./scripts/micro/_bin/probe_r6 > ${O}_probe_r6.txt 2>&1; cat ${O}_probe_r6.txt
timeout 300 python -m pytest tests/test_gpu_range_guard.py -x -q 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "test_conv_layer and _k1_" 2>&1 | tee ${O}_conv_k1_tests.log | tail -8
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "counted_waits_vs_full_drain or test_seam1x1" 2>&1 | tail -4
SWEEP_CASES=rn SWEEP_CHECK=1 timeout 900 python scripts/conv_sweep.py > ${O}_rn_sweep_b32.txt 2>&1
grep -A4 "^case" ${O}_rn_sweep_b32.txt | head -80
timeout 900 python bench.py --backbone resnet50 --tune-file gpurun_out/tune_resnet50_r6.json --no-cpu-baseline --no-end-to-end 2>${O}_resnet.err | grep '^{' | tail -1 > ${O}_bench_resnet50.json
python - <<'PY'
import json
r = json.load(open('gpurun_out/r06c2_bench_resnet50.json'))
print('resnet50', r['value'], r['config'].get('ms_per_call'), r['config'].get('maps_max_abs_vs_oracle'), r['config'].get('detections_equal'))
for k, v in sorted(r['kernel_classes'].items(), key=lambda kv: -kv[1]['ms']):
    if v['ms'] > 0.05: print('  %-44s n=%3d ms=%7.3f tflops=%7.1f gbs=%7.1f' % (k, v['launches'], v['ms'], v['tflops'], v['gbs']))
print('  serial sum', round(sum(v['ms'] for v in r['kernel_classes'].values()), 3))
PY
