#!/bin/bash
# Round 6, GPU call 4: conv_h2g's 3x3 variants (parity on every 3x3 case, sweeps of the stride-2 and stride-1 classes against
# conv_h2s / conv_h2r), then fresh variant tables for the four committed configurations with the new kernel list.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
O=gpurun_out/r06c4
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "test_conv_layer and _k3_ and h2" 2>&1 | tail -4
SWEEP_CASES=s2 SWEEP_FILTER=h2g,h2s,h2d_ timeout 900 python scripts/conv_sweep.py 2>/dev/null > ${O}_s2_sweep_b32.txt; grep -A4 "^case" ${O}_s2_sweep_b32.txt | grep -v "^--"
SWEEP_CASES=s1 SWEEP_FILTER=h2g,h2r timeout 900 python scripts/conv_sweep.py 2>/dev/null > ${O}_s1_sweep_b32.txt; grep -A4 "^case" ${O}_s1_sweep_b32.txt | grep -v "^--"
bash scripts/gpu_tables.sh
python - <<'PY'
import json
for W in ('default', 'b128', 'bev', 'resnet50'):
    try:
        r = json.loads([l for l in open('gpurun_out/tables_%s.log' % W) if l.startswith('{')][-1])
    except Exception as e:
        print(W, 'no line', e); continue
    print(W, r['value'], r['config'].get('ms_per_call'), r['config'].get('maps_max_abs_vs_oracle'), r['config'].get('detections_equal'), r['roofline']['kernel'], r['roofline']['frac'])
    for k, v in sorted(r['kernel_classes'].items(), key=lambda kv: -kv[1]['ms']):
        if v['ms'] > 0.08: print('    %-44s n=%3d ms=%7.3f tflops=%7.1f gbs=%7.1f' % (k, v['launches'], v['ms'], v['tflops'], v['gbs']))
    print('    serial sum', round(sum(v['ms'] for v in r['kernel_classes'].values()), 3))
PY
