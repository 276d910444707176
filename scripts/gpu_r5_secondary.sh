#!/bin/bash
# Round 5: the secondary bench lines (BEV, ResNet-50 with its new parity / cpu_baseline legs, B = 128, SMPL-only) with their
# FETCH_SIZE / WRITE_SIZE passes taken first (scripts/gpu_pmc_secondary.sh) so that every line's roofline.traffic is this build's.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
bash scripts/gpu_pmc_secondary.sh
for W in bev resnet50 b128; do
  for C in FETCH_SIZE WRITE_SIZE; do cp gpurun_out/prof_$W/pmc_${C}_by_kernel.csv profiles/r05_${W}_pmc_${C}_by_kernel.csv 2>/dev/null; done
  cp gpurun_out/prof_$W/pmc_traffic_by_op.json profiles/r05_${W}_pmc_traffic_by_op.json 2>/dev/null
done
bash scripts/gpu_bench_lines.sh > gpurun_out/r05_bench_lines.log 2>&1
python - <<'PY'
import json
for w in ('bev', 'resnet50', 'b128', 'smpl'):
    try:
        d = json.load(open('gpurun_out/bench_%s.json' % w)); r = d.get('roofline', {}); c = d['config']
        print(w, 'value', d['value'], d['unit'], 'roof', r.get('kernel'), r.get('bound'), r.get('frac'), 'traffic', r.get('traffic'),
              'cpu', d.get('cpu_baseline', {}).get('value'), d.get('cpu_baseline', {}).get('kind'),
              'parity', c.get('maps_max_abs_vs_oracle'), c.get('detections_equal'), c.get('mesh_max_abs_vs_oracle'), len(c.get('images_compared', [])))
    except Exception as e:
        print(w, 'parse failed', e)
PY
