#!/bin/bash
# round-end check as the driver runs it: smoke, the GPU suite, the default bench line (twice: run-to-run spread)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for i in 1 2; do
timeout 900 python bench.py 2>/dev/null | grep '^{' | tail -1 > gpurun_out/bench_default_$i.json
python -c "
import json; r=json.load(open('gpurun_out/bench_default_$i.json'))
print(r['value'], r['config']['ms_per_call'], r['roofline']['kernel'], r['roofline']['bound'], r['roofline']['frac'], 'e2e', r['end_to_end']['value'], 'f32', r['f32_mfma_companion']['value'], 'lat', r['single_image_latency']['fps'], 'cpu', r['cpu_baseline']['value'], 'parity', r['config']['maps_max_abs_vs_oracle'], r['config']['mesh_max_abs_vs_oracle'], r['config']['detections_equal'])"
done
