#!/bin/bash
# the default bench line exactly as the driver runs it
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 230 python bench.py 2>gpurun_out/bench_err.log | grep '^{' | tail -1 > gpurun_out/bench_default.json
python -c "
import json; r=json.load(open('gpurun_out/bench_default.json')); ro=r['roofline']
print(r['value'], r['config']['ms_per_call'], ro['kernel'], ro['bound'], ro['frac'], 'traffic', ro.get('traffic'), ro.get('traffic_over_algorithmic'), 'e2e', r['end_to_end']['value'], 'f32', r['f32_mfma_companion']['value'], 'lat', r['single_image_latency'], 'cpu', r['cpu_baseline']['value'])"
tail -2 gpurun_out/bench_err.log
