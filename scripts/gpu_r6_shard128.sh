#!/bin/bash
# rank 0's exact work in the 8-GPU job, on one GPU -- 1024 / 8 = 128 images per step walked in 4 calls of 32 (the
# forward_chunks pipeline fills and drains every step) -- beside the default N = 1 line of the same box; predicted_8gpu = 8 x the
# shard's images/s, efficiency_upper = predicted / (8 x N1): what the 8-GPU point can reach at most before any fabric cost.
# Usage: gpurun --timeout 900 -- 'bash scripts/gpu_r6_shard128.sh'  -> gpurun_out/r06_bench_shard128.json
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
LEGS="--no-cpu-baseline --no-parity --no-f32-companion --no-latency --no-end-to-end --no-roofline"
for r in 1 2; do
  timeout 600 python bench.py --steps 10 $LEGS 2>/dev/null | grep '^{' | tail -1 > gpurun_out/r06_shard_n1_$r.json
  timeout 600 python bench.py --global-batch 128 --batch 32 --steps 80 --warmup 10 $LEGS $SHARD_ARGS 2>/dev/null | grep '^{' | tail -1 > gpurun_out/r06_shard_128_$r.json
done
python - <<'PY'
import json
n1 = [json.load(open('gpurun_out/r06_shard_n1_%d.json' % r)) for r in (1, 2)]
sh = [json.load(open('gpurun_out/r06_shard_128_%d.json' % r)) for r in (1, 2)]
best1, bests = max(n1, key=lambda r: r['value']), max(sh, key=lambda r: r['value'])
out = dict(bests)
out['shard_prediction'] = {
    'what': "rank 0's work of the 8-GPU job on one GPU: --global-batch 128 --batch 32 (4 calls per step), against the default N = 1 line of the same box; two interleaved runs per arm, best of each",
    'n1_images_per_s': [r['value'] for r in n1], 'shard_images_per_s': [r['value'] for r in sh],
    'predicted_8gpu': round(8 * bests['value'], 1), 'efficiency_upper': round(bests['value'] / best1['value'], 4),
    'n1_ms_per_call': best1['config']['ms_per_call'], 'shard_ms_per_step': bests['ms_per_step'], 'shard_ms_per_call': bests['config']['ms_per_call']}
import os
json.dump(out, open('gpurun_out/r06_bench_shard128%s.json' % ('_nocross' if '--cross-step 0' in os.environ.get('SHARD_ARGS', '') else ''), 'w'))
print(json.dumps(out['shard_prediction']))
PY
