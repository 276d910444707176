#!/bin/bash
# conv-layer parity of every variant, then one bench run with the per-class table
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "conv_layer" 2>&1 | tail -2
A="--no-cpu-baseline --no-f32-companion --no-latency --no-end-to-end --no-parity --steps 6 --warmup 2"
python bench.py $A 2>/dev/null | tail -1 | python -c "
import sys,json; r=json.loads(sys.stdin.readline()); print(r['value'], r['config']['ms_per_call'], r['roofline']['kernel'], r['roofline']['frac'], r['roofline']['net_ms_per_batch'])
for k,v in sorted(r['kernel_classes'].items(), key=lambda kv:-kv[1]['ms'])[:8]: print('  ', k, v)"
