#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "split_k or net_b1" 2>&1 | tail -5
for sk in 0 64 128 256; do
echo "=== SPLIT_K=$sk"
SPLIT_K=$sk timeout 600 python scripts/latency_breakdown.py 2>&1 | grep -v "^$" | head -14
done
