#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -x -s -k "conv_layer" --timeout 600 > gpurun_out/bx3_tests.log 2>&1
echo "== conv tests exit $? :: $(tail -n 1 gpurun_out/bx3_tests.log)"
grep -E "conv_bx3" gpurun_out/bx3_tests.log | sort -t: -k2 | awk '{print $0}' | sort -k7 -g | tail -8
grep -E "FAILED|Error|assert" gpurun_out/bx3_tests.log | head -10
BENCH_ARGS="--conv-math bf16x3" bash scripts/gpu_quick.sh 2>&1 | grep -vE "k1s1" | head -30
