#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_resnet.py -m gpu -q --tb=short --timeout 900 -x -k "net_golden or net_vs_oracle or split_k or resnet" > gpurun_out/r4o_tests.log 2>&1
echo "== tests exit $? :: $(tail -n 1 gpurun_out/r4o_tests.log)"; grep -E "FAILED|Error|assert" gpurun_out/r4o_tests.log | head
bash scripts/gpu_bench_fresh.sh
