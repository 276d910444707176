#!/bin/bash
# full GPU suite + default bench (what the driver runs at round end)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
timeout 900 python bench.py 2>gpurun_out/bench_err.log | tail -1 | tee gpurun_out/bench_default.json
tail -3 gpurun_out/bench_err.log
