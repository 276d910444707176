#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -x -k "benchmark_batch" --timeout 900 -s > gpurun_out/r2b_tests.log 2>&1
echo "== tests exit $? :: $(tail -n 1 gpurun_out/r2b_tests.log)"
grep -E "FAILED|Error|error|assert|vs oracle|vs reference|persons in" gpurun_out/r2b_tests.log | head -40
timeout 600 python scripts/op_table.py 32 f16x2 > gpurun_out/r2b_optable.log 2>&1; echo "== op_table exit $?"; tail -n 45 gpurun_out/r2b_optable.log
ABLATE_KIND=h2,h2d timeout 900 python scripts/conv_ablate.py > gpurun_out/r2b_ablate.log 2>&1; echo "== ablate exit $?"; cat gpurun_out/r2b_ablate.log
