#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist1.py -m gpu -q --tb=short -x --timeout 600 -k "forward_chunks or forward_batch or dist" > gpurun_out/r2m_tests.log 2>&1
echo "== tests exit $? :: $(tail -n 1 gpurun_out/r2m_tests.log)"; grep -E "FAILED|Error|assert" gpurun_out/r2m_tests.log | head
timeout 900 python bench.py --no-cpu-baseline --no-f32-companion > gpurun_out/r2m_bench.log 2>&1; echo "== bench exit $?"; tail -n 1 gpurun_out/r2m_bench.log | cut -c1-900
