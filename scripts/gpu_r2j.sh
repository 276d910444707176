#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_bev_post.py tests/test_gpu_dist1.py -m gpu -q --tb=short -x --timeout 600 > gpurun_out/r2j_tests.log 2>&1
echo "== tests exit $? :: $(tail -n 1 gpurun_out/r2j_tests.log)"; grep -E "FAILED|Error|assert" gpurun_out/r2j_tests.log | head
timeout 1200 python bench.py > gpurun_out/r2j_bench.log 2>&1; echo "== bench exit $?"; tail -n 3 gpurun_out/r2j_bench.log | cut -c1-4000
timeout 600 python bench.py --workload bev > gpurun_out/r2j_bench_bev.log 2>&1; echo "== bench bev exit $?"; tail -n 2 gpurun_out/r2j_bench_bev.log | cut -c1-2500
timeout 600 python bench.py --batch 128 --no-f32-companion --no-cpu-baseline > gpurun_out/r2j_bench_b128.log 2>&1; echo "== bench b128 exit $?"; tail -n 2 gpurun_out/r2j_bench_b128.log | cut -c1-1500
