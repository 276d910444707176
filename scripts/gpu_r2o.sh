#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_bev.py -m gpu -q --tb=short -x --timeout 600 -k "benchmark_batch" -s > gpurun_out/r2o_tests.log 2>&1
echo "== tests exit $? :: $(tail -n 1 gpurun_out/r2o_tests.log)"; grep -E "FAILED|Error|assert|BEV B=32" gpurun_out/r2o_tests.log | head
timeout 300 python scripts/latency_b1.py > gpurun_out/r2o_latency.log 2>&1; tail -n 1 gpurun_out/r2o_latency.log
timeout 600 python bench.py --workload bev > gpurun_out/r2o_bench_bev.log 2>&1; echo "== bev exit $?"; tail -n 1 gpurun_out/r2o_bench_bev.log | cut -c1-1200
timeout 300 python bench.py --workload smpl > gpurun_out/r2o_bench_smpl.log 2>&1; echo "== smpl exit $?"; tail -n 1 gpurun_out/r2o_bench_smpl.log | cut -c1-900
timeout 600 python bench.py --backbone resnet50 --no-f32-companion > gpurun_out/r2o_bench_resnet50.log 2>&1; echo "== resnet exit $?"; tail -n 1 gpurun_out/r2o_bench_resnet50.log | cut -c1-500
timeout 600 python bench.py --batch 128 --no-f32-companion --no-cpu-baseline > gpurun_out/r2o_bench_b128.log 2>&1; echo "== b128 exit $?"; tail -n 1 gpurun_out/r2o_bench_b128.log | cut -c1-300
