#!/bin/bash
# the secondary bench lines of the round (committed under profiles/)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python bench.py --workload bev 2>/dev/null | grep '^{' | tail -1 > gpurun_out/bench_bev.json; cut -c1-400 gpurun_out/bench_bev.json
timeout 300 python bench.py --workload smpl 2>/dev/null | grep '^{' | tail -1 > gpurun_out/bench_smpl.json; cut -c1-400 gpurun_out/bench_smpl.json
timeout 900 python bench.py --backbone resnet50 --no-f32-companion --no-latency 2>/dev/null | grep '^{' | tail -1 > gpurun_out/bench_resnet50.json; cut -c1-300 gpurun_out/bench_resnet50.json
timeout 900 python bench.py --batch 128 --no-f32-companion --no-latency --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > gpurun_out/bench_b128.json; cut -c1-300 gpurun_out/bench_b128.json
