#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "smpl" -m gpu 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_bev.py tests/test_temporal.py tests/test_render.py -q -x -m gpu -s 2>&1 | grep -v "^$" | tail -25
timeout 300 python bench.py --workload smpl 2>&1 | tail -2 | tee gpurun_out/bench_smpl.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_smpl -o smpl -- python /root/repo/bench.py --workload smpl --no-cpu-baseline > /dev/null 2>&1
head -8 /root/repo/gpurun_out/prof_smpl/*/smpl_kernel_stats.csv 2>/dev/null || find /root/repo/gpurun_out/prof_smpl -name "*kernel_stats*" | head
