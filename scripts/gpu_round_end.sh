#!/bin/bash
# Everything a round's profiles/ needs, in one gpurun call: the driver's checks (smoke, GPU suite, default bench line twice), the
# rocprofv3 / PMC passes of the same command, the fused-block phase traces, the B=1 latency breakdown and the secondary bench lines.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
bash scripts/gpu_final.sh > gpurun_out/final_run.log 2>&1
tail -4 gpurun_out/final_run.log
bash scripts/gpu_profile.sh > gpurun_out/profile_run.log 2>&1
grep -E "exit|aligned" gpurun_out/profile_run.log | head -12
for c in 32 64; do
  (for d in 0 7 8 1 2 4; do echo "== ROMP_CONV_DEBUG=$d"; BB_C=$c BB_FUSED_ONLY=$([ $d = 0 ] && echo 0 || echo 1) ROMP_CONV_DEBUG=$d ROMP_CONV_TRACE=1 timeout 100 python scripts/bblock_bench.py 2>&1 | grep -v amdgpu.ids | head -22; done) > gpurun_out/bblock${c}_trace.txt 2>&1
done
timeout 200 python scripts/latency_b1.py 2>&1 | grep -v amdgpu.ids > gpurun_out/latency_b1.txt; cat gpurun_out/latency_b1.txt
bash scripts/gpu_bench_lines.sh 2>&1 | tail -8
