#!/bin/bash
# the ResNet-50 traffic passes and bench line again (scripts/pmc_by_op.py did not know stem7p_kernel when gpu_r6_profiles.sh ran)
cd "$(dirname "$0")/.."
PROFILE_ONLY=pmc PROF_TAG=_resnet50 BENCH_ARGS="--backbone resnet50" bash scripts/gpu_profile.sh > gpurun_out/profile_resnet50.log 2>&1
echo "== resnet50 :: $(grep -E 'ops aligned|no forward' gpurun_out/profile_resnet50.log | tail -1)"
cp gpurun_out/prof_resnet50/pmc_traffic_by_op.json profiles/r06_resnet50_pmc_traffic_by_op.json
timeout 900 python bench.py --backbone resnet50 --no-f32-companion --no-latency 2>/dev/null | grep '^{' | tail -1 > gpurun_out/bench_resnet50.json; cut -c1-300 gpurun_out/bench_resnet50.json
