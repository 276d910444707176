#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes per op for the secondary lines (committed tables): BEV, ResNet-50, B = 128
cd "$(dirname "$0")/.."
for W in bev resnet50 b128; do
  case $W in bev) A="--workload bev";; resnet50) A="--backbone resnet50";; b128) A="--batch 128 --global-batch 256";; esac
  PROFILE_ONLY=pmc PROF_TAG=_$W BENCH_ARGS="$A" bash scripts/gpu_profile.sh > gpurun_out/profile_$W.log 2>&1
  echo "== $W :: $(grep -E 'ops aligned|no forward' gpurun_out/profile_$W.log | tail -1)"
done
