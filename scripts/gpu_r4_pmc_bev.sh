#!/bin/bash
cd "$(dirname "$0")/.."
PROFILE_ONLY=pmc PROF_TAG=_bev BENCH_ARGS="--workload bev" bash scripts/gpu_profile.sh > gpurun_out/profile_bev.log 2>&1
echo "== bev :: $(grep -E 'ops aligned|no forward|Error' gpurun_out/profile_bev.log | tail -2)"
