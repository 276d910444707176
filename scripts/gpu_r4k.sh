#!/bin/bash
# Round 4, call K: conv_h2r ck32 (two 16-channel sub-stages per barrier): parity of every variant, sweep.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short --timeout 900 -x -k "conv_layer and k3_s1" > gpurun_out/r4k_tests.log 2>&1
echo "== tests exit $? :: $(tail -n 1 gpurun_out/r4k_tests.log)"; grep -E "FAILED|Error|assert" gpurun_out/r4k_tests.log | head
SWEEP_CASES=s1 SWEEP_FILTER=h2r,h2_k3s1_mt1_nt1_tw16_ck32 timeout 900 python scripts/conv_sweep.py > gpurun_out/r4k_sweep_s1.log 2>&1
grep -v amdgpu.ids gpurun_out/r4k_sweep_s1.log | awk '/^case/{n=0; print} !/^case/{if (n<7) print; n++}'
