#!/bin/bash
# Round 4, call C: where does conv_h2s spend its time?  sweep of every s2 variant + phase traces; saturation test.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short --timeout 600 -x -k "saturation" > gpurun_out/r4c_tests.log 2>&1
echo "== tests exit $? :: $(tail -n 1 gpurun_out/r4c_tests.log)"; grep -E "FAILED|Error|assert|default build" gpurun_out/r4c_tests.log | head
SWEEP_CASES=s2 SWEEP_FILTER=k3s2 timeout 900 python scripts/conv_sweep.py > gpurun_out/r4c_sweep.log 2>&1
grep -v amdgpu.ids gpurun_out/r4c_sweep.log | awk '/^case/{n=0; print} !/^case/{if (n<7) print; n++}'
ROMP_CONV_TRACE=1 TRACE_CASES="32,128,3,2,128,0;64,64,3,2,256,0;256,64,3,2,128,0" timeout 600 python scripts/conv_trace.py 32 h2s_k3s2_mt2_nt4 h2s_k3s2_mt1_nt2 h2s_k3s2_mt2_nt2 > gpurun_out/r4c_trace.log 2>&1
grep -v amdgpu.ids gpurun_out/r4c_trace.log | grep -v "timeline" | head -120
