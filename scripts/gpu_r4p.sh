#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
ROMP_CONV_TRACE=1 TRACE_CASES="128,128,3,1,32,1;256,256,3,1,16,1;64,64,3,1,128,0" timeout 600 python scripts/conv_trace.py 32 h2r_k3s1_mt2_nt4_tw16_ck32 h2r_k3s1_mt2_nt2_tw16_ck32 > gpurun_out/r4p_trace.log 2>&1
grep -v amdgpu.ids gpurun_out/r4p_trace.log | grep -v "timeline" | head -60
