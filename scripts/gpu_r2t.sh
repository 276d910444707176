#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
ROMP_CONV_TRACE=1 timeout 600 python scripts/conv_trace.py 1 h2_k3s1_mt1_nt1_tw16 2>&1 | grep -v "^$" | tee gpurun_out/trace_b1.txt
