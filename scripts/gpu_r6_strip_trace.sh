#!/bin/bash
# phase stamps (ROMP_CONV_TRACE=1) of the fused BasicBlock kernels, plain tile order against the strip form
for C in 64 32; do for run in 0 -1; do
if [ $run = 0 ]; then export ROMP_BBLOCK_RUN=0; else unset ROMP_BBLOCK_RUN; fi
echo "== C=$C ROMP_BBLOCK_RUN=${ROMP_BBLOCK_RUN:-auto}"
ROMP_CONV_TRACE=1 BB_C=$C BB_FUSED_ONLY=1 timeout 300 python scripts/bblock_bench.py 2>&1 | grep -v "Warn\|amdgpu.ids"
done; done > gpurun_out/r06s_trace3.txt 2>&1
