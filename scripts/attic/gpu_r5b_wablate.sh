#!/bin/bash
# Is the 128 / 256-channel 3x3 class bound by its weight stream from L2?  conv_h2r with ROMP_CONV_DEBUG=64 (every weight fragment
# load re-reads the first tap's 2 KB: same instructions, L1 hits), =1 (pixel DMA from the zero page), =65 (both), =4 (no epilogue)
# against the plain kernel, same process order per shape.  Outputs are wrong under the knock-outs.
cd "$(dirname "$0")/.."; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
for D in 0 64 1 65 4 0; do
  echo "== ROMP_CONV_DEBUG=$D"
  ROMP_CONV_DEBUG=$D SWEEP_CASES=s1 SWEEP_FILTER=h2r_k3s1_mt2_nt4_tw16_ck32,h2r_k3s1_mt2_nt2_tw16_ck32,h2r_k3s1_mt2_nt1_tw16_ck32 timeout 200 python scripts/conv_sweep.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r5b4_wablate.txt
