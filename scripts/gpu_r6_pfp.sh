#!/bin/bash
# strip form, the prologue of a run's first tile: fragments two units ahead (product) against one (-DROMP_BBLOCK_PFP1 build), 64 channels
O=gpurun_out
{
for rep in 1 2 3; do for arm in pfp2 pfp1; do
  if [ $arm = pfp1 ]; then export ROMP_HIP_LIB=romp_amd/libromp_hip_pfp1.so; else unset ROMP_HIP_LIB; fi
  echo -n "$arm rep $rep: "
  BB_C=64 BB_FUSED_ONLY=1 timeout 300 python scripts/bblock_bench.py 2>&1 | grep fuse= | sed 's/.*\(bblock[0-9]* [0-9.]* us\).*/\1/'
done; done
unset ROMP_HIP_LIB
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "fused_basic_block_strip" 2>&1 | tail -2
} > $O/r06s_pfp_ab.txt 2>&1
cat $O/r06s_pfp_ab.txt
