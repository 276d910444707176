#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for skew in 0 1500 3000 5000; do
echo "=== skew $skew"
ROMP_CONV_SKEW=$skew ABLATE_KIND=h2_,h2d ABLATE_DBG=0 timeout 600 python scripts/conv_ablate.py 2>&1 | grep -E "case|h2_k3s1_mt1_nt2_tw16|h2_k3s1_mt1_nt1_tw16|h2d_k3s1_mt2_nt2_tw16_ck16|h2d_k3s1_mt1_nt2_tw16"
done
