#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
ABLATE_KIND=h2p,h2w ABLATE_DBG=0,1,8,9,4,12,13 timeout 900 python scripts/conv_ablate.py > gpurun_out/r2g_ablate.log 2>&1; cat gpurun_out/r2g_ablate.log
