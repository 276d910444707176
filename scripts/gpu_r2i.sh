#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -x --timeout 900 -k "conv_layer and h2 and k3_s1" > gpurun_out/r2i_tests.log 2>&1
echo "== tests exit $? :: $(tail -n 1 gpurun_out/r2i_tests.log)"
ABLATE_KIND=h2p,h2w ABLATE_DBG=0,4 timeout 900 python scripts/conv_ablate.py > gpurun_out/r2i_ablate.log 2>&1; cat gpurun_out/r2i_ablate.log
