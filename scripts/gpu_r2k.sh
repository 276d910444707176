#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_bev_post.py -m gpu -q --tb=short -x --timeout 900 -k "(conv_layer and h2 and k3_s1) or preprocess" > gpurun_out/r2k_tests.log 2>&1
echo "== tests exit $? :: $(tail -n 1 gpurun_out/r2k_tests.log)"; grep -E "FAILED|Error|assert" gpurun_out/r2k_tests.log | head
ABLATE_KIND=h2q ABLATE_DBG=0,4,1,8,13 timeout 900 python scripts/conv_ablate.py > gpurun_out/r2k_ablate.log 2>&1; cat gpurun_out/r2k_ablate.log
ROMP_CONV_TRACE=1 timeout 600 python scripts/conv_trace.py 32 h2q_k3s1_mt2_nt2_tw16 h2q_k3s1_mt4_nt1 > gpurun_out/r2k_trace.log 2>&1; grep -v "timeline\|^  wave" gpurun_out/r2k_trace.log | head -60
