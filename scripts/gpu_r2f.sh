#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -x --timeout 900 -s -k "conv_layer and h2 and k3_s1" > gpurun_out/r2f_tests.log 2>&1
echo "== tests exit $? :: $(tail -n 1 gpurun_out/r2f_tests.log)"
grep -E "FAILED|Error|error|assert|h2p" gpurun_out/r2f_tests.log | head -40
ROMP_CONV_TRACE=1 timeout 600 python scripts/conv_trace.py 32 h2p_ > gpurun_out/r2f_trace.log 2>&1; echo "trace exit $?"; grep -v "timeline" gpurun_out/r2f_trace.log | head -120
ABLATE_KIND=h2p ABLATE_DBG=0,32,7,4 timeout 600 python scripts/conv_ablate.py > gpurun_out/r2f_ablate.log 2>&1; cat gpurun_out/r2f_ablate.log
