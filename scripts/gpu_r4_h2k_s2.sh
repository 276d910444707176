#!/bin/bash
# conv_h2k for the stride-2 convs of a single-image plan: parity of every stride-2 variant, the single-image network, latency A/B
cd "$(dirname "$0")/.."; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short --timeout 400 -x -k "(conv_layer and k3_s2) or (split_k and f16x2)" > gpurun_out/h2k_s2_tests.log 2>&1
echo "== tests exit $? :: $(tail -n 1 gpurun_out/h2k_s2_tests.log)"; grep -E "FAILED|Error|assert" gpurun_out/h2k_s2_tests.log | head
for m in 1 0 1; do
  ROMP_KSPLIT_S2=$m timeout 200 python scripts/latency_b1.py > gpurun_out/h2k_s2_latency_$m.txt 2>&1; echo "ROMP_KSPLIT_S2=$m :: $(grep 'ROMP(image)' gpurun_out/h2k_s2_latency_$m.txt | cut -c1-200)"
done
