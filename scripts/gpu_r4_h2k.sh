#!/bin/bash
# conv_h2k (K-split across the waves of a workgroup): parity of every variant, single-image network parity, latency A/B on one box
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short --timeout 900 -x -k "(conv_layer and k3_s1) or split_k or conv_math_all or romp_api or net_golden" > gpurun_out/h2k_tests.log 2>&1
echo "== tests exit $? :: $(tail -n 1 gpurun_out/h2k_tests.log)"; grep -E "FAILED|Error|assert" gpurun_out/h2k_tests.log | head
for m in 1 0 1 0; do
  ROMP_KSPLIT_WG=$m timeout 300 python scripts/latency_b1.py > gpurun_out/h2k_latency_$m.txt 2>&1; echo "ROMP_KSPLIT_WG=$m :: $(grep 'ROMP(image)' gpurun_out/h2k_latency_$m.txt | cut -c1-200)"
done
ROMP_KSPLIT_WG=1 timeout 600 python scripts/op_table.py 1 f16x2 > gpurun_out/h2k_optable_b1.log 2>&1
sed -n '/total serial/,$p' gpurun_out/h2k_optable_b1.log | head -16
