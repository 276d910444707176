#!/bin/bash
# conv_h2k (K-split across the waves of a workgroup): parity of every variant, single-image network parity, latency A/B on one box,
# kernel timeline of a single-image forward
REPO="$(cd "$(dirname "$0")/.." && pwd)"
cd "$REPO"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short --timeout 900 -x -k "(conv_layer and (k3_s1 or k1_s1 or k3_s2)) or split_k or conv_math_all or romp_api or net_golden" > gpurun_out/h2k_tests.log 2>&1
echo "== tests exit $? :: $(tail -n 1 gpurun_out/h2k_tests.log)"; grep -E "FAILED|Error|assert" gpurun_out/h2k_tests.log | head
for m in 1 0 1 0; do
  ROMP_KSPLIT_WG=$m timeout 300 python scripts/latency_b1.py > gpurun_out/h2k_latency_$m.txt 2>&1; echo "ROMP_KSPLIT_WG=$m :: $(grep 'ROMP(image)' gpurun_out/h2k_latency_$m.txt | cut -c1-200)"
done
for m in 1 0 1; do       # the stride-2 convs on conv_h2k (1) or split through ksum as before (0)
  ROMP_KSPLIT_S2=$m timeout 200 python scripts/latency_b1.py > gpurun_out/h2k_s2_latency_$m.txt 2>&1; echo "ROMP_KSPLIT_S2=$m :: $(grep 'ROMP(image)' gpurun_out/h2k_s2_latency_$m.txt | cut -c1-200)"
done
ROMP_KSPLIT_WG=1 timeout 600 python scripts/op_table.py 1 f16x2 > gpurun_out/h2k_optable_b1.log 2>&1
sed -n '/total serial/,$p' gpurun_out/h2k_optable_b1.log | head -16
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_b1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_b1 -o tl -- python $REPO/scripts/net_b1_loop.py 12 > $REPO/gpurun_out/b1_timeline_run.log 2>&1
echo "trace exit $? :: $(grep 'network alone' $REPO/gpurun_out/b1_timeline_run.log)"
f=$(find /tmp/rp_b1 -name "*kernel_trace.csv" | head -1)
python $REPO/scripts/timeline.py "$f" 8 $REPO/gpurun_out/b1_timeline_kernels.txt | tee $REPO/gpurun_out/b1_timeline.txt
NET_GRAPH=0 timeout 300 python $REPO/scripts/net_b1_loop.py 50 2>&1 | tail -1
