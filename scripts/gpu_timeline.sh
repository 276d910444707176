#!/bin/bash
# kernel timeline of the default workload (rocprofv3 --kernel-trace, no counters): where a forward's wall time goes
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$REPO/gpurun_out"
cd /tmp && export TMPDIR=/tmp PYTHONUNBUFFERED=1
rm -rf /tmp/rp_tl
BENCH="python $REPO/bench.py --no-cpu-baseline --no-f32-companion --no-parity --no-end-to-end --no-latency --no-roofline --global-batch 256 ${BENCH_ARGS}"
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_tl -o tl -- $BENCH --steps 2 --warmup 1 > "$OUT/timeline_run${TAG}.log" 2>&1
echo "trace pass exit $? :: $(grep -o '"value": [0-9.]*' "$OUT/timeline_run${TAG}.log" | head -1)"
f=$(find /tmp/rp_tl -name "*kernel_trace.csv" | head -1)
python "$REPO/scripts/timeline.py" "$f" 4 | tee "$OUT/timeline${TAG}.txt"
