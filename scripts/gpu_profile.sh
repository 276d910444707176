#!/bin/bash
# rocprofv3 passes over bench.py (one GPU): kernel-trace stats, then FETCH_SIZE and WRITE_SIZE in
# their own passes (PMC never combined with sys/runtime traces).  Summaries -> gpurun_out/prof/.
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$REPO/gpurun_out/prof"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export PYTHONUNBUFFERED=1
# tune once OUTSIDE the profiler and reuse the table, so the traces hold the forward's kernels only
# (no autotune measuring launches); ${BENCH_ARGS} e.g. "--conv-math f32" or "--workload bev"
TUNE=/tmp/romp_tune.json
rm -f $TUNE
BENCH="python $REPO/bench.py --no-cpu-baseline --no-f32-companion --tune-file $TUNE ${BENCH_ARGS}"
$BENCH --steps 2 --warmup 1 --no-roofline > "$OUT/bench_plain.log" 2>&1
echo "tune pass exit $? :: $(grep -o '"value": [0-9.]*' "$OUT/bench_plain.log" | head -1)"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_stats -o stats -- $BENCH --steps 5 --warmup 2 > "$OUT/bench_under_rocprof.log" 2>&1
echo "stats pass exit $?"
find /tmp/rp_stats -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
# same command with the HRNet branches serialised on one stream: per-kernel durations without the
# overlap of concurrent branch kernels -- these are the ones bench.py's roofline (serial per-op HIP events) must agree with
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_stats1 -o stats -- $BENCH --steps 5 --warmup 2 --streams 0 > "$OUT/bench_under_rocprof_serial.log" 2>&1
echo "serial stats pass exit $?"
find /tmp/rp_stats1 -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_serial.csv" \;
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/rp_$C -o pmc -- $BENCH --steps 1 --warmup 1 --no-roofline --streams 0 > "$OUT/pmc_$C.log" 2>&1
  echo "pmc $C exit $?"
  f=$(find /tmp/rp_$C -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python "$REPO/scripts/summarize_pmc.py" "$f" $C > "$OUT/pmc_${C}_by_kernel.csv"
done
head -14 "$OUT/kernel_stats.csv"; head -14 "$OUT/kernel_stats_serial.csv"
tail -3 "$OUT/bench_under_rocprof.log" | cut -c1-1500
head -30 "$OUT/pmc_FETCH_SIZE_by_kernel.csv"; head -30 "$OUT/pmc_WRITE_SIZE_by_kernel.csv"
