#!/bin/bash
# rocprofv3 passes over bench.py (one GPU): kernel-trace stats, then FETCH_SIZE / WRITE_SIZE / utilisation counters each in
# their own pass (PMC never combined with sys/runtime traces).  Summaries -> gpurun_out/prof/ (copy to profiles/<round>_*).
# The profiled command is bench.py's default workload cut to 2 forward calls of 32 images per step (--global-batch 64): the
# same kernels and variant table as the 1024-image job, a trace of a manageable size.
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$REPO/gpurun_out/prof${PROF_TAG}"                # PROF_TAG: e.g. _bev, _resnet50, _b128 for the secondary workloads' passes
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export PYTHONUNBUFFERED=1
# ONE variant table for every pass: the committed one of this configuration (romp_amd/tune/, bench.py's default) when there is
# one, else a table tuned once OUTSIDE the profiler (${TUNE_FILE}), so the traces hold the forward's kernels only (no autotune
# measuring launches); ${BENCH_ARGS} e.g. "--conv-math f32" or "--workload bev"
TUNE_ARG=""
if [ -n "${TUNE_FILE}" ]; then rm -f "$TUNE_FILE"; TUNE_ARG="--tune-file $TUNE_FILE"; fi
BENCH="python $REPO/bench.py --no-cpu-baseline --no-f32-companion --no-parity --no-end-to-end --no-latency --global-batch 64 $TUNE_ARG ${BENCH_ARGS}"
$BENCH --steps 2 --warmup 1 --no-roofline --dump-op-kernels "$OUT/op_kernels.json" > "$OUT/bench_plain.log" 2>&1
grep -o '"variant_table": "[^"]*"' "$OUT/bench_plain.log" | head -1
echo "tune pass exit $? :: $(grep -o '"value": [0-9.]*' "$OUT/bench_plain.log" | head -1)"
if [ "${PROFILE_ONLY}" != "pmc" ]; then
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_stats -o stats -- $BENCH --steps 5 --warmup 2 > "$OUT/bench_under_rocprof.log" 2>&1
echo "stats pass exit $?"
find /tmp/rp_stats -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
# same command with the HRNet branches serialised on one stream: per-kernel durations without the
# overlap of concurrent branch kernels -- these are the ones bench.py's roofline (serial per-op HIP events) must agree with
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_stats1 -o stats -- $BENCH --steps 5 --warmup 2 --streams 0 > "$OUT/bench_under_rocprof_serial.log" 2>&1
echo "serial stats pass exit $?"
find /tmp/rp_stats1 -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_serial.csv" \;
fi
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/rp_$C -o pmc -- $BENCH --steps 1 --warmup 1 --no-roofline --streams 0 > "$OUT/pmc_$C.log" 2>&1
  echo "pmc $C exit $?"
  f=$(find /tmp/rp_$C -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python "$REPO/scripts/summarize_pmc.py" "$f" $C > "$OUT/pmc_${C}_by_kernel.csv" && cp "$f" /tmp/pmc_$C.csv
done
# per OP INDEX (what bench.py's roofline.traffic reads): the two passes aligned with the op list of the same variant table
[ -f /tmp/pmc_FETCH_SIZE.csv ] && [ -f /tmp/pmc_WRITE_SIZE.csv ] && python "$REPO/scripts/pmc_by_op.py" "$OUT/op_kernels.json" /tmp/pmc_FETCH_SIZE.csv /tmp/pmc_WRITE_SIZE.csv "$OUT/pmc_traffic_by_op.json" | tee "$OUT/pmc_traffic_by_op.log"
[ "${PROFILE_ONLY}" = "pmc" ] && { cat "$OUT/pmc_traffic_by_op.log" 2>/dev/null; exit 0; }
for CS in "MfmaUtil LdsUtil LdsBankConflict" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  TAG=$(echo $CS | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/rp_multi
  timeout 300 rocprofv3 --kernel-trace --pmc $CS --output-format csv -d /tmp/rp_multi -o pmc -- $BENCH --steps 1 --warmup 1 --no-roofline --streams 0 > "$OUT/pmc_$TAG.log" 2>&1
  echo "pmc $TAG exit $?"
  f=$(find /tmp/rp_multi -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" $CS > "$OUT/pmc_${TAG}_by_kernel.csv" <<'PY'
import csv, re, sys
from collections import defaultdict
path, counters = sys.argv[1], sys.argv[2:]
agg = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
with open(path) as f:
    for row in csv.DictReader(f):
        name = re.sub(r'\(.*$', '', row['Kernel_Name']).replace('void romp::', '').replace('romp::', '')
        agg[name][row['Counter_Name']] += float(row['Counter_Value'])
        if row['Counter_Name'] == counters[0]:
            cnt[name] += 1
print('kernel,dispatches,' + ','.join(c + '_mean' for c in counters))
for k in sorted(agg, key=lambda k: -agg[k][counters[0]]):
    print('"%s",%d,' % (k, cnt[k]) + ','.join('%.1f' % (agg[k][c] / max(cnt[k], 1)) for c in counters))
PY
done
grep '^{' "$OUT/bench_under_rocprof.log" | tail -1 > "$OUT/bench_under_rocprof.json"
grep '^{' "$OUT/bench_under_rocprof_serial.log" | tail -1 > "$OUT/bench_under_rocprof_serial.json"
# the other workloads: kernel-trace stats only
if [ -z "${BENCH_ARGS}" ]; then
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_smpl -o stats -- python $REPO/bench.py --workload smpl --no-cpu-baseline > "$OUT/bench_smpl_under_rocprof.log" 2>&1
  find /tmp/rp_smpl -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_smpl.csv" \;
  rm -f /tmp/romp_tune_bev.json
  python $REPO/bench.py --workload bev --no-cpu-baseline --no-roofline --steps 2 --warmup 1 --tune-file /tmp/romp_tune_bev.json > "$OUT/bench_bev_plain.log" 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_bev -o stats -- python $REPO/bench.py --workload bev --no-cpu-baseline --steps 5 --warmup 2 --tune-file /tmp/romp_tune_bev.json > "$OUT/bench_bev_under_rocprof.log" 2>&1
  find /tmp/rp_bev -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_bev.csv" \;
  grep '^{' "$OUT/bench_bev_under_rocprof.log" | tail -1 > "$OUT/bench_bev_under_rocprof.json"
  head -8 "$OUT/kernel_stats_smpl.csv"; head -12 "$OUT/kernel_stats_bev.csv"
fi
head -14 "$OUT/kernel_stats.csv"; head -14 "$OUT/kernel_stats_serial.csv"
cut -c1-600 "$OUT/bench_under_rocprof.json"
head -16 "$OUT/pmc_FETCH_SIZE_by_kernel.csv"; head -16 "$OUT/pmc_WRITE_SIZE_by_kernel.csv"
head -12 "$OUT"/pmc_MfmaUtil*_by_kernel.csv; head -12 "$OUT"/pmc_SQ_WAVE*_by_kernel.csv
