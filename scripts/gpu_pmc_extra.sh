#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes for the ResNet-50 and BEV workloads (the default workload: gpu_profile.sh)
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$REPO/gpurun_out/prof_extra"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export PYTHONUNBUFFERED=1
for W in resnet50 bev; do
  if [ $W = resnet50 ]; then WA="--backbone resnet50"; else WA="--workload bev"; fi
  TUNE=/tmp/romp_tune_$W.json; rm -f $TUNE
  BENCH="python $REPO/bench.py --no-cpu-baseline --no-f32-companion --no-parity --no-end-to-end --no-latency --global-batch 64 --tune-file $TUNE $WA"
  $BENCH --steps 2 --warmup 1 --no-roofline > "$OUT/${W}_plain.log" 2>&1
  echo "$W tune pass exit $?"
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/rpx_$C
    timeout 900 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/rpx_$C -o pmc -- $BENCH --steps 1 --warmup 1 --no-roofline --streams 0 > "$OUT/${W}_pmc_$C.log" 2>&1
    echo "$W pmc $C exit $?"
    f=$(find /tmp/rpx_$C -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python "$REPO/scripts/summarize_pmc.py" "$f" $C > "$OUT/${W}_pmc_${C}_by_kernel.csv"
  done
  head -4 "$OUT/${W}_pmc_FETCH_SIZE_by_kernel.csv"
  # the bench line of the SAME variant table, looking its traffic up in the passes just made
  cp "$OUT/${W}_pmc_FETCH_SIZE_by_kernel.csv" "$REPO/profiles/r02_${W}_pmc_FETCH_SIZE_by_kernel.csv"
  cp "$OUT/${W}_pmc_WRITE_SIZE_by_kernel.csv" "$REPO/profiles/r02_${W}_pmc_WRITE_SIZE_by_kernel.csv"
  python $REPO/bench.py --tune-file $TUNE $WA --no-f32-companion --no-latency 2>/dev/null | grep '^{' | tail -1 > "$OUT/bench_$W.json"
  python -c "
import json; r=json.load(open('$OUT/bench_$W.json')); ro=r['roofline']
print('$W', r['value'], ro['kernel'], ro['bound'], ro['frac'], 'traffic', ro.get('traffic'), 'x', ro.get('traffic_over_algorithmic'))"
done
