#!/bin/bash
# Round 4, final library: smoke, the default bench line (committed table, CPU baseline) and the secondary lines, the single-image
# latency script, then rocprofv3 kernel traces of the default job (stats + timeline) and of a single-image forward (timeline).
REPO="$(cd "$(dirname "$0")/.." && pwd)"
cd "$REPO"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "== smoke: exit $? :: $(tail -n 1 gpurun_out/smoke.log)"
timeout 900 python bench.py > gpurun_out/bench_default.log 2>&1; echo "== bench: exit $?"
grep '^{' gpurun_out/bench_default.log | tail -1 > gpurun_out/bench_default.json
for W in bev resnet50 b128 smpl; do
  case $W in
    bev) A="--workload bev";; resnet50) A="--backbone resnet50";; b128) A="--batch 128";; smpl) A="--workload smpl";;
  esac
  timeout 600 python bench.py $A --no-f32-companion --no-latency > gpurun_out/bench_$W.log 2>&1; echo "== bench $W: exit $?"
  grep '^{' gpurun_out/bench_$W.log | tail -1 > gpurun_out/bench_$W.json
done
python - <<'PY'
import json
for w in ('default', 'bev', 'resnet50', 'b128', 'smpl'):
    try:
        d = json.load(open('gpurun_out/bench_%s.json' % w))
        r = d.get('roofline', {})
        print(w, 'value', d['value'], d['unit'], 'ms/step', d['ms_per_step'], 'roof', r.get('kernel'), r.get('bound'), r.get('achieved'), r.get('frac'), 'traffic', r.get('traffic'),
              'cpu', d.get('cpu_baseline', {}).get('value'), d.get('cpu_baseline', {}).get('kind'), 'lat', d.get('single_image_latency', {}).get('ms_per_frame'),
              'e2e', d.get('end_to_end', {}).get('value'), 'f32', d.get('f32_mfma_companion', {}).get('value'), d['config'].get('variant_table'))
    except Exception as e:
        print(w, 'parse failed', e)
PY
timeout 300 python scripts/latency_b1.py > gpurun_out/latency_b1.txt 2>&1; grep -E "ROMP\(image\)|stages" gpurun_out/latency_b1.txt | cut -c1-260
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --no-f32-companion --no-parity --no-end-to-end --no-latency --no-roofline --global-batch 256"
rm -rf /tmp/rp_tl
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_tl -o tl -- $BENCH --steps 2 --warmup 1 > $REPO/gpurun_out/final_trace_run.log 2>&1
echo "batch trace exit $? :: $(grep -o '"value": [0-9.]*' $REPO/gpurun_out/final_trace_run.log | head -1)"
grep '^{' $REPO/gpurun_out/final_trace_run.log | tail -1 > $REPO/gpurun_out/bench_under_rocprof.json
find /tmp/rp_tl -name "*kernel_stats.csv" -exec cp {} $REPO/gpurun_out/final_kernel_stats.csv \;
f=$(find /tmp/rp_tl -name "*kernel_trace.csv" | head -1)
python $REPO/scripts/timeline.py "$f" 4 | tee $REPO/gpurun_out/final_timeline_b32.txt
rm -rf /tmp/rp_b1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_b1 -o tl -- python $REPO/scripts/net_b1_loop.py 12 > $REPO/gpurun_out/final_b1_trace_run.log 2>&1
echo "b1 trace exit $? :: $(grep 'network alone' $REPO/gpurun_out/final_b1_trace_run.log)"
f=$(find /tmp/rp_b1 -name "*kernel_trace.csv" | head -1)
python $REPO/scripts/timeline.py "$f" 8 $REPO/gpurun_out/final_b1_timeline_kernels.txt | tee $REPO/gpurun_out/final_b1_timeline.txt
