#!/bin/bash
# Round 6 evidence run on the FINAL library and tables: the rocprofv3 passes of the default job (scripts/gpu_profile.sh: kernel stats
# concurrent + serial, FETCH_SIZE / WRITE_SIZE per op, utilisation counters, SMPL / BEV stats), the same traffic passes for the
# secondary configurations (BEV, ResNet-50, B = 128), the timeline of the concurrent job, then every bench line -- taken AFTER the
# traffic files are in profiles/ so that each line's roofline.traffic is this build's.
REPO="$(cd "$(dirname "$0")/.." && pwd)"
cd "$REPO"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
PROF_TAG=_r6 bash scripts/gpu_profile.sh 2>&1 | tail -30
P=gpurun_out/prof_r6
cp $P/pmc_traffic_by_op.json profiles/r06_pmc_traffic_by_op.json 2>/dev/null
bash scripts/gpu_pmc_secondary.sh
for W in bev resnet50 b128; do cp gpurun_out/prof_$W/pmc_traffic_by_op.json profiles/r06_${W}_pmc_traffic_by_op.json 2>/dev/null; done
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --no-f32-companion --no-parity --no-end-to-end --no-latency --no-roofline"
rm -rf /tmp/rp_tl
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_tl -o tl -- $BENCH --global-batch 256 --steps 2 --warmup 1 > $REPO/gpurun_out/r06_trace_run.log 2>&1
echo "== batch trace exit $? :: $(grep -o '"value": [0-9.]*' $REPO/gpurun_out/r06_trace_run.log | head -1)"
f=$(find /tmp/rp_tl -name "*kernel_trace.csv" | head -1)
python $REPO/scripts/timeline.py "$f" 4 | tee $REPO/gpurun_out/r06_timeline_b32.txt | head -12
python $REPO/scripts/timeline.py "$f" 4 $REPO/gpurun_out/r06_timeline_kernels.txt > /dev/null
cd $REPO
timeout 900 python bench.py 2>gpurun_out/r06_bench.err | grep '^{' | tail -1 > gpurun_out/r06_bench.json
bash scripts/gpu_bench_lines.sh > gpurun_out/r06_bench_lines.log 2>&1
python - <<'PY'
import json
for w in ('', '_bev', '_resnet50', '_b128', '_smpl'):
    try:
        d = json.load(open('gpurun_out/%s.json' % ('r06_bench' if not w else 'bench' + w))); r = d.get('roofline', {}); c = d['config']
        print(w or 'default', 'value', d['value'], d['unit'], 'roof', r.get('kernel'), r.get('bound'), r.get('frac'), 'traffic', r.get('traffic'), r.get('traffic_over_algorithmic'),
              'cpu', d.get('cpu_baseline', {}).get('value'), d.get('cpu_baseline', {}).get('kind'),
              'parity', c.get('maps_max_abs_vs_oracle'), c.get('detections_equal'), c.get('mesh_max_abs_vs_oracle'), 'clock', d.get('clock_mhz'), 'power', d.get('power_w'))
    except Exception as e:
        print(w, 'parse failed', e)
PY
