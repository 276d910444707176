#!/bin/bash
# Round 5 evidence run: the Winograd step micro-benchmark, the default bench line, the rocprofv3 passes of the default job
# (scripts/gpu_profile.sh: kernel stats concurrent + serial, FETCH / WRITE per op, utilisation counters), ONE counter pass with the
# branch streams ON and the timeline of the concurrent job.
REPO="$(cd "$(dirname "$0")/.." && pwd)"
cd "$REPO"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
if [ -x scripts/micro/_bin/wino_step_bench ]; then       # (first session of the round; the binary is built by hand from scripts/micro/wino_step_bench.hip)
  timeout 120 scripts/micro/_bin/wino_step_bench > gpurun_out/r05_wino_step_bench.txt 2>&1; echo "== wino_step_bench exit $?"; cat gpurun_out/r05_wino_step_bench.txt
fi
timeout 900 python bench.py > gpurun_out/r05_bench.log 2>&1; echo "== bench: exit $?"
grep '^{' gpurun_out/r05_bench.log | tail -1 > gpurun_out/r05_bench.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/r05_bench.json')); r = d['roofline']; c = d['config']
print('value', d['value'], 'ms/step', d['ms_per_step'], 'roof', r['kernel'], r['frac'], 'cpu', d['cpu_baseline']['value'], 'e2e', d['end_to_end']['value'], 'lat', d['single_image_latency'].get('ms_per_frame'))
print('parity: images', len(c['images_compared']), 'maps', c['maps_max_abs_vs_oracle'], 'detections_equal', c['detections_equal'], 'persons', c.get('persons_compared'), 'mesh', c.get('mesh_max_abs_vs_oracle'), c.get('mesh_images_compared'))
PY
PROF_TAG=_r5 bash scripts/gpu_profile.sh 2>&1 | tail -25
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --no-f32-companion --no-parity --no-end-to-end --no-latency --no-roofline --global-batch 64"
# (SQ_INST_CYCLES_VMEM has no gfx950 definition in rocprofiler-sdk's counter_defs.yaml: SQ_ACTIVE_INST_VMEM is the issue-cycle counter that has)
CS="SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
rm -rf /tmp/rp_conc
timeout 900 rocprofv3 --kernel-trace --pmc $CS --output-format csv -d /tmp/rp_conc -o pmc -- $BENCH --steps 2 --warmup 1 > $REPO/gpurun_out/r05_pmc_concurrent_run.log 2>&1
echo "== concurrent pmc pass exit $?"
cc=$(find /tmp/rp_conc -name "*counter_collection.csv" | head -1); kt=$(find /tmp/rp_conc -name "*kernel_trace.csv" | head -1)
[ -n "$cc" ] && python $REPO/scripts/pmc_concurrent.py "$cc" "$kt" $CS | tee $REPO/gpurun_out/r05_pmc_concurrent.csv | head -40
rm -rf /tmp/rp_tl
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_tl -o tl -- $BENCH --global-batch 256 --steps 2 --warmup 1 > $REPO/gpurun_out/r05_trace_run.log 2>&1
echo "== batch trace exit $? :: $(grep -o '"value": [0-9.]*' $REPO/gpurun_out/r05_trace_run.log | head -1)"
f=$(find /tmp/rp_tl -name "*kernel_trace.csv" | head -1)
python $REPO/scripts/timeline.py "$f" 4 | tee $REPO/gpurun_out/r05_timeline_b32.txt | head -12
# the last forward kernel by kernel, and the engine clock / socket power the job sustains (second session)
python $REPO/scripts/timeline.py "$f" 4 $REPO/gpurun_out/r05_timeline_kernels.txt > /dev/null
cd $REPO && bash scripts/gpu_clocks.sh
