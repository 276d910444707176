#!/bin/bash
# Round 5, second session: transition1.1 on conv_h2s in the committed B = 32 table -- the traffic passes per op re-taken (the table
# changed one kernel), two default-job runs, and the net parity tests that run the committed table.
REPO="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$REPO"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "committed_table or net_golden or benchmark_batch" 2>&1 | tail -3
PROFILE_ONLY=pmc PROF_TAG=_r5 bash scripts/gpu_profile.sh 2>&1 | tail -6
cd "$REPO"
for run in 1 2; do
  timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-f32-companion --no-end-to-end --no-latency 2>/dev/null | grep '^{' | tail -1 > gpurun_out/r5b10_bench_$run.json
  python - gpurun_out/r5b10_bench_$run.json <<'PY'
import json, sys
r = json.load(open(sys.argv[1])); kc = r['kernel_classes']
print('%.1f images/s  ms/call %s  serial %.3f  parity %.2e %s  %s' % (r['value'], r['config']['ms_per_call'], r['roofline']['net_ms_per_batch'], r['config']['maps_max_abs_vs_oracle'],
      r['config']['detections_equal'], {k: (v['launches'], round(v['ms'], 4)) for k, v in kc.items() if 'k3s2' in k}))
PY
done
