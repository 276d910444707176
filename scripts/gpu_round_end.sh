#!/bin/bash
# THE way a round ends: on the FINAL commit, after the last code change -- smoke, the FULL GPU suite exactly as the
# driver runs it (-x -q; a second pass without -x only if it failed, to see everything that is red), then the default bench line.
# Usage: gpurun --timeout 1500 -- 'bash scripts/gpu_round_end.sh'
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1200 python -m pytest tests -m gpu -x -q --durations=12 2>&1 | tee gpurun_out/round_end_tests.log | tail -6
if ! grep -q " passed" gpurun_out/round_end_tests.log || grep -q " failed" gpurun_out/round_end_tests.log; then
    timeout 900 python -m pytest tests -m gpu -q 2>&1 | tee gpurun_out/round_end_tests_all.log | tail -30
fi
timeout 600 python bench.py 2>gpurun_out/round_end_bench.err | grep '^{' | tail -1 > gpurun_out/round_end_bench.json
python - <<'PY'
import json
r = json.load(open('gpurun_out/round_end_bench.json'))
print(r['value'], r['unit'], 'ms/call', r['config'].get('ms_per_call'), 'roofline', r['roofline']['kernel'], r['roofline']['frac'],
      'cpu', r['cpu_baseline']['value'], 'parity', r['config'].get('maps_max_abs_vs_oracle'), r['config'].get('mesh_max_abs_vs_oracle'), r['config'].get('detections_equal'))
PY
