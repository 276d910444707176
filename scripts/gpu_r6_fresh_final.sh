#!/bin/bash
# the final build as the driver runs it: the default line FIRST on a fresh lease, then twice more on the warm box
cd "$(dirname "$0")/.."
O=gpurun_out/r06_fresh_final
for t in fresh warm1 warm2; do
  timeout 600 python bench.py 2>$O.$t.err | grep '^{' | tail -1 > $O.$t.json
done
python - <<'PY'
import json
for tag in ('fresh', 'warm1', 'warm2'):
    r = json.load(open('gpurun_out/r06_fresh_final.%s.json' % tag))
    print(tag, r['value'], 'ms/step', r['ms_per_step'], 'steps', r['step_ms']['min'], r['step_ms']['median'], r['step_ms']['max'], 'preheat', r['preheat_s'], r['preheat_step_ms'],
          'clock', r.get('clock_mhz'), 'power', r.get('power_w'), 'roof', r['roofline']['frac'], r['roofline']['avg_launch_ms'])
PY
