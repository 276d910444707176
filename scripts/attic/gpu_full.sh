#!/bin/bash
# One gpurun call: the whole `pytest -m gpu` suite (as the driver runs it), smoke, and the bench lines
# (ROMP headline in both conv-math modes, BEV).  Logs land in gpurun_out/.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q --tb=short -x --timeout 900 > gpurun_out/full_tests.log 2>&1
echo "== pytest -m gpu: exit $? :: $(tail -n 1 gpurun_out/full_tests.log)"
grep -hE "FAILED|Error" gpurun_out/full_tests.log | head -20
timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "== smoke: exit $? :: $(tail -n 1 gpurun_out/smoke.log)"
timeout 900 python bench.py > gpurun_out/bench.log 2>&1; echo "== bench: exit $?"
python - <<'EOF'
import json
for f in ('gpurun_out/bench.log',):
    for line in open(f):
        if line.startswith('{'):
            d = json.loads(line)
            r = d.get('roofline', {})
            print('value', d['value'], 'ms/step', d['ms_per_step'], 'roof', r.get('kernel'), r.get('achieved'), r.get('peak'), r.get('frac'),
                  'cpu', d.get('cpu_baseline', {}).get('value'))
EOF
timeout 600 python bench.py --conv-math f32 --no-cpu-baseline > gpurun_out/bench_f32.log 2>&1
echo "== bench f32: exit $? :: $(grep -o '"value": [0-9.]*' gpurun_out/bench_f32.log | head -1)"
timeout 300 python bench.py --workload smpl > gpurun_out/bench_smpl.log 2>&1
echo "== bench smpl: exit $? :: $(tail -c 1500 gpurun_out/bench_smpl.log)"
timeout 600 python bench.py --workload bev > gpurun_out/bench_bev.log 2>&1
echo "== bench bev: exit $? :: $(tail -c 700 gpurun_out/bench_bev.log)"
