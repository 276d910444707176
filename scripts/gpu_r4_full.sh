#!/bin/bash
# Round 4: the whole `pytest -m gpu` suite as the driver runs it, smoke, the default bench line (committed table, CPU baseline),
# and the secondary lines with freshly measured variant tables (copied into romp_amd/tune/ afterwards).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q --tb=short -x --timeout 900 --durations=12 > gpurun_out/full_tests.log 2>&1
echo "== pytest -m gpu: exit $? :: $(tail -n 1 gpurun_out/full_tests.log)"; grep -E "^[0-9.]+s (call|setup)" gpurun_out/full_tests.log | head -12
grep -hE "FAILED|Error" gpurun_out/full_tests.log | head -20
timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "== smoke: exit $? :: $(tail -n 1 gpurun_out/smoke.log)"
timeout 900 python bench.py > gpurun_out/bench_default.log 2>&1; echo "== bench: exit $?"
grep '^{' gpurun_out/bench_default.log | tail -1 > gpurun_out/bench_default.json
for W in bev resnet50 b128 smpl; do
  case $W in
    bev) A="--workload bev";; resnet50) A="--backbone resnet50";; b128) A="--batch 128";; smpl) A="--workload smpl";;
  esac

  T=""
  timeout 900 python bench.py $A $T --no-f32-companion --no-latency > gpurun_out/bench_$W.log 2>&1; echo "== bench $W: exit $?"
  grep '^{' gpurun_out/bench_$W.log | tail -1 > gpurun_out/bench_$W.json
done
python - <<'PY'
import json
for w in ('default', 'bev', 'resnet50', 'b128', 'smpl'):
    try:
        d = json.load(open('gpurun_out/bench_%s.json' % w))
        r = d.get('roofline', {})
        print(w, 'value', d['value'], d['unit'], 'ms/step', d['ms_per_step'], 'roof', r.get('kernel'), r.get('bound'), r.get('achieved'), r.get('frac'),
              'cpu', d.get('cpu_baseline', {}).get('value'), d.get('cpu_baseline', {}).get('kind'), 'lat', d.get('single_image_latency', {}).get('ms_per_frame'),
              'e2e', d.get('end_to_end', {}).get('value'), 'f32', d.get('f32_mfma_companion', {}).get('value'))
    except Exception as e:
        print(w, 'parse failed', e)
PY
