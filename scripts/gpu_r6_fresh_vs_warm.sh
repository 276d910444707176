#!/bin/bash
# the driver's bench is the FIRST thing a fresh lease runs; the round-end script runs it after minutes of pytest.
# This runs the default line both ways on one box: (1) first thing on the fresh lease, exactly as the driver does, (2) after the
# full GPU suite, (3) once more right after -- plus a dump of the sensor files the in-process clock / power sampler can read.
# Usage: gpurun --timeout 1500 -- 'bash scripts/gpu_r6_fresh_vs_warm.sh'  -> gpurun_out/r06_fresh_vs_warm.*
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
O=gpurun_out/r06_fresh_vs_warm
timeout 600 python bench.py 2>$O.fresh.err | grep '^{' | tail -1 > $O.fresh.json
{ echo "== sensors"; for d in /sys/bus/pci/devices/*/hwmon/hwmon*; do echo "$d: $(ls $d | tr '\n' ' ')"; for f in freq1_input freq1_label power1_average power1_input power1_label power1_cap; do [ -e $d/$f ] && echo "   $f = $(cat $d/$f 2>&1)"; done; done; rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Power" | head -4; } > $O.sensors.txt 2>&1
timeout 300 python -m pytest tests/test_gpu_range_guard.py -x -q 2>&1 | tee $O.guard_tests.log | tail -15
timeout 1200 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_range_guard.py 2>&1 | tee $O.tests.log | tail -6
timeout 600 python bench.py 2>$O.warm.err | grep '^{' | tail -1 > $O.warm.json
timeout 600 python bench.py --no-cpu-baseline --no-parity --no-f32-companion --no-latency --no-end-to-end 2>>$O.warm.err | grep '^{' | tail -1 > $O.warm2.json
# what the range guard costs: interleaved, two runs per arm (ROMP_RANGE_GUARD is a measurement switch, not a product option)
for r in 1 2; do for g in 1 0 nofused; do
  ROMP_RANGE_GUARD=$g timeout 300 python bench.py --steps 10 --no-cpu-baseline --no-parity --no-f32-companion --no-latency --no-end-to-end --no-roofline 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('guard=$g run $r value', d['value'], 'step_ms', d['step_ms']['min'], d['step_ms']['median'], d['step_ms']['max'], 'clock', d.get('clock_mhz'), 'power', d.get('power_w'))" ; done; done > $O.guard_cost.txt 2>&1
python - <<'PY'
import json
for tag in ('fresh', 'warm', 'warm2'):
    try:
        r = json.load(open('gpurun_out/r06_fresh_vs_warm.%s.json' % tag))
    except Exception as e:
        print(tag, 'no line:', e); continue
    print(tag, r['value'], 'ms/step', r['ms_per_step'], 'steps', r['step_ms']['all'], 'preheat', r['preheat_s'], r['preheat_step_ms'], 'clock', r.get('clock_mhz'), 'power', r.get('power_w'),
          r.get('sensor_source'), 'roof', r.get('roofline', {}).get('frac'), r.get('roofline', {}).get('avg_launch_ms'))
PY
cat $O.guard_cost.txt
