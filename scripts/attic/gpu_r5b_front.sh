#!/bin/bash
# Round 5, second session, call 3: conv_h2s with the next stage's DMA pieces front-loaded onto the first four tap ends -- same-box
# sweep of every stride-2 shape against the spread schedule of rounds 3-4 (build_ab/libromp_hip_spread.so = -DROMP_H2S_FRONT=0),
# parity of the stride-2 variants, then the default job on both libraries.
# Usage: gpurun --timeout 1200 -- 'bash scripts/gpu_r5b_front.sh'
REPO="$(cd "$(dirname "$0")/.." && pwd)"
cd "$REPO"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
ALT=$REPO/romp_amd/build_ab/libromp_hip_spread.so
SWEEP_CASES=s2 SWEEP_FILTER=h2s,h2d_ SWEEP_CHECK=1 timeout 300 python scripts/conv_sweep.py > gpurun_out/r5b3_sweep_front.txt 2>&1; echo "sweep front exit $?"
ROMP_HIP_LIB=$ALT SWEEP_CASES=s2 SWEEP_FILTER=h2s,h2d_ timeout 300 python scripts/conv_sweep.py > gpurun_out/r5b3_sweep_spread.txt 2>&1; echo "sweep spread exit $?"
python - <<'PY'
import re
def parse(path):
    out, case = {}, None
    for l in open(path):
        m = re.match(r'^case (\(.*?\))', l) or re.match(r'^== (\(.*?\))', l)
        if l.startswith('case') or l.startswith('=='):
            case = l.strip()
        m = re.match(r'\s*(conv_\S+)\s+.*?([0-9.]+) us', l)
        if m and case:
            out.setdefault(case, {})[m.group(1)] = float(m.group(2))
    return out
a, b = parse('gpurun_out/r5b3_sweep_front.txt'), parse('gpurun_out/r5b3_sweep_spread.txt')
for case in a:
    for k in sorted(a[case], key=lambda k: a[case][k])[:4]:
        print('%-46s %-40s front %7.1f  spread %7.1f' % (case[:46], k, a[case][k], b.get(case, {}).get(k, float('nan'))))
PY
head -30 gpurun_out/r5b3_sweep_front.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "conv_layer or net_golden or benchmark_batch" 2>&1 | tee gpurun_out/r5b3_tests.log | tail -4
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-f32-companion --no-end-to-end --no-latency"
report() {
python - "$1" "$2" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1]))
except Exception as e:
    print(sys.argv[2], 'FAILED', e); sys.exit(0)
kc = r['kernel_classes']
pick = {k.replace('conv_h2s_k3s2_', 's2_'): (v['launches'], round(v['ms'], 4)) for k, v in kc.items() if 'h2s' in k or 'h2d_k3s2' in k}
print('%-8s %.1f images/s  ms/call %s  net_ms_serial %.3f  parity %.2e %s  %s' % (sys.argv[2], r['value'], r['config'].get('ms_per_call'), r['roofline']['net_ms_per_batch'],
      r['config'].get('maps_max_abs_vs_oracle'), r['config'].get('detections_equal'), pick))
PY
}
for run in 1 2; do
  timeout 300 $B 2>gpurun_out/r5b3_front_$run.err | grep '^{' | tail -1 > gpurun_out/r5b3_front_$run.json; report gpurun_out/r5b3_front_$run.json front
  ROMP_HIP_LIB=$ALT timeout 300 $B 2>gpurun_out/r5b3_spread_$run.err | grep '^{' | tail -1 > gpurun_out/r5b3_spread_$run.json; report gpurun_out/r5b3_spread_$run.json spread
done
