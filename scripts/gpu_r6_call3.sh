#!/bin/bash
# Round 6, GPU call 3: packed saturation tracking in the fused-block kernels (guard cost again), conv_h2g's 2x2 form + forked parity convs,
# the early-drain A/B of conv_h2g, cross-call priming, the shard-128 prediction.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
O=gpurun_out/r06c3
timeout 900 python -m pytest tests/test_gpu_range_guard.py tests/test_gpu_resnet.py -x -q 2>&1 | tail -6
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "primed_across_calls or saturation_is_observable or fused_basic_block or forward_chunks or (test_conv_layer and _k1_)" 2>&1 | tail -6
LEGS="--no-cpu-baseline --no-parity --no-f32-companion --no-latency --no-end-to-end --no-roofline"
for r in 1 2 3; do for g in 1 0 nofused; do
  ROMP_RANGE_GUARD=$g timeout 300 python bench.py --steps 10 $LEGS 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('guard=$g run $r value', d['value'], 'step_ms', d['step_ms']['min'], d['step_ms']['median'], d['step_ms']['max'], 'clock', d['clock_mhz']['median'], 'power', d['power_w']['median'])" ; done; done > ${O}_guard_cost.txt 2>&1
cat ${O}_guard_cost.txt
for lib in "" romp_amd/libromp_hip_drainlate.so; do
  echo "== ROMP_HIP_LIB=$lib"; ROMP_HIP_LIB=$lib SWEEP_CASES=rn SWEEP_FILTER=h2g timeout 600 python scripts/conv_sweep.py 2>/dev/null | grep -A2 "^case" | grep -v "^--"
done > ${O}_h2g_drain_ab.txt 2>&1
cat ${O}_h2g_drain_ab.txt
timeout 900 python bench.py --backbone resnet50 --tune-file gpurun_out/tune_resnet50_r6b.json --no-cpu-baseline --no-end-to-end 2>${O}_resnet.err | grep '^{' | tail -1 > ${O}_bench_resnet50.json
python - <<'PY'
import json
r = json.load(open('gpurun_out/r06c3_bench_resnet50.json'))
print('resnet50', r['value'], r['config'].get('ms_per_call'), r['config'].get('maps_max_abs_vs_oracle'), r['config'].get('detections_equal'))
for k, v in sorted(r['kernel_classes'].items(), key=lambda kv: -kv[1]['ms']):
    if v['ms'] > 0.05: print('  %-44s n=%3d ms=%7.3f tflops=%7.1f gbs=%7.1f' % (k, v['launches'], v['ms'], v['tflops'], v['gbs']))
print('  serial sum', round(sum(v['ms'] for v in r['kernel_classes'].values()), 3))
PY
bash scripts/gpu_r6_shard128.sh
SHARD_ARGS="--cross-step 0" bash scripts/gpu_r6_shard128.sh 2>&1 | sed 's/^/cross-step 0: /'
