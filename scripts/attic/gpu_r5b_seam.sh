#!/bin/bash
# Round 5, second session: the pipelined seam kernel and the folded downsample (csrc/conv_h2x.hip) -- parity, then same-box A/B
# of the default job: ROMP_SEAM_DS=0 (downsample a launch of its own) vs the default, two runs per arm.
# Usage: gpurun --timeout 900 -- 'bash scripts/gpu_r5b_seam.sh'
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "seam1x1 or net_golden or net_bf16x3_parity or benchmark_batch" 2>&1 | tee gpurun_out/r5b_seam_tests.log | tail -8
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-f32-companion --no-end-to-end --no-latency"
for run in 1 2; do
  for arm in 0 1; do
    ROMP_SEAM_DS=$arm timeout 300 $B 2>gpurun_out/r5b_bench_ds${arm}_$run.err | grep '^{' | tail -1 > gpurun_out/r5b_bench_ds${arm}_$run.json
    python - gpurun_out/r5b_bench_ds${arm}_$run.json $arm <<'PY'
import json, sys
r = json.load(open(sys.argv[1]))
kc = r['kernel_classes']
seam = {k: (v['launches'], round(v['ms'], 4), round(v['gbs'])) for k, v in kc.items() if 'seam' in k or k.startswith('conv_h2_k1s1_mt2_nt2_tw32')}
print('SEAM_DS=%s  %.1f images/s  ms/call %s  net_ms_serial %.3f  parity %s %s  %s' % (sys.argv[2], r['value'], r['config'].get('ms_per_call'), r['roofline']['net_ms_per_batch'],
      r['config'].get('maps_max_abs_vs_oracle'), r['config'].get('detections_equal'), seam))
PY
  done
done
