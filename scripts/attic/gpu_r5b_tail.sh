#!/bin/bash
# Round 5, second session, call 2: seam tail form + counted end-of-tile wait of the fused BasicBlock kernels -- parity, same-box A/Bs
# (ROMP_SEAM_TAIL=0 | the full-drain library build_ab/libromp_hip_drain0.so | ROMP_PIPE_NETS=2), then the kernel-by-kernel timeline.
# Usage: gpurun --timeout 1200 -- 'bash scripts/gpu_r5b_tail.sh'
REPO="$(cd "$(dirname "$0")/.." && pwd)"
cd "$REPO"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "seam1x1 or net_golden or fused_basic_block or benchmark_batch or committed_table" 2>&1 | tee gpurun_out/r5b_tail_tests.log | tail -6
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-f32-companion --no-end-to-end --no-latency"
report() {
python - "$1" "$2" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1]))
except Exception as e:
    print(sys.argv[2], 'FAILED', e); sys.exit(0)
kc = r['kernel_classes']
pick = {k: (v['launches'], round(v['ms'], 4)) for k, v in kc.items() if 'seam' in k or 'bblock' in k or k.startswith('conv_h2_k1s1_mt1_nt2')}
print('%-14s %.1f images/s  ms/call %s  net_ms_serial %.3f  parity %.2e %s  %s' % (sys.argv[2], r['value'], r['config'].get('ms_per_call'), r['roofline']['net_ms_per_batch'],
      r['config'].get('maps_max_abs_vs_oracle'), r['config'].get('detections_equal'), pick))
PY
}
for run in 1 2; do
  timeout 300 $B 2>gpurun_out/r5b2_default_$run.err | grep '^{' | tail -1 > gpurun_out/r5b2_default_$run.json; report gpurun_out/r5b2_default_$run.json default
  ROMP_SEAM_TAIL=0 timeout 300 $B 2>gpurun_out/r5b2_tail0_$run.err | grep '^{' | tail -1 > gpurun_out/r5b2_tail0_$run.json; report gpurun_out/r5b2_tail0_$run.json tail0
  ROMP_HIP_LIB=$REPO/romp_amd/build_ab/libromp_hip_drain0.so timeout 300 $B 2>gpurun_out/r5b2_drain0_$run.err | grep '^{' | tail -1 > gpurun_out/r5b2_drain0_$run.json; report gpurun_out/r5b2_drain0_$run.json drain0
done
ROMP_PIPE_NETS=2 timeout 300 $B 2>gpurun_out/r5b2_pipe2.err | grep '^{' | tail -1 > gpurun_out/r5b2_pipe2.json; report gpurun_out/r5b2_pipe2.json pipe_nets2
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --no-f32-companion --no-parity --no-end-to-end --no-latency --no-roofline"
rm -rf /tmp/rp_tl
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_tl -o tl -- $BENCH --global-batch 256 --steps 2 --warmup 1 > $REPO/gpurun_out/r5b2_trace_run.log 2>&1
echo "== batch trace exit $? :: $(grep -o '"value": [0-9.]*' $REPO/gpurun_out/r5b2_trace_run.log | head -1)"
f=$(find /tmp/rp_tl -name "*kernel_trace.csv" | head -1)
python $REPO/scripts/timeline.py "$f" 4 $REPO/gpurun_out/r5b2_timeline_kernels.txt | tee $REPO/gpurun_out/r5b2_timeline_b32.txt | head -24
