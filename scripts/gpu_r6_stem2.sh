#!/bin/bash
# Round 6: the fused stem (csrc/stem2.hip): parity, the network tests that go through it, then the interleaved same-box A/B
# ROMP_FUSE_STEM2=1 / 0 of the default job (three runs per arm) and the per-op times of the serial head.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
O=gpurun_out/r06_stem2
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "stem2 or test_net_golden or saturation_is_observable or test_net_benchmark_batch_vs_oracle or plan_file or romp_api or range_calibration" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_range_guard.py tests/test_gpu_bev.py -x -q 2>&1 | tail -3
LEGS="--no-cpu-baseline --no-parity --no-f32-companion --no-latency --no-end-to-end"
for r in 1 2 3; do for f in 1 0; do
  ROMP_FUSE_STEM2=$f timeout 300 python bench.py --steps 10 $LEGS 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); k = d['kernel_classes']
head = {n: (v['launches'], v['ms']) for n, v in k.items() if n in ('stem2', 'stem_conv') or 'k3s2_mt2_nt2' in n}
print('stem2=$f run $r value', d['value'], 'ms/call', d['config']['ms_per_call'], 'step_ms', d['step_ms']['median'], head)"; done; done > ${O}_ab.txt 2>&1
cat ${O}_ab.txt
timeout 300 python bench.py --steps 5 --no-cpu-baseline --no-f32-companion --no-end-to-end --no-roofline 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); c = d['config']; print('parity with the fused stem: maps', c['maps_max_abs_vs_oracle'], 'detections', c['detections_equal'], 'mesh', c['mesh_max_abs_vs_oracle'], 'latency', d['single_image_latency']['ms_per_frame'], d['single_image_latency']['network_ms'])"
