#!/bin/bash
# stem2, second pass: four waves with two m blocks in flight per wave (default) against eight waves (ROMP_STEM2_WAVES=8) and the
# unfused pair; parity of both forms first.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
O=gpurun_out/r06_stem2b
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "stem2 or test_net_golden" 2>&1 | tail -3
ROMP_STEM2_WAVES=8 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "stem2 or test_net_golden" 2>&1 | tail -3
LEGS="--no-cpu-baseline --no-parity --no-f32-companion --no-latency --no-end-to-end"
for r in 1 2 3; do for arm in "1 4" "1 8" "0 4"; do
  set -- $arm
  ROMP_FUSE_STEM2=$1 ROMP_STEM2_WAVES=$2 timeout 300 python bench.py --steps 10 $LEGS 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); k = d['kernel_classes']
head = {n: (v['launches'], v['ms']) for n, v in k.items() if n in ('stem2', 'stem_conv') or 'k3s2_mt2_nt2_tw16_ck16' in n}
print('fuse=$1 waves=$2 run $r value', d['value'], 'ms/call', d['config']['ms_per_call'], head)"; done; done > ${O}_ab.txt 2>&1
cat ${O}_ab.txt
