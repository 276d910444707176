#!/bin/bash
# strip form A/B, second pass: per channel count (ROMP_BBLOCK_RUN64 / RUN32), block alone with phase trace, then the headline
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "fused_basic_block" > $O/r06s_tests.log 2>&1; echo "tests rc=$?" >> $O/r06s_tests.log; tail -2 $O/r06s_tests.log
{
for C in 64 32; do for run in 0 -1; do
if [ $run = 0 ]; then export ROMP_BBLOCK_RUN=0; else unset ROMP_BBLOCK_RUN; fi
echo "== C=$C ROMP_BBLOCK_RUN=${ROMP_BBLOCK_RUN:-auto}"
ROMP_CONV_TRACE=1 BB_C=$C BB_FUSED_ONLY=1 timeout 300 python scripts/bblock_bench.py 2>&1 | grep -v "Warn\|amdgpu.ids"
done; done; } > $O/r06s_trace2.txt 2>&1
unset ROMP_BBLOCK_RUN
{
for rep in 1 2 3; do for C in 64 32; do for run in 0 -1; do
  if [ $run = 0 ]; then export ROMP_BBLOCK_RUN=0; else unset ROMP_BBLOCK_RUN; fi
  echo -n "C=$C run=${ROMP_BBLOCK_RUN:-auto} rep $rep: "
  BB_C=$C BB_FUSED_ONLY=1 timeout 300 python scripts/bblock_bench.py 2>&1 | grep fuse= | sed 's/.*\(bblock[0-9]* [0-9.]* us\).*/\1/'
done; done; done; } > $O/r06s_block_ab2.txt 2>&1
cat $O/r06s_block_ab2.txt
unset ROMP_BBLOCK_RUN
{
for rep in 1 2; do
  for arm in "0 0" "a 0" "0 a" "a a"; do
    set -- $arm
    unset ROMP_BBLOCK_RUN64 ROMP_BBLOCK_RUN32
    [ $1 = 0 ] && export ROMP_BBLOCK_RUN64=0
    [ $2 = 0 ] && export ROMP_BBLOCK_RUN32=0
    echo -n "headline strip64=$1 strip32=$2 rep $rep: "
    timeout 600 python bench.py --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('maps_max_abs_vs_oracle'), d['config'].get('detections_equal'))"
  done
done
} > $O/r06s_headline_ab2.txt 2>&1
cat $O/r06s_headline_ab2.txt
