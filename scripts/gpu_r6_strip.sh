#!/bin/bash
# Round 6, second session: the halo-carrying ("strip") form of the fused BasicBlock kernels (conv_h2c.h, RCfg<C, STRIP>).
# 1. parity: the plain and the strip form against torch; 2. the block alone, strip form off / on, interleaved; 3. the headline, off / on.
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "fused_basic_block" > $O/r06s_tests.log 2>&1
echo "tests rc=$?" >> $O/r06s_tests.log
tail -3 $O/r06s_tests.log
{
for rep in 1 2; do
  for C in 64 32; do
    for run in 0 -1; do
      if [ $run = 0 ]; then export ROMP_BBLOCK_RUN=0; else unset ROMP_BBLOCK_RUN; fi
      echo "== C=$C ROMP_BBLOCK_RUN=${ROMP_BBLOCK_RUN:-auto} rep $rep"
      BB_C=$C BB_FUSED_ONLY=1 timeout 300 python scripts/bblock_bench.py 2>&1 | grep fuse=
    done
  done
done
} > $O/r06s_block_ab.txt 2>&1
cat $O/r06s_block_ab.txt
{
for rep in 1 2; do
  for run in 0 -1; do
    if [ $run = 0 ]; then export ROMP_BBLOCK_RUN=0; else unset ROMP_BBLOCK_RUN; fi
    echo "== headline ROMP_BBLOCK_RUN=${ROMP_BBLOCK_RUN:-auto} rep $rep"
    timeout 600 python bench.py --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('maps_max_abs_vs_oracle'), d['config'].get('detections_equal'))"
  done
done
} > $O/r06s_headline_ab.txt 2>&1
cat $O/r06s_headline_ab.txt
