#!/bin/bash
# after the set-up change (arguments in one round trip, weights requested first): parity, phase trace, block A/B, headline
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "fused_basic_block" > $O/r06s_tests.log 2>&1; echo "tests rc=$?" >> $O/r06s_tests.log; tail -2 $O/r06s_tests.log
bash scripts/gpu_r6_strip_trace.sh; mv $O/r06s_trace3.txt $O/r06s_trace4.txt
{
for rep in 1 2 3; do for C in 64 32; do for run in 0 -1; do
  if [ $run = 0 ]; then export ROMP_BBLOCK_RUN=0; else unset ROMP_BBLOCK_RUN; fi
  echo -n "C=$C run=${ROMP_BBLOCK_RUN:-auto} rep $rep: "
  BB_C=$C BB_FUSED_ONLY=1 timeout 300 python scripts/bblock_bench.py 2>&1 | grep fuse= | sed 's/.*\(bblock[0-9]* [0-9.]* us\).*/\1/'
done; done; done; } > $O/r06s_block_ab3.txt 2>&1
cat $O/r06s_block_ab3.txt
unset ROMP_BBLOCK_RUN
for rep in 1 2 3; do
timeout 600 python bench.py --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['config'].get('maps_max_abs_vs_oracle'), d['config'].get('detections_equal'))"
done > $O/r06s_headline3.txt 2>&1
cat $O/r06s_headline3.txt
