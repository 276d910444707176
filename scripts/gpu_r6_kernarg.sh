#!/bin/bash
# Where do kernel arguments live?  The conv kernels read their ~350-byte ConvParams in 4-8 serialised scalar-memory round trips before
# their first memory request (profiles/r06s_trace3.txt: 3.5 us between a wave's entry and its first halo DMA in the fused blocks).
# HIP_FORCE_DEV_KERNARG=1 places the argument segment in device memory; A/B on the block and on the headline.
O=gpurun_out
{
for kv in unset 0 1; do
  if [ $kv = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$kv; fi
  echo "== HIP_FORCE_DEV_KERNARG=$kv"
  ROMP_CONV_TRACE=1 BB_C=64 BB_FUSED_ONLY=1 timeout 300 python scripts/bblock_bench.py 2>&1 | grep -v "Warn\|amdgpu.ids\|wave "
  for rep in 1 2; do
  timeout 600 python bench.py --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['config'].get('maps_max_abs_vs_oracle'))"
  done
done
} > $O/r06s_kernarg_ab.txt 2>&1
cat $O/r06s_kernarg_ab.txt
