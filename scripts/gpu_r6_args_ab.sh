#!/bin/bash
# conv_args_now (conv_common.h): the set-up's kernel arguments in one scalar-memory round trip.  Same box, interleaved: the product
# library against a -DROMP_NO_ARGS_BATCH build of the same sources (romp_amd/libromp_hip_noab.so, built by the caller).
O=gpurun_out
{
for rep in 1 2 3; do
  for arm in batch noab; do
    if [ $arm = noab ]; then export ROMP_HIP_LIB=romp_amd/libromp_hip_noab.so; else unset ROMP_HIP_LIB; fi
    echo -n "$arm rep $rep: "
    timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end --no-f32-companion --no-latency 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], 'dominant us', round(d['roofline'].get('launch_us', 0) or 0, 2), d['roofline']['achieved'], d['config'].get('maps_max_abs_vs_oracle'), d['config'].get('detections_equal'))"
  done
done
for arm in batch noab; do
  if [ $arm = noab ]; then export ROMP_HIP_LIB=romp_amd/libromp_hip_noab.so; else unset ROMP_HIP_LIB; fi
  echo "== $arm: ResNet-50 line"
  timeout 600 python bench.py --backbone resnet50 --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
} > $O/r06s_args_ab.txt 2>&1
cat $O/r06s_args_ab.txt
