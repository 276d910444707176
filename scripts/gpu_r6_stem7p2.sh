#!/bin/bash
# stem7p, second form (K rows of 24 = three 8-half groups, one 16-byte DS read per group at a 4-byte-aligned address, reciprocal
# normalisation): first the question the form rests on (scripts/micro/lds_unaligned.hip), then parity, then the ResNet-50 line off / on
O=gpurun_out
timeout 60 scripts/micro/_bin/lds_unaligned > $O/r06s_lds_unaligned.txt 2>&1; echo "probe rc=$?" >> $O/r06s_lds_unaligned.txt; cat $O/r06s_lds_unaligned.txt
timeout 900 python -m pytest tests/test_gpu_resnet.py -q -x -k "stem7p or fixture or oracle" > $O/r06s_stem7p2_tests.log 2>&1; echo "tests rc=$?" >> $O/r06s_stem7p2_tests.log; tail -4 $O/r06s_stem7p2_tests.log
{
for rep in 1 2; do for f in 0 1; do
  export ROMP_FUSE_STEM7P=$f
  echo -n "ROMP_FUSE_STEM7P=$f rep $rep: "
  timeout 600 python bench.py --backbone resnet50 --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end --no-f32-companion --no-latency 2>$O/r06s_stem7p_bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
kc=d.get('kernel_classes') or {}
st={k:v for k,v in kc.items() if 'stem' in k or 'maxpool' in k}
print(d['value'], d['ms_per_step'], d['config'].get('ms_per_call'), d['config'].get('maps_max_abs_vs_oracle'), d['config'].get('detections_equal'), st)"
done; done
} > $O/r06s_stem7p2_ab.txt 2>&1
cat $O/r06s_stem7p2_ab.txt
