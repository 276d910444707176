#!/bin/bash
# ResNet-50's stem as one MFMA kernel (csrc/stem7p.hip): parity, then the ResNet-50 line with the fusion off / on, interleaved
O=gpurun_out
timeout 1200 python -m pytest tests/test_gpu_resnet.py -q -x > $O/r06s_stem7p_tests.log 2>&1; echo "tests rc=$?" >> $O/r06s_stem7p_tests.log; tail -4 $O/r06s_stem7p_tests.log
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
} > $O/r06s_stem7p_ab.txt 2>&1
cat $O/r06s_stem7p_ab.txt
