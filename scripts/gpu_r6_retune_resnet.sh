#!/bin/bash
# ResNet-50's variant table again (layer1 reads an H2 tensor since the stem is one kernel: its first convs have other candidates now),
# then the committed table against the fresh one on the same box
O=gpurun_out
LEGS="--no-f32-companion --no-latency --no-cpu-baseline --no-end-to-end"
rm -f $O/tune_resnet50_s2.json
timeout 900 python bench.py --backbone resnet50 --tune-file $O/tune_resnet50_s2.json $LEGS > $O/tables_resnet50_s2.log 2>&1
echo "== tuned: $(grep -o '"value": [0-9.]*' $O/tables_resnet50_s2.log | head -1)"
{
for rep in 1 2; do
  echo -n "committed rep $rep: "; timeout 600 python bench.py --backbone resnet50 --steps 5 --warmup 2 $LEGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['config'].get('ms_per_call'), d['config'].get('variant_table'), d['config'].get('maps_max_abs_vs_oracle'), d['config'].get('detections_equal'))"
  echo -n "fresh     rep $rep: "; timeout 600 python bench.py --backbone resnet50 --steps 5 --warmup 2 $LEGS --tune-file $O/tune_resnet50_s2.json 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['config'].get('ms_per_call'), d['config'].get('variant_table'), d['config'].get('maps_max_abs_vs_oracle'), d['config'].get('detections_equal'))"
done
} > $O/r06s_retune_resnet_ab.txt 2>&1
cat $O/r06s_retune_resnet_ab.txt
