#!/bin/bash
# Do the runtime's graph / queue switches move the ~15-us cross-queue hand-overs at the HRNet module boundaries?
# Same box, interleaved: default, DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 (no pre-recorded AQL packets), GPU_MAX_HW_QUEUES=8.
cd "$(dirname "$0")/.."
LEGS="--steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end --no-f32-companion --no-latency --no-roofline"
run() { timeout 600 python bench.py $LEGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('ms_per_call'), d.get('clock_mhz',{}).get('median'), d['config'].get('detections_equal'))"; }
for rep in 1 2; do
  echo -n "default rep $rep: "; run
  echo -n "PACKET_CAPTURE=0 rep $rep: "; DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 run
  echo -n "GPU_MAX_HW_QUEUES=8 rep $rep: "; GPU_MAX_HW_QUEUES=8 run
done
