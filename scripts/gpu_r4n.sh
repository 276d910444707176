#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for cfg in "2 1" "2 0" "1 0"; do
  set -- $cfg
  if [ $2 = 0 ]; then unset ROMP_PIPE_WGCAP; else export ROMP_PIPE_WGCAP=$2; fi
  ROMP_PIPE_NETS=$1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-companion --no-latency --no-end-to-end --no-roofline > gpurun_out/r4n_bench_$1_$2.log 2>&1
  echo "== nets=$1 wgcap=$2: exit $? :: $(grep -o '"value": [0-9.]*' gpurun_out/r4n_bench_$1_$2.log | head -1) $(grep -o '"detections_equal": [a-z]*' gpurun_out/r4n_bench_$1_$2.log | head -1)"
done
