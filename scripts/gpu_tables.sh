#!/bin/bash
# Round 4: fresh variant tables for the four committed configurations (default, B = 128, BEV, ResNet-50) -> gpurun_out/tune_*.json
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for W in default b128 bev resnet50; do
  case $W in
    default) A="";; bev) A="--workload bev";; resnet50) A="--backbone resnet50";; b128) A="--batch 128";;
  esac
  rm -f gpurun_out/tune_$W.json
  timeout 900 python bench.py $A --tune-file gpurun_out/tune_$W.json --no-f32-companion --no-latency --no-cpu-baseline --no-end-to-end > gpurun_out/tables_$W.log 2>&1
  echo "== $W: exit $? :: $(grep -o '"value": [0-9.]*' gpurun_out/tables_$W.log | head -1)"
done
