#!/bin/bash
# the BEV line again (cpu_baseline now the staged reference BEVv1), and the stage-region test
cd "$(dirname "$0")/.."; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 600 python bench.py --workload bev --no-f32-companion --no-latency > gpurun_out/bench_bev.log 2>&1; echo "== bench bev: exit $?"
grep '^{' gpurun_out/bench_bev.log | tail -1 > gpurun_out/bench_bev.json
python -c "
import json; d = json.load(open('gpurun_out/bench_bev.json')); print(d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline'])"
timeout 600 python -m pytest tests -m gpu -q --tb=short --timeout 600 -x -k "stage_region or c_host or plan_file" 2>&1 | tail -3
