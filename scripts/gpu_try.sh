#!/bin/bash
REPO="$(cd "$(dirname "$0")/.." && pwd)"
cd $REPO
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist1.py -q -x -m gpu -k "split_k or plan_file or fast_path or parse or romp_api or c_host or dist" 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
cd /tmp
python $REPO/scripts/latency_from_plan.py make /tmp/romp_b1.plan 2>&1 | tail -1
python $REPO/scripts/latency_from_plan.py run /tmp/romp_b1.plan 2>&1 | tail -1
