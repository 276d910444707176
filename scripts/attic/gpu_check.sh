#!/bin/bash
# One gpurun call: GPU parity tests in separate processes (a faulting kernel must not hide the
# other groups), smoke, and a short bench.  Logs land in gpurun_out/.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
rocminfo | grep -E "Marketing Name|Compute Unit" | head -4 > gpurun_out/rocminfo.txt 2>&1
nproc >> gpurun_out/rocminfo.txt
run() { # name, pytest -k expr
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -s -k "$2" --timeout 600 > gpurun_out/test_$1.log 2>&1
  echo "== $1: exit $? :: $(tail -n 1 gpurun_out/test_$1.log)"
}
run smpl "smpl"
run parse "parse or rot6d"
run conv "conv_layer"
run net "net_"
run e2e "romp_api or forward_batch"
timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "== smoke: exit $? :: $(tail -n 1 gpurun_out/smoke.log)"
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.log 2>&1; echo "== bench: exit $? :: $(tail -c 3000 gpurun_out/bench.log)"
grep -hE "FAILED|Error|error|max-abs|err " gpurun_out/test_*.log | head -60
