#!/bin/bash
# the whole GPU suite on the final library, then single-image A/B runs of the plan switches under the open stage region
REPO="$(cd "$(dirname "$0")/.." && pwd)"
cd "$REPO"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 1100 python -m pytest tests -m gpu -q --tb=short --timeout 900 > gpurun_out/suite.log 2>&1
echo "== suite exit $? :: $(tail -n 1 gpurun_out/suite.log)"; grep -E "^FAILED|^ERROR" gpurun_out/suite.log | head
for kv in "X=1" "ROMP_FUSE_SEAMS=all" "ROMP_FUSEUP=all" "ROMP_MERGE_S2=0" "X=1"; do
  env $kv NET_GRAPH=1 timeout 120 python scripts/net_b1_loop.py 100 2>&1 | tail -n 1 | sed "s/^/$kv :: /"
done
