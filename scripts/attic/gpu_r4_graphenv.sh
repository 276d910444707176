#!/bin/bash
# single-image replay time under the HIP runtime's graph switches (found in libamdhip64's strings)
cd "$(dirname "$0")/.."; export PYTHONUNBUFFERED=1
for kv in "X=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "DEBUG_HIP_FORCE_GRAPH_QUEUES=1" "DEBUG_HIP_FORCE_GRAPH_QUEUES=2" "DEBUG_HIP_FORCE_GRAPH_QUEUES=4" "DEBUG_HIP_FORCE_GRAPH_QUEUES=8" "DEBUG_HIP_GRAPH_BATCH_SIZE=1" "DEBUG_HIP_GRAPH_BATCH_SIZE=1024" "GPU_MAX_HW_QUEUES=8" "X=2"; do
  env $kv NET_GRAPH=1 timeout 120 python scripts/net_b1_loop.py 100 2>&1 | tail -n 1 | sed "s/^/$kv :: /"
done
