#!/bin/bash
cd "$(dirname "$0")/.."
export PYTHONUNBUFFERED=1
for i in 1 2; do
python scripts/latency_breakdown.py 2>&1 | grep "graph 1 streams 1"
ROMP_CONV_DEBUG=512 python scripts/latency_breakdown.py 2>&1 | grep "graph 1 streams 1" | sed 's/^/prefetch off: /'
done
