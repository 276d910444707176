#!/bin/bash
# same-box A/B of a ROMP_CONV_DEBUG experiment bit: one tune pass, then alternating runs with the bit off / on
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
BIT=${BIT:-256}
T=/tmp/tune_ab.json; rm -f $T
A="--no-cpu-baseline --no-f32-companion --no-latency --no-end-to-end --no-parity --no-roofline --tune-file $T --steps 6 --warmup 2"
python bench.py $A 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('tune run', r['value'])"
for i in 1 2; do
python bench.py $A 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('bit off', r['value'], r['config']['ms_per_call'])"
ROMP_CONV_DEBUG=$BIT python bench.py $A 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('bit on ', r['value'], r['config']['ms_per_call'])"
done
