#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-companion --no-roofline --tune-file gpurun_out/tune_b32.json > gpurun_out/r2h_b0.log 2>&1; tail -n 1 gpurun_out/r2h_b0.log | cut -c1-120
for dbg in 32 15 7 4; do
ROMP_CONV_DEBUG=$dbg python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-companion --no-roofline --tune-file gpurun_out/tune_b32.json > gpurun_out/r2h_b$dbg.log 2>&1; echo "dbg $dbg: $(tail -n 1 gpurun_out/r2h_b$dbg.log | cut -c1-120)"
done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-companion --no-roofline --tune-file gpurun_out/tune_b32.json --streams 0 > gpurun_out/r2h_s0.log 2>&1; echo "streams0: $(tail -n 1 gpurun_out/r2h_s0.log | cut -c1-120)"
ROMP_CONV_DEBUG=32 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-companion --no-roofline --tune-file gpurun_out/tune_b32.json --streams 0 > gpurun_out/r2h_s0_32.log 2>&1; echo "streams0 dbg32: $(tail -n 1 gpurun_out/r2h_s0_32.log | cut -c1-120)"
