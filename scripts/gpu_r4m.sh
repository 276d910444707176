#!/bin/bash
# Round 4, call M: two networks in flight (RompNet.twin): byte-equality test, same-box A/B, timeline.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short --timeout 900 -x -k "forward_chunks or forward_batch_matches" > gpurun_out/r4m_tests.log 2>&1
echo "== tests exit $? :: $(tail -n 1 gpurun_out/r4m_tests.log)"; grep -E "FAILED|Error|assert" gpurun_out/r4m_tests.log | head
for mode in 2 1 2 1; do
  ROMP_PIPE_NETS=$mode timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-companion --no-latency --no-end-to-end --no-roofline > gpurun_out/r4m_bench_$mode.log 2>&1
  echo "== ROMP_PIPE_NETS=$mode: exit $? :: $(grep -o '"value": [0-9.]*' gpurun_out/r4m_bench_$mode.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4m_bench_$mode.log | head -1) $(grep -o '"detections_equal": [a-z]*' gpurun_out/r4m_bench_$mode.log | head -1)"
done
