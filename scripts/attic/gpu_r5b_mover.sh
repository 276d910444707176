#!/bin/bash
# Round 5, second session: conv_h2m (the mover form of conv_h2s: a fifth wave moves pixels two stages ahead) -- parity of every
# stride-2 conv case on every variant, then the sweep of the stride-2 shapes (interleaved twice: first-run effects).
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "conv_layer and _s2_" 2>&1 | tee gpurun_out/r5b9_tests.log | tail -5
for run in 1 2; do
  SWEEP_CASES=s2 SWEEP_FILTER=h2s_k3s2_mt2_nt2,h2m,h2d_k3s2_mt1_nt2_tw16 SWEEP_CHECK=1 timeout 300 python scripts/conv_sweep.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r5b9_sweep.txt | grep -A4 "64, 64, 3, 2, 256\|256, 64, 3, 2, 128\|48, 192, 3, 2, 128"
