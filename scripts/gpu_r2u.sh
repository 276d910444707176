#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "translation or api or parse or romp_end or forward_chunks or batch_vs" 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_dist1.py tests/test_temporal.py tests/test_render.py -q -x -m gpu 2>&1 | tail -2
timeout 300 python scripts/latency_b1.py 2>&1 | tail -2 | tee gpurun_out/latency_b1.txt
