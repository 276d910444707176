#!/bin/bash
cd "$(dirname "$0")/.."
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -s -k "fast_path or romp_api or romp_end" 2>&1 | tail -12
timeout 300 python scripts/latency_b1.py 2>&1 | tail -2
