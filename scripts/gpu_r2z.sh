#!/bin/bash
cd "$(dirname "$0")/.."
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "plan_file or c_host" 2>&1 | tail -6; timeout 600 python -m pytest tests/test_gpu_bev.py -q -x -m gpu -k "plan_file" 2>&1 | tail -8 2>&1 | tail -15
