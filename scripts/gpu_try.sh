#!/bin/bash
# bench.py under the driver's multi-GPU launcher (one rank here: the RCCL init / all-gather path with world size 1)
cd "$(dirname "$0")/.."
export PYTHONUNBUFFERED=1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-f32-companion --no-latency --no-end-to-end 2>&1 | tail -3 | cut -c1-900
