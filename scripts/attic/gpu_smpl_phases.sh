#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bev.py -q -x -k "smpl" -m gpu 2>&1 | tail -3
for dbg in 0 1 2 4 8 10 15; do
echo "=== ROMP_SMPL_DBG=$dbg"
ROMP_SMPL_DBG=$dbg timeout 300 python bench.py --workload smpl --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print(r['config']['gpu_ms_per_launch'])"
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_smpl -o smpl -- python /root/repo/bench.py --workload smpl --no-cpu-baseline > /dev/null 2>&1
python /root/repo/scripts/rocpd_stats.py /root/repo/gpurun_out/prof_smpl/
