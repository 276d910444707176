#!/bin/bash
# the open stage region (RECORD / WAIT edges) against the barrier form: identity test, then same-box A/B of the single-image
# latency and of the default bench line
REPO="$(cd "$(dirname "$0")/.." && pwd)"
cd "$REPO"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short --timeout 900 -x -k "stage_region or split_k or committed_table or net_golden or forward_chunks" > gpurun_out/df_tests.log 2>&1
echo "== tests exit $? :: $(tail -n 1 gpurun_out/df_tests.log)"; grep -E "FAILED|Error|assert" gpurun_out/df_tests.log | head
for m in 1 0 1 0; do
  ROMP_DATAFLOW=$m timeout 300 python scripts/latency_b1.py > gpurun_out/df_latency_$m.txt 2>&1; echo "ROMP_DATAFLOW=$m :: $(grep 'ROMP(image)' gpurun_out/df_latency_$m.txt | cut -c1-200)"
done
B="--steps 10 --warmup 3 --no-cpu-baseline --no-f32-companion --no-end-to-end --no-latency --no-roofline --no-parity"
for m in 1 0 1 0; do
  ROMP_DATAFLOW=$m timeout 600 python bench.py $B > gpurun_out/df_bench_$m.log 2>&1
  echo "ROMP_DATAFLOW=$m :: $(tail -n 1 gpurun_out/df_bench_$m.log | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('variant_table'))")"
done
