#!/bin/bash
# round-2 first GPU pass: hardware probes, conv-layer sweep incl. the f16x2 kernels, network parity, bench.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 120 scripts/micro/_bin/probe_r2 > gpurun_out/probe_r2.log 2>&1; echo "== probe exit $?"; cat gpurun_out/probe_r2.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -x -k "conv_layer or bf16x3_parity or benchmark_batch" --timeout 900 -s > gpurun_out/r2a_tests.log 2>&1
echo "== tests exit $? :: $(tail -n 1 gpurun_out/r2a_tests.log)"
grep -E "FAILED|Error|error|assert|vs oracle|vs reference|persons in" gpurun_out/r2a_tests.log | head -40
for m in f16x2 bf16x3; do
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-companion --conv-math $m > gpurun_out/r2a_bench_$m.log 2>&1
echo "== bench $m exit $?"
tail -n 1 gpurun_out/r2a_bench_$m.log | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    print('value', d['value'], 'ms/step', d['ms_per_step'], 'roof', d['roofline']['kernel'], d['roofline']['achieved'], d['roofline']['frac'], 'net_ms', d['roofline']['net_ms_per_batch'])
    for k, v in sorted(d['kernel_classes'].items(), key=lambda kv: -kv[1]['ms']):
        print('  %-40s n=%3d ms=%8.3f tflops=%7.2f gbs=%7.1f' % (k, v['launches'], v['ms'], v['tflops'], v['gbs']))
except Exception as e:
    print('parse failed', e)
"
tail -n 5 gpurun_out/r2a_bench_$m.log | cut -c1-300 | grep -v '^{'
done
