#!/bin/bash
# Quick GPU iteration: conv-layer + network parity, then bench (no CPU baseline).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -x -k "conv_layer or net_" --timeout 600 > gpurun_out/quick_tests.log 2>&1
echo "== tests exit $? :: $(tail -n 1 gpurun_out/quick_tests.log)"
grep -E "FAILED|Error|error|assert" gpurun_out/quick_tests.log | head -20
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline $BENCH_ARGS > gpurun_out/quick_bench.log 2>&1
echo "== bench exit $?"
tail -n 1 gpurun_out/quick_bench.log | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    print('value', d['value'], 'ms/step', d['ms_per_step'], 'roof', d['roofline']['kernel'], d['roofline']['achieved'], d['roofline']['frac'], 'net_ms', d['roofline']['net_ms_per_batch'])
    for k, v in sorted(d['kernel_classes'].items(), key=lambda kv: -kv[1]['ms']):
        print('  %-40s n=%3d ms=%8.3f tflops=%7.2f gbs=%7.1f' % (k, v['launches'], v['ms'], v['tflops'], v['gbs']))
except Exception as e:
    print('parse failed', e)
"
tail -n 5 gpurun_out/quick_bench.log | cut -c1-300 | grep -v '^{' 
