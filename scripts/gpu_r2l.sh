#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -x --timeout 900 -k "(conv_layer and h2 and k3_s1)" > gpurun_out/r2l_tests.log 2>&1
echo "== tests exit $? :: $(tail -n 1 gpurun_out/r2l_tests.log)"; grep -E "FAILED|Error|assert" gpurun_out/r2l_tests.log | head
ABLATE_KIND=h2p_k3s1_mt2_nt2 ABLATE_DBG=0,4,1 timeout 900 python - <<'PY' > gpurun_out/r2l_ablate.log 2>&1
import os, sys, subprocess, json
sys.path.insert(0, 'scripts')
import conv_ablate as A
cases = [((64, 64, 3, 1, 64, True), ['h2p_k3s1_mt2_nt2_tw32', 'h2p_k3s1_mt2_nt2_tw16', 'h2d_k3s1_mt2_nt2_tw16_ck16']),
         ((128, 128, 3, 1, 32, True), ['h2p_k3s1_mt2_nt2_tw32', 'h2p_k3s1_mt2_nt2_tw16', 'h2p_k3s1_mt1_nt2_tw16']),
         ((256, 256, 3, 1, 16, True), ['h2p_k3s1_mt1_nt2_tw16', 'h2p_k3s1_mt1_nt1_tw16', 'h2_k3s1_mt1_nt2_tw16', 'h2_k3s1_mt1_nt1_tw16'])]
for case, variants in cases:
    print('case', case)
    for dbg in (0, 4):
        env = dict(os.environ, ROMP_CONV_DEBUG=str(dbg))
        r = subprocess.run([sys.executable, 'scripts/conv_ablate.py', 'child', json.dumps(case), json.dumps(variants)], env=env, capture_output=True, text=True)
        print(r.stdout.rstrip() or r.stderr[-500:])
PY
cat gpurun_out/r2l_ablate.log
