cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
for D in 0 1 4 5 8 0; do
  echo "== ROMP_CONV_DEBUG=$D"
  ROMP_CONV_DEBUG=$D SWEEP_CASES=s2 SWEEP_FILTER=h2s_k3s2_mt2_nt2_tw16_ck16,h2s_k3s2_mt2_nt4_tw16_ck16,h2s_k3s2_mt2_nt3_tw16_ck16 timeout 200 python scripts/conv_sweep.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r5b8_s2_ablate.txt | grep -A3 "64, 64, 3, 2, 256\|256, 64, 3, 2, 128\|32, 128, 3, 2, 128\|== ROMP"
