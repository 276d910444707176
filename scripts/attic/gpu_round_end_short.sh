cd /root/repo
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
bash scripts/attic/gpu_final.sh > gpurun_out/final_run.log 2>&1
tail -4 gpurun_out/final_run.log
bash scripts/gpu_profile.sh > gpurun_out/profile_run.log 2>&1
grep -E "exit|aligned" gpurun_out/profile_run.log | head -12
