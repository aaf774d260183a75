cd /root/repo
python -m pytest tests -m gpu -q --durations=12 2>&1 | grep -vE "^\s*$|amdgpu.ids|RCCL|NCCL" | tail -30 > gpurun_out/r4_full_gpu.txt
grep -E "passed|failed" gpurun_out/r4_full_gpu.txt
