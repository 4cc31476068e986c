timeout 3000 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_shplonk.py -q -x 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -25
