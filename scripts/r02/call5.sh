#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids > gpurun_out/r02/pytest_gpu_full.txt
grep -n "^FAILED\|^ERROR\|passed\|failed" gpurun_out/r02/pytest_gpu_full.txt | tail -20
grep -B5 -A60 "^___" gpurun_out/r02/pytest_gpu_full.txt | head -200 > gpurun_out/r02/pytest_failures.txt
for i in 1 2; do
python scripts/reparam_ab.py 2>/dev/null | tee -a gpurun_out/r02/reparam_rounds.txt
CPLXAMD_LIB=/root/repo/cplxmodule_amd/libcplxamd_r10.so python scripts/reparam_ab.py 2>/dev/null | tee -a gpurun_out/r02/reparam_rounds.txt
done
