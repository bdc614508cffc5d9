#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids > gpurun_out/r02/pytest_gpu_full.txt
grep -n "^FAILED\|^ERROR\|passed\|failed" gpurun_out/r02/pytest_gpu_full.txt | tail -20
grep -B5 -A40 "^___" gpurun_out/r02/pytest_gpu_full.txt | head -150 > gpurun_out/r02/pytest_failures.txt
bash scripts/r02/pmc_traffic.sh
mkdir -p profiles; cp gpurun_out/r02/gemm_traffic.json profiles/r02_gemm_traffic.json
timeout 400 python bench.py 2>/dev/null | tee gpurun_out/r02/bench_c.json
