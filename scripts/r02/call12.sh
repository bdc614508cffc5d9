#!/bin/bash
cd /root/repo
L=cplxmodule_amd
ONLY=c_wgrad,c_wgrad_kl,r_wgrad,r_wgrad_kl timeout 300 python scripts/gemm_ab.py before=$L/libcplxamd_nopersist.so accinit=$L/libcplxamd.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02/gemm_ab5.txt
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -3
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tee gpurun_out/r02/bench_h.json
