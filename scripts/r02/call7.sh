#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02
L=cplxmodule_amd
rm -f gpurun_out/r02/gemm_check.txt
for v in "" _nopersist; do timeout 120 python scripts/gemm_check.py $L/libcplxamd$v.so 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r02/gemm_check.txt; done
timeout 300 python scripts/gemm_ab.py prev=$L/libcplxamd_prev.so nopersist=$L/libcplxamd_nopersist.so persist=$L/libcplxamd.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02/gemm_ab3.txt
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -5
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tee gpurun_out/r02/bench_e.json
