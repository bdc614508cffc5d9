#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r04; mkdir -p $out
for rep in 1 2; do for v in "" _rp_nt _rp_g1k _rp_g4k _rp_nt4k; do CPLXAMD_LIB=$PWD/cplxmodule_amd/libcplxamd$v.so timeout 300 python scripts/r04/hbm_ab.py 2>&1 | tail -1; done; done | tee $out/hbm_ab10.txt
timeout 600 python -m pytest tests/test_gpu_r04.py -x -q 2>&1 | tail -3
