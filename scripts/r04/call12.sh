#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r04; mkdir -p $out
for rep in 1 2 3; do for v in "" _klold; do CPLXAMD_LIB=$PWD/cplxmodule_amd/libcplxamd$v.so timeout 300 python scripts/r04/hbm_ab.py 2>&1 | tail -1; done; done | tee $out/hbm_ab12.txt
