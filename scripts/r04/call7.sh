#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r04; mkdir -p $out
for v in "" _p1 _p1w; do echo "== lib$v"; W4O=1 SHAPES=bench ROUNDS=4 PER=8 ONLY=c_fwd,c_dgrad,c_dgrad_lrt,c_wgrad_kl,r_fwd timeout 300 python scripts/r04/w4_ab.py cplxmodule_amd/libcplxamd$v.so 2>&1 | grep -E "^w[48]|^family|RESULT"; done > $out/w4_variants7.txt 2>&1
cat $out/w4_variants7.txt
for v in "" _p1; do echo "== cfg4 lib$v"; W4O=1 SHAPES=cfg4 ROUNDS=3 PER=3 ONLY=c_fwd,c_dgrad,c_dgrad_lrt,r_fwd,r_dgrad timeout 300 python scripts/r04/w4_ab.py cplxmodule_amd/libcplxamd$v.so 2>&1 | grep -E "^w[48]|^family|RESULT"; done > $out/w4_variants7c.txt 2>&1
cat $out/w4_variants7c.txt
bash scripts/r04/w4_pmc.sh p1 cplxmodule_amd/libcplxamd_p1.so "c_fwd" "1"
bash scripts/r04/w4_pmc.sh p3 cplxmodule_amd/libcplxamd.so "c_fwd" "1"
