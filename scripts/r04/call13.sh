#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r04; mkdir -p $out
for v in "" _s2 _s2w; do echo "== lib$v"; SHAPES=bench ROUNDS=4 PER=8 timeout 300 python scripts/r04/w4_ab.py cplxmodule_amd/libcplxamd$v.so 2>&1 | grep -E "^w[48]|^family|RESULT|MISMATCH"; done > $out/w4_variants13.txt 2>&1
cat $out/w4_variants13.txt
for v in "" _s2; do echo "== cfg4 lib$v"; SHAPES=cfg4 ROUNDS=3 PER=3 ONLY=c_fwd,c_dgrad,c_dgrad_lrt,r_fwd,r_dgrad timeout 300 python scripts/r04/w4_ab.py cplxmodule_amd/libcplxamd$v.so 2>&1 | grep -E "^w[48]|^family|RESULT|MISMATCH"; done > $out/w4_variants13c.txt 2>&1
cat $out/w4_variants13c.txt
bash scripts/r04/w4_pmc.sh s2 cplxmodule_amd/libcplxamd_s2.so "c_fwd r_fwd" "1"
bash scripts/r04/w4_pmc.sh base cplxmodule_amd/libcplxamd.so "c_fwd r_fwd" "1"
