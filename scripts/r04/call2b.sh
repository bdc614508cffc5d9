#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r04; mkdir -p $out
for v in _noepi _floor; do echo "== lib$v"; SHAPES=bench ROUNDS=3 PER=6 ONLY=c_fwd,c_dgrad,c_wgrad_kl,r_fwd,r_wgrad_kl timeout 300 python scripts/r04/w4_ab.py cplxmodule_amd/libcplxamd$v.so 2>&1 | tail -12; done > $out/w4_variants_b.txt 2>&1
cat $out/w4_variants_b.txt
bash scripts/r04/w4_pmc.sh prod cplxmodule_amd/libcplxamd.so "c_fwd c_wgrad_kl r_fwd r_wgrad_kl" "0 1"
bash scripts/r04/w4_pmc.sh floor cplxmodule_amd/libcplxamd_floor.so "c_fwd" "1"
bash scripts/r04/w4_pmc.sh noepi cplxmodule_amd/libcplxamd_noepi.so "c_fwd" "1"
