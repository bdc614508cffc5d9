#!/bin/bash
# round 4, GPU call 2: where the w4 family's time goes -- load placement variants, epilogue / staging ablations, tile
# order, PMC cycles + clock + matrix-pipe occupancy of both families.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r04; mkdir -p $out
ab() { SHAPES=bench ROUNDS=3 PER=6 ONLY=c_fwd,c_dgrad,c_wgrad_kl,r_fwd,r_wgrad_kl timeout 300 python scripts/r04/w4_ab.py $1 2>&1 | grep -E "^w[48]|^family"; }
for v in "" _ld11 _ld12 _ld13 _noepi _floor; do echo "== lib$v"; ab cplxmodule_amd/libcplxamd$v.so; done > $out/w4_variants.txt 2>&1
for gm in 2 8 16; do echo "== GROUP_M=$gm"; CPLXAMD_GEMM_GROUP_M=$gm ab cplxmodule_amd/libcplxamd.so; done >> $out/w4_variants.txt 2>&1
cat $out/w4_variants.txt
bash scripts/r04/w4_pmc.sh prod cplxmodule_amd/libcplxamd.so "c_fwd c_wgrad_kl r_fwd r_wgrad_kl" "0 1" 
bash scripts/r04/w4_pmc.sh floor cplxmodule_amd/libcplxamd_floor.so "c_fwd" "1"
