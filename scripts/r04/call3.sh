#!/bin/bash
# round 4, GPU call 3: write-window / wait variants of the w4 family, hot-source and epilogue ablations, PMC.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r04; mkdir -p $out
ab() { SHAPES=bench ROUNDS=3 PER=6 ONLY=c_fwd,c_dgrad,c_wgrad_kl,r_fwd,r_wgrad_kl timeout 300 python scripts/r04/w4_ab.py $1 2>&1 | grep -E "^w[48]|^family|Error|error|RESULT"; }
for v in "" _wait1 _burst _burstw _hot _noepi _floor; do echo "== lib$v"; ab cplxmodule_amd/libcplxamd$v.so; done > $out/w4_variants3.txt 2>&1
cat $out/w4_variants3.txt
bash scripts/r04/w4_pmc.sh prod cplxmodule_amd/libcplxamd.so "c_fwd c_wgrad_kl r_fwd r_wgrad_kl" "0 1"
bash scripts/r04/w4_pmc.sh burstw cplxmodule_amd/libcplxamd_burstw.so "c_fwd r_fwd" "1"
bash scripts/r04/w4_pmc.sh floor cplxmodule_amd/libcplxamd_floor.so "c_fwd" "1"
bash scripts/r04/w4_pmc.sh noepi cplxmodule_amd/libcplxamd_noepi.so "c_fwd" "1"
bash scripts/r04/w4_pmc.sh hot cplxmodule_amd/libcplxamd_hot.so "c_fwd" "1"
