#!/bin/bash
# Round 4, call 14: (a) the instruction mix of pass 1 of a two-pass one-loop 3M (W4_DBG=32: 32 MFMAs per K tile beside the full
# staging + fragment traffic), (b) the ablations of section 3 repeated for the REAL kernel (r_fwd), same box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r04; mkdir -p $out
for v in "" _m3 _nomfma _noload _nostage _hot; do echo "== lib$v"; SHAPES=bench ROUNDS=4 PER=8 ONLY=c_fwd,r_fwd,r_wgrad_kl timeout 300 python scripts/r04/w4_ab.py cplxmodule_amd/libcplxamd$v.so 2>&1 | grep -E "^w[48]|^family"; done > $out/w4_variants14.txt 2>&1
cat $out/w4_variants14.txt
for v in base:"" m3:_m3 nomfma:_nomfma noload:_noload nostage:_nostage; do
  bash scripts/r04/w4_pmc.sh ${v%%:*} cplxmodule_amd/libcplxamd${v##*:}.so "c_fwd r_fwd" "1" | tail -3
done
