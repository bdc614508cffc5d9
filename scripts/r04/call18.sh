#!/bin/bash
# Round 4, call 18: what the per-K-tile barrier costs (W4_DBG=64: no s_barrier in the K loop; results are wrong), with and
# without staging (64 + 2 + 4)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r04; mkdir -p $out
for v in "" _nobar _nobar_nostage "" _nobar; do echo "== lib$v"; SHAPES=bench ROUNDS=4 PER=8 ONLY=c_fwd,r_fwd,c_wgrad_kl timeout 300 python scripts/r04/w4_ab.py cplxmodule_amd/libcplxamd$v.so 2>&1 | grep -E "^w4|^family"; done > $out/w4_variants18.txt 2>&1
cat $out/w4_variants18.txt
bash scripts/r04/w4_pmc.sh nobar cplxmodule_amd/libcplxamd_nobar.so "c_fwd r_fwd" "1" | tail -3
