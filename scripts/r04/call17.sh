#!/bin/bash
# Round 4, call 17: epilogue operands of round r + 1 requested ahead of the stores of round r (W4_EPI_PIPE 1, production)
# against behind them (0): the launches with operand-reading epilogues, same process order, bit-exactness vs the 8-wave family
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r04; mkdir -p $out
for v in _nopipe "" _nopipe ""; do echo "== lib$v"; SHAPES=bench ROUNDS=4 PER=8 ONLY=c_fwd,c_dgrad_lrt,c_wgrad_kl,r_wgrad_kl,r_dgrad_lrt,c_wgrad timeout 300 python scripts/r04/w4_ab.py cplxmodule_amd/libcplxamd$v.so 2>&1 | grep -E "^w[48]|^family|MISMATCH|RESULT"; done > $out/w4_variants17.txt 2>&1
cat $out/w4_variants17.txt
