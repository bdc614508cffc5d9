#!/bin/bash
# Round 4, call 15: the one-tile kernel body inside a tile-walking loop (W4_PGRID=256: one workgroup per CU, tiles lin,
# lin + 256, ...; no ring continuation across tiles) -- what the workgroup dispatch between a CU's tiles costs
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r04; mkdir -p $out
for v in "" _pg "" _pg; do echo "== lib$v"; SHAPES=bench ROUNDS=4 PER=8 timeout 300 python scripts/r04/w4_ab.py cplxmodule_amd/libcplxamd$v.so 2>&1 | grep -E "^w[48]|^family|MISMATCH"; done > $out/w4_variants15.txt 2>&1
cat $out/w4_variants15.txt
