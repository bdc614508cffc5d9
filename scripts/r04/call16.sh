#!/bin/bash
# Round 4, call 16: the tile-walking wrapper with DYNAMIC tile assignment (per-XCD ticket counters, W4_PDYN) against the
# static list of call 15 and the production one-tile launch -- is the static list's +1..4 % load imbalance?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r04; mkdir -p $out
for v in "" _pg _pdyn "" _pdyn; do echo "== lib$v"; SHAPES=bench ROUNDS=4 PER=8 timeout 300 python scripts/r04/w4_ab.py cplxmodule_amd/libcplxamd$v.so 2>&1 | grep -E "^w4|^family|MISMATCH"; done > $out/w4_variants16.txt 2>&1
cat $out/w4_variants16.txt
