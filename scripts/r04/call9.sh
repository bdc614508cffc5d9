#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r04; mkdir -p $out
WS=1 SHAPES=cfg4 ROUNDS=3 PER=3 ONLY=c_wgrad_ws,r_wgrad_ws,c_fwd timeout 400 python scripts/r04/w4_ab.py > $out/w4_split9.txt 2>&1; grep -v "^##\|amdgpu" $out/w4_split9.txt | tail -8
WS=1 SHAPES=small ONLY=c_wgrad_ws,r_wgrad_ws timeout 400 python scripts/r04/w4_ab.py 2>&1 | tail -12
python - <<'PY'
# split-K at a shape where it triggers with small tensors: M = N = 256, K = 65536
import ctypes, os, sys, torch
sys.path.insert(0, "scripts/r04"); import w4_ab
os.environ["WS"] = "1"
lib = w4_ab.load("cplxmodule_amd/libcplxamd.so")
w4_ab.ONLY = "c_wgrad_ws,r_wgrad_ws"
print("bad:", w4_ab.run(lib, "split-K: 1 tile, K = 32768", 32768, 256, 256, time_it=False))
print("bad:", w4_ab.run(lib, "split-K: 4 tiles, K = 16384", 16384, 256, 512, time_it=False))
PY
