#!/bin/bash
# round 4, GPU call 1: first run of the one-wave-per-SIMD GEMM family -- bit-exactness vs the 8-wave family, A/B timing,
# ablation builds, then the GPU test tier with the new family as the default.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r04; mkdir -p $out
SHAPES=small timeout 300 python scripts/r04/w4_ab.py > $out/w4_small.txt 2>&1; echo "small rc=$?"
tail -3 $out/w4_small.txt
SHAPES=bench ROUNDS=7 PER=8 timeout 400 python scripts/r04/w4_ab.py > $out/w4_bench.txt 2>&1; echo "bench rc=$?"
tail -4 $out/w4_bench.txt
SHAPES=cfg4 ROUNDS=3 PER=3 timeout 400 python scripts/r04/w4_ab.py > $out/w4_cfg4.txt 2>&1; echo "cfg4 rc=$?"
tail -4 $out/w4_cfg4.txt
for v in nomfma noload nostage; do
  SHAPES=bench ROUNDS=3 PER=6 ONLY=c_fwd,c_dgrad,c_wgrad_kl,r_fwd,r_wgrad_kl timeout 300 python scripts/r04/w4_ab.py cplxmodule_amd/libcplxamd_$v.so > $out/w4_$v.txt 2>&1
  echo "== $v"; tail -3 $out/w4_$v.txt
done
timeout 600 python -m pytest tests -m gpu -x -q > $out/gpu_tests_1.txt 2>&1; echo "pytest rc=$?"
tail -5 $out/gpu_tests_1.txt
