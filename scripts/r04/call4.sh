#!/bin/bash
# round 4, GPU call 4: w4 with the reworked epilogues (operands prefetched per round, one test outside, buffer addressing)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r04; mkdir -p $out
SHAPES=small timeout 300 python scripts/r04/w4_ab.py > $out/w4_small4.txt 2>&1; tail -2 $out/w4_small4.txt
SHAPES=bench ROUNDS=5 PER=8 timeout 400 python scripts/r04/w4_ab.py > $out/w4_bench4.txt 2>&1; tail -4 $out/w4_bench4.txt
SHAPES=cfg4 ROUNDS=3 PER=3 timeout 400 python scripts/r04/w4_ab.py > $out/w4_cfg44.txt 2>&1; tail -4 $out/w4_cfg44.txt
timeout 900 python -m pytest tests/test_gpu_r04.py tests/test_gpu_fullsize.py tests/test_gpu_batchnorm.py -x -q > $out/gpu_tests_4.txt 2>&1; echo "pytest rc=$?"; tail -15 $out/gpu_tests_4.txt
