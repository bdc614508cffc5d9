#!/bin/bash
# round 4, GPU call 6: the persistent w4 kernels (ring through the output-tile boundaries)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r04; mkdir -p $out
SHAPES=small timeout 300 python scripts/r04/w4_ab.py > $out/w4_small6.txt 2>&1; tail -3 $out/w4_small6.txt; grep -c "identical" $out/w4_small6.txt; grep MISMATCH $out/w4_small6.txt | head
W4O=1 SHAPES=bench ROUNDS=5 PER=8 timeout 400 python scripts/r04/w4_ab.py > $out/w4_bench6.txt 2>&1; tail -5 $out/w4_bench6.txt; grep MISMATCH $out/w4_bench6.txt | head
W4O=1 SHAPES=cfg4 ROUNDS=3 PER=3 timeout 400 python scripts/r04/w4_ab.py > $out/w4_cfg46.txt 2>&1; tail -5 $out/w4_cfg46.txt; grep MISMATCH $out/w4_cfg46.txt | head
timeout 900 python -m pytest tests/test_gpu_r04.py tests/test_gpu_fullsize.py -x -q > $out/gpu_tests_6.txt 2>&1; echo "pytest rc=$?"; tail -5 $out/gpu_tests_6.txt
