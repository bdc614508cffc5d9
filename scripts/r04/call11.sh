#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r04; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/gpu_tests_11.txt 2>&1; echo "pytest rc=$?"; tail -4 $out/gpu_tests_11.txt
for rep in 1 2 3; do timeout 300 python scripts/r04/hbm_ab.py 2>&1 | tail -1; done | tee $out/hbm_ab11.txt
