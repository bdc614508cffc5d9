#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r04; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_r04.py tests/test_gpu_fullsize.py tests/test_gpu_batchnorm.py -q > $out/gpu_tests_5.txt 2>&1; echo "pytest rc=$?"; tail -25 $out/gpu_tests_5.txt
