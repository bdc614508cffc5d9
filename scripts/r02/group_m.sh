#!/bin/bash
# tile-group shape of the complex GEMM (rows of 256 x columns of 128 per XCD round): time and L2<->fabric traffic
mkdir -p gpurun_out/r02
for gm in 2 4 8; do
  export CPLXAMD_GEMM_GROUP_M=$gm
  echo "GROUP_M=$gm"
  ITERS=50 timeout 100 python scripts/gemm_one.py 2>&1 | tail -1
  bash scripts/r02/pmc_traffic.sh 2>&1 | tail -1 | cut -c1-400
  cp gpurun_out/r02/gemm_traffic.json gpurun_out/r02/gemm_traffic_gm$gm.json
done
