#!/bin/bash
# CplxConv2dVD(64, 64, 3) training step on 256 x 256 bf16 images: input gradient fused into the data-gradient kernel
# (cplxamd_conv2d_cl2_lrt_dx) vs data gradient + cplxamd_lrt_dx_accum, same box, alternating.
cd "$(dirname "$0")/../.."
for rep in 1 2 3; do
  for f in 1 0; do
    echo -n "CPLXAMD_LRT_DX_FUSE=$f  "
    CPLXAMD_LRT_DX_FUSE=$f python scripts/lrt_conv_bench.py 32 cl 2>&1 | grep channels-last
  done
done
