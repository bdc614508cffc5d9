#!/bin/bash
# A/B helper: build an alternative libcplxamd_<name>.so whose GEMM translation units come from the EXPERIMENT copies of
# the kernels (scripts/gemm_experiments/: every ablation / variant switch is still in those headers; the production
# headers in cplxmodule_amd/csrc hold the shipped kernels only), compiled with extra flags; select it at run time with
# CPLXAMD_LIB=<path>.
#   scripts/ab_build.sh classic -DCPLXAMD_GEMM_CLASSIC
set -e
name=$1; shift
exp="$(cd "$(dirname "$0")/gemm_experiments" && pwd)"
cd "$(dirname "$0")/../cplxmodule_amd/csrc"
tmp=$(mktemp -d)
for f in gemm_bf16 gemm_bf16_cplx; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -I "$PWD" "$@" -c $exp/$f.hip -o $tmp/$f.o &
done
wait
objs=$(ls build/*.o | grep -v "build/gemm_bf16.o\|build/gemm_bf16_cplx.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $tmp/gemm_bf16.o $tmp/gemm_bf16_cplx.o -o ../libcplxamd_$name.so
rm -rf $tmp
echo "built $(realpath ../libcplxamd_$name.so)"
