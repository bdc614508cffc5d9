#!/bin/bash
# Ablation builds of the one-wave-per-SIMD GEMM family: libcplxamd_<name>.so = the production objects with
# gemm_bf16_w4.hip recompiled with extra flags (e.g. -DW4_DBG=1: no MFMA, 2: no global loads after the prologue,
# 4: no LDS writes after the prologue).   scripts/r04/w4_build.sh nomfma -DW4_DBG=1
set -e
name=$1; shift
cd "$(dirname "$0")/../../cplxmodule_amd/csrc"
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -I "$PWD" "$@" -c gemm_bf16_w4.hip -o $tmp/w4.o
objs=$(ls build/*.o | grep -v "build/gemm_bf16_w4.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $tmp/w4.o -o ../libcplxamd_$name.so
rm -rf $tmp
echo "built $(realpath ../libcplxamd_$name.so)"
