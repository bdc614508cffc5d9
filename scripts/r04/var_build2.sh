#!/bin/bash
# A/B builds of one translation unit: libcplxamd_<name>.so = the production objects with <file>.hip recompiled with extra flags
#   scripts/r04/var_build2.sh <name> <file.hip> [flags...]
set -e
name=$1; file=$2; shift 2
cd "$(dirname "$0")/../../cplxmodule_amd/csrc"
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -I "$PWD" "$@" -c $file -o $tmp/v.o
objs=$(ls build/*.o | grep -v "build/${file%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $tmp/v.o -o ../libcplxamd_$name.so
rm -rf $tmp
echo "built $(realpath ../libcplxamd_$name.so)"
