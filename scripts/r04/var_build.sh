#!/bin/bash
# libcplxamd_<name>.so = the production objects with ONE translation unit recompiled with extra flags (experiments).
#   scripts/r04/var_build.sh rp_nt reparam.hip -DRP_NT=1
set -e
name=$1; tu=$2; shift 2
cd "$(dirname "$0")/../../cplxmodule_amd/csrc"
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -I "$PWD" "$@" -c $tu -o $tmp/x.o
objs=$(ls build/*.o | grep -v "build/${tu%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $tmp/x.o -o ../libcplxamd_$name.so
rm -rf $tmp
echo "built $(realpath ../libcplxamd_$name.so)"
