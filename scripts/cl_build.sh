#!/bin/bash
# A/B helper for the channels-last conv kernel: libcplxamd_<name>.so with conv_cl.hip compiled with extra flags
#   scripts/cl_build.sh nostore -DCPLXAMD_CL_DBG=1        (select with CPLXAMD_LIB=<path>)
set -e
name=$1; shift
cd "$(dirname "$0")/../cplxmodule_amd/csrc"
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value "$@" -c conv_cl.hip -o $tmp/conv_cl.o
objs=$(ls build/*.o | grep -v "build/conv_cl.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $tmp/conv_cl.o -o ../libcplxamd_$name.so
rm -rf $tmp
echo "built $(realpath ../libcplxamd_$name.so)"
