#!/bin/bash
# Build libcplxamd.so (gfx950 only) in-tree: cplxmodule_amd/libcplxamd.so
set -e
cd "$(dirname "$0")"
OUT=../libcplxamd.so
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value"
mkdir -p build
pids=()
for f in *.hip; do
  o=build/${f%.hip}.o
  stale=0
  # (the half-operand translation units are wrappers that #include a .hip source: that source is a dependency too)
  inc=$(sed -n 's/^#include "\(.*\.hip\)".*/\1/p' "$f")
  for h in "$f" $inc *.h ../../include/cplxamd.h; do
    if [ ! -f "$o" ] || [ "$h" -nt "$o" ]; then stale=1; fi
  done
  if [ $stale = 1 ]; then
    $HIPCC $FLAGS -c "$f" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC build/*.o -o $OUT
echo "built $(realpath $OUT)"
