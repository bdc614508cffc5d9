#!/bin/bash
# conv_cl ablation: same box, one process per build (kernel-only timings of scripts/conv_cl_bench.py)
for v in "" _d1 _d2 _d4 _d8 _d12; do
  echo "== build ${v:-full}"
  CPLXAMD_LIB=$PWD/cplxmodule_amd/libcplxamd$v.so ONLY="cl kernel" timeout 120 python scripts/conv_cl_bench.py 64 2>&1 | grep "cl kernel"
done
