#!/bin/bash
for st in 0 25 50 100 200; do
  echo "== CPLXAMD_CL_STAGGER=$st"
  CPLXAMD_CL_STAGGER=$st ONLY="cl kernel" timeout 120 python scripts/conv_cl_bench.py 64 2>&1 | grep "cl kernel"
done
