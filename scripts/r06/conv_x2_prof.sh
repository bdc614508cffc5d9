#!/bin/bash
# rocprofv3 kernel stats of configs[2] in float32 on half pieces (and exact): 7 steps of conv + batch-norm, batch 256
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r06; mkdir -p $O
for m in x2 exact; do
  MODES=$m rocprofv3 --kernel-trace --stats -d $O/prof_cx2 -- python $R/scripts/r06/conv_x2_bench.py > $O/conv_$m.log 2>&1
  python $R/scripts/rocprof_summary.py $O/prof_cx2/*/*_results.db > $O/conv_${m}_kernel_stats.txt
  rm -rf $O/prof_cx2
done
grep -h "ms_per_step" $O/conv_x2.log $O/conv_exact.log | cut -c1-200; head -24 $O/conv_x2_kernel_stats.txt | cut -c1-170
