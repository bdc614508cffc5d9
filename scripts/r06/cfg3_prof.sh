#!/bin/bash
# rocprofv3 kernel stats of configs[2] in bf16 with and without the batch-norm backward folded into the weight gradient
# (same box, back to back, twice)
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r06; mkdir -p $O
for f in 1 0; do
  FOLD=$f STEPS=20 python $R/scripts/r06/cfg3_step.py > $O/cfg3_fold${f}_plain.log 2>&1
  FOLD=$f rocprofv3 --kernel-trace --stats -d $O/prof_cfg3 -- python $R/scripts/r06/cfg3_step.py > $O/cfg3_fold$f.log 2>&1
  python $R/scripts/rocprof_summary.py $O/prof_cfg3/*/*_results.db > $O/cfg3_fold${f}_kernel_stats.txt
  rm -rf $O/prof_cfg3
done
FOLD=1 STEPS=20 python $R/scripts/r06/cfg3_step.py > $O/cfg3_fold1_plain2.log 2>&1
FOLD=0 STEPS=20 python $R/scripts/r06/cfg3_step.py > $O/cfg3_fold0_plain2.log 2>&1
grep -h "ms_per_step" $O/cfg3_fold1_plain.log $O/cfg3_fold0_plain.log $O/cfg3_fold1_plain2.log $O/cfg3_fold0_plain2.log $O/cfg3_fold1.log $O/cfg3_fold0.log
head -12 $O/cfg3_fold1_kernel_stats.txt | cut -c1-170; head -12 $O/cfg3_fold0_kernel_stats.txt | cut -c1-170
