#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r03; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/prof_cfg3 -- python $R/scripts/bench_configs.py --only cfg3b --cfg3-batch 256 > $O/cfg3_prof.log 2>&1
python $R/scripts/rocprof_summary.py $O/prof_cfg3/*/*_results.db > $O/cfg3_kernel_stats.txt
rm -rf $O/prof_cfg3
grep -v "^$" $O/cfg3_prof.log | tail -2 | cut -c1-300; head -16 $O/cfg3_kernel_stats.txt | cut -c1-170
