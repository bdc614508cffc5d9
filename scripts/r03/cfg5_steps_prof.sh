#!/bin/bash
# kernel trace of 100 hipGraph replays of the configs[4] step -> gpurun_out/r03/cfg5_steps_kernel_stats.txt
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r03; mkdir -p $O
python $R/scripts/r03/cfg5_steps_prof.py --graph > $O/cfg5_steps.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_cfg5s -- python $R/scripts/r03/cfg5_steps_prof.py --graph >> $O/cfg5_steps.log 2>&1
python $R/scripts/rocprof_summary.py $O/prof_cfg5s/*/*_results.db > $O/cfg5_steps_kernel_stats.txt
rm -rf $O/prof_cfg5s
cat $O/cfg5_steps.log | grep "ms / step"
head -70 $O/cfg5_steps_kernel_stats.txt | cut -c1-170
