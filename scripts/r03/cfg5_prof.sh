#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r03; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/prof_cfg5 -- python $R/scripts/r03/cfg5_graph.py > $O/cfg5_prof.log 2>&1
python $R/scripts/rocprof_summary.py $O/prof_cfg5/*/*_results.db > $O/cfg5_kernel_stats.txt
rm -rf $O/prof_cfg5
head -45 $O/cfg5_kernel_stats.txt | cut -c1-180
