#!/bin/bash
# launch-ordered kernel list of the last configs[4] step (hipGraph replay) -> gpurun_out/r03/cfg5_timeline.txt
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r03; mkdir -p $O
rocprofv3 --kernel-trace -d $O/prof_cfg5t -- python $R/scripts/r03/cfg5_steps_prof.py --graph --steps 20 > $O/cfg5_tl.log 2>&1
python $R/scripts/rocprof_timeline.py $O/prof_cfg5t/*/*_results.db ${1:-160} > $O/cfg5_timeline.txt
rm -rf $O/prof_cfg5t
grep conv_kernel $O/cfg5_timeline.txt
