#!/bin/bash
# rocprofv3 kernel stats of one float32 CplxLinearVD step on split operands (x3): B = 2^17, 2048 -> 2048
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r06; mkdir -p $O
X3_ONLY=${1:-x3} rocprofv3 --kernel-trace --stats -d $O/prof_x3 -- python $R/scripts/r06/x3_bench.py ${2:-17} ${3:-2048} > $O/x3_${1:-x3}.log 2>&1
python $R/scripts/rocprof_summary.py $O/prof_x3/*/*_results.db > $O/x3_${1:-x3}_kernel_stats.txt
rm -rf $O/prof_x3
tail -3 $O/x3_${1:-x3}.log; head -40 $O/x3_${1:-x3}_kernel_stats.txt
