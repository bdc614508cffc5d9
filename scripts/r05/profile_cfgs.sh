#!/bin/bash
# Round 5: per-kernel times of BASELINE configs[2] (batch 256, bf16 channels-last) and configs[3] (batch 2^20, bf16) at HEAD:
# rocprofv3 --kernel-trace --stats of scripts/bench_configs.py, one config per run -> gpurun_out/r05p/cfg{3,4}_kernel_stats.txt
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r05p; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/prof_cfg3 -- python $R/scripts/bench_configs.py --only cfg3b --cfg3-batch 256 > $O/cfg3.jsonl 2> $O/cfg3.err
python $R/scripts/rocprof_summary.py $O/prof_cfg3/*/*_results.db > $O/cfg3_kernel_stats.txt
rm -rf $O/prof_cfg3
rocprofv3 --kernel-trace --stats -d $O/prof_cfg4 -- python $R/scripts/bench_configs.py --only cfg4b > $O/cfg4.jsonl 2> $O/cfg4.err
python $R/scripts/rocprof_summary.py $O/prof_cfg4/*/*_results.db > $O/cfg4_kernel_stats.txt
rm -rf $O/prof_cfg4
cat $O/cfg3.jsonl $O/cfg4.jsonl | cut -c1-300
head -16 $O/cfg3_kernel_stats.txt | cut -c1-180; head -22 $O/cfg4_kernel_stats.txt | cut -c1-180
