#!/bin/bash
# cfg3 (B from $1, default 64): timing with channels-last and NCHW input, then a kernel trace of the channels-last run
B=${1:-64}
cd /tmp && export TMPDIR=/tmp; cd - >/dev/null
timeout 300 python scripts/bench_configs.py --only cfg3b --cfg3-batch $B 2>&1 | tail -1
timeout 300 python scripts/bench_configs.py --only cfg3b --cfg3-batch $B --cfg3-layout nchw 2>&1 | tail -1
rm -rf gpurun_out/cfg3prof; mkdir -p gpurun_out/cfg3prof
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/cfg3prof -o cfg3 -- python scripts/bench_configs.py --only cfg3b --cfg3-batch $B > gpurun_out/cfg3prof/log.txt 2>&1
DB=$(ls gpurun_out/cfg3prof/*.db gpurun_out/cfg3prof/*/*.db 2>/dev/null | head -1)
python scripts/rocprof_summary.py $DB > gpurun_out/cfg3_kernel_stats_B$B.txt
rm -rf gpurun_out/cfg3prof
head -30 gpurun_out/cfg3_kernel_stats_B$B.txt
