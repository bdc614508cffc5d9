#!/bin/bash
# kernel trace of the CplxConv2dVD(64,64,3) LRT step (channels-last path), B from $1 (default 32)
B=${1:-32}
cd /tmp && export TMPDIR=/tmp; cd - >/dev/null
timeout 300 python scripts/lrt_conv_bench.py $B 2>&1 | tail -2
rm -rf gpurun_out/lrtprof; mkdir -p gpurun_out/lrtprof
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/lrtprof -o lrt -- python scripts/lrt_conv_bench.py $B > gpurun_out/lrtprof/log.txt 2>&1
DB=$(ls gpurun_out/lrtprof/*.db gpurun_out/lrtprof/*/*.db 2>/dev/null | head -1)
python scripts/rocprof_summary.py $DB > gpurun_out/lrt_conv_kernel_stats_B$B.txt
rm -rf gpurun_out/lrtprof
head -40 gpurun_out/lrt_conv_kernel_stats_B$B.txt
