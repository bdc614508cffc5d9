#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02
python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -2
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r02/prof_c1 -- python /root/repo/scripts/bench_configs.py --only cfg1 > /root/repo/gpurun_out/r02/prof_c1.log 2>&1
cd /root/repo
python scripts/rocprof_summary.py gpurun_out/r02/prof_c1/*/*_results.db > gpurun_out/r02/cfg1_kernel_stats.txt
rm -rf gpurun_out/r02/prof_c1
head -50 gpurun_out/r02/cfg1_kernel_stats.txt | cut -c1-150
python scripts/bench_configs.py --only cfg1,cfg1g 2>/dev/null
