#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02
CPLXAMD_PARITY_REPORT=/root/repo/gpurun_out/r02/parity_report.txt timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -4
python scripts/bench_configs.py --cfg3-batch 64 --only cfg1,cfg1g,cfg3b,cfg4b 2>/dev/null | tee gpurun_out/r02/other_configs.jsonl
python scripts/bench_configs.py --cfg3-batch 256 --only cfg3b 2>/dev/null | tee -a gpurun_out/r02/other_configs.jsonl
timeout 400 python bench.py 2>/dev/null | tee gpurun_out/r02/bench_g.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r02/prof_bench -- python /root/repo/bench.py --steps 25 --warmup 5 --no-cpu-baseline > /root/repo/gpurun_out/r02/prof_bench.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r02/prof_c4 -- python /root/repo/scripts/bench_configs.py --only cfg4b > /root/repo/gpurun_out/r02/prof_c4.log 2>&1
cd /root/repo
python scripts/rocprof_summary.py gpurun_out/r02/prof_bench/*/*_results.db > gpurun_out/r02/bench_n1_kernel_stats.txt
python scripts/rocprof_summary.py gpurun_out/r02/prof_c4/*/*_results.db > gpurun_out/r02/cfg4_kernel_stats.txt
rm -rf gpurun_out/r02/prof_bench gpurun_out/r02/prof_c4
head -30 gpurun_out/r02/bench_n1_kernel_stats.txt
