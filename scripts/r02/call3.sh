#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests -m gpu -q -s 2>&1 | grep -v amdgpu.ids | tail -60 > gpurun_out/r02/pytest_gpu.txt
tail -25 gpurun_out/r02/pytest_gpu.txt
timeout 300 python bench.py --steps 50 --warmup 10 2>/dev/null | tee gpurun_out/r02/bench_b.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r02/prof_bench -- python /root/repo/bench.py --steps 25 --warmup 5 --no-cpu-baseline > /root/repo/gpurun_out/r02/prof_bench.log 2>&1
cd /root/repo
python scripts/rocprof_summary.py gpurun_out/r02/prof_bench/*/*_results.db > gpurun_out/r02/bench_kernel_stats_b.txt
rm -rf gpurun_out/r02/prof_bench
head -40 gpurun_out/r02/bench_kernel_stats_b.txt
