set -x
cd /root/repo
mkdir -p gpurun_out/r01
python bench.py > gpurun_out/r01/bench_n1.json 2> gpurun_out/r01/bench_n1.err
python scripts/bench_configs.py --cfg3-batch 64 > gpurun_out/r01/other_configs.jsonl 2>/dev/null
python scripts/bench_configs.py --only cfg3b --cfg3-batch 256 >> gpurun_out/r01/other_configs.jsonl 2>/dev/null
python scripts/microbench.py > gpurun_out/r01/microbench.txt 2>&1
python scripts/gemm_bench.py > gpurun_out/r01/gemm_bench.txt 2>&1
python scripts/rgemm_bench.py >> gpurun_out/r01/gemm_bench.txt 2>&1
python scripts/vendor_gemm.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r01/vendor_gemm.txt
python scripts/conv_one.py > gpurun_out/r01/conv_one.txt 2>&1
python scripts/conv_wgrad_one.py >> gpurun_out/r01/conv_one.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r01/prof_bench -- python /root/repo/bench.py > /root/repo/gpurun_out/r01/prof_bench.log 2>&1
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r01/prof_c3 -- python /root/repo/scripts/bench_configs.py --only cfg3b --cfg3-batch 64 > /root/repo/gpurun_out/r01/prof_c3.log 2>&1
cd /root/repo
python scripts/rocprof_summary.py gpurun_out/r01/prof_bench/*/*_results.db > gpurun_out/r01/bench_n1_kernel_stats.txt
python scripts/rocprof_summary.py gpurun_out/r01/prof_c3/*/*_results.db > gpurun_out/r01/cfg3_kernel_stats.txt
rm -rf gpurun_out/r01/prof_bench gpurun_out/r01/prof_c3
