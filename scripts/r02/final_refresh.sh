#!/bin/bash
# round-2 evidence refresh with the final binary -> gpurun_out/final/ (copied into profiles/r02_* afterwards)
O=gpurun_out/final; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd - >/dev/null
R=$PWD
# 1. parity report (whole GPU suite)
CPLXAMD_PARITY_REPORT=$R/$O/parity_report.txt timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -2 > $O/pytest_gpu.txt
# 2. headline bench: two plain runs + kernel trace of the same command
timeout 300 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
timeout 300 python bench.py > $O/bench_n1_b.json 2>> $O/bench_n1.err
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_bench -o bench -- python $R/bench.py > $R/$O/prof_bench.log 2>&1)
python scripts/rocprof_summary.py $(ls $O/prof_bench/*.db $O/prof_bench/*/*.db 2>/dev/null | head -1) > $O/bench_n1_kernel_stats.txt
rm -rf $O/prof_bench
# 3. convolution kernels (cfg3 layer), both forward / data-gradient kernels
{ for e in cplxamd_conv2d_cl2 cplxamd_conv2d_cl; do echo "## forward / data gradient through $e"; ENTRY=$e PAD=0 timeout 200 python scripts/conv_cl_bench.py 64; done
  echo "## padding 1"; ENTRY=cplxamd_conv2d_cl2 PAD=1 ONLY="cl " timeout 200 python scripts/conv_cl_bench.py 64
  echo "## batch 256"; ENTRY=cplxamd_conv2d_cl2 PAD=0 ONLY="cl " timeout 200 python scripts/conv_cl_bench.py 256; } 2>&1 | grep -v amdgpu.ids > $O/conv_cl_kernels.txt
# 4. cfg3 end to end, chain, LRT conv layer
for B in 64 256; do for lay in channels_last nchw; do timeout 300 python scripts/bench_configs.py --only cfg3b --cfg3-batch $B --cfg3-layout $lay 2>&1 | tail -1; done; done > $O/cfg3.jsonl
timeout 400 python scripts/cl_chain_bench.py 32 3 256 2>&1 | tail -2 > $O/conv_chain.txt
scripts/r02/cfg3_prof.sh 64 > /dev/null 2>&1; cp gpurun_out/cfg3_kernel_stats_B64.txt $O/cfg3_kernel_stats.txt
scripts/r02/lrt_conv_prof.sh 32 2>&1 | head -2 > $O/lrt_conv.txt; cp gpurun_out/lrt_conv_kernel_stats_B32.txt $O/lrt_conv_kernel_stats.txt
# 5. the other configs + cfg4 kernel trace
timeout 600 python scripts/bench_configs.py --cfg3-batch 256 --only cfg1,cfg1g,cfg3f,cfg4b,cfg4f 2>/dev/null > $O/other_configs.jsonl
scripts/r02/cfg4_prof.sh > /dev/null 2>&1; cp gpurun_out/cfg4_kernel_stats.txt $O/cfg4_kernel_stats.txt
cat $O/pytest_gpu.txt; cut -c1-600 $O/bench_n1.json; cut -c1-200 $O/cfg3.jsonl $O/other_configs.jsonl; cat $O/conv_chain.txt $O/lrt_conv.txt; head -12 $O/conv_cl_kernels.txt
