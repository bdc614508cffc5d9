#!/bin/bash
# round-2 conv evidence: kernel timings (pad 0 = the cfg3 layer, pad 1), ablation A/B in one process, cfg3 end to end
# (both input layouts, B = 64 and 256) and the kernel trace of the B = 64 channels-last run
mkdir -p gpurun_out/cl
L=cplxmodule_amd/libcplxamd
{ PAD=0 timeout 200 python scripts/conv_cl_bench.py 64; PAD=1 ONLY="cl " timeout 200 python scripts/conv_cl_bench.py 64; PAD=0 ONLY="cl " timeout 200 python scripts/conv_cl_bench.py 256; } 2>&1 | grep -v amdgpu.ids > gpurun_out/cl/kernels.txt
timeout 300 python scripts/conv_cl_ab.py full=$L.so no_global_stores=${L}_d1.so stores_to_dump=${L}_d2.so no_epilogue=${L}_d4.so no_lds_dma=${L}_d8.so neither=${L}_d12.so contiguous_A=${L}_d32.so contigA_no_epi=${L}_d40.so 2>&1 | grep -v amdgpu.ids > gpurun_out/cl/ablation.txt
for B in 64 256; do for lay in channels_last nchw; do timeout 300 python scripts/bench_configs.py --only cfg3b --cfg3-batch $B --cfg3-layout $lay 2>&1 | tail -1; done; done > gpurun_out/cl/cfg3.jsonl
timeout 400 python scripts/cl_chain_bench.py 32 3 256 2>&1 | tail -2 > gpurun_out/cl/chain.txt
scripts/r02/cfg3_prof.sh 64 > /dev/null 2>&1
cp gpurun_out/cfg3_kernel_stats_B64.txt gpurun_out/cl/
cat gpurun_out/cl/kernels.txt gpurun_out/cl/ablation.txt gpurun_out/cl/chain.txt; cut -c1-170 gpurun_out/cl/cfg3.jsonl
