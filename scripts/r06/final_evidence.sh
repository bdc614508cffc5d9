mkdir -p gpurun_out/r06p
CPLXAMD_PARITY_REPORT=$PWD/gpurun_out/r06p/parity_report.txt timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r06p/gpu_tier.txt
bash scripts/r06/profile_bench.sh > gpurun_out/r06p/profile_bench.log 2>&1
bash scripts/r06/cfg3_prof.sh > gpurun_out/r06p/cfg3_prof.log 2>&1
bash scripts/r06/conv_x2_prof.sh > gpurun_out/r06p/conv_x2_prof.log 2>&1
bash scripts/r06/cfg3_pmc.sh > gpurun_out/r06p/cfg3_pmc.txt 2>&1
CPLXAMD_BN_FOLD=0 CPLXAMD_CLW_WALK=0 CPLXAMD_X2_ONE=0 timeout 900 python -m pytest tests -m gpu -q -k "conv or bn or batchnorm or x3" 2>&1 | tail -3 > gpurun_out/r06p/gpu_tier_switches_off.txt
tail -3 gpurun_out/r06p/gpu_tier.txt; tail -2 gpurun_out/r06p/gpu_tier_switches_off.txt
