#!/bin/bash
# Round 4, call 19 (HEAD): both families at configs[3]'s shapes (K = 2048, batch 2^16 rows per launch here) and the bench
# shapes once more, same process; then the default bench line of this box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r04; mkdir -p $out
SHAPES=cfg4 ROUNDS=3 PER=4 timeout 600 python scripts/r04/w4_ab.py cplxmodule_amd/libcplxamd.so 2>&1 | grep -E "^w[48]|^family|RESULT|MISMATCH" > $out/w4_head_cfg4.txt
SHAPES=bench ROUNDS=4 PER=8 timeout 600 python scripts/r04/w4_ab.py cplxmodule_amd/libcplxamd.so 2>&1 | grep -E "^w[48]|^family|RESULT|MISMATCH" > $out/w4_head_bench.txt
cat $out/w4_head_cfg4.txt $out/w4_head_bench.txt
python bench.py > $out/bench_head.json 2> $out/bench_head.err
python -c "
import json;d=json.load(open('$out/bench_head.json'));print(d['ms_per_step'],d['roofline']['frac'],d['conv_cfg3']['ms_per_step'],d['cfg4_lrt']['ms_per_step'],d['cfg2_linear']['4m']['ms_per_step'],d['hbm_kernels_GBps'])"
