#!/bin/bash
# one GPU call: round-3 evidence (bench line + kernel stats + PMC, parity report, the other BASELINE configs, eager vs graph)
R=/root/repo; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
bash scripts/r03/profile_bench.sh > $O/profile_bench.log 2>&1
CPLXAMD_PARITY_REPORT=$O/parity_report.txt python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
python scripts/bench_configs.py --cfg3-batch 64 > $O/other_configs.jsonl 2>/dev/null
python scripts/r03/bench_graph.py 2>/dev/null | tail -1 > $O/graph_replay.txt
python scripts/r03/bench_graph.py --rccl1 2>/dev/null | grep "^rccl1" >> $O/graph_replay.txt
python scripts/r03/cfg5_graph.py 2>/dev/null | grep "^plain" >> $O/graph_replay.txt
python scripts/r03/cfg5_graph.py --rccl1 2>/dev/null | grep "^rccl1" >> $O/graph_replay.txt
bash scripts/r03/cfg5_steps_prof.sh > /dev/null 2>&1      # -> cfg5_steps_kernel_stats.txt (100 graph replays)
bash scripts/r03/cfg5_timeline.sh 140 > /dev/null 2>&1    # -> cfg5_timeline.txt (the kernels of one step, in order)
for f in 1 0; do CPLXAMD_LRT_DX_FUSE=$f python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lrt_dx_fuse=$f', d['ms_per_step'], d['step_ms'], d['roofline']['launch_ms'])"; done > $O/lrt_dx_fuse_ab.txt
cat $O/graph_replay.txt $O/lrt_dx_fuse_ab.txt; head -c 300 $O/bench_n1.json; echo; cat $O/other_configs.jsonl | cut -c1-250
