#!/bin/bash
# round 4, GPU call 8: whole GPU test tier + bench.py with the w4 family on (default) and off
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r04; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/gpu_tests_8.txt 2>&1; echo "pytest rc=$?"; tail -4 $out/gpu_tests_8.txt
timeout 600 python bench.py > $out/bench_w4.json 2> $out/bench_w4.err; echo "bench rc=$?"
CPLXAMD_GEMM_W4=0 timeout 600 python bench.py > $out/bench_w8.json 2> $out/bench_w8.err; echo "bench w8 rc=$?"
python - <<'PY'
import json
for n in ("w4","w8"):
    try:
        d=json.loads(open(f"gpurun_out/r04/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["launch_ms"])
        print("   cfg2", d.get("cfg2_linear")); print("   cfg4", d.get("cfg4_lrt")); print("   cfg3", d.get("conv_cfg3"))
    except Exception as e: print(n, "ERR", e)
PY
