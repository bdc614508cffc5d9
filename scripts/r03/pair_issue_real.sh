#!/bin/bash
# the pair-issue probe of scripts/r03/pair_issue.sh on the REAL 256 x 256 kernel (forward variance GEMM 8192 x 4096 x 4096)
cd /tmp && export TMPDIR=/tmp
R=/root/repo
for d in ${KSET:-0 2 130 258}; do
  timeout 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmcr_$d -- python $R/scripts/rgemm_one_lib.py $R/cplxmodule_amd/libcplxamd_k$d.so 8 > $R/gpurun_out/pmcr_$d.log 2>&1
done
cd $R
python - <<'PY' | tee gpurun_out/gemm_pair_issue_real.txt
import glob, csv, collections, re
val = collections.defaultdict(list); dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmcr_*/**/*counter_collection.csv", recursive=True):
    v = re.search(r"pmcr_(\d+)", f).group(1)
    for r in csv.DictReader(open(f)):
        if "gemm_bf16" in r["Kernel_Name"]:
            val[(r["Counter_Name"], v)].append(float(r["Counter_Value"]))
for f in glob.glob("gpurun_out/pmcr_*/**/*kernel_trace.csv", recursive=True):
    v = re.search(r"pmcr_(\d+)", f).group(1)
    for r in csv.DictReader(open(f)):
        if "gemm_bf16" in r["Kernel_Name"]:
            dur[v].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
names = {"0": "full (one-tile kernel)", "2": "no MFMA, production half lines", "130": "no MFMA, whole lines (one instruction)", "258": "no MFMA, halves back to back"}
med = lambda x: sorted(x)[len(x) // 2] if x else float("nan")
print(f"{'build':42s} {'cycles/XCD':>11s} {'per K tile':>11s} {'B/clk/CU':>9s} {'us':>8s} {'GHz':>6s} {'MFMA busy':>10s}")
for v in sorted(dur, key=int):
    cyc = med(val[("GRBM_GUI_ACTIVE", v)][2:]) / 8
    us = med(dur[v][2:])
    mf = med(val[("SQ_VALU_MFMA_BUSY_CYCLES", v)][2:])
    print(f"{names.get(v, v):42s} {cyc:11.4g} {cyc / 256:11.0f} {32768 / (cyc / 256):9.1f} {us:8.1f} {cyc / us / 1e3:6.2f} {mf / 1024 / cyc * 100:9.1f}%")
PY
rm -rf gpurun_out/pmcr_*/
