#!/bin/bash
# Round 3: does requesting the two 64-byte halves of a 128-byte line close together in time (same wave, back to back, or
# three instructions apart) give the LDS-DMA stream the throughput of whole-line requests?  Cycles per launch and XCD of
# the one-tile complex forward GEMM WITHOUT its MFMAs (CPLXAMD_GEMM_DBG_BUILD bit 2) with the request pattern varied:
#   2 production half lines (the other half one K tile = 48 KiB later) | 130 whole lines in one instruction
#   258 halves back to back | 770 halves three instructions apart
# builds (round-3 tree, git show 1a9fb01:scripts/ab_build.sh): scripts/ab_build.sh k<N> -DCPLXAMD_GEMM_DBG_BUILD=<N> -DCPLXAMD_GEMM_NO_PERSIST
cd /tmp && export TMPDIR=/tmp
R=/root/repo
for d in ${KSET:-2 130 258 770}; do
  timeout 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmcp_$d -- python $R/scripts/gemm_one_lib.py $R/cplxmodule_amd/libcplxamd_k$d.so 8 > $R/gpurun_out/pmcp_$d.log 2>&1
done
cd $R
python - <<'PY' | tee gpurun_out/gemm_pair_issue.txt
import glob, csv, collections, re
val = collections.defaultdict(list); dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmcp_*/**/*counter_collection.csv", recursive=True):
    v = re.search(r"pmcp_(\d+)", f).group(1)
    for r in csv.DictReader(open(f)):
        if "gemm_bf16" in r["Kernel_Name"]:
            val[(r["Counter_Name"], v)].append(float(r["Counter_Value"]))
for f in glob.glob("gpurun_out/pmcp_*/**/*kernel_trace.csv", recursive=True):
    v = re.search(r"pmcp_(\d+)", f).group(1)
    for r in csv.DictReader(open(f)):
        if "gemm_bf16" in r["Kernel_Name"]:
            dur[v].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
names = {"2": "no MFMA, production half lines", "130": "no MFMA, whole lines (one instruction)",
         "258": "no MFMA, halves back to back", "770": "no MFMA, halves 3 instructions apart", "0": "full"}
med = lambda x: sorted(x)[len(x) // 2] if x else float("nan")
print(f"{'build':42s} {'cycles/XCD':>11s} {'per K tile':>11s} {'B/clk/CU':>9s} {'us':>8s} {'GHz':>6s}")
for v in sorted(dur, key=int):
    cyc = med(val[("GRBM_GUI_ACTIVE", v)][2:]) / 8
    us = med(dur[v][2:])
    print(f"{names.get(v, v):42s} {cyc:11.4g} {cyc / 512:11.0f} {49152 / (cyc / 512):9.1f} {us:8.1f} {cyc / us / 1e3:6.2f}")
PY
rm -rf gpurun_out/pmcp_*/
