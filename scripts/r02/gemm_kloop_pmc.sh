#!/bin/bash
# cycles (not time: the chip is power-limited, removed work raises the clock) of the one-tile complex GEMM with parts of
# its K loop compiled out (CPLXAMD_GEMM_DBG_BUILD bits: 1 no LDS-DMA after the prologue, 2 no MFMA, 4 no barrier,
# 16 no sign XORs, 64 no vmcnt wait).  One --pmc pass per build, kernel-trace only.
cd /tmp && export TMPDIR=/tmp
R=/root/repo
for d in ${KSET:-0 2 4 64 68 1 5 16}; do
  timeout 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA --kernel-trace --output-format csv -d $R/gpurun_out/pmck_$d -- python $R/scripts/gemm_one_lib.py $R/cplxmodule_amd/libcplxamd_k$d.so 8 > $R/gpurun_out/pmck_$d.log 2>&1
done
cd $R
python - <<'PY' | tee gpurun_out/gemm_kloop_pmc.txt
import glob, csv, collections, re
val = collections.defaultdict(list); dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmck_*/**/*counter_collection.csv", recursive=True):
    v = re.search(r"pmck_(\d+)", f).group(1)
    for r in csv.DictReader(open(f)):
        if "gemm_bf16" in r["Kernel_Name"]:
            val[(r["Counter_Name"], v)].append(float(r["Counter_Value"]))
for f in glob.glob("gpurun_out/pmck_*/**/*kernel_trace.csv", recursive=True):
    v = re.search(r"pmck_(\d+)", f).group(1)
    for r in csv.DictReader(open(f)):
        if "gemm_bf16" in r["Kernel_Name"]:
            dur[v].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
order = [v for v in ["0", "16", "4", "64", "68", "1", "5", "2", "34", "66", "3", "130"] if dur.get(v)]
names = {"130": "no MFMA, whole-line requests", "34": "no MFMA, hot source", "66": "no MFMA, no vmcnt wait", "3": "no MFMA, no LDS-DMA", "0": "full", "16": "no XOR", "4": "no barrier", "64": "no vmcnt wait", "68": "no barrier, no wait", "1": "no LDS-DMA", "5": "no DMA, no barrier", "2": "no MFMA"}
med = lambda x: sorted(x)[len(x) // 2] if x else float("nan")
print(f"{'build':22s} {'cycles/XCD':>11s} {'us':>8s} {'GHz':>6s} {'MFMA busy':>10s} {'WAIT_ANY/WAVE':>14s} {'WAIT_LDS/WAVE':>14s}")
for v in order:
    cyc = med(val[("GRBM_GUI_ACTIVE", v)][2:]) / 8
    us = med(dur[v][2:])
    mf = med(val[("SQ_VALU_MFMA_BUSY_CYCLES", v)][2:])
    wc = med(val[("SQ_WAVE_CYCLES", v)][2:])
    print(f"{names[v]:22s} {cyc:11.4g} {us:8.1f} {cyc / us / 1e3:6.2f} {mf / 1024 / cyc * 100:9.1f}% {med(val[('SQ_WAIT_ANY', v)][2:]) / wc * 100:13.1f}% {med(val[('SQ_WAIT_INST_LDS', v)][2:]) / wc * 100:13.1f}%")
PY
rm -rf gpurun_out/pmck_*/
