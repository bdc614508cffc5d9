#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r03; mkdir -p $O
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/pmc_rp -- python $R/scripts/r03/reparam_pmc.py > $O/pmc_rp.log 2>&1
cd $R
python - <<'PY' | tee gpurun_out/r03/reparam_pmc.txt
import csv, glob, collections
val = collections.defaultdict(list); dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/r03/pmc_rp/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "reparam" in r["Kernel_Name"]:
            val[(r["Kernel_Name"].split("(")[0][-70:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for f in glob.glob("gpurun_out/r03/pmc_rp/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "reparam" in r["Kernel_Name"]:
            dur[r["Kernel_Name"].split("(")[0][-70:]].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
n = 1 << 27
med = lambda x: sorted(x)[len(x) // 2]
print(f"# noise injection at 2^27 outputs, rocprofv3 --pmc (one pass), median of 4 launches")
print(f"{'kernel':72s} {'us':>8s} {'VALU instr/output':>18s} {'lane-ops/output':>16s} {'VALU busy':>10s} {'GHz':>6s}")
for k in sorted(dur):
    iv = med(val[(k, "SQ_INSTS_VALU")]); av = med(val[(k, "SQ_ACTIVE_INST_VALU")]); g = med(val[(k, "GRBM_GUI_ACTIVE")]) / 8
    us = med(dur[k])
    # SQ_INSTS_VALU counts wave instructions; SQ_ACTIVE_INST_VALU counts quad-cycles the VALU executes (4 cycles each), summed over SIMDs
    print(f"{k:72s} {us:8.1f} {iv / n:18.3f} {iv * 64 / n:16.1f} {av * 4 / (1024 * g) * 100:9.1f}% {g / us / 1e3:6.2f}")
PY
rm -rf gpurun_out/r03/pmc_rp
