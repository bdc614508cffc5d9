#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r06; mkdir -p $O
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_kl -- python $R/scripts/r03/kl_pmc.py > $O/pmc_kl.log 2>&1
cd $R
python - <<'PY' | tee gpurun_out/r06/kl_pmc.txt
import csv, glob, collections
val = collections.defaultdict(list); dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/r06/pmc_kl/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "kl_kernel" in r["Kernel_Name"]:
            val[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for f in glob.glob("gpurun_out/r06/pmc_kl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "kl_kernel" in r["Kernel_Name"]:
            dur[r["Kernel_Name"].split("(")[0][-40:]].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
n = 8192 * 8192
med = lambda x: sorted(x)[len(x) // 2]
print("# KL kernels on a 8192^2 complex weight (kl_kernel<KIND, VALUE, GRAD>; KIND 2 = complex VD, 3 = complex ARD), rocprofv3 --pmc, median")
print(f"{'kernel':42s} {'calls':>5s} {'us':>8s} {'lane-ops/elt':>13s} {'VALU busy':>10s} {'GHz':>6s}")
for k in sorted(dur):
    iv = med(val[(k, "SQ_INSTS_VALU")]); av = med(val[(k, "SQ_ACTIVE_INST_VALU")]); g = med(val[(k, "GRBM_GUI_ACTIVE")]) / 8
    us = med(dur[k])
    print(f"{k:42s} {len(dur[k]):5d} {us:8.1f} {iv * 64 / n:13.1f} {av * 4 / (1024 * g) * 100:9.1f}% {g / us / 1e3:6.2f}")
PY
rm -rf gpurun_out/r06/pmc_kl
