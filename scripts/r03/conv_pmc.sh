#!/bin/bash
# PMC view of the generic NCHW conv kernels on cfg5's six layers (scripts/r03/cfg5_conv_layers.py): clock, VALU / MFMA busy,
# wave wait share per (mode, grid).  Two counter passes (own runs, kernel trace only).
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r03; mkdir -p $O
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc_conv1 -- python $R/scripts/r03/cfg5_conv_layers.py > $O/pmc_conv.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_conv2 -- python $R/scripts/r03/cfg5_conv_layers.py >> $O/pmc_conv.log 2>&1
cd $R
python - <<'PY' | tee gpurun_out/r03/conv_pmc.txt
import csv, glob, collections, re
val = collections.defaultdict(list); dur = collections.defaultdict(list)
def key(r):
    m = re.search(r"conv_kernel<float, true, (\d), (\d+)", r["Kernel_Name"])
    if not m: return None
    if "Grid_Size" in r:
        return (int(m.group(1)), int(r["Grid_Size"]) // int(r["Workgroup_Size"]))
    return (int(m.group(1)), int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // 256)
for d in ("pmc_conv1", "pmc_conv2"):
    for f in glob.glob(f"gpurun_out/r03/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = key(r)
            if k: val[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
for f in glob.glob("gpurun_out/r03/pmc_conv1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = key(r)
        if k: dur[k].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
med = lambda x: sorted(x)[len(x) // 2] if x else float("nan")
print("# conv_kernel<float, complex, MODE, 32, TN, NARROW> on cfg5's layers (MODE 0 fwd, 1 dgrad, 2 wgrad), rocprofv3 --pmc, medians per launch")
print(f"{'mode':>4s} {'wgs':>6s} {'us':>7s} {'GHz':>5s} {'VALU inst/wave':>15s} {'VALU busy':>10s} {'MFMA busy':>10s} {'wave wait':>10s}")
for k in sorted(dur):
    g = med(val[(k, "GRBM_GUI_ACTIVE")]) / 8; us = med(dur[k])
    iv = med(val[(k, "SQ_INSTS_VALU")]); av = med(val[(k, "SQ_ACTIVE_INST_VALU")])
    mf = med(val[(k, "SQ_VALU_MFMA_BUSY_CYCLES")]); bc = med(val[(k, "SQ_BUSY_CYCLES")])
    wc = med(val[(k, "SQ_WAVE_CYCLES")]); wa = med(val[(k, "SQ_WAIT_INST_ANY")])
    waves = k[1] * 4
    print(f"{k[0]:4d} {k[1]:6d} {us:7.1f} {g / us / 1e3:5.2f} {iv / waves:15.0f} {av * 4 / (1024 * g) * 100:9.1f}% {mf / (1024 * g) * 100:9.1f}% {wa / wc * 100 if wc else float('nan'):9.1f}%")
PY
rm -rf gpurun_out/r03/pmc_conv1 gpurun_out/r03/pmc_conv2
