#!/bin/bash
# does fetch volume buy clock on the (power-limited) complex GEMM?  The same launch under tile orders
# that change FETCH_SIZE (CPLXAMD_GEMM_GROUP_M sweep) and with B pre-loaded into the Infinity Cache; separate --pmc passes
# (kernel-trace only): FETCH_SIZE | WRITE_SIZE | GRBM_GUI_ACTIVE + SQ_VALU_MFMA_BUSY_CYCLES; wall time from the kernel trace.
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r06/cvf; mkdir -p $O
for LAYOUT in NN NT; do
for cfg in "1 0" "2 0" "4 0" "8 0" "16 0" "64 0" "4 1"; do
  set -- $cfg; GM=$1; TOUCH=$2
  for c in FETCH_SIZE WRITE_SIZE "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"; do
    n=$(echo $c | tr ' ' '_')
    LAYOUT=$LAYOUT TOUCH=$TOUCH CPLXAMD_GEMM_GROUP_M=$GM timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv \
      -d $O/${LAYOUT}_gm${GM}_t${TOUCH}_$n -- python $R/scripts/r06/gemm_nn_one.py > $O/${LAYOUT}_gm${GM}_t${TOUCH}_$n.log 2>&1
  done
done
done
cd $R
python - <<'PY' | tee gpurun_out/r06/gemm_clock_vs_fetch.txt
import csv, glob, re, statistics
O = "gpurun_out/r06/cvf"
rows = []
for lay in ("NN", "NT"):
  for gm, t in ((1,0),(2,0),(4,0),(8,0),(16,0),(64,0),(4,1)):
    val, dur = {}, []
    for f in glob.glob(f"{O}/{lay}_gm{gm}_t{t}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_bf16" in r["Kernel_Name"]:
                val.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for f in glob.glob(f"{O}/{lay}_gm{gm}_t{t}_GRBM*/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_bf16" in r["Kernel_Name"]:
                dur.append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
    med = lambda k: statistics.median(val[k][2:]) if k in val and len(val[k]) > 2 else float("nan")
    us = statistics.median(dur[2:]) if len(dur) > 2 else float("nan")
    cyc = med("GRBM_GUI_ACTIVE") / 8
    rows.append((lay, gm, t, med("FETCH_SIZE") * 2 * 1024 / 1e9, med("WRITE_SIZE") * 1024 / 1e9, us, cyc, cyc / us / 1e3, med("SQ_VALU_MFMA_BUSY_CYCLES") / 1024 / cyc))
print("# complex GEMM 8192 x 4096 x 4096 bf16 (NN = forward, NT = input gradient), 14 launches per pass, median after the first two")
print("# FETCH = FETCH_SIZE x 2 (gfx950 note) in GB, algorithmic operand bytes 0.201 GB + 0.134 GB written; cycles = GRBM_GUI_ACTIVE / 8 XCDs")
print(f"{'layout':6s} {'GROUP_M':>7s} {'B touched':>9s} {'FETCH GB':>9s} {'WRITE GB':>9s} {'us':>8s} {'cycles/XCD':>11s} {'GHz':>6s} {'MFMA busy':>9s}")
for r in rows:
    print(f"{r[0]:6s} {r[1]:7d} {r[2]:9d} {r[3]:9.3f} {r[4]:9.3f} {r[5]:8.1f} {r[6]:11.0f} {r[7]:6.3f} {r[8]:9.3f}")
PY
rm -rf $O
