#!/bin/bash
# Cycles, clock and matrix-pipe occupancy of the persistent one-wave-per-SIMD GEMM against the one-tile form and the 8-wave
# family (separate --pmc pass per launch kind, kernel trace only: gpurun rule).
#   usage: scripts/r05/w4p_pmc.sh <tag> "<shapes>" "<forms>" [B I O]      -> gpurun_out/r05/pmc_<tag>.txt
tag=$1; shapes=${2:-"c_fwd0 r_fwd"}; forms=${3:-"w8 w4 w4p"}; shift 3
R="${GRAFT_REPO_ROOT:-/root/repo}"
out=$R/gpurun_out/r05; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for s in $shapes; do for f in $forms; do
  d=$out/pmc_${tag}_${s}_$f
  timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
      --kernel-trace --output-format csv -d $d -- python $R/scripts/r05/w4p_one.py $f $s 8 "$@" > $d.log 2>&1
done; done
cd $R
python - "$tag" <<'PY' | tee $out/pmc_$tag.txt
import glob, csv, collections, re, sys
tag = sys.argv[1]
val = collections.defaultdict(list); dur = collections.defaultdict(list)
for f in glob.glob(f"gpurun_out/r05/pmc_{tag}_*/**/*counter_collection.csv", recursive=True):
    key = re.search(rf"pmc_{tag}_(\w+?)_(w8|w4p|w4)/", f).groups()
    for r in csv.DictReader(open(f)):
        if "gemm_bf16" in r["Kernel_Name"]:
            val[key + (r["Counter_Name"],)].append(float(r["Counter_Value"]))
for f in glob.glob(f"gpurun_out/r05/pmc_{tag}_*/**/*kernel_trace.csv", recursive=True):
    key = re.search(rf"pmc_{tag}_(\w+?)_(w8|w4p|w4)/", f).groups()
    for r in csv.DictReader(open(f)):
        if "gemm_bf16" in r["Kernel_Name"]:
            dur[key].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
med = lambda x: sorted(x)[len(x) // 2] if x else float("nan")
print(f"# {tag}: median of launches 3..8 under rocprofv3 --pmc (profiled runs clock lower than unprofiled ones)")
print(f"{'shape':10s} {'form':>4s} {'us':>8s} {'cyc/XCD':>10s} {'GHz':>6s} {'MFMA busy':>10s} {'wave cyc':>10s} {'wait_any':>9s} {'wait_inst':>9s} {'active':>8s}")
for key in sorted(dur):
    g = lambda c: med(val[key + (c,)][2:])
    cyc = g("GRBM_GUI_ACTIVE") / 8
    us = med(dur[key][2:])
    busy = g("SQ_VALU_MFMA_BUSY_CYCLES") / 1024 / cyc
    wc = g("SQ_WAVE_CYCLES")
    print(f"{key[0]:10s} {key[1]:>4s} {us:8.1f} {cyc:10.4g} {cyc / us / 1e3:6.2f} {100 * busy:9.1f}% {wc:10.4g} "
          f"{100 * g('SQ_WAIT_ANY') / wc:8.1f}% {100 * g('SQ_WAIT_INST_ANY') / wc:8.1f}% {100 * g('SQ_ACTIVE_INST_ANY') / wc:7.1f}%")
PY
rm -rf $out/pmc_${tag}_*/
