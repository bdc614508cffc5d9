#!/bin/bash
# device durations (rocprofv3 kernel trace, median over ~55 launches) of the generic conv kernels on cfg5's six layers,
# for the library in $CPLXAMD_LIB (default: the in-tree one).   bash scripts/r03/conv_layers_prof.sh [label]
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r03; mkdir -p $O
D=$O/prof_cl_$$
rocprofv3 --kernel-trace --output-format csv -d $D -- python $R/scripts/r03/cfg5_conv_layers.py > /dev/null 2>&1
python - "$D" "${1:-in-tree}" <<'PY'
import csv, glob, sys, re, collections
d = collections.OrderedDict()
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    for r in rows:
        m = re.search(r"conv_kernel<float, true, (\d), (\d+), (\d+), (\d)>", r["Kernel_Name"])
        if not m: continue
        wgs = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // 256
        d.setdefault((int(m.group(1)), wgs, m.group(2) + "x" + m.group(3) + ("n" + m.group(4) if m.group(4) != "0" else "")), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
med = lambda x: sorted(x)[len(x) // 2]
print(f"# {sys.argv[2]}: mode/wgs/tile -> median us  (0 fwd, 1 dgrad, 2 wgrad; layers in launch order)")
print("  ".join(f"{k[0]}/{k[1]}/{k[2]}:{med(v):.1f}" for k, v in d.items()))
PY
rm -rf $D
