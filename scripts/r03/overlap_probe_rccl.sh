#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=/root/repo
for m in "side 1 hi" "side 1 lo" "side 0 hi"; do
  rm -rf $R/gpurun_out/ov
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ov -- python $R/scripts/r03/overlap_probe_rccl.py $m > $R/gpurun_out/ov.log 2>&1
  echo "== $m"
  python - <<PY
import csv, glob
rows = []
for f in glob.glob("$R/gpurun_out/ov/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Queue_Id", "?")))
rows.sort()
rows = [r for r in rows if "gemm_bf16" in r[2] or "Reduce" in r[2]]
last = rows[-6:]
t0 = last[0][0]
for s, e, n, q in last:
    print(f"  {(s - t0) / 1e3:8.1f} .. {(e - t0) / 1e3:8.1f} us  q{q}  {n}")
PY
done
rm -rf $R/gpurun_out/ov
