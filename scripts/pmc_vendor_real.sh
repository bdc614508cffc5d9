cd /tmp && export TMPDIR=/tmp
for c in GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /root/repo/gpurun_out/pmcr_$c -- python /root/repo/scripts/rgemm_vs_vendor_one.py > /root/repo/gpurun_out/pmcr_$c.log 2>&1
done
cd /root/repo
python - <<'PY'
import glob, csv, collections
for c in ("GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES"):
    dur = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmcr_{c}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"][:60]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    val = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmcr_{c}/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        if rows and c == "GRBM_GUI_ACTIVE":
            print("columns:", list(rows[0].keys()))
        for r in rows:
            val[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    for k in val:
        if "gemm" in k.lower() or "cijk" in k.lower():
            d = dur.get(k, [0])
            print(c, k, f"n={len(val[k])} avg={sum(val[k])/len(val[k]):.5g} dur_avg={sum(d)/len(d):.0f} ns")
PY
rm -rf gpurun_out/pmcr_*/
