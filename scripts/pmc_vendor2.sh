cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU" "SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_ACTIVE_INST_SCA" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  ITERS=6 timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /root/repo/gpurun_out/pmcw_$i -- python /root/repo/scripts/gemm_vs_vendor_one.py > /root/repo/gpurun_out/pmcw_$i.log 2>&1
done
cd /root/repo
python - <<'PY'
import glob, csv, collections
val = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmcw_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = "ours" if "gemm_bf16" in r["Kernel_Name"] else ("vendor" if "Cijk" in r["Kernel_Name"] else None)
        if k:
            val[(r["Counter_Name"], k)].append(float(r["Counter_Value"]))
names = sorted({n for n, _ in val})
print(f"{'counter':32s} {'ours':>14s} {'vendor':>14s}")
for n in names:
    o, v = val.get((n, "ours"), [0]), val.get((n, "vendor"), [0])
    print(f"{n:32s} {sum(o)/len(o):14.5g} {sum(v)/len(v):14.5g}")
PY
rm -rf gpurun_out/pmcw_*/
