#!/bin/bash
# round 2, GPU call 1: ablations + cheap variants of the complex bf16 GEMM, in-process interleaved,
# then PMC passes (SQ wait / issue breakdown) of our kernel and the vendor's concatenated real GEMM.
cd /root/repo
mkdir -p gpurun_out/r02
L=cplxmodule_amd
python scripts/gemm_ab.py base=$L/libcplxamd.so nodma=$L/libcplxamd_nodma.so nobar=$L/libcplxamd_nobar.so \
   noepi=$L/libcplxamd_noepi.so noxor=$L/libcplxamd_noxor.so floor=$L/libcplxamd_floor.so prio=$L/libcplxamd_prio.so \
   u3=$L/libcplxamd_u3.so u3prio=$L/libcplxamd_u3prio.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02/gemm_ab1.txt
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum"; do
  i=$((i+1))
  ITERS=6 timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /root/repo/gpurun_out/pmcw_$i -- python /root/repo/scripts/gemm_vs_vendor_one.py > /root/repo/gpurun_out/pmcw_$i.log 2>&1
done
cd /root/repo
python - <<'PY' | tee gpurun_out/r02/gemm_pmc1.txt
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
grep -il "error\|fail" gpurun_out/pmcw_*.log | head
rm -rf gpurun_out/pmcw_*/
