#!/bin/bash
# PMC table of the channels-last convolution kernels on cfg3's layer (B = 64): clock (GRBM_GUI_ACTIVE per XCD), MFMA pipe
# busy, instruction mix, waits, LDS bank conflicts.  Counter groups in separate passes, kernel-trace only.
cd /tmp && export TMPDIR=/tmp
R=/root/repo
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  for e in cplxamd_conv2d_cl2 cplxamd_conv2d_cl; do
    ENTRY=$e PAD=0 ONLY="cl kernel fwd,cl wgrad" timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmcc_${e}_$i -- python $R/scripts/conv_cl_bench.py 64 > $R/gpurun_out/pmcc_${e}_$i.log 2>&1
  done
done
cd $R
python - <<'PY' | tee gpurun_out/conv_pmc.txt
import glob, csv, collections, re
val = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmcc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        name = "conv_cl2 (fwd)" if "conv_cl2_kernel" in k else "conv_cl (fwd)" if "conv_cl_kernel" in k else "conv_cl_wgrad" if "conv_cl_wgrad_kernel" in k else None
        if name:
            val[(r["Counter_Name"], name)].append(float(r["Counter_Value"]))
names = sorted({n for n, _ in val}); vs = sorted({v for _, v in val})
print("# cfg3 layer (64 -> 64 channels, 3 x 3, 256 x 256, B = 64, padding 0), average per launch; rocprofv3 --pmc, kernel-trace only")
print(f"{'counter':28s}" + "".join(f"{v:>18s}" for v in vs))
for n in names:
    print(f"{n:28s}" + "".join(f"{(sum(val[(n,v)])/max(len(val[(n,v)]),1)):18.5g}" for v in vs))
PY
rm -rf gpurun_out/pmcc_*/
