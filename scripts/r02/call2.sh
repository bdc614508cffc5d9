#!/bin/bash
# round 2, GPU call 2: full GPU test suite (new parity tests), GEMM ablations on the unrolled-ring kernel,
# PMC clock / MFMA-busy / wait breakdown per build, a bench run.
cd /root/repo
mkdir -p gpurun_out/r02
timeout 600 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v amdgpu.ids | tail -40 > gpurun_out/r02/pytest_gpu.txt
tail -15 gpurun_out/r02/pytest_gpu.txt
L=cplxmodule_amd
ONLY=c_fwd,c_dgrad,c_wgrad timeout 300 python scripts/gemm_ab.py base=$L/libcplxamd.so u3=$L/libcplxamd_u3.so u3hot=$L/libcplxamd_u3hot.so \
   u3nowait=$L/libcplxamd_u3nowait.so u3nodma=$L/libcplxamd_u3nodma.so u3noepi=$L/libcplxamd_u3noepi.so \
   u3floor=$L/libcplxamd_u3floor.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02/gemm_ab2.txt
timeout 300 python bench.py --steps 50 --warmup 10 2>/dev/null | tee gpurun_out/r02/bench_a.json
cd /tmp && export TMPDIR=/tmp
for v in "" _u3 _u3nodma _u3floor; do
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
             "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /root/repo/gpurun_out/pmcv${v}_$i -- python /root/repo/scripts/gemm_one_lib.py /root/repo/cplxmodule_amd/libcplxamd$v.so 8 > /root/repo/gpurun_out/pmcv${v}_$i.log 2>&1
  done
done
cd /root/repo
python - <<'PY' | tee gpurun_out/r02/gemm_pmc2.txt
import glob, csv, collections, re
val = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmcv*/**/*counter_collection.csv", recursive=True):
    v = re.search(r"pmcv(_\w+?)?_\d", f).group(1) or "_base"
    for r in csv.DictReader(open(f)):
        if "gemm_bf16" in r["Kernel_Name"]:
            val[(r["Counter_Name"], v)].append(float(r["Counter_Value"]))
names = sorted({n for n, _ in val}); vs = sorted({v for _, v in val})
print(f"{'counter':28s}" + "".join(f"{v:>16s}" for v in vs))
for n in names:
    print(f"{n:28s}" + "".join(f"{(sum(val[(n,v)])/max(len(val[(n,v)]),1)):16.5g}" for v in vs))
PY
rm -rf gpurun_out/pmcv*/
