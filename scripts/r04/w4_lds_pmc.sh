#!/bin/bash
# LDS-side counters of the w4 launches (complex / real forward): bank-conflict cycles against the cycles the LDS was busy.
R="${GRAFT_REPO_ROOT:-/root/repo}"; out=$R/gpurun_out/r04; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for s in c_fwd r_fwd r_wgrad_kl; do
  d=$out/ldspmc_$s
  timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE \
      --kernel-trace --output-format csv -d $d -- python $R/scripts/r04/w4_one.py $R/cplxmodule_amd/libcplxamd.so 1 $s 8 > $d.log 2>&1
done
cd $R
python - <<'PY' | tee $out/w4_lds_pmc.txt
import glob, csv, collections, re
val = collections.defaultdict(list)
for f in glob.glob("gpurun_out/r04/ldspmc_*/**/*counter_collection.csv", recursive=True):
    key = re.search(r"ldspmc_(\w+?)/", f).group(1)
    for r in csv.DictReader(open(f)):
        if "gemm_bf16" in r["Kernel_Name"]:
            val[(key, r["Counter_Name"])].append(float(r["Counter_Value"]))
med = lambda x: sorted(x)[len(x) // 2] if x else float("nan")
names = sorted({k[1] for k in val})
print("shape        " + " ".join(f"{n:>22s}" for n in names))
for s in sorted({k[0] for k in val}):
    print(f"{s:12s} " + " ".join(f"{med(val[(s, n)][2:]):22.4g}" for n in names))
PY
rm -rf $out/ldspmc_*/
