#!/bin/bash
# HBM traffic of the complex bf16 GEMM (bench shape, forward launch): FETCH_SIZE and WRITE_SIZE in separate
# --pmc passes (TCC slots), kernel-trace only (MI355X_MICROARCH.md, HBM / rocprofv3 section: FETCH_SIZE counts
# 64 B per 128-B request on gfx950 -> doubled; WRITE_SIZE calibrated 1:1 on streaming writes in r01).
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /root/repo/gpurun_out/pmct_$c -- \
    python /root/repo/scripts/gemm_one_lib.py /root/repo/cplxmodule_amd/libcplxamd.so 12 > /root/repo/gpurun_out/pmct_$c.log 2>&1
done
cd /root/repo
python - <<'PY'
import glob, csv, json, statistics
val = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = []
    for f in glob.glob(f"gpurun_out/pmct_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_bf16" in r["Kernel_Name"] and r["Counter_Name"] == c:
                v.append(float(r["Counter_Value"]))
    val[c] = v[2:]            # drop the first (cold) launches
fetch = statistics.median(val["FETCH_SIZE"]) * 1024 * 2      # KiB, x2: gfx950 correction
write = statistics.median(val["WRITE_SIZE"]) * 1024
out = {"kernel": "gemm_bf16_kernel<bf16 out, CPLX> forward launch 8192x4096x4096", "launches": len(val["FETCH_SIZE"]),
       "FETCH_SIZE_KiB_median": statistics.median(val["FETCH_SIZE"]), "WRITE_SIZE_KiB_median": statistics.median(val["WRITE_SIZE"]),
       "fetch_bytes_corrected_x2": fetch, "write_bytes": write, "traffic_bytes_per_launch": fetch + write,
       "algorithmic_bytes": 2 * (8192 * 4096 + 4096 * 4096) * 2 + 2 * 8192 * 4096 * 2}
json.dump(out, open("gpurun_out/r02/gemm_traffic.json", "w"), indent=1)
print(json.dumps(out))
PY
rm -rf gpurun_out/pmct_*/
