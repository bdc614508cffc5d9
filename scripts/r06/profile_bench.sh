#!/bin/bash
# Round-6 evidence for bench.py (run on the GPU box, outputs under gpurun_out/r06p/ -> copied to profiles/):
#  1. rocprofv3 --kernel-trace --stats of the DEFAULT bench command           -> bench_n1_kernel_stats.txt (+ the bench line)
#  2. separate --pmc passes (kernel-trace only) of the eager bench step: FETCH_SIZE / WRITE_SIZE per GEMM launch kind
#     (FETCH_SIZE doubled per the gfx950 note) and MFMA-pipe busy (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / per-XCD cycles)
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r06p; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/prof_bench -- python $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python $R/scripts/rocprof_summary.py $O/prof_bench/*/*_results.db > $O/bench_n1_kernel_stats.txt
rm -rf $O/prof_bench
# the same command without the extra points of the line (cfg2 / cfg3 / cfg4 / HBM sizes / CPU leg share kernel names with the
# headline step and would blur its per-kernel averages): the headline step's kernels only
rocprofv3 --kernel-trace --stats -d $O/prof_head -- python $R/bench.py --no-cpu-baseline > $O/bench_n1_headline.json 2> $O/bench_n1_headline.err
python $R/scripts/rocprof_summary.py $O/prof_head/*/*_results.db > $O/bench_n1_headline_kernel_stats.txt
rm -rf $O/prof_head
for c in FETCH_SIZE WRITE_SIZE "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"; do
  n=$(echo $c | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$n -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --graph off > $O/pmc_$n.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, json, re, statistics
O = "gpurun_out/r06p"
kind = lambda n: ("cgemm_NN" if re.search(r"w4_kernel<unsigned short, true, false, false, false", n) else
                  "cgemm_NT" if re.search(r"w4_kernel<unsigned short, true, true, false, true", n) else
                  "cgemm_TT" if re.search(r"w4_kernel<float, true, true, true, true", n) else
                  "rgemm_NN" if re.search(r"w4_kernel<unsigned short, false, false, false, false|w4p_kernel<false, false, false>", n) else
                  "rgemm_NT" if re.search(r"(w4|persist)_kernel<unsigned short, false, false, false, true|w4p_kernel<false, false, true>", n) else
                  "rgemm_TT" if re.search(r"w4_kernel<float, false, false, true, true", n) else None)
val = {}
for f in glob.glob(f"{O}/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = kind(r["Kernel_Name"])
        if k:
            val.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
med = lambda k, c: statistics.median(val[(k, c)][2:]) if (k, c) in val else None
B, I, Oo = 8192, 4096, 4096
alg_c = {"cgemm_NN": 2 * (B * I + Oo * I) * 2 + 2 * B * Oo * 2, "cgemm_NT": 2 * (B * Oo + Oo * I) * 2 + 2 * B * I * 2,
         "cgemm_TT": 2 * (B * Oo + B * I) * 2 + 2 * Oo * I * 4 * 2}      # TT: fp32 out + the accumulate operand it reads
out = {"command": "bench.py --steps 6 --warmup 3 --no-cpu-baseline --graph off (eager), rocprofv3 --pmc <counter> --kernel-trace, one pass per counter set; median over the launches after the first two",
       "note": "FETCH_SIZE in KiB, doubled (gfx950: 64 B tallied per 128-B request, MI355X_MICROARCH.md); WRITE_SIZE in KiB; Infinity-Cache hits are not excluded",
       "per_launch": {}, "detail": {}}
for k in ("cgemm_NN", "cgemm_NT", "cgemm_TT", "rgemm_NN", "rgemm_NT", "rgemm_TT"):
    f, w = med(k, "FETCH_SIZE"), med(k, "WRITE_SIZE")
    g, m = med(k, "GRBM_GUI_ACTIVE"), med(k, "SQ_VALU_MFMA_BUSY_CYCLES")
    d = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w}
    if f is not None and w is not None:
        d["traffic_bytes"] = f * 1024 * 2 + w * 1024
        if k in alg_c:
            out["per_launch"][k] = d["traffic_bytes"]
            d["algorithmic_bytes"] = alg_c[k]
    if g and m:
        d["cycles_per_xcd"] = g / 8
        d["mfma_pipe_busy"] = round(m / 1024 / (g / 8), 4)
    out["detail"][k] = d
json.dump(out, open(f"{O}/gemm_traffic.json", "w"), indent=1)
print(json.dumps(out["detail"], indent=1))
PY
rm -rf $O/pmc_*/
head -c 400 $O/bench_n1.json; echo; head -25 $O/bench_n1_kernel_stats.txt | cut -c1-200
