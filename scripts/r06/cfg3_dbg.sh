#!/bin/bash
# A/B builds of the library (build_dbg/libcplxamd_<tag>.so): cfg3 kernel times by rocprofv3
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r06; mkdir -p $O
for d in default nt0 nt1 nt2 bnnt0; do
  L=$R/cplxmodule_amd/libcplxamd.so; [ $d != default ] && L=$R/build_dbg/libcplxamd_$d.so
  CPLXAMD_LIB=$L STEPS=8 rocprofv3 --kernel-trace --stats -d $O/prof_dbg -- python $R/scripts/r06/cfg3_step.py > $O/dbg.log 2>&1
  python $R/scripts/rocprof_summary.py $O/prof_dbg/*/*_results.db > $O/dbg_stats.txt
  echo "== $d $(grep ms_per_step $O/dbg.log)"; grep 'conv_cl_wgrad_kernel<true>\|bn_apply_rows\|bn_reduce_rows<unsigned short, 6' $O/dbg_stats.txt | cut -c1-150
  rm -rf $O/prof_dbg
done
