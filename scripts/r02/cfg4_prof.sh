#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd - >/dev/null
rm -rf gpurun_out/cfg4prof; mkdir -p gpurun_out/cfg4prof
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/cfg4prof -o cfg4 -- python scripts/bench_configs.py --only cfg4b > gpurun_out/cfg4prof/log.txt 2>&1
tail -1 gpurun_out/cfg4prof/log.txt | cut -c1-200
DB=$(ls gpurun_out/cfg4prof/*.db gpurun_out/cfg4prof/*/*.db 2>/dev/null | head -1)
python scripts/rocprof_summary.py $DB > gpurun_out/cfg4_kernel_stats.txt
python - "$DB" <<'P'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
for name, n, tot in c.execute("select name, count(*), sum(end-start) from kernels where name like '%at::%' group by name order by 3 desc limit 8"):
    print(n, round(tot/1e6, 3), name[:400])
P
rm -rf gpurun_out/cfg4prof
head -24 gpurun_out/cfg4_kernel_stats.txt | cut -c1-140
