#!/bin/bash
# kernel timeline of the headline step, eager: plain and with the RCCL exchange forced in a world of one
cd /tmp && export TMPDIR=/tmp
R=/root/repo
rm -rf $R/gpurun_out/tl_plain $R/gpurun_out/tl_rccl
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_plain -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline > $R/gpurun_out/tl_plain.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_rccl -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --force-collectives > $R/gpurun_out/tl_rccl.log 2>&1
cd $R
python scripts/r03/step_timeline.py gpurun_out/tl_plain > gpurun_out/tl_plain.txt
python scripts/r03/step_timeline.py gpurun_out/tl_rccl > gpurun_out/tl_rccl.txt
rm -rf gpurun_out/tl_plain gpurun_out/tl_rccl
tail -25 gpurun_out/tl_plain.txt; tail -32 gpurun_out/tl_rccl.txt
