#!/bin/bash
# kernel-trace summary of tools/dbg/one_star.py -> gpurun_out/star_kernel_stats.txt
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/star_kt -- python3 $R/tools/dbg/one_star.py > $R/gpurun_out/star_kt.log 2>&1
grep "one star" $R/gpurun_out/star_kt.log
DB=$(find $R/gpurun_out/star_kt -name "*.db" | head -1)
python3 $R/tools/rocprof_summary.py $DB $R/gpurun_out/star_kernel_stats.txt > /dev/null
head -9 $R/gpurun_out/star_kernel_stats.txt
rm -rf $R/gpurun_out/star_kt
