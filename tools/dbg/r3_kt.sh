#!/bin/bash
# kernel-trace summary of the C3 bench step under the current environment -> gpurun_out/$1_kernel_stats.txt
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_$1 -- python3 $R/bench.py --config ${2:-c3} --steps 5 --warmup 2 --no-cpu-baseline --no-cold > $R/gpurun_out/kt_$1.log 2>&1
DB=$(find $R/gpurun_out/kt_$1 -name "*.db" | head -1)
python3 $R/tools/rocprof_summary.py $DB $R/gpurun_out/$1_kernel_stats.txt > /dev/null
head -${3:-14} $R/gpurun_out/$1_kernel_stats.txt | cut -c1-150
rm -rf $R/gpurun_out/kt_$1
