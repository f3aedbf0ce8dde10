#!/bin/bash
# kernel-trace summary of one bench config: bash tools/dbg/kt.sh <tag> [config] -> gpurun_out/<tag>_<config>_kernel_stats.txt
TAG=${1:-kt}
CFG=${2:-c3}
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_${CFG}_kt -- python3 $R/bench.py --config $CFG --steps 5 --warmup 2 --no-cpu-baseline --no-cold > $R/gpurun_out/${TAG}_${CFG}_kt.log 2>&1
DB=$(find $R/gpurun_out/${TAG}_${CFG}_kt -name "*.db" | head -1)
python3 $R/tools/rocprof_summary.py $DB $R/gpurun_out/${TAG}_${CFG}_kernel_stats.txt > /dev/null
rm -rf $R/gpurun_out/${TAG}_${CFG}_kt
tail -1 $R/gpurun_out/${TAG}_${CFG}_kt.log | cut -c1-200
head -9 $R/gpurun_out/${TAG}_${CFG}_kernel_stats.txt | cut -c1-160
