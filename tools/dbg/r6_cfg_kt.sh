#!/bin/bash
# kernel-trace stats of one bench config: bash tools/dbg/r6_cfg_kt.sh <config> <steps> [tag]
CFG=$1; S=${2:-3}; TAG=${3:-r6}
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_$CFG
rocprofv3 --kernel-trace --stats -d /tmp/kt_$CFG -- python3 $R/bench.py --config $CFG --steps $S --warmup 1 --no-cpu-baseline --no-cold --no-extra-configs > $R/gpurun_out/${TAG}_${CFG}_kt.log 2>&1
tail -1 $R/gpurun_out/${TAG}_${CFG}_kt.log | cut -c1-300
DB=$(find /tmp/kt_$CFG -name "*.db" | head -1)
python3 $R/tools/rocprof_summary.py $DB $R/gpurun_out/${TAG}_${CFG}_kernel_stats.txt > /dev/null
head -24 $R/gpurun_out/${TAG}_${CFG}_kernel_stats.txt | cut -c1-160
