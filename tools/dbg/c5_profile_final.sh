#!/bin/bash
# the C5 records of profiles/: bench line (189 CCDs), kernel stats (189 CCDs, 7 calls), SQ / HBM counters (48 CCDs)
ulimit -c 0
R=$PWD
mkdir -p $R/gpurun_out
python3 $R/bench.py --config c5 --no-extra-configs --steps 5 --warmup 2 > $R/gpurun_out/round5_c5_bench.json 2> $R/gpurun_out/round5_c5_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/round5_c5_kt -- python3 $R/bench.py --config c5 --no-extra-configs --steps 5 --warmup 2 --no-cpu-baseline --no-cold > $R/gpurun_out/round5_c5_kt.log 2>&1
DB=$(find $R/gpurun_out/round5_c5_kt -name "*.db" | head -1)
python3 $R/tools/rocprof_summary.py $DB $R/gpurun_out/round5_c5_kernel_stats.txt > /dev/null
rm -rf $R/gpurun_out/round5_c5_kt
cd $R
bash tools/dbg/c5_counters.sh > /dev/null 2>&1
head -16 gpurun_out/round5_c5_kernel_stats.txt | cut -c1-170
cut -c1-400 gpurun_out/round5_c5_bench.json
