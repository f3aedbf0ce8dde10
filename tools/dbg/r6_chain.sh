#!/bin/bash
# kernel trace of the C5 step over n CCDs -> chain_stats
N=${1:-64}
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export R5_CALLS=3
rm -rf /tmp/kt_c
rocprofv3 --kernel-trace --stats -d /tmp/kt_c -- python3 $R/tools/dbg/c5_full.py $N > $R/gpurun_out/r6_chain.log 2>&1
grep "call" $R/gpurun_out/r6_chain.log
DB=$(find /tmp/kt_c -name "*.db" | head -1)
python3 $R/tools/dbg/chain_stats.py $DB 0.3 | tee $R/gpurun_out/r6_chain_stats_$N.txt
