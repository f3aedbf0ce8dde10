#!/bin/bash
# kernel-trace stats of the C5 step over n CCDs for the values of one knob:  bash tools/dbg/r6_kt.sh <n_ccd> <ENV_NAME> <v1> <v2> ...
N=${1:-64}; NAME=$2; shift 2
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for V in "$@"; do
  export $NAME=$V
  export R5_CALLS=3
  rm -rf /tmp/kt_$V
  rocprofv3 --kernel-trace --stats -d /tmp/kt_$V -- python3 $R/tools/dbg/c5_full.py $N > $R/gpurun_out/r6_kt_${NAME}_$V.log 2>&1
  grep "call" $R/gpurun_out/r6_kt_${NAME}_$V.log
  DB=$(find /tmp/kt_$V -name "*.db" | head -1)
  python3 $R/tools/rocprof_summary.py $DB $R/gpurun_out/r6_kt_${NAME}_${V}_kernel_stats.txt > /dev/null
  head -16 $R/gpurun_out/r6_kt_${NAME}_${V}_kernel_stats.txt | cut -c1-150
done
