#!/bin/bash
# kernel trace of a few CCDs of the C5 step: tools/dbg/c5_trace.sh <n_ccd>   (under gpurun)
R=$PWD
cd /tmp && export TMPDIR=/tmp
C5_ONLY=${2:-3} rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/c5_kt -- python3 $R/tools/dbg/c5_profile.py ${1:-12} > $R/gpurun_out/c5_kt.log 2>&1
python3 $R/tools/dbg/c5_queues.py "$R/gpurun_out/c5_kt/*/*kernel_trace.csv"
grep "concurrent" $R/gpurun_out/c5_kt.log
rm -rf $R/gpurun_out/c5_kt
