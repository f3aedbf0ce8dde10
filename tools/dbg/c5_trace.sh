#!/bin/bash
# kernel trace of the C5 step on the first N CCDs: per-queue busy fractions (tools/dbg/c5_queue_busy.py)   (under gpurun)
ulimit -c 0
R=$PWD
cd /tmp && export TMPDIR=/tmp
export IMS_C5_CCDS=${1:-64}
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/c5_kt -- python3 $R/bench.py --config c5 --no-extra-configs --steps 2 --warmup 1 --no-cpu-baseline --no-cold > $R/gpurun_out/c5_kt.log 2>&1
python3 $R/tools/dbg/c5_queue_busy.py "$R/gpurun_out/c5_kt/*/*kernel_trace.csv" 0.3
rm -rf $R/gpurun_out/c5_kt
