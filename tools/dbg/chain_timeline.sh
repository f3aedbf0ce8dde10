#!/bin/bash
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ctl -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-cold > /dev/null 2>&1
python3 $R/tools/dbg/chain_timeline.py "$R/gpurun_out/ctl/**/*kernel_trace.csv"
rm -rf $R/gpurun_out/ctl
