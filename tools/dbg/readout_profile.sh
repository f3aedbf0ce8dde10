#!/bin/bash
# kernel trace of the CCD readout chain (tools/dbg/readout_time.py): which kernels its 1.5 ms per CCD are   (under gpurun)
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ro_kt
python3 $R/tools/dbg/readout_time.py
rocprofv3 --kernel-trace --stats -d /tmp/ro_kt -- python3 $R/tools/dbg/readout_time.py > $R/gpurun_out/readout_kt.log 2>&1
DB=$(find /tmp/ro_kt -name "*.db" | head -1)
python3 $R/tools/rocprof_summary.py $DB $R/gpurun_out/readout_kernel_stats.txt > /dev/null
head -30 $R/gpurun_out/readout_kernel_stats.txt | cut -c1-170
rm -rf /tmp/ro_kt
