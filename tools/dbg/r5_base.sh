#!/bin/bash
# round 5 baseline: GPU tests, then C5 on 64 CCDs: host/device trace of the joint path and the kernel trace per queue
R=$PWD
mkdir -p $R/gpurun_out
timeout 1500 python3 -m pytest tests -m gpu -x -q > $R/gpurun_out/r5_gputest.log 2>&1; tail -3 $R/gpurun_out/r5_gputest.log
IMS_FOCAL_TRACE=1 R4_SKIP_SINGLE=1 R4_CONC=4 timeout 600 python3 tools/dbg/r4_c5.py 64 > $R/gpurun_out/r5_c5_trace64.log 2>&1; tail -5 $R/gpurun_out/r5_c5_trace64.log
timeout 900 bash tools/dbg/r4_joint_trace.sh 16 64 > $R/gpurun_out/r5_c5_jt64.log 2>&1; tail -60 $R/gpurun_out/r5_c5_jt64.log
