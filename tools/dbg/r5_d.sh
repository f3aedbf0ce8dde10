#!/bin/bash
# round 5: host profile of a steady-state C5 step, the focal trace, and the per-queue kernel trace of the new pipeline
R=$PWD
mkdir -p $R/gpurun_out
ulimit -c 0
export HSA_ENABLE_COREDUMP=0
timeout 300 python3 -m pytest tests/test_parity_gpu.py tests/test_fft_cpu.py -m gpu -x -q -k "fft or focal" > $R/gpurun_out/r5d_tests.log 2>&1; tail -3 $R/gpurun_out/r5d_tests.log
timeout 200 python3 tools/dbg/r5_c5.py 64 profile > $R/gpurun_out/r5d_profile.log 2>&1; grep -v amdgpu.ids $R/gpurun_out/r5d_profile.log | cut -c1-220 | head -75
IMS_FOCAL_TRACE=1 timeout 200 python3 tools/dbg/r5_c5.py 64 > $R/gpurun_out/r5d_trace.log 2>&1; grep -c CCD $R/gpurun_out/r5d_trace.log
timeout 200 python3 tools/dbg/r5_cold.py 48 > $R/gpurun_out/r5d_cold.log 2>&1; grep -v amdgpu.ids $R/gpurun_out/r5d_cold.log | cut -c1-200 | head -12
timeout 600 bash tools/dbg/r4_joint_trace.sh 16 64 > $R/gpurun_out/r5d_jt64.log 2>&1; tail -70 $R/gpurun_out/r5d_jt64.log | cut -c1-400
