#!/bin/bash
R=$PWD
mkdir -p $R/gpurun_out
ulimit -c 0
export HSA_ENABLE_COREDUMP=0
timeout 400 python3 tools/dbg/r5_cold.py 189 2>&1 | grep -v amdgpu.ids | cut -c1-200 | head -40 | tee $R/gpurun_out/round5_c5_cold.log
R5_CALLS=3 timeout 300 python3 tools/dbg/r5_c5full.py 189 2>&1 | grep -v amdgpu.ids | cut -c1-330 | tee $R/gpurun_out/r5l_c5full.log
timeout 600 python3 -m pytest tests/test_parity_gpu.py tests/test_fft_cpu.py -m gpu -x -q -k "focal or joint or fft" > $R/gpurun_out/r5l_tests.log 2>&1; tail -3 $R/gpurun_out/r5l_tests.log
