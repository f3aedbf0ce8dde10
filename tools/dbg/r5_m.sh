#!/bin/bash
R=$PWD
mkdir -p $R/gpurun_out
ulimit -c 0
export HSA_ENABLE_COREDUMP=0
timeout 600 python3 -m pytest tests/test_parity_gpu.py tests/test_fft_cpu.py tests/test_native_plan.py -m gpu -x -q -k "focal or joint or fft or native" > $R/gpurun_out/r5m_tests.log 2>&1; tail -3 $R/gpurun_out/r5m_tests.log
timeout 400 python3 tools/dbg/r5_cold.py 189 2>&1 | grep -v amdgpu.ids | cut -c1-200 | head -14 | tee $R/gpurun_out/round5_c5_cold.log
for V in "X=1" "IMS_FOCAL_COARSE_SLICES=0"; do
env $V R5_CALLS=3 timeout 300 python3 tools/dbg/r5_c5full.py 189 2>&1 | grep -v amdgpu.ids | cut -c1-330 | tee -a $R/gpurun_out/r5m_c5full.log
done
