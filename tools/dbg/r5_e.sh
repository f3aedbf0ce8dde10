#!/bin/bash
R=$PWD
mkdir -p $R/gpurun_out
ulimit -c 0
timeout 200 python3 tools/dbg/r5_fftcold.py 2>&1 | grep -v amdgpu.ids | tee $R/gpurun_out/r5e_fftcold.log
