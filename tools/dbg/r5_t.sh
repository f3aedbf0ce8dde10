#!/bin/bash
# round 5: the rocFFT kernel file and the first call of a fresh process when NOTHING hides the warm-up (R5_COLD_IMMEDIATE)
ulimit -c 0
mkdir -p gpurun_out
rm -rf /tmp/imsim_amd_* ~/.cache/imsim_amd 2>/dev/null
L=gpurun_out/round5_rocfft_kernel_cache.log
echo "== first call right behind the warm-up's start, no kernel file" > $L
R5_COLD_IMMEDIATE=1 ROCFFT_RTC_CACHE_PATH=/tmp/none_$$.db timeout 300 python tools/dbg/r5_cold.py 189 2>&1 | grep -v amdgpu.ids | head -9 | cut -c1-160 >> $L
echo "== making the kernel file" >> $L
timeout 600 python tools/make_fft_cache.py gpurun_out/rocfft_kernels.db 2>&1 | grep -v amdgpu.ids >> $L
mkdir -p imsim_amd/lib && cp gpurun_out/rocfft_kernels.db imsim_amd/lib/rocfft_kernels.db
echo "== first call right behind the warm-up's start, a fresh process that starts from the seed in lib/" >> $L
rm -rf /tmp/imsim_amd_* ~/.cache/imsim_amd 2>/dev/null
R5_COLD_IMMEDIATE=1 timeout 300 python tools/dbg/r5_cold.py 189 2>&1 | grep -v amdgpu.ids | head -9 | cut -c1-160 >> $L
echo "== the usual order (warm-up first thing, set-up, first call), kernel file there" >> $L
timeout 300 python tools/dbg/r5_cold.py 189 2>&1 | grep -v amdgpu.ids | head -4 | cut -c1-160 >> $L
cat $L
