#!/bin/bash
# round 5: the rocFFT kernel file (made here, then used by fresh processes) and the cold start of a visit with and without it
ulimit -c 0
mkdir -p gpurun_out
rm -rf /tmp/imsim_amd_* ~/.cache/imsim_amd 2>/dev/null
L=gpurun_out/round5_rocfft_kernel_cache.log
: > $L
echo "== fresh box, no kernel file (cold start of a visit: warm-up first thing, catalogs, then the first call)" >> $L
ROCFFT_RTC_CACHE_PATH=/tmp/none_$$.db timeout 300 python tools/dbg/r5_cold.py 189 2>&1 | grep -v amdgpu.ids | head -12 | cut -c1-160 >> $L
echo "== making the kernel file" >> $L
timeout 600 python tools/make_fft_cache.py gpurun_out/rocfft_kernels.db 2>&1 | grep -v amdgpu.ids >> $L
mkdir -p imsim_amd/lib && cp gpurun_out/rocfft_kernels.db imsim_amd/lib/rocfft_kernels.db
echo "== a fresh process that starts from the seed in lib/ (no ROCFFT_RTC_CACHE_PATH set)" >> $L
rm -rf /tmp/imsim_amd_* ~/.cache/imsim_amd 2>/dev/null
timeout 300 python tools/dbg/r5_cold.py 189 2>&1 | grep -v amdgpu.ids | head -12 | cut -c1-160 >> $L
echo "== and the next one (its own kernel file is there)" >> $L
timeout 300 python tools/dbg/r5_cold.py 189 2>&1 | grep -v amdgpu.ids | head -4 | cut -c1-160 >> $L
timeout 300 python bench.py --config c5 --no-extra-configs --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r5s_c5.log 2>&1
timeout 300 python bench.py --config fft --no-extra-configs --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r5s_fft.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q -k "focal or fft or two_segment" 2>&1 | tail -3 >> $L
python - <<'PY' >> $L
import json
for f in ("gpurun_out/r5s_c5.log", "gpurun_out/r5s_fft.log"):
    for line in open(f):
        if line.startswith("{"):
            d = json.loads(line); print(f, d["ms_per_step"], d.get("extra", {}).get("step_ms"))
PY
cat $L
