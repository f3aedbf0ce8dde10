#!/bin/bash
# round 5: why bench.py --config c5 takes 2.4 s a step where tools/dbg/r5_c5full.py takes 1.84 s on the same tree
ulimit -c 0
mkdir -p gpurun_out
L=gpurun_out/r5p_c5_bench_variants.log
: > $L
run() { # label, env...
  label=$1; shift
  env "$@" timeout 300 python bench.py --config c5 --no-extra-configs --steps 2 --warmup 1 --no-cpu-baseline --no-cold > /tmp/o.json 2>/tmp/o.err
  python - "$label" <<'PY' >> gpurun_out/r5p_c5_bench_variants.log
import json, sys
for line in open("/tmp/o.json"):
    if line.startswith("{"):
        d = json.loads(line); print(sys.argv[1], d["ms_per_step"], d["extra"].get("step_ms"))
PY
}
run "as is" X=1
run "IMS_FOCAL_TOPS=4 exported" IMS_FOCAL_TOPS=4
run "no early warm-up" IMS_BENCH_EARLY_WARM=0
run "no early warm-up, TOPS=4" IMS_BENCH_EARLY_WARM=0 IMS_FOCAL_TOPS=4
run "TOPS=2 everywhere, no early warm-up" IMS_BENCH_EARLY_WARM=0 IMS_FOCAL_TOPS=2
cat $L
# the rocFFT kernel cache on disk: does a second process find the first one's kernels?
export ROCFFT_RTC_CACHE_PATH=/tmp/ims_rocfft_cache.db
python tools/dbg/r5_fftinit.py > gpurun_out/r5p_fft_cache_first.log 2>&1
python tools/dbg/r5_fftinit.py > gpurun_out/r5p_fft_cache_second.log 2>&1
ls -la /tmp/ims_rocfft_cache.db >> gpurun_out/r5p_fft_cache_second.log
grep -v amdgpu.ids gpurun_out/r5p_fft_cache_first.log gpurun_out/r5p_fft_cache_second.log
cp /tmp/ims_rocfft_cache.db gpurun_out/ 2>/dev/null
