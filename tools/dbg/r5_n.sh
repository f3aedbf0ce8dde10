#!/bin/bash
# round 5: ring per device + two-segment round kernel; parallel FFT warm-up cold start; coarse slices A/B
ulimit -c 0
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -k "focal or joint or fft or native or round or plan" 2>&1 | tail -5 > gpurun_out/r5n_tests.log
timeout 300 python tools/dbg/r5_cold.py 189 > gpurun_out/round5_c5_cold.log 2>&1
timeout 300 python tools/dbg/r5_c5full.py 189 > gpurun_out/r5n_c5_coarse1.log 2>&1
IMS_FOCAL_COARSE_SLICES=0 timeout 300 python tools/dbg/r5_c5full.py 189 > gpurun_out/r5n_c5_coarse0.log 2>&1
timeout 300 python bench.py --config c3 --no-extra-configs --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r5n_c3_seg1.log 2>&1
IMS_ROUND_TWO_SEGMENTS=1 IMS_BENCH_DUMP=gpurun_out/r5n_c3_two.npz timeout 300 python bench.py --config c3 --no-extra-configs --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r5n_c3_seg2.log 2>&1
IMS_BENCH_DUMP=gpurun_out/r5n_c3_one.npz timeout 300 python bench.py --config c3 --no-extra-configs --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python - <<'PY' > gpurun_out/r5n_c3_same.log 2>&1
import numpy as np
a = np.load("gpurun_out/r5n_c3_one.npz"); b = np.load("gpurun_out/r5n_c3_two.npz")
print({k: bool(np.array_equal(a[k], b[k])) for k in a.files})
PY
rm -f gpurun_out/r5n_c3_one.npz gpurun_out/r5n_c3_two.npz
timeout 300 python bench.py --config c5 --no-extra-configs --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r5n_c5_seg1.log 2>&1
IMS_ROUND_TWO_SEGMENTS=1 timeout 300 python bench.py --config c5 --no-extra-configs --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r5n_c5_seg2.log 2>&1
tail -3 gpurun_out/r5n_*.log gpurun_out/round5_c5_cold.log
