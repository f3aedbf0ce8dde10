#!/bin/bash
R=$PWD
mkdir -p $R/gpurun_out
ulimit -c 0
export HSA_ENABLE_COREDUMP=0
timeout 1500 python3 -m pytest tests -m gpu -x -q > $R/gpurun_out/r5j_tests.log 2>&1; tail -3 $R/gpurun_out/r5j_tests.log
timeout 100 python3 tools/dbg/r5_fftinit.py first 2>&1 | grep -v amdgpu.ids | tee $R/gpurun_out/r5j_fftinit1.log
timeout 100 python3 tools/dbg/r5_fftinit.py second 2>&1 | grep -v amdgpu.ids | tee $R/gpurun_out/r5j_fftinit2.log
for V in "X=1" "IMS_FOCAL_DIRECT_COPY=0" "IMS_PHOTON_LDS=41984"; do
  env $V R5_CALLS=3 timeout 300 python3 tools/dbg/r5_c5full.py 189 2>&1 | grep -v amdgpu.ids | cut -c1-330 | tee -a $R/gpurun_out/r5j_c5full.log
done
( time python3 bench.py --steps 10 --warmup 3 ) > $R/gpurun_out/r5j_bench.json 2> $R/gpurun_out/r5j_bench.err; tail -5 $R/gpurun_out/r5j_bench.err; python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r5j_bench.json').read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["roofline"].get("step"))
for k,v in d["extra"].get("configs",{}).items(): print(k, {a:b for a,b in v.items() if a in ("ms_per_step","objects_per_s","bit_identical","within_tolerance","wall_s","failed","skipped","stderr_tail")})
print({k:v for k,v in d["extra"].items() if k!="configs"})
PY
