#!/bin/bash
# round 5: new focal-plane defaults (four batches alive, photon kernels capped at three workgroups per CU): tests, C5, memory, cold call
ulimit -c 0
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -k "focal or joint or arena" 2>&1 | tail -3 > gpurun_out/r5aa_tests.log
timeout 300 python bench.py --config c5 --no-extra-configs --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r5aa_c5.json 2> gpurun_out/r5aa_c5.err
R5_CALLS=3 timeout 300 python tools/dbg/r5_c5full.py 189 2>&1 | grep -v amdgpu.ids | cut -c1-300 > gpurun_out/r5aa_c5full.log
timeout 300 python tools/dbg/r5_cold.py 189 2>&1 | grep -v amdgpu.ids | head -12 | cut -c1-160 > gpurun_out/r5aa_cold.log
R5_COLD_IMMEDIATE=1 timeout 300 python tools/dbg/r5_cold.py 189 2>&1 | grep -v amdgpu.ids | head -4 | cut -c1-160 >> gpurun_out/r5aa_cold.log
timeout 300 python bench.py --config c3 --no-extra-configs --steps 6 --warmup 2 --no-cpu-baseline --no-cold > gpurun_out/r5aa_c3.json 2>/dev/null
cat gpurun_out/r5aa_tests.log gpurun_out/r5aa_c5full.log gpurun_out/r5aa_cold.log
python - <<'PY'
import json
for f in ("gpurun_out/r5aa_c5.json", "gpurun_out/r5aa_c3.json"):
    for line in open(f):
        if line.startswith("{"):
            d = json.loads(line); print(f, round(d["ms_per_step"], 2), d["value"], d.get("extra", {}).get("step_ms"))
PY
