#!/bin/bash
# end of a round: the GPU suite, smoke, and the default bench line as the driver runs it
ulimit -c 0
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/final_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > gpurun_out/final_smoke.log
SECONDS=0
timeout 900 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
echo "default bench wall: $SECONDS s" > gpurun_out/final_bench_wall.log
cat gpurun_out/final_tests.log gpurun_out/final_smoke.log gpurun_out/final_bench_wall.log
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/final_bench.json") if l.startswith("{")][-1])
print(d["ms_per_step"], d["value"], d["roofline"].get("frac"), d["roofline"].get("step", {}).get("frac"))
for k, c in d["extra"]["configs"].items():
    print(k, c.get("ms_per_step"), c.get("objects_per_s"), c.get("bit_identical"), c.get("wall_s"))
print(d["cpu_baseline"]["parity"])
PY
