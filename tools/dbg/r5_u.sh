#!/bin/bash
# round 5: the whole GPU suite, smoke, and the default bench line as the driver runs it
ulimit -c 0
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r5u_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > gpurun_out/r5u_smoke.log
/usr/bin/time -v timeout 900 python bench.py > gpurun_out/r5u_bench.json 2> gpurun_out/r5u_bench.err
grep -E "Elapsed" gpurun_out/r5u_bench.err > gpurun_out/r5u_bench_wall.log
cat gpurun_out/r5u_tests.log gpurun_out/r5u_smoke.log gpurun_out/r5u_bench_wall.log
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r5u_bench.json") if l.startswith("{")][-1])
print(d["ms_per_step"], d["value"], d["roofline"].get("frac"), d["roofline"].get("step"))
for c in d["extra"]["configs"]:
    print(c)
PY
