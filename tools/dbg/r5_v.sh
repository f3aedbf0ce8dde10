#!/bin/bash
# round 5: the default bench line as the driver runs it (+ C5 and FFT lines for profiles/)
ulimit -c 0
mkdir -p gpurun_out
t0=$(date +%s.%N)
timeout 900 python bench.py > gpurun_out/r5u_bench.json 2> gpurun_out/r5u_bench.err
echo "default bench wall: $(echo "$(date +%s.%N) - $t0" | bc) s" > gpurun_out/r5u_bench_wall.log
timeout 600 python bench.py --config c5 --no-extra-configs > gpurun_out/r5u_c5.json 2> /dev/null
cat gpurun_out/r5u_bench_wall.log
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r5u_bench.json") if l.startswith("{")][-1])
print(d["ms_per_step"], d["value"], d["roofline"].get("frac"), d["roofline"].get("step"))
for c in d["extra"]["configs"]:
    print(c)
d = json.loads([l for l in open("gpurun_out/r5u_c5.json") if l.startswith("{")][-1])
print("c5", d["ms_per_step"], d["value"], d["extra"].get("step_ms"), d["roofline"], d.get("cpu_baseline"))
PY
