#!/bin/bash
ulimit -c 0
R=$PWD
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r5aj_tests.log
timeout 300 python bench.py --config c5 --no-extra-configs --steps 6 --warmup 1 --no-cpu-baseline --no-cold > gpurun_out/r5aj_c5.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
export IMS_C5_CCDS=64
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r5aj_kt -- python3 $R/bench.py --config c5 --no-extra-configs --steps 2 --warmup 1 --no-cpu-baseline --no-cold > $R/gpurun_out/r5aj_kt.log 2>&1
DB=$(find $R/gpurun_out/r5aj_kt -name "*.db" | head -1)
python3 $R/tools/rocprof_summary.py $DB $R/gpurun_out/r5aj_kernel_stats.txt > /dev/null
rm -rf $R/gpurun_out/r5aj_kt
cd $R
cat gpurun_out/r5aj_tests.log; head -12 gpurun_out/r5aj_kernel_stats.txt | cut -c1-160
python - <<'PY'
import json, statistics
d = json.loads([l for l in open("gpurun_out/r5aj_c5.json") if l.startswith("{")][-1]); st = d["extra"]["step_ms"]
print("c5", round(d["ms_per_step"], 1), "median", statistics.median(st), st)
PY
