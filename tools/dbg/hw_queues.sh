#!/bin/bash
# round 5: more hardware queues than HIP's four, now that the order of first use is fixed (C3 and C5)
ulimit -c 0
mkdir -p gpurun_out
L=gpurun_out/r5x_hw_queues.log
: > $L
for q in 4 5 6 8; do
  GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --config c3 --no-extra-configs --steps 6 --warmup 2 --no-cpu-baseline --no-cold > /tmp/o.json 2>/tmp/o.err
  python - "c3 GPU_MAX_HW_QUEUES=$q" <<'PY' >> gpurun_out/r5x_hw_queues.log
import json, sys
for line in open("/tmp/o.json"):
    if line.startswith("{"):
        d = json.loads(line); print(sys.argv[1], round(d["ms_per_step"], 2))
PY
done
for q in 4 5 6 8; do
  GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --config c5 --no-extra-configs --steps 2 --warmup 1 --no-cpu-baseline --no-cold > /tmp/o.json 2>/tmp/o.err
  python - "c5 GPU_MAX_HW_QUEUES=$q" <<'PY' >> gpurun_out/r5x_hw_queues.log
import json, sys
for line in open("/tmp/o.json"):
    if line.startswith("{"):
        d = json.loads(line); print(sys.argv[1], round(d["ms_per_step"], 1), d["extra"].get("step_ms"))
PY
done
cat $L
