#!/bin/bash
# round 5: tile lists appended to by the pixel search (IMS_JOINT_SEARCH_LISTS=1) against the list builder launch (0)
ulimit -c 0
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "joint or focal" 2>&1 | tail -2
L=gpurun_out/r5ah_search_lists_long.log
: > $L
for v in 1 0 1 0; do
  IMS_JOINT_SEARCH_LISTS=$v timeout 300 python bench.py --config c5 --no-extra-configs --steps 8 --warmup 1 --no-cpu-baseline --no-cold > /tmp/o.json 2>/tmp/o.err
  python - "IMS_JOINT_SEARCH_LISTS=$v" <<'PY' >> gpurun_out/r5ah_search_lists_long.log
import json, sys, statistics
for line in open("/tmp/o.json"):
    if line.startswith("{"):
        d = json.loads(line); st = d["extra"].get("step_ms"); print(sys.argv[1], "mean", round(d["ms_per_step"], 1), "median", statistics.median(st), "min", min(st), st)
PY
done
cat $L
