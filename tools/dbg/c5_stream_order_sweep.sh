#!/bin/bash
# round 5: C5 step time for every order of first use of the four role streams (bench.py --config c5, early warm-up, 2 steps)
ulimit -c 0
mkdir -p gpurun_out
L=gpurun_out/r5r_c5_stream_order_pre.log
: > $L
for order in $(python - <<'PY'
import itertools
print(" ".join(",".join(p) for p in itertools.permutations(["top0", "pre", "bulk", "mid"])))
PY
); do
  IMS_FOCAL_TOUCH=$order timeout 300 python bench.py --config c5 --no-extra-configs --steps 2 --warmup 1 --no-cpu-baseline --no-cold > /tmp/o.json 2>/tmp/o.err
  python - "$order" <<'PY' >> gpurun_out/r5r_c5_stream_order_pre.log
import json, sys
for line in open("/tmp/o.json"):
    if line.startswith("{"):
        d = json.loads(line); print(sys.argv[1], d["ms_per_step"], d["extra"].get("step_ms"))
PY
done
sort -k2 -n $L
