#!/bin/bash
# round 5: C3 step time for every order of first use of the four plan streams in use (chain, bulk, chain1, chain2)
ulimit -c 0
mkdir -p gpurun_out
L=gpurun_out/r5w_c3_stream_order.log
: > $L
for order in $(python - <<'PY'
import itertools
print(" ".join(",".join(p) for p in itertools.permutations(["0", "1", "2", "3"])))
PY
); do
  IMS_STREAM_TOUCH=$order timeout 300 python bench.py --config c3 --no-extra-configs --steps 6 --warmup 2 --no-cpu-baseline --no-cold > /tmp/o.json 2>/tmp/o.err
  python - "$order" <<'PY' >> gpurun_out/r5w_c3_stream_order.log
import json, sys
for line in open("/tmp/o.json"):
    if line.startswith("{"):
        d = json.loads(line); print(sys.argv[1], round(d["ms_per_step"], 2))
PY
done
sort -k2 -n $L
