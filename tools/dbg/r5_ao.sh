#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out
L=gpurun_out/r5ao_c4_overlap_lds.log
: > $L
run() { label=$1; shift
  env "$@" timeout 600 python bench.py --config c4 --no-extra-configs --steps 3 --warmup 1 --no-cpu-baseline --no-cold > /tmp/o.json 2>/tmp/o.err
  python - "$label" <<'PY' >> gpurun_out/r5ao_c4_overlap_lds.log
import json, sys
ok = False
for line in open("/tmp/o.json"):
    if line.startswith("{"):
        d = json.loads(line); print(sys.argv[1], round(d["ms_per_step"], 1), d["roofline"].get("timed_launches_per_step"), round(d["roofline"].get("mean_launch_ms") or 0, 2)); ok = True
if not ok: print(sys.argv[1], "FAILED", open("/tmp/o.err").read()[-500:])
PY
}
run "overlap, LDS 41984" IMS_POOL_OVERLAP=1 IMS_PHOTON_LDS=41984
run "overlap, LDS 57344" IMS_POOL_OVERLAP=1 IMS_PHOTON_LDS=57344
run "one launch, LDS 41984" IMS_POOL_OVERLAP=0 IMS_PHOTON_LDS=41984
run "overlap, small max 640" IMS_POOL_OVERLAP=1 IMS_POOL_SMALL_MAX=640
cat $L
