#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out
L=gpurun_out/r5ai_active_fraction.log
: > $L
run() { label=$1; shift
  env "$@" timeout 300 python bench.py --config c5 --no-extra-configs --steps 6 --warmup 1 --no-cpu-baseline --no-cold > /tmp/o.json 2>/tmp/o.err
  python - "$label" <<'PY' >> gpurun_out/r5ai_active_fraction.log
import json, sys, statistics
ok = False
for line in open("/tmp/o.json"):
    if line.startswith("{"):
        d = json.loads(line); st = d["extra"].get("step_ms"); print(sys.argv[1], "mean", round(d["ms_per_step"], 1), "median", statistics.median(st), "min", min(st)); ok = True
if not ok: print(sys.argv[1], "FAILED", open("/tmp/o.err").read()[-300:])
PY
}
run "fraction 0.25 (default)" X=1
run "fraction 0.5" IMS_ACTIVE_FRACTION=0.5
run "fraction 1.0" IMS_ACTIVE_FRACTION=1.0
run "fraction 0.125" IMS_ACTIVE_FRACTION=0.125
run "list_min 256" IMS_JOINT_LIST_MIN=256
run "list_min 4096" IMS_JOINT_LIST_MIN=4096
run "upd_dpp_max 512" IMS_UPD_DPP_MAX=512
run "fraction 0.25 again" X=1
cat $L
