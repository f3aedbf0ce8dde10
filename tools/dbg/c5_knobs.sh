#!/bin/bash
# round 5: the focal-plane knobs again, now that every role stream has a hardware queue of its own (bench.py --config c5, 2 steps)
ulimit -c 0
mkdir -p gpurun_out
L=gpurun_out/r5z_c5_knobs6.log
: > $L
run() { # label, env...
  label=$1; shift
  env "$@" timeout 300 python bench.py --config c5 --no-extra-configs --steps 4 --warmup 1 --no-cpu-baseline --no-cold > /tmp/o.json 2>/tmp/o.err
  python - "$label" <<'PY' >> gpurun_out/r5z_c5_knobs6.log
import json, sys
ok = False
for line in open("/tmp/o.json"):
    if line.startswith("{"):
        d = json.loads(line); print(sys.argv[1], round(d["ms_per_step"], 1), d["extra"].get("step_ms")); ok = True
if not ok:
    print(sys.argv[1], "FAILED", open("/tmp/o.err").read()[-300:])
PY
}
run "default" X=1
run "ALIVE=3" IMS_FOCAL_ALIVE=3
run "ALIVE=5" IMS_FOCAL_ALIVE=5
run "JOINT=16" IMS_FOCAL_JOINT=16
run "JOINT=24" IMS_FOCAL_JOINT=24
run "JOINT=21" IMS_FOCAL_JOINT=21
run "no LDS cap" IMS_FOCAL_PHOTON_LDS=
run "LDS 57344" IMS_FOCAL_PHOTON_LDS=57344
run "AHEAD=pre:2" IMS_FOCAL_AHEAD=pre:2
run "AHEAD=pre:0" IMS_FOCAL_AHEAD=pre:0
run "JOINT_INIT=bulk" IMS_FOCAL_JOINT_INIT=bulk
run "default again" X=1
cat $L
