#!/bin/bash
# round 5: the focal-plane knobs again, now that every role stream has a hardware queue of its own (bench.py --config c5, 2 steps)
ulimit -c 0
mkdir -p gpurun_out
L=gpurun_out/r5z_c5_knobs3.log
: > $L
run() { # label, env...
  label=$1; shift
  env "$@" timeout 300 python bench.py --config c5 --no-extra-configs --steps 2 --warmup 1 --no-cpu-baseline --no-cold > /tmp/o.json 2>/tmp/o.err
  python - "$label" <<'PY' >> gpurun_out/r5z_c5_knobs3.log
import json, sys
ok = False
for line in open("/tmp/o.json"):
    if line.startswith("{"):
        d = json.loads(line); print(sys.argv[1], round(d["ms_per_step"], 1), d["extra"].get("step_ms")); ok = True
if not ok:
    print(sys.argv[1], "FAILED", open("/tmp/o.err").read()[-300:])
PY
}
B="IMS_FOCAL_ALIVE=4 IMS_PHOTON_LDS=41984"
run "ALIVE=4 LDS (one joint stream)" $B
run "2 joint streams, bulk,top0,top1,mid,pre" $B IMS_FOCAL_JOINT_STREAMS=2 IMS_FOCAL_TOUCH=bulk,top0,top1,mid,pre
run "2 joint streams, mid,top0,pre,bulk,top1" $B IMS_FOCAL_JOINT_STREAMS=2 IMS_FOCAL_TOUCH=mid,top0,pre,bulk,top1
run "2 joint streams, top0,top1,pre,bulk,mid" $B IMS_FOCAL_JOINT_STREAMS=2 IMS_FOCAL_TOUCH=top0,top1,pre,bulk,mid
run "2 joint streams, pre,top0,top1,bulk,mid" $B IMS_FOCAL_JOINT_STREAMS=2 IMS_FOCAL_TOUCH=pre,top0,top1,bulk,mid
run "2 joint streams, mid,top0,top1,bulk,pre" $B IMS_FOCAL_JOINT_STREAMS=2 IMS_FOCAL_TOUCH=mid,top0,top1,bulk,pre
run "2 joint streams, top0,pre,bulk,mid,top1" $B IMS_FOCAL_JOINT_STREAMS=2 IMS_FOCAL_TOUCH=top0,pre,bulk,mid,top1
run "2 joint streams, bulk,top0,top1,mid,pre, ALIVE=5" IMS_FOCAL_ALIVE=5 IMS_PHOTON_LDS=41984 IMS_FOCAL_JOINT_STREAMS=2 IMS_FOCAL_TOUCH=bulk,top0,top1,mid,pre
run "ALIVE=3 JOINT=24 LDS (one joint stream)" IMS_FOCAL_ALIVE=3 IMS_FOCAL_JOINT=24 IMS_PHOTON_LDS=41984
run "ALIVE=4 JOINT=24 LDS" IMS_FOCAL_ALIVE=4 IMS_FOCAL_JOINT=24 IMS_PHOTON_LDS=41984
run "ALIVE=5 LDS" IMS_FOCAL_ALIVE=5 IMS_PHOTON_LDS=41984
run "ALIVE=4 LDS=57344" IMS_FOCAL_ALIVE=4 IMS_PHOTON_LDS=57344
cat $L
