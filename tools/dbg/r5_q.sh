#!/bin/bash
# round 5: does the order of first use of the focal-plane role streams decide the C5 step time? (bench.py --config c5, 2 steps)
ulimit -c 0
mkdir -p gpurun_out
L=gpurun_out/r5q_c5_stream_order.log
: > $L
run() { # label, env...
  label=$1; shift
  env "$@" timeout 300 python bench.py --config c5 --no-extra-configs --steps 2 --warmup 1 --no-cpu-baseline --no-cold > /tmp/o.json 2>/tmp/o.err
  python - "$label" <<'PY' >> gpurun_out/r5q_c5_stream_order.log
import json, sys
ok = False
for line in open("/tmp/o.json"):
    if line.startswith("{"):
        d = json.loads(line); print(sys.argv[1], d["ms_per_step"], d["extra"].get("step_ms")); ok = True
if not ok:
    print(sys.argv[1], "FAILED", open("/tmp/o.err").read()[-400:])
PY
}
run "early warm, no touch" X=1
run "early warm, top0,top1,bulk,mid" IMS_FOCAL_TOUCH=top0,top1,bulk,mid
run "early warm, bulk,mid,top0,top1" IMS_FOCAL_TOUCH=bulk,mid,top0,top1
run "early warm, mid,bulk,top1,top0" IMS_FOCAL_TOUCH=mid,bulk,top1,top0
run "early warm, top0,bulk,top1,mid" IMS_FOCAL_TOUCH=top0,bulk,top1,mid
run "early warm, null,top0,top1,bulk,mid" IMS_FOCAL_TOUCH=null,top0,top1,bulk,mid
run "late warm, no touch" IMS_BENCH_EARLY_WARM=0
run "late warm, top0,top1,bulk,mid" IMS_BENCH_EARLY_WARM=0 IMS_FOCAL_TOUCH=top0,top1,bulk,mid
run "late warm, bulk,mid,top0,top1" IMS_BENCH_EARLY_WARM=0 IMS_FOCAL_TOUCH=bulk,mid,top0,top1
run "late warm, mid,top1,bulk,top0" IMS_BENCH_EARLY_WARM=0 IMS_FOCAL_TOUCH=mid,top1,bulk,top0
cat $L
