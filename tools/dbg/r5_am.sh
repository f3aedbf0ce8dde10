#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out
L=gpurun_out/r5am_c3_lds.log
: > $L
run() { label=$1; shift
  env "$@" timeout 300 python bench.py --config c3 --no-extra-configs --steps 8 --warmup 2 --no-cpu-baseline --no-cold > /tmp/o.json 2>/tmp/o.err
  python - "$label" <<'PY' >> gpurun_out/r5am_c3_lds.log
import json, sys
ok = False
for line in open("/tmp/o.json"):
    if line.startswith("{"):
        d = json.loads(line); print(sys.argv[1], round(d["ms_per_step"], 2)); ok = True
if not ok: print(sys.argv[1], "FAILED", open("/tmp/o.err").read()[-300:])
PY
}
run "c3 default" X=1
run "c3 IMS_PHOTON_LDS=41984" IMS_PHOTON_LDS=41984
run "c3 IMS_PHOTON_LDS=33000" IMS_PHOTON_LDS=33000
run "c3 IMS_STREAM_PRIORITIES=-1,0,-1,0,0" IMS_STREAM_PRIORITIES=-1,0,-1,0,0
run "c3 IMS_CHAIN_CLASSES=40" IMS_CHAIN_CLASSES=40
run "c3 IMS_CHAIN_CLASSES=100,10" IMS_CHAIN_CLASSES=100,10
run "c3 default again" X=1
timeout 300 python bench.py --config c3b --no-extra-configs --steps 4 --warmup 1 --no-cpu-baseline --no-cold 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(\"c3b default\", round(d[\"ms_per_step\"],2))" >> $L
cat $L
