#!/bin/bash
# round 5: the focal-plane knobs again, now that every role stream has a hardware queue of its own (bench.py --config c5, 2 steps)
ulimit -c 0
mkdir -p gpurun_out
L=gpurun_out/r5z_c5_knobs7.log
: > $L
run() { # label, env...
  label=$1; shift
  env "$@" timeout 300 python bench.py --config c5 --no-extra-configs --steps 4 --warmup 1 --no-cpu-baseline --no-cold > /tmp/o.json 2>/tmp/o.err
  python - "$label" <<'PY' >> gpurun_out/r5z_c5_knobs7.log
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
run "HSA_ENABLE_SDMA=1" HSA_ENABLE_SDMA=1
run "HSA_ENABLE_SDMA=0" HSA_ENABLE_SDMA=0
run "IMS_FOCAL_DIRECT_COPY=1" IMS_FOCAL_DIRECT_COPY=1
run "IMS_FOCAL_COARSE_SLICES=0" IMS_FOCAL_COARSE_SLICES=0
run "IMS_FOCAL_FFT=bulk" IMS_FOCAL_FFT=bulk
run "IMS_FOCAL_FFT=top" IMS_FOCAL_FFT=top
run "default again" X=1
cat $L
