#!/bin/bash
# round 5: a CCD's static pixel-boundary state not made (IMS_FOCAL_LAZY_STATIC): parity tests, C5 A/B, CRCs of all images
ulimit -c 0
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -k "lazy_static or focal or joint" 2>&1 | tail -15 > gpurun_out/lazy_tests.log
L=gpurun_out/round5_c5_lazy_static.log
: > $L
for v in 1 0 1 0; do
  IMS_FOCAL_LAZY_STATIC=$v timeout 300 python bench.py --config c5 --no-extra-configs --steps 4 --warmup 1 --no-cpu-baseline --no-cold > /tmp/o.json 2>/tmp/o.err
  python - "IMS_FOCAL_LAZY_STATIC=$v" <<'PY' >> gpurun_out/round5_c5_lazy_static.log
import json, sys
ok = False
for line in open("/tmp/o.json"):
    if line.startswith("{"):
        d = json.loads(line); print(sys.argv[1], round(d["ms_per_step"], 1), d["extra"].get("step_ms")); ok = True
if not ok: print(sys.argv[1], "FAILED", open("/tmp/o.err").read()[-800:])
PY
done
for v in 1 0; do
  IMS_FOCAL_LAZY_STATIC=$v IMS_BENCH_DUMP=gpurun_out/lazy_crc_$v.json timeout 300 python bench.py --config c5 --no-extra-configs --steps 1 --warmup 0 --no-cpu-baseline --no-cold > /dev/null 2>&1
done
python - <<'PY' >> $L
import json
a = json.load(open("gpurun_out/lazy_crc_1.json")); b = json.load(open("gpurun_out/lazy_crc_0.json"))
print("per-CCD CRCs of the 189 float32 images, static state not made vs made: equal =", a == b)
PY
cat gpurun_out/lazy_tests.log $L
