#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r5an_tests.log
L=gpurun_out/r5an_c4_overlap.log
: > $L
for v in 1 0 1 0; do
  IMS_POOL_OVERLAP=$v IMS_BENCH_DUMP=gpurun_out/r5an_img_$v.npz timeout 600 python bench.py --config c4 --no-extra-configs --steps 4 --warmup 1 --no-cpu-baseline --no-cold > /tmp/o.json 2>/tmp/o.err
  python - "IMS_POOL_OVERLAP=$v" <<'PY' >> gpurun_out/r5an_c4_overlap.log
import json, sys
ok = False
for line in open("/tmp/o.json"):
    if line.startswith("{"):
        d = json.loads(line); print(sys.argv[1], round(d["ms_per_step"], 1), d["roofline"].get("timed_launches_per_step"), d["roofline"].get("mean_launch_ms")); ok = True
if not ok: print(sys.argv[1], "FAILED", open("/tmp/o.err").read()[-800:])
PY
done
python - <<'PY' >> $L
import numpy as np
a = np.load("gpurun_out/r5an_img_1.npz"); b = np.load("gpurun_out/r5an_img_0.npz")
print("C4 image, overlapped vs one launch:", {k: bool(np.array_equal(a[k], b[k])) for k in a.files})
PY
rm -f gpurun_out/r5an_img_*.npz
cat gpurun_out/r5an_tests.log $L
