#!/bin/bash
# round 5: tile lists of a joint round appended to by the pixel search (IMS_JOINT_SEARCH_LISTS): parity tests, C5 A/B
ulimit -c 0
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q -k "focal or joint" 2>&1 | tail -3 > gpurun_out/r5ag_tests.log
L=gpurun_out/r5ag_search_lists.log
: > $L
for v in 1 0 1 0 1 0; do
  IMS_JOINT_SEARCH_LISTS=$v timeout 300 python bench.py --config c5 --no-extra-configs --steps 3 --warmup 1 --no-cpu-baseline --no-cold > /tmp/o.json 2>/tmp/o.err
  python - "IMS_JOINT_SEARCH_LISTS=$v" <<'PY' >> gpurun_out/r5ag_search_lists.log
import json, sys
ok = False
for line in open("/tmp/o.json"):
    if line.startswith("{"):
        d = json.loads(line); print(sys.argv[1], round(d["ms_per_step"], 1), d["extra"].get("step_ms")); ok = True
if not ok:
    print(sys.argv[1], "FAILED", open("/tmp/o.err").read()[-600:])
PY
done
for v in 1 0; do
  IMS_JOINT_SEARCH_LISTS=$v IMS_BENCH_DUMP=gpurun_out/r5ag_crc_$v.json timeout 300 python bench.py --config c5 --no-extra-configs --steps 1 --warmup 0 --no-cpu-baseline --no-cold > /dev/null 2>&1
done
python - <<'PY' >> $L
import json
a = json.load(open("gpurun_out/r5ag_crc_1.json")); b = json.load(open("gpurun_out/r5ag_crc_0.json"))
print("per-CCD CRCs of the 189 float32 images, search-side lists vs list builder: equal =", a == b)
PY
cat gpurun_out/r5ag_tests.log $L
