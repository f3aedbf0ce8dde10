#!/bin/bash
# every memory copy of a few focal-plane steps with its duration (rocprofv3 --memory-copy-trace): which image copies are the slow ones?   (under gpurun)
ulimit -c 0
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/c5_copy
R5_CALLS=${1:-8} timeout 900 rocprofv3 --memory-copy-trace --output-format csv -d /tmp/c5_copy -- python3 $R/tools/dbg/c5_full.py 189 > $R/gpurun_out/c5_copy_trace.log 2>&1
grep "call" $R/gpurun_out/c5_copy_trace.log | cut -c1-70
F=$(find /tmp/c5_copy -name "*memory_copy_trace.csv" | head -1)
head -3 $F
python3 - "$F" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print(len(rows), "copies; columns:", list(rows[0].keys()))
big = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = int(r.get("Bytes", r.get("Size", 0)) or 0) if any(k in r for k in ("Bytes", "Size")) else 0
    big.append((s, e - s, r.get("Direction", ""), n, r.get("Source_Agent_Id", ""), r.get("Destination_Agent_Id", "")))
big.sort()
img = [b for b in big if b[1] > 800_000]
print(len(img), "copies longer than 0.8 ms; duration histogram [ms]:")
h = collections.Counter(round(b[1] / 1e6 * 2) / 2 for b in img)
print(sorted(h.items()))
print("by direction:", collections.Counter(b[2] for b in img))
# runs of slow copies in time order
t0 = img[0][0]
print("time [ms], duration [ms] of the image-sized copies of the LAST 400 ms of the trace:")
tend = img[-1][0]
print(" ".join(f"{(b[0]-t0)/1e6:.0f}:{b[1]/1e6:.1f}" for b in img if b[0] > tend - 400e6))
PY
rm -rf /tmp/c5_copy
