#!/bin/bash
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ctl -- python3 $R/tools/dbg/top41.py $1 > /dev/null 2>&1
python3 - "$R/gpurun_out/ctl" <<'PY'
import csv, glob, sys, collections
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0])))
K = [(r["Kernel_Name"].split("(")[0].replace("void ", "")[:22], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)) for r in rows]
K.sort(key=lambda k: k[1])
acc = [k for k in K if k[0].startswith("k_accumulate_round")]
n = len(acc) // 3
acc = acc[-n:]                       # the last replay
t0 = acc[0][1]
step = [k for k in K if k[1] >= t0 - 3e6]
print("rounds", n, "span ms", (acc[-1][2] - t0) / 1e6)
for i in range(0, n, 12):
    a = acc[i]
    nxt = acc[i + 1][1] if i + 1 < n else a[2]
    inside = [k for k in step if a[1] <= k[1] < nxt]
    print(f"  round {i:3d} at {(a[1]-t0)/1e6:6.2f} ms, {a[3]:5d} wgs: " + "  ".join(f"{k[0][:14]} {(k[2]-k[1])/1e3:5.1f}" for k in inside) + f"  | total {(nxt-a[1])/1e3:6.1f} us")
PY
rm -rf $R/gpurun_out/ctl
