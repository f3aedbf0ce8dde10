"""Per hardware queue of a rocprofv3 kernel trace (csv): busy fraction over the last `span` of the trace, the kernels that hold it, and how
many queues are busy at a time.  python tools/dbg/c5_queue_busy.py "<glob of kernel_trace.csv>" [fraction of the trace, from its end]"""
import csv
import glob
import sys
from collections import defaultdict

rows = []
for f in glob.glob(sys.argv[1]):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"].split("(")[0][:40]))
rows.sort()
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
t_end = max(r[1] for r in rows)
t_beg = t_end - int(frac * (t_end - rows[0][0]))
rows = [r for r in rows if r[0] >= t_beg]
span = t_end - t_beg
print(f"window: the last {1e-6 * span:.1f} ms of the trace, {len(rows)} kernels")
by_q = defaultdict(list)
for a, b, q, k in rows:
    by_q[q].append((a, b, k))
for q, ks in sorted(by_q.items()):
    busy = 0
    last = 0
    per = defaultdict(int)
    for a, b, k in ks:
        a = max(a, last)
        if b > a:
            busy += b - a
            per[k] += b - a
            last = b
    top = ", ".join(f"{k} {100 * v / span:.0f}%" for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:4])
    print(f"queue {q}: busy {100 * busy / span:5.1f}%  ({len(ks)} kernels)  {top}")
ev = []
for a, b, q, k in rows:
    ev.append((a, 1, q))
    ev.append((b, -1, q))
ev.sort()
depth = defaultdict(int)
active = defaultdict(int)
last_t = t_beg
hist = defaultdict(int)
for t, d, q in ev:
    n = sum(1 for v in active.values() if v > 0)
    hist[n] += t - last_t
    last_t = t
    active[q] += d
print("queues busy at a time:", ", ".join(f"{n}: {100 * v / span:.0f}%" for n, v in sorted(hist.items())))
