"""Percentiles of the round kernels of a C5 kernel trace and the idle time of the queue that runs them:
   python tools/dbg/chain_stats.py <results.db> [last fraction of the trace, default 0.3]"""
import sqlite3
import sys
from collections import defaultdict

import numpy as np

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
if not cols:
    cols = [d[0] for d in cur.execute("select * from kernels limit 1").description]
print("# columns:", cols)
qcol = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else None)
scol = "stream_id" if "stream_id" in cols else None
key = qcol or scol
rows = list(cur.execute(f"select name, start, end, {key} from kernels order by start"))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
t_end = max(r[2] for r in rows)
t_beg = t_end - int(frac * (t_end - rows[0][1]))
rows = [r for r in rows if r[1] >= t_beg]
span = t_end - t_beg
print(f"# window: the last {span * 1e-6:.1f} ms, {len(rows)} kernels")
by = defaultdict(list)
for n, a, b, q in rows:
    by[n.split("(")[0][:48]].append(b - a)
print("%-50s %7s %9s %8s %8s %8s %8s %8s  %s" % ("kernel", "calls", "total_ms", "p10", "p50", "p90", "p99", "max", "share of its time in launches > 4 x p50"))
for n, d in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:12]:
    d = np.array(d, dtype=np.float64) * 1e-3
    p = np.percentile(d, [10, 50, 90, 99])
    print("%-50s %7d %9.1f %8.1f %8.1f %8.1f %8.1f %8.1f  %.2f" % (n, len(d), d.sum() * 1e-3, p[0], p[1], p[2], p[3], d.max(), d[d > 4 * p[1]].sum() / d.sum()))
# the queue of the rounds
byq = defaultdict(list)
for n, a, b, q in rows:
    byq[q].append((a, b, n.split("(")[0][:40]))
for q, ks in byq.items():
    n_round = sum(1 for k in ks if "k_accumulate_round_j" in k[2])
    if n_round < 100:
        continue
    ks.sort()
    busy = sum(b - a for a, b, _ in ks)
    gaps = np.array([max(ks[i + 1][0] - ks[i][1], 0) for i in range(len(ks) - 1)], dtype=np.float64) * 1e-3
    print(f"queue {q}: {len(ks)} kernels, {n_round} pixel searches, busy {100 * busy / span:.1f} % of the window; gaps between consecutive kernels: "
          f"p50 {np.percentile(gaps, 50):.1f} us, p90 {np.percentile(gaps, 90):.1f}, sum {gaps.sum() * 1e-3:.1f} ms ({100 * gaps.sum() * 1e3 / span:.1f} %), "
          f"in gaps > 100 us {gaps[gaps > 100].sum() * 1e-3:.1f} ms")
    # a round = from one pixel search to the next
    starts = [a for a, b, n in ks if "k_accumulate_round_j" in n]
    per = np.diff(np.array(starts, dtype=np.float64)) * 1e-3
    print(f"   round period: p10 {np.percentile(per, 10):.0f} us, p50 {np.percentile(per, 50):.0f}, p90 {np.percentile(per, 90):.0f}, mean {per.mean():.0f}")
