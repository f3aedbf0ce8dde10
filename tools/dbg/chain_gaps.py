#!/usr/bin/env python
"""Gaps between the dependent kernels of the longest brighter-fatter chain (from a rocprofv3 --kernel-trace csv):
   python tools/dbg/chain_gaps.py <kernel_trace.csv>"""
import collections
import csv
import sys

import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
byq = collections.defaultdict(list)
for r in rows:
    byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-26:]))
q = max(byq, key=lambda k: sum(1 for x in byq[k] if "accumulate_segments" in x[2]))
ks = sorted(byq[q])
tail = ks[-3 * 150:]                      # the last 150 rounds of the last step: few objects left
dur = collections.defaultdict(list)
gaps = []
for a, b in zip(tail[:-1], tail[1:]):
    gaps.append(b[0] - a[1])
    dur[a[2]].append(a[1] - a[0])
print("queue", q, "kernels", len(ks))
for k, v in dur.items():
    print("  %-28s n %4d  median %.1f us  min %.1f us" % (k, len(v), np.median(v) / 1e3, np.min(v) / 1e3))
g = np.array(gaps) / 1e3
print("  gap between dependent kernels: median %.1f us, mean %.1f us, p90 %.1f us" % (np.median(g), g.mean(), np.percentile(g, 90)))
span = (tail[-1][1] - tail[0][0]) / 1e3
print("  %d kernels in %.1f us: %.1f us per round; kernel time %.0f %%, gaps %.0f %%" %
      (len(tail), span, span / (len(tail) / 3), 100 * sum(sum(v) for v in dur.values()) / 1e3 / span, 100 * g.sum() / span))
