#!/usr/bin/env python
"""Timeline analysis of the last bench step in a rocprofv3 rocpd database."""
import collections
import glob
import sqlite3
import sys

import numpy as np

db = sqlite3.connect(glob.glob(sys.argv[1])[0])
cur = db.cursor()
rows = list(cur.execute("select name,start,end,queue_id from kernels order by start"))
names = [r[0].split('(')[0][:28] for r in rows]
st = np.array([r[1] for r in rows]); en = np.array([r[2] for r in rows]); q = np.array([r[3] for r in rows])
idx = [i for i, n in enumerate(names) if 'k_shoot_photons' in n]
i0 = idx[-1]; t0 = st[i0]; qc = q[i0]
chain = np.array([i for i in range(i0, len(rows)) if q[i] == qc])
for i in range(max(i0 - 6, 0), len(rows)):
    if 'k_shoot_accumulate' in names[i] and st[i] > t0 - 5e6:
        print('BULK start %.3f dur %.3f ms' % ((st[i] - t0) / 1e6, (en[i] - st[i]) / 1e6))
print('pool dur %.3f ms, chain end %.3f ms' % ((en[i0] - st[i0]) / 1e6, (en[chain[-1]] - t0) / 1e6))
gap = (st[chain[1:]] - en[chain[:-1]]) / 1e3
agg = collections.defaultdict(list)
for a, b, g in zip(chain[:-1], chain[1:], gap):
    agg[names[a][:14] + '->' + names[b][:14]].append(g)
for t, g in agg.items():
    g = np.array(g)
    print(t.ljust(32), 'n', len(g), 'mean %.1f us  late-mean %.1f' % (g.mean(), g[-40:].mean()))
for nm in ('k_accumulate_segments', 'k_update_distort', 'k_refresh_changed'):
    d = np.array([(en[i] - st[i]) / 1e3 for i in chain if nm in names[i]])
    if len(d):
        print(nm.ljust(24), 'first %s  late mean %.1f us  total %.2f ms' % (np.round(d[:4]), d[-40:].mean(), d.sum() / 1e3))
