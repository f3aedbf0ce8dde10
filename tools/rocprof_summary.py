#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a text table:
   python tools/rocprof_summary.py <results.db> [out.txt]"""
import sqlite3
import sys

import numpy as np


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    lines = ["# rocprofv3 --kernel-trace --stats summary of " + path,
             "%-72s %8s %12s %12s %12s %12s %7s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct")]
    rows = list(cur.execute("select name, duration, vgpr_count, sgpr_count, scratch_size, lds_size from kernels"))
    by = {}
    for name, dur, vg, sg, sc, lds in rows:
        by.setdefault(name, []).append(dur)
    total = sum(sum(v) for v in by.values())
    for name, d in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        d = np.array(d, dtype=np.float64)
        lines.append("%-72s %8d %12.3f %12.2f %12.2f %12.2f %6.2f%%" % (name[:72], len(d), d.sum() / 1e6, d.mean() / 1e3,
                                                                    d.min() / 1e3, d.max() / 1e3, 100 * d.sum() / total))
    res = {}
    for name, dur, vg, sg, sc, lds in rows:
        res[name] = (vg, sg, sc, lds)
    lines.append("")
    lines.append("%-72s %6s %6s %8s %8s" % ("kernel", "vgpr", "sgpr", "scratch", "lds"))
    for name, (vg, sg, sc, lds) in res.items():
        lines.append("%-72s %6s %6s %8s %8s" % (name[:72], vg, sg, sc, lds))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
