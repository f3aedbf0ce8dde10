#!/usr/bin/env python
"""Batch timeline of focal-plane steps from the lines IMS_FOCAL_TRACE=1 prints (focal_plane.render_focal_plane): per step and batch
the host time of its fronts, when they were through on the device, when its joint run was enqueued and ended, the run's length
(end - max(previous end, fronts done)) and when its last image was on the host.
  IMS_FOCAL_TRACE=1 R5_CALLS=8 python3 tools/dbg/c5_full.py 189 | python3 tools/dbg/c5_batches.py"""
import re
import sys

pat = re.compile(r"CCD\s+(\d+) host\s+([\d.]+)\s+([\d.]+) \| bulk\s+([\d.]+) pre\s+([\d.]+) mid\s+([\d.]+) \| joint end\s+([\d.]+) \(enqueued\s+([\d.]+)\) done\s+([\d.]+)")
steps, cur = [], None
for line in sys.stdin:
    if line.startswith("focal trace"):
        cur = []
        steps.append(cur)
        continue
    m = pat.search(line)
    if m and cur is not None:
        cur.append([float(x) for x in m.groups()])
    elif "call " in line and " s for " in line:
        print(line.strip()[:110])
        if cur:
            batches = []
            for row in cur:
                if batches and batches[-1][0][6] == row[6]:
                    batches[-1].append(row)
                else:
                    batches.append([row])
            prev_end = 0.0
            for b in batches:
                fronts = max(max(r[3], r[4], r[5]) for r in b)
                end = b[0][6]
                run = end - max(prev_end, fronts)
                print(f"   {len(b):3d} host {min(r[1] for r in b):7.1f}..{max(r[2] for r in b):7.1f} fronts {fronts:7.1f} enq {b[0][7]:7.1f} end {end:7.1f} "
                      f"run {run:6.1f} done {max(r[8] for r in b):7.1f}")
                prev_end = end
        cur = None
