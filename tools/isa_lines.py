#!/usr/bin/env python
"""Static instruction attribution: histogram of the VALU/SALU/memory instructions of one kernel
by source (file, function) using the .loc line tables of a -gline-tables-only -save-temps build.
   python tools/isa_lines.py <file.s> <kernel-substring>"""
import bisect
import collections
import re
import sys

s = open(sys.argv[1]).read().splitlines()
kern = sys.argv[2]
files = {}
for ln in s:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', ln)
    if m:
        files[int(m.group(1))] = m.group(2) + "/" + m.group(3)
# function start lines of the source files (top-level IMS_DEV / __device__ / __global__ definitions)
func_tab = {}
for fid, path in files.items():
    try:
        src = open(path).read().splitlines()
    except OSError:
        continue
    starts = []
    for i, l in enumerate(src, 1):
        m = re.match(r'^(?:template.*>\s*)?(?:IMS_DEV|__device__|__global__|static|inline|__host__)[^;]*?\b(\w+)\s*\(', l)
        if m and not l.strip().endswith(";"):
            starts.append((i, m.group(1)))
    func_tab[fid] = starts
inside = False
cur = (0, 0)
hist = collections.Counter()
kinds = collections.Counter()
for ln in s:
    if re.match(r'^[_\w]*' + re.escape(kern) + r'[_\w]*:', ln):
        inside = True
        continue
    if inside and ".end_amdhsa_kernel" in ln:
        break
    if not inside:
        continue
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', ln)
    if m:
        cur = (int(m.group(1)), int(m.group(2)))
        continue
    m = re.match(r'\s+([vs]_\w+|global_\w+|scratch_\w+|ds_\w+|buffer_\w+|flat_\w+)', ln)
    if not m:
        continue
    op = m.group(1)
    kind = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "mem"
    fid, line = cur
    name = "?"
    st = func_tab.get(fid, [])
    if st:
        k = bisect.bisect_right([a for a, _ in st], line) - 1
        if k >= 0:
            name = st[k][1]
    hist[(files.get(fid, "?").split("/")[-1], name, kind)] += 1
    kinds[kind] += 1
print(dict(kinds))
agg = collections.defaultdict(lambda: [0, 0, 0])
for (f, n, k), c in hist.items():
    agg[(f, n)][["valu", "salu", "mem"].index(k)] += c
for (f, n), (v, sa, me) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print("%-28s %-28s valu %6d salu %6d mem %5d" % (f, n, v, sa, me))
