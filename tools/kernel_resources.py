#!/usr/bin/env python3
"""Register / scratch / LDS / occupancy table of every kernel in csrc/imsim_hip.hip.

Compiles the translation unit for gfx950 with -Rpass-analysis=kernel-resource-usage (no GPU needed) and prints
one line per kernel.  Usage: python tools/kernel_resources.py [extra hipcc flags ...]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "imsim_amd", "csrc", "imsim_hip.hip")


def main():
    with tempfile.TemporaryDirectory() as d:
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               "-Rpass-analysis=kernel-resource-usage", SRC, "-o", os.path.join(d, "x.so")] + sys.argv[1:]
        err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark:\s+(?:Function )?Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/lane\]| \[bytes/block\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    print(f"{'kernel':<60} {'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'scratch':>8} {'LDS':>7} {'occ':>4}")
    for r in rows:
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip().split("(")[0]
        print(f"{name[:60]:<60} {r.get('VGPRs', 0):>5} {r.get('AGPRs', 0):>5} {r.get('TotalSGPRs', 0):>5} "
              f"{r.get('ScratchSize', 0):>8} {r.get('LDS Size', 0):>7} {r.get('Occupancy', 0):>4}")


if __name__ == "__main__":
    main()
