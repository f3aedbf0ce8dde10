#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel.

   python tools/pmc_summary.py out.txt [out.json] dir1 [dir2 ...]

Each dir holds one pass (one --pmc set).  Per kernel: number of dispatches and the SUM and MEAN of
every counter.  For FETCH_SIZE / WRITE_SIZE (unit: KiB) the JSON also carries the per-launch HBM
bytes with the gfx950 correction of MI355X_MICROARCH.md (HBM): bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024
(FETCH_SIZE tallies 128-B read requests at 64 B)."""
import csv
import glob
import json
import os
import sys


def collect(dirs):
    acc = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    k = acc.setdefault(row["Kernel_Name"], {})
                    c = k.setdefault(row["Counter_Name"], [0, 0.0, set()])
                    c[0] += 1
                    c[1] += float(row["Counter_Value"])
                    c[2].add(row["Dispatch_Id"])
    return acc


def main(argv):
    out_txt = argv[0]
    out_json = argv[1] if len(argv) > 1 and argv[1].endswith(".json") else None
    dirs = argv[2:] if out_json else argv[1:]
    acc = collect(dirs)
    lines = ["# rocprofv3 --pmc summary of " + " ".join(dirs),
             "%-64s %-22s %9s %18s %16s" % ("kernel", "counter", "launches", "sum", "mean/launch")]
    js = {}
    for name in sorted(acc, key=lambda n: -sum(v[1] for v in acc[n].values())):
        for cname, (rows, total, disp) in sorted(acc[name].items()):
            n = len(disp)
            lines.append("%-64s %-22s %9d %18.1f %16.2f" % (name[:64], cname, n, total, total / max(n, 1)))
            js.setdefault(name, {})[cname] = {"launches": n, "sum": total, "mean": total / max(n, 1)}
        k = js[name]
        if "FETCH_SIZE" in k and "WRITE_SIZE" in k:
            k["hbm_bytes_per_launch"] = (2.0 * k["FETCH_SIZE"]["mean"] + k["WRITE_SIZE"]["mean"]) * 1024.0
            lines.append("%-64s %-22s %9s %18s %16.0f" % (name[:64], "HBM bytes (2F+W)*1024", "", "", k["hbm_bytes_per_launch"]))
    text = "\n".join(lines) + "\n"
    open(out_txt, "w").write(text)
    if out_json:
        json.dump(js, open(out_json, "w"), indent=1, sort_keys=True)
    print(text)


if __name__ == "__main__":
    main(sys.argv[1:])
