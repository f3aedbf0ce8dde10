#!/usr/bin/env python
"""Sustained shader clock of a counter pass: GRBM_GUI_ACTIVE (GPU-active cycles of a dispatch) / the dispatch's duration, per kernel
and over the kernels that last long enough to measure (MI355X_MICROARCH.md, DVFS: effective clock = GRBM_GUI_ACTIVE / wall).
   python tools/clock_summary.py out.json <dir of a rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv pass>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


N_XCD = 8


def main(out, d):
    dur = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    per = defaultdict(lambda: [0.0, 0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
                    continue
                t = dur.get(r["Dispatch_Id"])
                if t is None and "Start_Timestamp" in r and r.get("End_Timestamp"):
                    t = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                if not t or t < 50_000:                 # dispatches under 50 us: the counter's window is not the kernel's
                    continue
                p = per[r["Kernel_Name"].split("(")[0][:60]]
                p[0] += float(r["Counter_Value"]); p[1] += t; p[2] += 1
    res = {"kernels": {k: {"ghz": v[0] / v[1], "dispatches": v[2], "ms": v[1] * 1e-6} for k, v in per.items() if v[1] > 0}}
    tot_c = sum(v[0] for v in per.values()); tot_t = sum(v[1] for v in per.values())
    # (the counter is the sum over the 8 XCDs of the part: one GRBM per XCD)
    for v in res["kernels"].values():
        v["ghz"] /= N_XCD
    res["sustained_clock_ghz"] = tot_c / tot_t / N_XCD if tot_t else None
    photon = [v for k, v in per.items() if "k_shoot" in k]
    if photon:
        res["photon_kernels_clock_ghz"] = sum(v[0] for v in photon) / sum(v[1] for v in photon) / N_XCD
    res["note"] = "GRBM_GUI_ACTIVE / duration over dispatches of >= 50 us; counted per dispatch while other streams' kernels run beside it"
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps({k: v for k, v in res.items() if k != "kernels"}))
    for k, v in sorted(res["kernels"].items(), key=lambda kv: -kv[1]["ms"])[:8]:
        print(f"  {k:60s} {v['ghz']:.3f} GHz  {v['dispatches']:6d} dispatches {v['ms']:9.1f} ms")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
