#!/usr/bin/env python
"""Rebuild profiles/hbm_traffic.json (what bench.py reports as roofline.traffic) from the committed
per-config PMC summaries profiles/round1_<config>_final_hbm_pmc.json (tools/pmc_summary.py output)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = {"k_shoot_accumulate(ims_render_params)": "k_shoot_accumulate",
           "void k_shoot_photons<true>(ims_render_params, long const*, ims_photons)": "k_shoot_photons<true>"}


def main(tag="round1"):
    out = {}
    for cfg in ("c2", "c3", "c3b"):
        src = f"profiles/{tag}_{cfg}_final_hbm_pmc.json"
        path = os.path.join(ROOT, src)
        if not os.path.exists(path):
            continue
        d = json.load(open(path))
        for long_name, short in KERNELS.items():
            if long_name not in d:
                continue
            k = d[long_name]
            if k["FETCH_SIZE"]["launches"] == 0:
                continue
            out.setdefault(cfg, {})[short] = {
                "hbm_bytes_per_launch": k["hbm_bytes_per_launch"],
                "fetch_size_kib_mean": k["FETCH_SIZE"]["mean"],
                "write_size_kib_mean": k["WRITE_SIZE"]["mean"],
                "launches_sampled": k["FETCH_SIZE"]["launches"],
                "source": src,
                "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024",
            }
    json.dump(out, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:])
