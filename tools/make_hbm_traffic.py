#!/usr/bin/env python
"""Rebuild profiles/hbm_traffic.json (what bench.py reports as roofline.traffic and uses for the f64-issue fraction) from
the committed per-config PMC summaries (tools/pmc_summary.py output): the newest of profiles/round4_<config>_hbm_pmc.json, round3_<config>_hbm_pmc.json, round2e_<config>_hbm_pmc.json
(end of round 2: converted pool, k_shoot_photons<2>), round2_<config>_hbm_pmc.json and round1_<config>_final_hbm_pmc.json, with
the SQ pass of the same tag where present."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def short_name(long_name):
    """bench.py's label of a kernel: the template arguments after the pool mode (chain / PSF / optics-layout specialisations
    of one and the same kernel) are dropped"""
    n = long_name.replace("void ", "")
    if n.startswith("k_shoot_accumulate"):
        return "k_shoot_accumulate"
    if n.startswith("k_shoot_photons<true>"):
        return "k_shoot_photons<true>"
    if n.startswith("k_shoot_photons<2"):
        return "k_shoot_photons<2>"
    if n.startswith("k_accumulate_round"):
        return "k_accumulate_round<4>"
    return None


def main():
    out = {}
    for cfg in ("c2", "c3", "c3b", "c5"):
        cands = [(f"profiles/round5_{cfg}_hbm_pmc.json", f"profiles/round5_{cfg}_sq_pmc.json"),
                 (f"profiles/round4_{cfg}_hbm_pmc.json", f"profiles/round4_{cfg}_sq_pmc.json"),
                 (f"profiles/round3_{cfg}_hbm_pmc.json", f"profiles/round3_{cfg}_sq_pmc.json"),
                 (f"profiles/round2e_{cfg}_hbm_pmc.json", f"profiles/round2e_{cfg}_sq_pmc.json"),
                 (f"profiles/round2_{cfg}_hbm_pmc.json", f"profiles/round2_{cfg}_sq_pmc.json"),
                 (f"profiles/round1_{cfg}_final_hbm_pmc.json", f"profiles/round1_{cfg}_final_sq_pmc.json")]
        found = [c for c in cands if os.path.exists(os.path.join(ROOT, c[0]))]
        if not found:
            continue
        src, sq_src = found[0]
        path = os.path.join(ROOT, src)
        d = json.load(open(path))
        sq = json.load(open(os.path.join(ROOT, sq_src))) if os.path.exists(os.path.join(ROOT, sq_src)) else {}
        for long_name in d:
            short = short_name(long_name)
            if short is None:
                continue
            k = d[long_name]
            if "FETCH_SIZE" not in k or k["FETCH_SIZE"]["launches"] == 0:
                continue
            out.setdefault(cfg, {})[short] = {
                "hbm_bytes_per_launch": k["hbm_bytes_per_launch"],
                "fetch_size_kib_mean": k["FETCH_SIZE"]["mean"],
                "write_size_kib_mean": k["WRITE_SIZE"]["mean"],
                "launches_sampled": k["FETCH_SIZE"]["launches"],
                "source": src,
                "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024",
            }
            if long_name in sq and "SQ_INSTS_VALU" in sq[long_name] and "SQ_WAVES" in sq[long_name]:
                q = sq[long_name]
                out[cfg][short].update(valu_insts_per_wave=q["SQ_INSTS_VALU"]["sum"] / q["SQ_WAVES"]["sum"],
                                       valu_busy_frac_of_wave_cycles=q["SQ_ACTIVE_INST_VALU"]["sum"] / q["SQ_WAVE_CYCLES"]["sum"],
                                       sq_source=sq_src)
        # the whole step's vector work (bench.py's roofline.step): SQ_INSTS_VALU summed over EVERY kernel of the profiled run,
        # per step -- a step of these configs holds exactly one fused launch (k_shoot_accumulate), whose launch count
        # is therefore the number of steps the profiler saw
        steps = sum(v["SQ_INSTS_VALU"]["launches"] for n, v in sq.items() if short_name(n) == "k_shoot_accumulate" and "SQ_INSTS_VALU" in v)
        if steps > 0 and cfg in ("c3", "c3b"):
            total = sum(v["SQ_INSTS_VALU"]["sum"] for v in sq.values() if "SQ_INSTS_VALU" in v)
            out.setdefault(cfg, {})["_step"] = {"valu_wave_insts_per_step": total / steps, "steps_sampled": steps, "sq_source": sq_src,
                                                "note": "sum of SQ_INSTS_VALU over all kernels of the run / its steps"}
        if steps > 0 and cfg == "c5":
            # a focal plane: one fused launch per CCD render; the counters were taken on a part of the visit (the profiler's counter
            # pass does not survive the whole one), so the figure kept is per CCD and bench.py multiplies by the CCDs of its step
            total = sum(v["SQ_INSTS_VALU"]["sum"] for v in sq.values() if "SQ_INSTS_VALU" in v)
            out.setdefault(cfg, {})["_step"] = {"valu_wave_insts_per_ccd": total / steps, "ccd_renders_sampled": steps, "sq_source": sq_src,
                                                "note": "sum of SQ_INSTS_VALU over all kernels of the run / its CCD renders"}
    json.dump(out, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
