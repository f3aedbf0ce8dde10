#!/usr/bin/env python
"""Rebuild profiles/hbm_traffic.json -- what bench.py reports as roofline.traffic, roofline.f64_valu_issue, roofline.step (both
floors, the sustained clock) and roofline.phases -- from the committed per-config summaries of tools/profile_config.sh:
profiles/<tag>_<config>_{hbm_pmc,sq_pmc,mix_pmc,clock,bench}.json and <tag>_<config>_kernel_stats.txt, newest tag first."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAGS = ("round6", "round5", "round4", "round3", "round2e", "round2")
CONFIGS = ("c2", "c3", "c3b", "c4", "c5", "fft", "fftx", "fftxs")
# the kernel that is launched a known number of times per step (bench line: roofline.timed_launches_per_step for the FFT branch)
ONCE = {"c2": "k_shoot_accumulate", "c3": "k_shoot_accumulate", "c3b": "k_shoot_accumulate", "c5": "k_shoot_accumulate",
        "c4": "k_shoot_photons<2>", "fft": "k_fft_kspace_fill", "fftx": "k_fft_kspace_fill", "fftxs": "k_fft_kspace_fill"}
MIX = {"fma_f64": "SQ_INSTS_VALU_FMA_F64", "mul_f64": "SQ_INSTS_VALU_MUL_F64", "add_f64": "SQ_INSTS_VALU_ADD_F64",
       "trans_f64": "SQ_INSTS_VALU_TRANS_F64", "int32": "SQ_INSTS_VALU_INT32", "int64": "SQ_INSTS_VALU_INT64", "cvt": "SQ_INSTS_VALU_CVT"}


def short_name(long_name):
    """bench.py's label of a kernel: the template arguments after the pool mode (chain / PSF / optics-layout specialisations
    of one and the same kernel) are dropped"""
    n = long_name.replace("void ", "")
    for head, short in (("k_shoot_accumulate", "k_shoot_accumulate"), ("k_shoot_photons<true>", "k_shoot_photons<true>"),
                        ("k_shoot_photons<2", "k_shoot_photons<2>"), ("k_accumulate_round", "k_accumulate_round<4>"),
                        ("k_accumulate_segments", "k_accumulate_segments"), ("k_accumulate_small", "k_accumulate_small"),
                        ("k_fft_kspace_fill", "k_fft_kspace_fill"), ("k_fft_spikes", "k_fft_spikes"), ("k_fft_finish", "k_fft_finish"),
                        ("k_update_distortions_q3<", "k_update_distortions_q3"), ("k_refresh_changed<", "k_refresh_changed"),
                        ("k_update_list_j", "k_update_list_j"), ("k_refresh_list_j", "k_refresh_list_j")):
        if n.startswith(head):
            return short
    return None


def load(tag, cfg, what):
    p = os.path.join(ROOT, "profiles", f"{tag}_{cfg}_{what}.json")
    return (json.load(open(p)), f"profiles/{tag}_{cfg}_{what}.json") if os.path.exists(p) else (None, None)


def first(cfg, what):
    for tag in TAGS:
        d, src = load(tag, cfg, what)
        if d is not None:
            return d, src, tag
    p = os.path.join(ROOT, "profiles", f"round1_{cfg}_final_{what}.json")
    if os.path.exists(p):
        return json.load(open(p)), f"profiles/round1_{cfg}_final_{what}.json", "round1"
    return None, None, None


def launches(sq, short, counter="SQ_INSTS_VALU"):
    return sum(v[counter]["launches"] for n, v in sq.items() if short_name(n) == short and counter in v)


def c4_phases(tag):
    """per-phase figures of the C4 step from the kernel trace of the same tag: ms per step, algorithmic bytes, GB/s,
    fraction of 8 TB/s"""
    p = os.path.join(ROOT, "profiles", f"{tag}_c4_kernel_stats.txt")
    bench, _ = load(tag, "c4", "bench")
    if not os.path.exists(p) or bench is None:
        return None
    rows = {}
    for line in open(p):
        m = re.match(r"^(?:void )?(k_[a-z_0-9]+).*?\s(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+[\d.]+%", line)
        if m and m.group(1) not in rows:
            rows[m.group(1)] = (int(m.group(2)), float(m.group(3)))
    if "k_shoot_photons" not in rows:
        return None
    steps = rows["k_shoot_photons"][0]
    phot = bench["config"]["n_photons"]
    n_obj = bench["config"]["n_objects"]
    cells = 4097 * 4097
    # algorithmic bytes per step: shoot = 32-B converted record written + 256-B row; search = record read + the pixel's 64-B bounds
    # line + one f64 read-modify-write of the delta image (16 B); whole-CCD update = 160 B of owned points read and written + 8 B delta
    # per cell, per recalculation; refresh = 160 B points + 64 B bounds line per cell
    recalcs = rows.get("k_update_distortions_q3", (0, 0))[0] / max(steps, 1)
    spec = {"shoot (k_shoot_photons<2>)": ("k_shoot_photons", phot * 32 + n_obj * 256),
            "pixel search, shares of more than a wavefront (k_accumulate_segments)": ("k_accumulate_segments", None),
            "pixel search, one wavefront per object (k_accumulate_small)": ("k_accumulate_small", None),
            "whole-CCD updatePixelDistortions (k_update_distortions_q3)": ("k_update_distortions_q3", recalcs * cells * 328),
            "whole-CCD bounds refresh + target += delta (k_refresh_changed)": ("k_refresh_changed", recalcs * cells * 240),
            "initial pixel-boundary state (k_init_tiles)": ("k_init_tiles", cells * 232)}
    search_ms = sum(rows.get(k, (0, 0))[1] for k in ("k_accumulate_segments", "k_accumulate_small")) / steps
    out = {}
    for label, (k, nbytes) in spec.items():
        if k not in rows:
            continue
        ms = rows[k][1] / steps
        if nbytes is None:                           # the two searches share the photons: bytes in proportion to their time
            nbytes = phot * 112 * ms / max(search_ms, 1e-9)
        out[label] = {"ms_per_step": ms, "algorithmic_bytes": nbytes, "GBps": nbytes / (ms * 1e-3) / 1e9, "frac_of_8TBps": nbytes / (ms * 1e-3) / 8e12,
                      "launches_per_step": rows[k][0] / steps}
    out["_source"] = f"profiles/{tag}_c4_kernel_stats.txt ({steps} steps) + profiles/round6_atomic_rate.txt (the f64 atomic ceilings: 24 G/s random, 29 G/s within 32 x 32 pixels)"
    return out


def main():
    out = {}
    for cfg in CONFIGS:
        d, src, tag = first(cfg, "hbm_pmc")
        if d is None:
            continue
        sq, sq_src = load(tag, cfg, "sq_pmc")
        if sq is None and tag == "round1":
            p = os.path.join(ROOT, "profiles", f"round1_{cfg}_final_sq_pmc.json")
            sq, sq_src = (json.load(open(p)), f"profiles/round1_{cfg}_final_sq_pmc.json") if os.path.exists(p) else (None, None)
        sq = sq or {}
        for long_name, k in d.items():
            short = short_name(long_name)
            if short is None or "FETCH_SIZE" not in k or k["FETCH_SIZE"]["launches"] == 0 or "WRITE_SIZE" not in k:
                continue
            e = out.setdefault(cfg, {}).setdefault(short, {})
            if e:                                       # several specialisations of one kernel: the one with more launches
                if e["launches_sampled"] >= k["FETCH_SIZE"]["launches"]:
                    continue
                e.clear()
            e.update(hbm_bytes_per_launch=k["hbm_bytes_per_launch"], fetch_size_kib_mean=k["FETCH_SIZE"]["mean"],
                     write_size_kib_mean=k["WRITE_SIZE"]["mean"], launches_sampled=k["FETCH_SIZE"]["launches"], source=src,
                     formula="(2*FETCH_SIZE + WRITE_SIZE) * 1024")
            if long_name in sq and "SQ_INSTS_VALU" in sq[long_name] and "SQ_WAVES" in sq[long_name]:
                q = sq[long_name]
                e.update(valu_insts_per_wave=q["SQ_INSTS_VALU"]["sum"] / q["SQ_WAVES"]["sum"],
                         valu_busy_frac_of_wave_cycles=q["SQ_ACTIVE_INST_VALU"]["sum"] / q["SQ_WAVE_CYCLES"]["sum"],
                         wait_frac_of_wave_cycles=q["SQ_WAIT_ANY"]["sum"] / q["SQ_WAVE_CYCLES"]["sum"] if "SQ_WAIT_ANY" in q else None,
                         sq_source=sq_src)
        # the whole step's vector work (bench.py's roofline.step): SQ_INSTS_VALU summed over EVERY kernel of the profiled run / its steps
        bench, _ = load(tag, cfg, "bench")
        per_step = 1
        if cfg.startswith("fft") and bench is not None:
            per_step = max(int(bench["roofline"].get("timed_launches_per_step", 1)), 1)
        steps = launches(sq, ONCE[cfg]) / per_step if sq else 0
        if steps > 0:
            total = sum(v["SQ_INSTS_VALU"]["sum"] for v in sq.values() if "SQ_INSTS_VALU" in v)
            unit = "ccd" if cfg == "c5" else "step"
            st = {f"valu_wave_insts_per_{unit}": total / steps, ("ccd_renders_sampled" if cfg == "c5" else "steps_sampled"): steps,
                  "sq_source": sq_src, "note": f"sum of SQ_INSTS_VALU over all kernels of the run / its {'CCD renders' if cfg == 'c5' else 'steps'}"}
            mix, mix_src = load(tag, cfg, "mix_pmc")
            if mix:
                msteps = launches(mix, ONCE[cfg]) / per_step
                if msteps > 0:
                    st[f"class_mix_per_{unit}"] = {k: sum(v[c]["sum"] for v in mix.values() if c in v) / msteps for k, c in MIX.items()}
                    st["mix_source"] = mix_src
            clk, clk_src = load(tag, cfg, "clock")
            if clk and clk.get("sustained_clock_ghz"):
                st["sustained_clock_ghz"] = clk.get("photon_kernels_clock_ghz") or clk["sustained_clock_ghz"]
                st["clock_all_kernels_ghz"] = clk["sustained_clock_ghz"]
                st["clock_source"] = clk_src
            out.setdefault(cfg, {})["_step"] = st
        if cfg == "c4":
            ph = c4_phases(tag)
            if ph:
                out.setdefault(cfg, {})["_phases"] = ph
    json.dump(out, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)
    print(json.dumps({c: sorted(v) for c, v in out.items()}, indent=1))


if __name__ == "__main__":
    main()
