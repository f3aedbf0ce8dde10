#!/usr/bin/env python
"""The one place where reference-held numbers and the HIP path disagree, bounded: the reference's STORED
ray-tracing spike statistics of one bright star (tests/golden/fft-diffraction/*.npz, made by
/root/reference/tests/test_diffraction_fft.py:276-294) against this build's photon path, swept over
  (a) the photon count N (the reference's N -- Vega through LSST_r.dat -- is not on file; the estimators' r_max and bins depend on it),
  (b) the pupil sampled inside ONE rim only, by 0 / 5 / 10 / 20 mm,
  (c) the wavelength mix (flat-in-photons r band, monochromatic 577.6 nm, a Vega-like black body through the r band).
On the GPU box:  python tools/spike_sweep.py > gpurun_out/spike_sweep.log ; the table goes to profiles/round6_spike_sweep.log."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import spike_stats as ss  # noqa: E402

RIM_INNER = 2.558   # the inner circle of RUBIN_SPIDER_GEOMETRY (imsim/diffraction.py:41); the sampler starts at 2.55, INSIDE it
KEYS = ("c", "angle_deg", "angle_stddev_deg", "slope", "intercept")


def line(label, got, want):
    ok = ss.within(got, want)
    dc = float(abs(got["c"] - want["c"]).max())
    cells = [f"{dc:6.2f}{'' if ok['c'] else '*'}"]
    for k in KEYS[1:]:
        cells.append(f"{got[k]:9.4f}{' ' if ok[k] else '*'}")
    n_ok = sum(ok.values())
    return f"{label:<46s} " + " ".join(cells) + f"  se {got['slope_stderr']:.3f}/{got['intercept_stderr']:.3f}  {n_ok}/5"


def main():
    print("# columns: |centre - stored| [px], folded angle [deg], spread of the folded angle [deg], slope, intercept, the two standard")
    print("# errors, how many of the five stored statistics are met under the reference's tolerances (centre 2 px, angle 1 deg, spread")
    print("# 2 deg, slope 0.1, intercept 0.5); * = outside the tolerance")
    for exptime in (0.0, 300.0):
        ref = ss.stored(exptime)
        want = ss.stored_stats(ref)
        print(f"\n## exptime {exptime:g} s   stored: angle {want['angle_deg']:.4f}  spread {want['angle_stddev_deg']:.4f}  slope {want['slope']:.4f}"
              f"  intercept {want['intercept']:.4f}  (se {float(ref['slope_stderr']):.3f}/{float(ref['intercept_stderr']):.3f})")
        t0 = time.time()
        runs = []
        for n in (1_000_000, 2_000_000, 6_000_000, 20_000_000):
            runs.append((f"(a) nominal pupil 2.55..4.18, r-flat, N={n:.0e}", dict(), n))
        for mm in (5, 10, 20):
            runs.append((f"(b) inner rim only +{mm} mm (R_inner {RIM_INNER + mm * 1e-3:.3f})", dict(r_inner=RIM_INNER + mm * 1e-3), ss.N_PHOT))
        for mm in (5, 10, 20):
            runs.append((f"(b) outer rim only -{mm} mm (R_outer {ss.R_OUTER - mm * 1e-3:.3f})", dict(r_outer=ss.R_OUTER - mm * 1e-3), ss.N_PHOT))
        for mm in (5, 10, 20):
            runs.append((f"(b) both rims {mm} mm", dict(r_inner=RIM_INNER + mm * 1e-3, r_outer=ss.R_OUTER - mm * 1e-3), ss.N_PHOT))
        runs.append(("(b) the emulation of the test: 2.58..4.16", dict(r_inner=2.58, r_outer=4.16), ss.N_PHOT))
        for sed in ("mono", "vega-r"):
            runs.append((f"(c) nominal pupil, SED {sed}, N=6e6", dict(sed=sed), ss.N_PHOT))
        for label, kw, n in runs:
            img = ss.render(ss.scene(exptime, **kw), ref, n)
            got = ss.image_stats(img)
            print(line(label, got, want), flush=True)
        print(f"# {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
