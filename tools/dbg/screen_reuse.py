#!/usr/bin/env python
"""Row N1 ("LDS-staged phase-screen tiles"), asked of the data: what is there to stage?

For the C3b bench workload (100 k objects, ~1e8 photons, imSim's six-layer atmosphere on 8192^2 screens of 0.1 m;
imsim/atmPSF.py:164-336) this counts, on the host, the 64-byte lines of the screen table that the gathers of
  (a) ONE object's photons,
  (b) the 256 photons a workgroup holds at a time,
  (c) all photons of one launch (one of the 14 shares of a step)
fall on, per layer, against the number of gathers -- in the layout the kernels use (one 16-byte 2 x 2 cell per sample: four samples
per line) and in the plain one (4-byte samples: sixteen per line, two rows per gather).  A photon's sample is at
(pupil point + altitude x field angle - wind x arrival time) / 0.1 m: the pupil point uniform on the annulus, the time uniform on
the exposure -- drawn here with numpy (the distribution is what matters for the count, not the kernels' bits).
No GPU:  python tools/dbg/screen_reuse.py > profiles/round6_screen_reuse.txt"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from imsim_amd import atm_psf, catalog, configs  # noqa: E402


def cells(rng, n, thx, thy, A, layer):
    """sample numbers (iy, ix) of n photons of an object at field angle (thx, thy) on one layer"""
    r = np.sqrt(A["ri"] ** 2 + rng.random(n) * (A["ro"] ** 2 - A["ri"] ** 2))
    ph = 2.0 * math.pi * rng.random(n)
    t = A["exptime"] * rng.random(n)
    x = r * np.cos(ph) - t * A["vx"][layer] + A["alt"][layer] * thx
    y = r * np.sin(ph) - t * A["vy"][layer] + A["alt"][layer] * thy
    ix = np.floor((x - A["x0"]) / A["scale"]).astype(np.int64) % A["npix"]
    iy = np.floor((y - A["x0"]) / A["scale"]).astype(np.int64) % A["npix"]
    return iy, ix


def lines_quads(iy, ix, npix):
    return np.unique((iy * npix + ix) >> 2).size                 # 16 B per sample, 4 samples per 64-B line


def lines_plain(iy, ix, npix):
    a = (iy * npix + ix) >> 4                                     # 4 B per sample, 16 per line; the cell's second row too
    b = (((iy + 1) % npix) * npix + ix) >> 4
    return np.unique(np.concatenate([a, b])).size


def main():
    scene = configs.scene_c3(nx=4096, ny=4096, sensor=False)
    cat = catalog.synthetic_catalog(100000, nx=scene.nx, ny=scene.ny)
    phot = catalog.realize_fluxes(cat["nominal_flux"], scene.seed)
    keep = phot > 0
    thx, thy = configs.field_angles(scene, cat["x"][keep], cat["y"][keep])
    phot = phot[keep]
    atm = atm_psf.AtmosphericPSF(configs.VISIT["airmass"], configs.VISIT["raw_seeing"], configs.VISIT["band"], seed=398414,
                                 exptime=configs.VISIT["exptime"], screen_size=12.8, screen_scale=0.1)     # the layers; no 8192^2 screens needed
    npix = 8192
    A = dict(npix=npix, scale=0.1, x0=-0.5 * npix * 0.1, exptime=atm.exptime, ro=atm.diam / 2, ri=atm.diam * atm.obscuration / 2,
             vx=atm.speeds * np.cos(atm.directions), vy=atm.speeds * np.sin(atm.directions), alt=atm.altitudes * 1000.0)
    rng = np.random.default_rng(7)
    print(f"# C3b workload: {len(phot)} objects, {int(phot.sum())} photons (median {int(np.median(phot))}, mean {phot.mean():.0f}, max {int(phot.max())} per object)")
    print(f"# field angles span {np.ptp(thx) * 206265 / 60:.1f}' x {np.ptp(thy) * 206265 / 60:.1f}';  exposure {atm.exptime} s;  screens {npix}^2 x 0.1 m; LDS of a CU: 160 KB = 2 560 lines")
    print("# layer: altitude [m], wind [m/s], the strip one object's photons fall on (pupil 8.36 m wide, wind x exposure long) in samples and MB")
    for l in range(6):
        sp = float(atm.speeds[l])
        strip = (84 + 1) * (sp * atm.exptime / 0.1 + 84)
        print(f"#   {l}: {A['alt'][l]:7.0f} m  {sp:5.2f} m/s   {strip / 1e3:7.0f} k samples = {strip * 16 / 1e6:6.2f} MB (16-B cells) / {strip * 4 / 1e6:5.2f} MB (plain)")
    print("\n## (a) one object: distinct 64-byte lines / gathers, summed over the six layers")
    print(f"{'photons of the object':>24s} {'objects like it':>16s} {'share of all photons':>21s} {'16-B cells':>11s} {'plain':>7s}")
    edges = [1, 100, 1000, 10000, 100000, 1000000, 10 ** 9]
    for lo, hi in zip(edges[:-1], edges[1:]):
        sel = np.flatnonzero((phot >= lo) & (phot < hi))
        if sel.size == 0:
            continue
        pick = rng.choice(sel, size=min(12, sel.size), replace=False)
        q = p = g = 0
        for o in pick:
            n = int(min(phot[o], 2_000_000))
            for l in range(6):
                iy, ix = cells(rng, n, thx[o], thy[o], A, l)
                q += lines_quads(iy, ix, npix); p += lines_plain(iy, ix, npix); g += n
        print(f"{lo:>11d} ..{hi:>10d} {sel.size:>16d} {phot[sel].sum() / phot.sum():>21.3f} {q / g:>11.3f} {p / g:>7.3f}")
    print("\n## (b) the 256 photons a workgroup holds at a time (one object's, or 4 x 64 of four faint objects): lines / gathers")
    o = int(np.argmax(phot))
    q = p = 0
    for l in range(6):
        iy, ix = cells(rng, 256, thx[o], thy[o], A, l)
        q += lines_quads(iy, ix, npix); p += lines_plain(iy, ix, npix)
    print(f"   {q / (6 * 256):.3f} (16-B cells)   {p / (6 * 256):.3f} (plain): every gather a line of its own -- a tile staged in LDS would be read once")
    print("\n## (c) one launch (a share of 1/14 of the objects, every photon): distinct lines / gathers per layer, and the footprint")
    share = np.arange(0, len(phot), 14)
    tot = int(phot[share].sum())
    print(f"   {share.size} objects, {tot} photons")
    for l in range(6):
        iys, ixs = [], []
        for o in share:
            iy, ix = cells(rng, int(phot[o]), thx[o], thy[o], A, l)
            iys.append(iy); ixs.append(ix)
        iy, ix = np.concatenate(iys), np.concatenate(ixs)
        q, p = lines_quads(iy, ix, npix), lines_plain(iy, ix, npix)
        print(f"   layer {l}: {q / iy.size:6.3f} lines per gather = {q * 64 / 1e6:7.1f} MB touched (16-B cells);  {p / iy.size:6.3f} = {p * 64 / 1e6:6.1f} MB (plain)")
    print("\n# Reading: within an object and within a workgroup there is NO reuse to stage (a); (b) -- a photon's sample is a random place on a strip")
    print("# of 1 .. 70 MB.  Reuse exists only ACROSS the objects of a launch (c), at the scale of megabytes per layer: the business of the L2 /")
    print("# Infinity Cache, which ims_screen_prepass manages (EXPERIMENTS, round 3: 6 x less traffic, slower) -- not of a 160-KB LDS.")


if __name__ == "__main__":
    main()
