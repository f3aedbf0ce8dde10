#!/usr/bin/env python
"""Time the CCD readout chain (CcdReadout.build_amp_images) for one full E2V CCD on the GPU, and the oracle beside it."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from imsim_amd import readout, camera


def main():
    cam = camera.Camera("LsstCamSim")
    det = "R22_S11"
    ny, nx = cam[det].bounds.numpyShape()
    rng = np.random.default_rng(1)
    e = rng.poisson(800.0, size=(ny, nx)).astype(np.float64)
    for _ in range(200):
        y, x, s = rng.integers(0, ny), rng.integers(0, nx), rng.integers(1, 5)
        e[max(y - s, 0):y + s, max(x - s, 0):x + s] += np.round(rng.uniform(0.2, 30.0) * 1e5)
    hdr = readout.eimage_header(det, 30.0)
    base = torch.from_numpy(e).cuda()
    eimg = readout.EImage(base.clone(), hdr)
    ro = readout.CcdReadout(eimg, camera_obj=cam)
    for _ in range(2):
        eimg.array.copy_(base)
        ro.build_amp_images(3)
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        eimg.array.copy_(base)
        ro.build_amp_images(3)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(f"GPU readout chain, 4096x4004 E2V CCD -> 16 x 2048 x 576 int32: {ms:.2f} ms per CCD")
    if "--oracle" in sys.argv:
        from oracle import orc_loader
        t0 = time.perf_counter()
        orc_loader.readout_chain(e, ro.descriptor(), ro.full_well, ro.midline_stop(), ro.dark_level(), readout.DARK_STREAM, 3,
                                 ro.pcte_band, ro.scte_band)
        print(f"oracle (1 core): {(time.perf_counter() - t0) * 1e3:.0f} ms per CCD")


if __name__ == "__main__":
    main()
