#!/usr/bin/env python
"""Time the three kernels of the pooled (LSST_Photons) path on a slice of the C3 catalog: shoot -> apply ops -> accumulate."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from imsim_amd import configs, catalog
from imsim_amd.engine import Renderer

scene = configs.scene_c3()
scene.track_static_delta = 1
cat = catalog.synthetic_catalog(100000, nx=scene.nx, ny=scene.ny)
phot = catalog.realize_fluxes(cat["nominal_flux"], scene.seed)
objects, _ = configs.c3_objects(cat, phot, scene)
objects = objects[objects["n_phot"] <= 20000][:60000]
objects["bf_state"] = 0
n = int(objects["n_phot"].sum())
r = Renderer(scene)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


t1, pool = timed(lambda: r.shoot_photons(objects))
t2, _ = timed(lambda: r.apply_ops(pool))
t3, _ = timed(lambda: r.accumulate(pool))
print(f"{n} photons: shoot {t1:.2f} ms ({n / t1 / 1e6:.2f} Gphot/s), apply_ops {t2:.2f} ms ({n / t2 / 1e6:.2f}), "
      f"accumulate {t3:.2f} ms ({n / t3 / 1e6:.2f})")
