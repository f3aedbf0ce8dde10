"""Predicted strong scaling of C5 over the GPUs of a node: the CCDs of rank r of an N-rank run (CCD i -> rank i mod N, no exchange)
rendered on ONE GPU, for every rank of N = 1, 2, 4, 8 -- the slowest rank is the time of the N-GPU step (the ranks do not
interact).  Run under gpurun: python tools/dbg/r4_c5_shards.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from imsim_amd import configs, catalog  # noqa: E402
from imsim_amd.engine import Renderer  # noqa: E402

cfg = configs.BENCH_CONFIGS["c5"]
scene = cfg["scene"]()
cat = cfg["catalog"](cfg["n_objects"], scene)
phot = catalog.realize_fluxes(cat["nominal_flux"], scene.seed)
objects, _ = cfg["objects"](cat, phot, scene)
r = Renderer(scene, "cuda:0")
n_ccd = len(objects.ccd_offsets) - 1
for world in (1, 2, 4, 8):
    times = []
    for rank in range(world if world <= 2 else 2):          # (the ranks' shards are statistically alike: two of them are timed)
        step = configs._c5_step(r, objects, rank=rank, world=world)
        step()
        torch.cuda.synchronize()
        ts = []
        for _ in range(2):
            t0 = time.perf_counter()
            step()
            torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
        times.append(min(ts))
        del step
    t = max(times)
    print(f"world {world}: {len(range(0, n_ccd, world))} CCDs on the slowest rank timed, {t:.0f} ms -> predicted {n_ccd * 10000 / (t * 1e-3):.3g} objects/s "
          f"(ranks timed: {np.round(times, 0)})")
