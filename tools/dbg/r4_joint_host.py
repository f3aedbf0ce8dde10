"""Where the host time of a focal plane's CCD goes in joint mode: cProfile of one C5 step over n CCDs (second step; the first warms
the allocator).  Run under gpurun: python tools/dbg/r4_joint_host.py [n_ccd]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from imsim_amd import configs, catalog, focal_plane  # noqa: E402
from imsim_amd.engine import Renderer  # noqa: E402

n_ccd = int(sys.argv[1]) if len(sys.argv) > 1 else 16
scene = configs.BENCH_CONFIGS["c5"]["scene"]()
cat = configs._c5_catalog(n_ccd * 10000, scene, n_ccd=n_ccd)
phot = catalog.realize_fluxes(cat["nominal_flux"], scene.seed)
objects, _ = configs._c5_objects(cat, phot, scene)
r = Renderer(scene, "cuda:0")
step = configs._c5_step(r, objects, concurrent=4)
for k in range(2):
    step()
    torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
step()
torch.cuda.synchronize()
pr.disable()
print(f"{1e3 * (time.perf_counter() - t0) / n_ccd:.1f} ms per CCD, host enqueue {focal_plane.render_focal_plane.last_host_ms_per_ccd:.1f} ms per CCD")
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
