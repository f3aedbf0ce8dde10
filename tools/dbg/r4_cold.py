"""Cold start of a focal plane: the first call of a process (every renderer's gigabytes come out of hipMalloc) against the second,
for joint batches of 16 / 8 CCDs and a chain per CCD.  Run under gpurun: python tools/dbg/r4_cold.py [n_ccd] [joint]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
n_ccd = int(sys.argv[1]) if len(sys.argv) > 1 else 48
os.environ["IMS_FOCAL_JOINT"] = sys.argv[2] if len(sys.argv) > 2 else "16"
from imsim_amd import configs, catalog, focal_plane  # noqa: E402
from imsim_amd.engine import Renderer  # noqa: E402

scene = configs.BENCH_CONFIGS["c5"]["scene"]()
cat = configs._c5_catalog(n_ccd * 10000, scene, n_ccd=n_ccd)
phot = catalog.realize_fluxes(cat["nominal_flux"], scene.seed)
objects, _ = configs._c5_objects(cat, phot, scene)
r = Renderer(scene, "cuda:0")
step = configs._c5_step(r, objects, concurrent=4)
torch.cuda.synchronize()
for k in range(3):
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    print(f"joint {os.environ['IMS_FOCAL_JOINT']}: call {k}: {time.perf_counter() - t0:.2f} s for {n_ccd} CCDs "
          f"({1e3 * (time.perf_counter() - t0) / n_ccd:.1f} ms per CCD), reserved {torch.cuda.memory_reserved() / 2**30:.0f} GiB")
