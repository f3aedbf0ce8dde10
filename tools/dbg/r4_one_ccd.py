"""One CCD of the C5 catalog (round 4: bright tail), photon-shot part only, alone on the GPU -- for a kernel trace.
python tools/dbg/r4_one_ccd.py <det> [n_ccd]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from imsim_amd import configs, catalog, lsst_image  # noqa: E402
from imsim_amd.engine import Renderer  # noqa: E402

det = int(sys.argv[1]) if len(sys.argv) > 1 else 11
n_ccd = int(sys.argv[2]) if len(sys.argv) > 2 else 12
scene = configs.BENCH_CONFIGS["c5"]["scene"]()
cat = configs._c5_catalog(n_ccd * 10000, scene, n_ccd=n_ccd)
phot = catalog.realize_fluxes(cat["nominal_flux"], scene.seed)
objects, _ = configs._c5_objects(cat, phot, scene)
a, b = objects.cat_offsets[det], objects.cat_offsets[det + 1]
sub = {k: v[a:b] for k, v in cat.items() if isinstance(v, np.ndarray)}
job = configs.c5_job(scene, sub, phot[a:b], np.asarray(objects[objects.ccd_offsets[det]:objects.ccd_offsets[det + 1]]))
rows = job.objects
if os.environ.get("R4_TOP_ONLY"):
    rows = rows[np.argsort(-rows["n_phot"])[:int(os.environ["R4_TOP_ONLY"])]]
print("brightest", np.sort(rows["n_phot"])[::-1][:4], "stamps", (rows["stamp_xmax"] - rows["stamp_xmin"] + 1)[np.argsort(-rows["n_phot"])[:4]])
r = Renderer(scene, "cuda:0")
side = torch.cuda.Stream()
for k in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(side):
        r.render_lsst_image(rows, nrecalc=10000)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"run {k}: {1e3 * (time.perf_counter() - t0):.2f} ms (host enqueue {1e3 * (t1 - t0):.2f} ms)")
