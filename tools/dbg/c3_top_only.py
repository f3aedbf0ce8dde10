"""C3 without the MIDDLE class of bright objects (ranks 41 .. 1600): do the long chains run faster when the wide rounds of the
moderately bright objects are not on the GPU?"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from imsim_amd import configs, catalog
from imsim_amd.engine import Renderer
cfg = configs.BENCH_CONFIGS["c3"]
scene = cfg["scene"]()
cat = catalog.synthetic_catalog(100000, nx=scene.nx, ny=scene.ny)
phot = catalog.realize_fluxes(cat["nominal_flux"], scene.seed)
objects, _ = configs.c3_objects(cat, phot, scene)
order = np.argsort(-objects["n_phot"])
r = Renderer(scene)
for name, keep in (("all", order), ("top 41 + ordinary", np.concatenate([order[:41], order[1600:]])), ("top 41 only", order[:41]),
                   ("middle + ordinary", order[41:])):
    sub = objects[np.sort(keep)]
    step = r.prepared_lsst_image(sub)
    for _ in range(2):
        r.image.zero_(); step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        r.image.zero_(); step(); torch.cuda.synchronize()
    print(f"{name:22s} {1e3 * (time.perf_counter() - t0) / 5:6.2f} ms per step, {int(sub['n_phot'].sum()) / 1e6:6.1f} M photons", flush=True)
    del step
