"""What bounds the C3 step: the step time with the N brightest objects (the longest brighter-fatter chains) left out."""
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
for drop in (0, 1, 8, 41, 200, 1600):
    sub = objects[np.sort(order[drop:])]
    step = r.prepared_lsst_image(sub)
    for _ in range(2):
        r.image.zero_(); step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        r.image.zero_(); step()
        torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    print(f"without the {drop:4d} brightest: {ms:6.2f} ms per step, {int(sub['n_phot'].sum()) / 1e6:6.1f} M photons, "
          f"{int(sub['n_phot'].sum()) / ms / 1e6:.2f} G photons/s, longest chain {int((sub['n_phot'].max() + 9999) // 10000)} rounds", flush=True)
    del step
