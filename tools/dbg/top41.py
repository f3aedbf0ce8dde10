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
sub = objects[np.sort(order[:int(sys.argv[1]) if len(sys.argv) > 1 else 41])]
step = r.prepared_lsst_image(sub)
for _ in range(3):
    r.image.zero_(); step(); torch.cuda.synchronize()
