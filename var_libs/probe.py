import os, sys, ctypes as C
sys.path.insert(0, ".")
import numpy as np, torch
from imsim_amd import configs, catalog, _abi
from imsim_amd.engine import Renderer
cfg = configs.BENCH_CONFIGS["c3"]
scene = cfg["scene"]()
cat = catalog.synthetic_catalog(100000, nx=scene.nx, ny=scene.ny)
phot = catalog.realize_fluxes(cat["nominal_flux"], scene.seed)
objects, _ = configs.c3_objects(cat, phot, scene)
top = objects[np.argsort(-objects["n_phot"])[:1]]
r = Renderer(scene)
step = r.prepared_lsst_image(top)
for _ in range(3):
    r.image.zero_(); step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter(); step(); torch.cuda.synchronize(); print("one star step ms", 1e3 * (time.perf_counter() - t0), "rounds", int((top["n_phot"][0] + 9999) // 10000))
lib = _abi.load()
out = (C.c_ulonglong * 16)()
lib.ims_upd_probe.argtypes = [C.c_void_p]
print("rc", lib.ims_upd_probe(out))
v = np.array(list(out), dtype=np.int64)
print("stamps (10 ns ticks rel. to 0):", (v[:7] - v[0]).tolist())
