"""Per-phase time of the brightest object's chain in its last persistent launch (library built with -DIMS_BFC_PROFILE):
   IMSIM_HIP_LIB=imsim_amd/lib/variants/libimsim_hip_prof.so python tools/dbg/chain_prof.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from imsim_amd import configs, catalog
from imsim_amd.engine import Renderer

cfg = configs.BENCH_CONFIGS["c3"]
scene = cfg["scene"]()
cat = catalog.synthetic_catalog(100000, nx=scene.nx, ny=scene.ny)
phot = catalog.realize_fluxes(cat["nominal_flux"], scene.seed)
objects, _ = cfg["objects"](cat, phot, scene)
if len(sys.argv) > 1 and sys.argv[1] == "bright":
    objects = objects[objects["n_phot"] > 400000]
r = Renderer(scene, "cuda:0")
step = r.prepared_lsst_image(objects)
for k in range(3):
    r.image.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    print("step ms %.2f" % (1e3 * (time.perf_counter() - t0)))
names = ["phaseA", "barA", "assign1", "update", "barB1", "refresh", "barB2"]
for name, t in r._chain_ctl.items():
    raw = t.cpu().numpy()
    prof = raw[-128:].view(np.uint64)
    rounds = max(int(prof[12]), 1)
    print(name, "team", int(prof[8]), "tiles", int(prof[9]), "rounds", rounds, "tiles/round upd %.1f ref %.1f" % (prof[10] / rounds, prof[11] / rounds))
    print("   us/round:", " ".join("%s %.1f" % (n, prof[k] * 0.01 / rounds) for k, n in enumerate(names)), " total %.1f" % (prof[:7].sum() * 0.01 / rounds))
