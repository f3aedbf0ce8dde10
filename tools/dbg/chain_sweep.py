"""Step time of the C3 bench (whole catalog, and the shards of an 8-rank run replayed on one GPU) for several settings
of the persistent brighter-fatter chain."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
from imsim_amd import configs, catalog, parallel
from imsim_amd.engine import Renderer
cfg = configs.BENCH_CONFIGS["c3"]
scene = cfg["scene"]()
cat = catalog.synthetic_catalog(100000, nx=scene.nx, ny=scene.ny)
phot = catalog.realize_fluxes(cat["nominal_flux"], scene.seed)
objects, _ = configs.c3_objects(cat, phot, scene)
r = Renderer(scene)
settings = [dict(use_chain=False)] + [dict(use_chain=True, chain_workers=w, chain_team=t)
                                      for (w, t) in [(160, 40), (320, 40), (480, 40), (320, 20), (480, 64)]]
if len(sys.argv) > 1:
    settings = [eval("dict(%s)" % a) for a in sys.argv[1:]]


def timed(mine):
    step = r.prepared_lsst_image(mine)
    for _ in range(2):
        r.image.zero_(); step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        r.image.zero_(); step()
    r.synchronize()
    return (time.perf_counter() - t0) / 3 * 1e3


for st in settings:
    for k, v in st.items():
        setattr(r, k, v)
    t1 = timed(parallel.shard_objects(objects, 0, 1))
    t8 = [timed(parallel.shard_objects(objects, rank, 8)) for rank in range(8)]
    print(st, "1 rank %.2f ms | 8 ranks: %s max %.2f ms" % (t1, np.round(t8, 2), max(t8)), flush=True)
