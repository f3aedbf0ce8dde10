"""Per-rank step times of the C3 bench when the catalog is sharded over N ranks, measured one rank after the other
on ONE GPU (no reduce): predicts the strong-scaling curve and exercises the planner on every shard."""
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
for world in (1, 2, 4, 8):
    times, host = [], []
    for rank in range(world):
        mine = parallel.shard_objects(objects, rank, world)
        step = r.prepared_lsst_image(mine)
        for _ in range(2):
            r.image.zero_(); step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            r.image.zero_(); step()
        host.append((time.perf_counter() - t0) / 3 * 1e3)          # host time to enqueue a step
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) / 3 * 1e3)
        del step
    print(f"world {world}: per-rank ms {np.round(times, 2)}  -> max {max(times):.2f} ms, predicted {100000 / max(times) * 1e3:.3g} objects/s; "
          f"host enqueue per step {np.round(host, 2)} ms", flush=True)
