"""The brighter-fatter chain of the brightest star of the C3 catalog alone on the GPU: step time and per-round latency.
Run under `rocprofv3 --kernel-trace --stats` (tools/dbg/one_star.sh) for the three kernels' durations."""
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
top = objects[np.argsort(-objects["n_phot"])[:1]].copy()
print("stamp", top["stamp_xmin"], top["stamp_xmax"], top["stamp_ymin"], top["stamp_ymax"], "centre", top["x0"], top["y0"])
if os.environ.get("STAMP"):            # experiment: a smaller stamp (private region) around the star
    h = int(os.environ["STAMP"]) // 2
    cx, cy = int(top["x0"][0]), int(top["y0"][0])
    top["stamp_xmin"], top["stamp_xmax"] = max(cx - h, 1), min(cx + h, scene.nx)
    top["stamp_ymin"], top["stamp_ymax"] = max(cy - h, 1), min(cy + h, scene.ny)
r = Renderer(scene)
step = r.prepared_lsst_image(top)
for _ in range(3):
    r.image.zero_(); step()
torch.cuda.synchronize()
n = 5
enq = 0.0
t0 = time.perf_counter()
for _ in range(n):
    r.image.zero_()
    t1 = time.perf_counter()
    step()
    enq += time.perf_counter() - t1
    torch.cuda.synchronize()
ms = 1e3 * (time.perf_counter() - t0) / n
print(f"host enqueue {1e3 * enq / n:.3f} ms per step")
rounds = int((top["n_phot"][0] + 9999) // 10000)
print(f"one star: {ms:.3f} ms per step, {rounds} rounds, {1e3 * ms / rounds:.1f} us per round", flush=True)
