"""cProfile of the cold launch-plan construction of the C3 bench catalog (host planning + uploads)."""
import cProfile, pstats, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from imsim_amd import configs, catalog
from imsim_amd.engine import Renderer
cfg = configs.BENCH_CONFIGS["c3"]
scene = cfg["scene"]()
cat = catalog.synthetic_catalog(100000, nx=scene.nx, ny=scene.ny)
phot = catalog.realize_fluxes(cat["nominal_flux"], scene.seed)
objects, _ = configs.c3_objects(cat, phot, scene)
r = Renderer(scene)
r.plan_lsst_image(objects)          # warm
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
plan, _ = r.plan_lsst_image(objects)
compiled = r._compile_plan(plan)
torch.cuda.synchronize()
pr.disable()
print("plan ms", 1e3 * (time.perf_counter() - t0), "items", len(plan))
pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
