"""Time parts of the C3 LSST_Image plan in isolation (which part is slowed by which)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
from imsim_amd import configs, catalog
from imsim_amd.engine import Renderer

cfg = configs.BENCH_CONFIGS["c3"]
scene = cfg["scene"]()
cat = catalog.synthetic_catalog(100000, nx=scene.nx, ny=scene.ny)
phot = catalog.realize_fluxes(cat["nominal_flux"], scene.seed)
objects, _ = configs.c3_objects(cat, phot, scene)
r = Renderer(scene)
plan, _ = r.plan_lsst_image(objects)
for it in plan:
    if it[0] == "slots":
        r.bound.set_private_slots(it[1])
plan = [it for it in plan if it[0] != "slots"]


def timeit(sub, name, n=3):
    comp = r._compile_plan(sub)
    for _ in range(2):
        r.execute_plan(sub, comp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    import time
    e0.record()
    t0 = time.perf_counter()
    for _ in range(n):
        r.execute_plan(sub, comp)
    t1 = time.perf_counter()
    e1.record()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:40s} {e0.elapsed_time(e1) / n:9.3f} ms   host enqueue {(t1 - t0) / n * 1e3:8.3f} ms  host total {(t2 - t0) / n * 1e3:8.3f} ms", flush=True)


only = sys.argv[1] if len(sys.argv) > 1 else None
kinds = lambda ks: [it for it in plan if it[0] in ks]
print({k: sum(1 for it in plan if it[0] == k) for k in set(it[0] for it in plan)})
if only == "chain":
    timeit([it for it in plan if it[0] in ("init", "acc_pool", "update")], "chain only", n=1)
    sys.exit(0)
if only == "full":
    timeit(plan, "full plan", n=1)
    sys.exit(0)
timeit(plan, "full plan")
timeit(kinds(("shoot_pool",)), "pool shoots only (2 streams)")
timeit(kinds(("render",)), "bulk fused only")
timeit(kinds(("shoot_pool", "render")), "pool shoots + bulk")
timeit(kinds(("init", "shoot_pool", "record", "wait", "acc_pool", "update")), "everything but bulk")
sub = [it for it in plan if it[0] in ("init", "acc_pool", "update")]
timeit(sub, "chain only (pool already filled)")
