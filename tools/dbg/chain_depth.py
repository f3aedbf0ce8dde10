"""Why do back-to-back replays of the chain run 3x slower than a single replay?"""
import os, sys, time
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
r.execute_plan(plan)                      # fill the pool
torch.cuda.synchronize()
sub = [it for it in plan if it[0] in ("init", "acc_pool", "update")]
comp = r._compile_plan(sub)


def run(name, n, sync_each=False, sleep=0.0):
    for _ in range(2):
        r.execute_plan(sub, comp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r.execute_plan(sub, comp)
        if sync_each:
            torch.cuda.synchronize()
        if sleep:
            time.sleep(sleep)
    torch.cuda.synchronize()
    print(f"{name:40s} {(time.perf_counter() - t0) / n * 1e3 - sleep * 1e3:9.3f} ms per replay", flush=True)


run("1 replay", 1)
if len(sys.argv) > 1:
    sys.exit(0)
run("2 replays, no sync", 2)
run("3 replays, no sync", 3)
run("6 replays, no sync", 6)
run("3 replays, sync each", 3, sync_each=True)
run("3 replays, 20 ms host sleep each", 3, sleep=0.02)
