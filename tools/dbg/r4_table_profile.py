"""Where the end-to-end path of C3 spends its host time: device table, native plan, render (cProfile of the table build)."""
import cProfile, pstats, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from imsim_amd import configs, catalog
from imsim_amd.engine import Renderer
from imsim_amd.device_table import DeviceTable
cfg = configs.BENCH_CONFIGS["c3"]
scene = cfg["scene"]()
cat = catalog.synthetic_catalog(100000, nx=scene.nx, ny=scene.ny)
r = Renderer(scene)
for _ in range(2):
    t = DeviceTable(r, cat, dict(configs.VISIT))
    p = r.native_plan(t); p.run(); torch.cuda.synchronize()
    del t, p
torch.cuda.synchronize()
for k in range(3):
    r.image.zero_(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    t = DeviceTable(r, cat, dict(configs.VISIT))
    t1 = time.perf_counter()
    p = r.native_plan(t)
    t2 = time.perf_counter()
    p.run()
    t3 = time.perf_counter()
    img = r.image_float()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print(f"table {1e3*(t1-t0):.2f} plan {1e3*(t2-t1):.2f} enqueue {1e3*(t3-t2):.2f} wait {1e3*(t4-t3):.2f} total {1e3*(t4-t0):.2f} ms")
    del t, p, img
pr = cProfile.Profile()
pr.enable()
t = DeviceTable(r, cat, dict(configs.VISIT))
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
