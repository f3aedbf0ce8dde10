"""Where the host time of the C4 end-to-end path goes (cProfile of DeviceTable + prepared_image from it)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from imsim_amd import configs, catalog, photon_pooling, stamp
from imsim_amd.engine import Renderer
from imsim_amd.device_table import DeviceTable
cfg = configs.BENCH_CONFIGS["c4"]
scene = cfg["scene"]()
cat = catalog.synthetic_catalog(cfg["n_objects"], nx=scene.nx, ny=scene.ny)
r = Renderer(scene)
for rep in range(2):
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    t0 = time.perf_counter()
    table = DeviceTable(r, cat, dict(configs.VISIT))
    modes = np.where(table.n_phot.astype(np.float64) < 100.0, stamp.ProcessingMode.FAINT.value, stamp.ProcessingMode.PHOT.value)
    step = photon_pooling.prepared_image(r, table, modes, nbatch=10, seed=scene.seed)
    t1 = time.perf_counter()
    pr.disable()
    print("host ms", 1e3 * (t1 - t0))
    if rep == 1:
        pstats.Stats(pr).sort_stats("tottime").print_stats(18)
    del step, table
