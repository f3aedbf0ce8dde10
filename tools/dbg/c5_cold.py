"""Where the first focal-plane call of a fresh process spends its time: host profile (cumulative) of the first step() of
bench config c5 on n CCDs, then the wall time of the second.  Run under gpurun: python tools/dbg/c5_cold.py [n_ccd]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from imsim_amd import configs, catalog, focal_plane  # noqa: E402
from imsim_amd.engine import Renderer  # noqa: E402

n_ccd = int(sys.argv[1]) if len(sys.argv) > 1 else 48
t_start = time.perf_counter()
torch.cuda.set_device(0)
if os.environ.get("R5_COLD_IMMEDIATE", "0") != "1":
    focal_plane.warm_fft("cuda:0")        # first thing, as config.Process and bench.py do: beside the host's own start-up work
scene = configs.BENCH_CONFIGS["c5"]["scene"]()
cat = configs._c5_catalog(n_ccd * 10000, scene, n_ccd=n_ccd)
phot = catalog.realize_fluxes(cat["nominal_flux"], scene.seed)
objects, _ = configs._c5_objects(cat, phot, scene)
r = Renderer(scene, "cuda:0")
step = configs._c5_step(r, objects, concurrent=4)
torch.cuda.synchronize()
print(f"set-up (catalogs, object tables, jobs of the CCDs): {time.perf_counter() - t_start:.1f} s; ROCFFT_RTC_CACHE_PATH = {os.environ.get('ROCFFT_RTC_CACHE_PATH')}")
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
step()
torch.cuda.synchronize()
pr.disable()
print(f"first call: {time.perf_counter() - t0:.2f} s for {n_ccd} CCDs")
t0 = time.perf_counter()
step()
torch.cuda.synchronize()
print(f"second call: {time.perf_counter() - t0:.2f} s")
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
pstats.Stats(pr).sort_stats("cumulative").print_stats(40)
