"""Upper bound of what a faster spike stencil can give the C5 visit: the step with stamp.diffraction_fft disabled (NOT the workload of
the bench: an experiment).  python tools/dbg/c5_no_spikes.py [n_ccd] [0|1 spikes]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from imsim_amd import configs, catalog
from imsim_amd.engine import Renderer
n_ccd = int(sys.argv[1]) if len(sys.argv) > 1 else 64
spikes = (sys.argv[2] != "0") if len(sys.argv) > 2 else False
v = configs.c5_visit_fft()
if not spikes:
    v["diffraction_fft"].enabled = False
scene = configs.BENCH_CONFIGS["c5"]["scene"]()
cat = configs._c5_catalog(n_ccd * 10000, scene, n_ccd=n_ccd)
phot = catalog.realize_fluxes(cat["nominal_flux"], scene.seed)
objects, _ = configs._c5_objects(cat, phot, scene)
r = Renderer(scene, "cuda:0")
step = configs._c5_step(r, objects, concurrent=4)
for k in range(4):
    t0 = time.perf_counter(); step(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"spikes {spikes} call {k}: {dt:.3f} s = {1e3 * dt / n_ccd:.2f} ms per CCD", flush=True)
