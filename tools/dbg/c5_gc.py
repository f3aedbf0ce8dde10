"""C5 over n CCDs with the cyclic garbage collector off during the step (is the host's time per CCD the collector's?)"""
import gc, os, sys, time, zlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from imsim_amd import configs, catalog, focal_plane
from imsim_amd.engine import Renderer
n_ccd = int(sys.argv[1]) if len(sys.argv) > 1 else 189
mode = sys.argv[2] if len(sys.argv) > 2 else "off"
scene = configs.BENCH_CONFIGS["c5"]["scene"]()
cat = configs._c5_catalog(n_ccd * 10000, scene, n_ccd=n_ccd)
phot = catalog.realize_fluxes(cat["nominal_flux"], scene.seed)
objects, _ = configs._c5_objects(cat, phot, scene)
r = Renderer(scene, "cuda:0")
step = configs._c5_step(r, objects, concurrent=4)
if mode == "freeze":
    gc.collect(); gc.freeze()
for k in range(5):
    if mode == "off":
        gc.collect(); gc.disable()
    t0 = time.perf_counter(); step(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    if mode == "off":
        gc.enable()
    print(f"gc {mode} call {k}: {dt:.3f} s = {1e3 * dt / n_ccd:.2f} ms per CCD; host {focal_plane.render_focal_plane.last_host_ms_per_ccd:.2f} ms per CCD, gc counts {gc.get_count()}", flush=True)
