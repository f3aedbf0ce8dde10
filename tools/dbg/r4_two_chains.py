"""Do the long brighter-fatter chains of two CCDs overlap on the GPU?  The brightest star of CCD a and of CCD b, each on its own
renderer / stream / host thread: one after the other against side by side (round 4, C5 analysis)."""
import os, sys, time, threading
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from imsim_amd import configs, catalog
from imsim_amd.engine import Renderer

n_ccd = 12
scene = configs.BENCH_CONFIGS["c5"]["scene"]()
cat = configs._c5_catalog(n_ccd * 10000, scene, n_ccd=n_ccd)
phot = catalog.realize_fluxes(cat["nominal_flux"], scene.seed)
objects, _ = configs._c5_objects(cat, phot, scene)
n_chains = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dets = [11, 9, 5, 1][:n_chains]
rows = []
for det in dets:
    r_ = np.asarray(objects[objects.ccd_offsets[det]:objects.ccd_offsets[det + 1]])
    r_ = r_[~objects.fft_mask[objects.ccd_offsets[det]:objects.ccd_offsets[det + 1]]]
    rows.append(r_[np.argsort(-r_["n_phot"])[:1]].copy())
    print("CCD", det, "star", rows[-1]["n_phot"], "stamp", rows[-1]["stamp_xmax"] - rows[-1]["stamp_xmin"] + 1)
os.environ["IMS_FOCAL_TOPS"] = str(n_chains)
if os.environ.get("R4_PRIVATE"):
    os.environ["IMS_PRIVATE_STREAMS"] = "1"            # every renderer its own five plan streams: nothing shared between the chains
    rs = [Renderer(scene, "cuda:0") for k in range(n_chains)]
else:
    rs = [Renderer(scene, "cuda:0", stream_roles="focal", top_index=k) for k in range(n_chains)]
side = [torch.cuda.Stream() for _ in range(n_chains)]


def go(k):
    torch.cuda.set_device(0)
    with torch.cuda.stream(side[k]):
        rs[k].render_lsst_image(rows[k], nrecalc=10000)


for k in range(n_chains):
    go(k)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(n_chains):
    go(k)
    torch.cuda.synchronize()
print(f"one after the other: {1e3 * (time.perf_counter() - t0):.1f} ms")
t0 = time.perf_counter()
for k in range(n_chains):
    go(k)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"one host thread, {n_chains} streams: {1e3 * (time.perf_counter() - t0):.1f} ms (host enqueue {1e3 * (t1 - t0):.1f} ms)")
t0 = time.perf_counter()
th = [threading.Thread(target=go, args=(k,)) for k in range(n_chains)]
[t.start() for t in th]
[t.join() for t in th]
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"{n_chains} host threads, {n_chains} streams: {1e3 * (time.perf_counter() - t0):.1f} ms (host enqueue {1e3 * (t1 - t0):.1f} ms)")
