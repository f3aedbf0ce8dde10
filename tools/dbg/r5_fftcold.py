"""Host time of the four library calls of an FFT draw (k-space fill, inverse transforms, spikes, finish) for the first CCDs of a
fresh process: which of them is slow the first time, and for how many CCDs.  Run under gpurun."""
import os
import sys
import time
import ctypes as C

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from imsim_amd import configs, catalog, fft_draw, _abi  # noqa: E402
from imsim_amd.engine import Renderer  # noqa: E402

n_ccd = 14
scene = configs.BENCH_CONFIGS["c5"]["scene"]()
cat = configs._c5_catalog(n_ccd * 10000, scene, n_ccd=n_ccd)
phot = catalog.realize_fluxes(cat["nominal_flux"], scene.seed)
objects, _ = configs._c5_objects(cat, phot, scene)
visit = configs.c5_visit_fft()
visit["diffraction_fft"].constants(visit["wavelength"])
r = Renderer(scene, "cuda:0")
lib = r.lib
real = {}
for name in ("ims_fft_kspace_fill", "ims_fft_inverse", "ims_fft_spikes", "ims_fft_finish"):
    real[name] = getattr(lib, name)


class Timed:
    def __init__(self, name):
        self.name, self.t = name, []

    def __call__(self, *a):
        t0 = time.perf_counter()
        rc = real[self.name](*a)
        self.t.append(1e3 * (time.perf_counter() - t0))
        return rc


timers = {n: Timed(n) for n in real}


class LibProxy:
    def __getattr__(self, n):
        return timers[n] if n in timers else getattr(lib, n)


r.lib = LibProxy()
side = torch.cuda.Stream()
for det in range(n_ccd):
    a, b = objects.cat_offsets[det], objects.cat_offsets[det + 1]
    sub = {k: v[a:b] for k, v in cat.items() if isinstance(v, np.ndarray)}
    job = configs.c5_job(scene, sub, phot[a:b], np.asarray(objects[objects.ccd_offsets[det]:objects.ccd_offsets[det + 1]]), visit=visit)
    if not job.n_fft:
        print(f"CCD {det}: no FFT object")
        continue
    for t in timers.values():
        t.t.clear()
    drawer = fft_draw.FftDrawer(r, job.kpsf, add_noise=True, diffraction_fft=job.diffraction_fft, wavelength=job.wavelength)
    t0 = time.perf_counter()
    state = drawer._upload(job.fft_rows)
    t1 = time.perf_counter()
    with torch.cuda.stream(side):
        drawer._run(state, None)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(f"CCD {det}: grids {job.fft_rows['nfft'].tolist()}: upload {1e3 * (t1 - t0):.1f} ms, run (host) {1e3 * (t2 - t1):.1f} ms, device {1e3 * (t3 - t2):.1f} ms; "
          + ", ".join(f"{n[8:]} {sum(t.t):.1f}" for n, t in timers.items()), flush=True)
