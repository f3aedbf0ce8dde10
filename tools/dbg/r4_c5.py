"""C5 with the bright tail (round 4): per-CCD cost against the CCD's content (FFT grids, longest chain), and the step at
several depths of the pipeline.  Run under gpurun: python tools/dbg/r4_c5.py [n_ccd]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from imsim_amd import configs, catalog, focal_plane, lsst_image  # noqa: E402
from imsim_amd.engine import Renderer  # noqa: E402

n_ccd = int(sys.argv[1]) if len(sys.argv) > 1 else 12
scene = configs.BENCH_CONFIGS["c5"]["scene"]()
cat = configs._c5_catalog(n_ccd * 10000, scene, n_ccd=n_ccd)
phot = catalog.realize_fluxes(cat["nominal_flux"], scene.seed)
objects, _ = configs._c5_objects(cat, phot, scene)
r = Renderer(scene, "cuda:0")
if os.environ.get("R4_SKIP_SINGLE", "0") == "0":
    visit = configs.c5_visit_fft()
    for det in range(n_ccd):
        a, b = objects.cat_offsets[det], objects.cat_offsets[det + 1]
        sub = {k: v[a:b] for k, v in cat.items() if isinstance(v, np.ndarray)}
        job = configs.c5_job(scene, sub, phot[a:b], np.asarray(objects[objects.ccd_offsets[det]:objects.ccd_offsets[det + 1]]), visit=visit)
        ts = []
        for what in ("all", "fft", "phot"):
            j = job
            if what == "fft":
                j = lsst_image.CcdJob(objects=job.objects[:0], nrecalc=job.nrecalc, fft_rows=job.fft_rows, kpsf=job.kpsf,
                                      diffraction_fft=job.diffraction_fft, wavelength=job.wavelength)
            elif what == "phot":
                j = lsst_image.CcdJob(objects=job.objects, nrecalc=job.nrecalc)
            rr = Renderer(scene, "cuda:0")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            lsst_image.draw_job(rr, j)
            torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
            del rr
        nf = job.fft_rows["nfft"].tolist() if job.n_fft else []
        top = np.sort(job.objects["n_phot"])[::-1][:3].tolist()
        print(f"CCD {det}: all {ts[0]:.1f} ms, fft only {ts[1]:.1f}, phot only {ts[2]:.1f}; FFT grids {nf}; brightest photon-shot {top}")
for conc in [int(v) for v in os.environ.get("R4_CONC", "1,2,3,4").split(",")]:
    step = configs._c5_step(r, objects, concurrent=conc)
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    print(f"concurrent {conc}: {1e3 * (time.perf_counter() - t0) / n_ccd:.1f} ms per CCD "
          f"(host enqueue {getattr(focal_plane.render_focal_plane, 'last_host_ms_per_ccd', float('nan')):.1f} ms per CCD)")
