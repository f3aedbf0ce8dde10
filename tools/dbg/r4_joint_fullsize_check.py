"""Full-size check of the joint rounds: CCDs of the C5 bench catalog (4096 x 4004, 10 k objects, FFT-drawn + photon-shot + faint,
stars of up to 1e7 photons) rendered (a) with the chains of the whole batch in joint launches and active-tile lists, (b) with a
chain per CCD -- every image must be the same bits; and CCD `ORC` against the CPU oracle's build of the same job (bit for bit
with the GPU's inverse transforms handed over, as in tests/test_parity_gpu.py).  Run under gpurun:
    python tools/dbg/r4_joint_fullsize_check.py [n_ccd] [oracle_ccd]"""
import copy
import hashlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from imsim_amd import configs, catalog, focal_plane  # noqa: E402
from imsim_amd.config import ccd_seed  # noqa: E402

n_ccd = int(sys.argv[1]) if len(sys.argv) > 1 else 12
orc_det = int(sys.argv[2]) if len(sys.argv) > 2 else 5
scene = configs.BENCH_CONFIGS["c5"]["scene"]()
cat = configs._c5_catalog(n_ccd * 10000, scene, n_ccd=n_ccd)
phot = catalog.realize_fluxes(cat["nominal_flux"], scene.seed)
objects, _ = configs._c5_objects(cat, phot, scene)
offs, coffs = objects.ccd_offsets, objects.cat_offsets
visit = configs.c5_visit_fft()
visit["diffraction_fft"].constants(visit["wavelength"])
jobs = {}
for det in range(n_ccd):
    sub = {k: v[coffs[det]:coffs[det + 1]] for k, v in cat.items() if isinstance(v, np.ndarray)}
    jobs[det] = configs.c5_job(scene, sub, phot[coffs[det]:coffs[det + 1]], np.asarray(objects[offs[det]:offs[det + 1]]), 10000, visit=visit)


def build(det):
    sc = copy.copy(scene)
    sc.seed = scene.seed if det == 0 else ccd_seed(scene.seed, det)
    return sc, jobs[det]


digests = {}
transforms = {}
for mode, joint in (("joint rounds, 16 CCDs per batch", "16"), ("a chain per CCD", "0")):
    os.environ["IMS_FOCAL_JOINT"] = joint
    out = {}

    def sink(det, image, out=out):
        out[det] = hashlib.sha256(np.ascontiguousarray(image).tobytes()).hexdigest()
        if det == orc_det and joint == "16":
            out["image"] = image.copy()

    def post(det, r):
        if det == orc_det and joint == "16" and getattr(r, "_keep_fft", None) is not None:
            transforms[det] = r._keep_fft[1].cpu().numpy()
    t0 = time.perf_counter()
    focal_plane.render_focal_plane(list(range(n_ccd)), build, concurrent=4, nrecalc=10000, sink=sink, post=post,
                                   chain_hint=lambda det: int(jobs[det].objects["n_phot"].max()))
    torch.cuda.synchronize()
    print(f"{mode}: {1e3 * (time.perf_counter() - t0) / n_ccd:.1f} ms per CCD (first run, allocator cold), "
          f"joint plans {getattr(focal_plane.render_focal_plane, 'last_joint_plans', 0) if joint != '0' else 0}")
    digests[mode] = out
a, b = digests["joint rounds, 16 CCDs per batch"], digests["a chain per CCD"]
same = [det for det in range(n_ccd) if a[det] == b[det]]
print(f"images identical (sha256 of the float32 CCD image), joint rounds vs a chain per CCD: {len(same)} of {n_ccd} CCDs")
for det in range(n_ccd):
    print(f"  CCD {det:3d}: {jobs[det].n_fft} FFT-drawn, brightest photon-shot {int(jobs[det].objects['n_phot'].max()):9d} photons "
          f"({int(jobs[det].objects['n_phot'].max()) // 10000 + 1} rounds)  {a[det][:16]}  {'==' if a[det] == b[det] else '!='}")
assert len(same) == n_ccd
# the oracle's build of one CCD
from oracle import orc_loader  # noqa: E402
sc, job = build(orc_det)
t0 = time.perf_counter()
orc = orc_loader.OracleScene(sc)
if job.n_fft:
    o = orc_loader.OracleFft(sc, job.kpsf, add_noise=True, diffraction_fft=job.diffraction_fft, wavelength=job.wavelength)
    o.finish(job.fft_rows, o.spikes(job.fft_rows, transforms[orc_det]))
    orc.image64 += o.image
orc.render_lsst_image(job.objects, nrecalc=job.nrecalc)
img = a["image"]
diff = int(np.count_nonzero(img != orc.image))
print(f"CCD {orc_det} through the joint rounds vs the CPU oracle ({time.perf_counter() - t0:.1f} s on one core): differing pixels {diff} of "
      f"{img.size}, flux {float(img.sum(dtype=np.float64)):.0f}")
assert diff == 0
