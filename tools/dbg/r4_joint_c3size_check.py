"""Joint rounds on CCDs of C3's size (100 k objects, ~1 600 of them with rounds of their own: a middle class of 1 550 regions and
33 k tiles per CCD): three such CCDs rendered in one joint batch and with a chain per CCD -- same bits?  Run under gpurun."""
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

cfg = configs.BENCH_CONFIGS["c3"]
scene = cfg["scene"]()
tables = {}
for det in range(3):
    cat = catalog.synthetic_catalog(100000, seed=20261001 + det, nx=scene.nx, ny=scene.ny)
    phot = catalog.realize_fluxes(cat["nominal_flux"], scene.seed + det)
    tables[det], _ = cfg["objects"](cat, phot, scene)
    n = tables[det]["n_phot"]
    print(f"CCD {det}: {len(n)} objects, {int((n > 10000).sum())} with rounds of their own, brightest {int(n.max())} photons")


def build(det):
    sc = copy.copy(scene)
    sc.seed = ccd_seed(scene.seed, det)
    return sc, tables[det]


digests = {}
for joint in ("16", "0"):
    os.environ["IMS_FOCAL_JOINT"] = joint
    # this check is about the joint batch itself: the default sends CCDs of this size through the rolling window
    os.environ["IMS_FOCAL_JOINT_MAX_BRIGHT"] = "1000000" if joint != "0" else "600"
    out = {}
    for rep in range(2):
        t0 = time.perf_counter()
        focal_plane.render_focal_plane([0, 1, 2], build, concurrent=3, nrecalc=10000,
                                       sink=lambda det, image, out=out: out.__setitem__(det, hashlib.sha256(np.ascontiguousarray(image).tobytes()).hexdigest()))
        torch.cuda.synchronize()
        dt = 1e3 * (time.perf_counter() - t0)
    print(f"IMS_FOCAL_JOINT={joint}: {dt / 3:.1f} ms per CCD (second call), joint plans {getattr(focal_plane.render_focal_plane, 'last_joint_plans', 0) if joint != '0' else 0}")
    digests[joint] = out
same = [d for d in range(3) if digests["16"][d] == digests["0"][d]]
print("identical images:", len(same), "of 3", [digests["16"][d][:12] for d in range(3)])
assert len(same) == 3
