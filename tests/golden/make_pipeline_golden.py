#!/usr/bin/env python
"""Golden digests of the numerics spec (spec v6), produced with the CPU oracle:

    python tests/golden/make_pipeline_golden.py        ->  tests/golden/pipeline_golden.json

Every entry is the SHA-256 of the raw little-endian bytes of an output array of a small, seeded case.  They do not
come from the reference (GalSim is not installable here); they freeze the spec, so that an unintended change of
the arithmetic -- in the oracle or in the HIP kernels -- is caught even when both sides change together.
Regenerate (and say so in DESIGN.md) when the spec is changed on purpose."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def cases(make_backend):
    """name -> array, computed with the given backend factory (scene -> object with the OracleScene API)."""
    from helpers import small_case, c3_small_case
    from imsim_amd import photon_pooling, stamp
    out = {}
    scene, objects, _ = small_case(n_obj=120, nx=192, ny=192, flux_seed=2)
    b = make_backend(scene)
    b.render(objects)
    out["c2_image"] = b.image64_host()
    scene, objects = c3_small_case(n_obj=60, n=160, flux_seed=3, sensor=False)
    b = make_backend(scene)
    pool = b.shoot_photons(objects)
    b.apply_ops(pool)
    h = pool.to_host()
    for f in ("x", "y", "dxdz", "dydz", "wavelength", "flux"):
        out["c3_photons_" + f] = h[f]
    scene, objects = c3_small_case(n_obj=80, n=160, flux_seed=4)
    b = make_backend(scene)
    b.render_lsst_image(objects)
    out["c3_lsst_image"] = b.image64_host()
    scene, objects = c3_small_case(n_obj=50, n=128, flux_seed=5, scratch=0)
    scene.track_static_delta = 1
    b = make_backend(scene)
    photon_pooling.build_image(b, objects, stamp.classify(objects["n_phot"].astype(float), 100.0), nbatch=3, nsubbatch=2, seed=7)
    out["pooling_image"] = b.image64_host()
    return out


def oracle_backend(scene):
    from oracle import orc_loader

    class B(orc_loader.OracleScene):
        def image64_host(self):
            return self.image64
    return B(scene)


if __name__ == "__main__":
    digests = {k: digest(v) for k, v in cases(oracle_backend).items()}
    with open(os.path.join(HERE, "pipeline_golden.json"), "w") as f:
        json.dump({"spec": "v6", "sha256": digests}, f, indent=1, sort_keys=True)
    print(json.dumps(digests, indent=1))
