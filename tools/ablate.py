#!/usr/bin/env python
"""Time the fused kernel on the ordinary-object part of C3 with parts of the chain removed
(ablation for optimisation work; not part of the product)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imsim_amd import configs, catalog, _abi
from imsim_amd.engine import Renderer


def main():
    n_obj = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    base = configs.scene_c3()
    cat = catalog.synthetic_catalog(n_obj, nx=base.nx, ny=base.ny)
    phot = catalog.realize_fluxes(cat["nominal_flux"], base.seed)
    objects, _ = configs.c3_objects(cat, phot, base)
    objects = objects[objects["n_phot"] <= 10000]
    nph = int(objects["n_phot"].sum())
    full = base.ops
    variants = {
        "full+silicon": (full, True),
        "full,no sensor": (full, False),
        "no diffraction": ([full[0], full[1], full[2], (_abi.IMS_OP_RUBIN_OPTICS, 0, [1.0, 0.0]), full[4], full[5]], False),
        "samplers+dcr only": (full[:3], False),
        "no ops": ([], False),
        "no ops+silicon": ([], True),
    }
    for name, (ops, sens) in variants.items():
        sc = configs.scene_c3(sensor=sens)
        sc.ops = list(ops)
        r = Renderer(sc)
        launch = r.prepared(objects)
        for _ in range(2):
            launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            launch()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        print(f"{name:22s} {ms:9.3f} ms  {nph / ms / 1e6:8.3f} Gphot/s   ({nph} photons, {len(objects)} objects)", flush=True)
        del r, launch
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
