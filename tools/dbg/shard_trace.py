"""Render one rank's shard of the C3 bench catalog a few times (for rocprofv3 --kernel-trace): python shard_trace.py <rank> <world>"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from imsim_amd import configs, catalog, parallel
from imsim_amd.engine import Renderer
rank, world = int(sys.argv[1]), int(sys.argv[2])
cfg = configs.BENCH_CONFIGS["c3"]
scene = cfg["scene"]()
cat = catalog.synthetic_catalog(100000, nx=scene.nx, ny=scene.ny)
phot = catalog.realize_fluxes(cat["nominal_flux"], scene.seed)
objects, _ = configs.c3_objects(cat, phot, scene)
r = Renderer(scene)
step = r.prepared_lsst_image(parallel.shard_objects(objects, rank, world))
for _ in range(3):
    r.image.zero_(); step()
torch.cuda.synchronize()
