import os, sys
sys.path.insert(0, os.getcwd())
os.environ["IMS_TILE_LISTS"]="1"
import numpy as np, torch
from imsim_amd import configs, catalog, _abi
from imsim_amd.engine import Renderer
cfg = configs.BENCH_CONFIGS["c3"]; scene = cfg["scene"]()
cat = catalog.synthetic_catalog(100000, nx=scene.nx, ny=scene.ny)
phot = catalog.realize_fluxes(cat["nominal_flux"], scene.seed)
objects, _ = configs.c3_objects(cat, phot, scene)
r = Renderer(scene)
plan, _ = r.plan_lsst_image(objects)
comp = r._compile_plan(plan)
r.execute_plan(plan, comp); torch.cuda.synchronize()
for k in comp[1]:
    if isinstance(k, tuple) and len(k)==3 and hasattr(k[2],'dtype') and k[2].dtype==torch.int32 and k[0].dtype==torch.uint8:
        counts=k[2].cpu().numpy(); nt=k[1].numel()//max(len(counts),1)
        print("class tiles", nt, "rounds", len(counts), "listed per round: first", counts[:8], "mid", counts[len(counts)//2-2:len(counts)//2+2], "mean frac", counts.mean()/nt)
