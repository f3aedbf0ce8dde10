"""Where one CCD of the C5 focal-plane step spends its host time (run under gpurun): cProfile of render_focal_plane over
a few CCDs, with and without stream overlap."""
import cProfile
import pstats
import sys
import time

import numpy as np
import torch

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from imsim_amd import configs, catalog, focal_plane  # noqa: E402
from imsim_amd.engine import Renderer  # noqa: E402

cfg = configs.BENCH_CONFIGS["c5"]
scene = cfg["scene"]()
n_ccd = int(sys.argv[1]) if len(sys.argv) > 1 else 12
cat = cfg["catalog"](configs.N_CCD_FOCAL_PLANE * 10000, scene)
keep = slice(0, int(cat.ccd_offsets[n_ccd]))
sub = configs._FocalPlaneCatalog({k: v[keep] for k, v in cat.items()})
sub.ccd_offsets = cat.ccd_offsets[:n_ccd + 1]
configs.N_CCD_FOCAL_PLANE = n_ccd
phot = catalog.realize_fluxes(sub["nominal_flux"], scene.seed)
objects, _ = cfg["objects"](sub, phot, scene)
r = Renderer(scene, "cuda:0")
if os.environ.get("C5_WITH_PG") == "1":
    # an RCCL communicator in the process, created after the device's streams (as bench.py does)
    import socket
    import torch.distributed as dist
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sk.getsockname()[1]))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    dist.barrier()
for conc in ((int(os.environ["C5_ONLY"]),) * 2 if "C5_ONLY" in os.environ else (1, 2, 3, 4)):
    step = configs._c5_step(r, objects, concurrent=conc)
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    print(f"concurrent {conc}: {1e3 * (time.perf_counter() - t0) / n_ccd:.1f} ms per CCD "
          f"(host enqueue {getattr(focal_plane.render_focal_plane, 'last_host_ms_per_ccd', float('nan')):.1f} ms per CCD)")
if "C5_ONLY" in os.environ:
    sys.exit(0)
step = configs._c5_step(r, objects, concurrent=3)
pr = cProfile.Profile()
pr.enable()
step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
