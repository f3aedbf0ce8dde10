"""C5 at its full size (189 CCDs, or n) for the environment given: 1 warm-up + 3 timed steps.  Run under gpurun."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from imsim_amd import configs, catalog, focal_plane  # noqa: E402
from imsim_amd.engine import Renderer  # noqa: E402

n_ccd = int(sys.argv[1]) if len(sys.argv) > 1 else 189
t00 = time.perf_counter()
scene = configs.BENCH_CONFIGS["c5"]["scene"]()
cat = configs._c5_catalog(n_ccd * 10000, scene, n_ccd=n_ccd)
phot = catalog.realize_fluxes(cat["nominal_flux"], scene.seed)
objects, _ = configs._c5_objects(cat, phot, scene)
r = Renderer(scene, "cuda:0")
step = configs._c5_step(r, objects, concurrent=4)
if os.environ.get("R6_RESERVE"):
    print("reserved", focal_plane.reserve_host_images("cuda:0", (4096, 4096), int(os.environ["R6_RESERVE"])))
torch.cuda.synchronize()
print(f"set-up {time.perf_counter() - t00:.1f} s", flush=True)
tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("IMS_"))
import zlib
for k in range(int(os.environ.get("R5_CALLS", "4"))):
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"[{tag}] call {k}: {dt:.3f} s for {n_ccd} CCDs = {1e3 * dt / n_ccd:.2f} ms per CCD; host {focal_plane.render_focal_plane.last_host_ms_per_ccd:.2f} ms per CCD, "
          f"batch {getattr(focal_plane.render_focal_plane, 'last_joint_batch', 0)}, arena {getattr(focal_plane.render_focal_plane, 'last_arena_gib', 0):.1f} GiB, "
          f"reserved {torch.cuda.memory_reserved() / 2**30:.0f} GiB, crc {zlib.crc32(repr(sorted(step.checksums.items())).encode())}", flush=True)
