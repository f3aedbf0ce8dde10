"""C5 on n CCDs (default 64): steady-state ms per CCD for the environment given, the first (cold) call, and optionally a host
profile of one step.  Run under gpurun: python tools/dbg/r5_c5.py [n_ccd] [profile]"""
import os
import sys
import time

import numpy as np
import torch

import faulthandler
faulthandler.dump_traceback_later(int(os.environ.get("R5_WATCHDOG", "75")), exit=False)      # where the host sits if a call hangs
sys.path.insert(0, os.environ.get("R5_PKG_ROOT") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from imsim_amd import configs, catalog, focal_plane  # noqa: E402
from imsim_amd.engine import Renderer  # noqa: E402

n_ccd = int(sys.argv[1]) if len(sys.argv) > 1 else 64
scene = configs.BENCH_CONFIGS["c5"]["scene"]()
cat = configs._c5_catalog(n_ccd * 10000, scene, n_ccd=n_ccd)
phot = catalog.realize_fluxes(cat["nominal_flux"], scene.seed)
objects, _ = configs._c5_objects(cat, phot, scene)
r = Renderer(scene, "cuda:0")
step = configs._c5_step(r, objects, concurrent=4)
torch.cuda.synchronize()
tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("IMS_"))
sums = []
for k in range(3):
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sums.append(sorted(step.checksums.items()))
    print(f"[{tag}] call {k}: {dt:.3f} s for {n_ccd} CCDs = {1e3 * dt / n_ccd:.2f} ms per CCD; host {focal_plane.render_focal_plane.last_host_ms_per_ccd:.2f} ms per CCD, "
          f"batch {getattr(focal_plane.render_focal_plane, 'last_joint_batch', 0)}, arena {getattr(focal_plane.render_focal_plane, 'last_arena_gib', 0):.1f} GiB, "
          f"reserved {torch.cuda.memory_reserved() / 2**30:.0f} GiB", flush=True)
print("checksums equal over the calls:", sums[0] == sums[1] == sums[2], " checksum of the step:", float(sum(v for _, v in sums[0])), flush=True)
import zlib
print("crc of all CCD checksums:", zlib.crc32(repr(sums[0]).encode()), flush=True)
if len(sys.argv) > 2 and sys.argv[2] == "profile":
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    step()
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
