"""What a listed tile of a joint round holds and what it costs ("show the same by TIME"): C5 over n CCDs with
a -DIMS_HIST build of the library, then the histogram over the number of charged cells in the update's 23 x 23 halo: tiles,
time inside update_listed_tile (thread 0, 10-ns ticks) and charged cells inside the tile itself.
   hipcc <flags of __graft_entry__> -DIMS_HIST imsim_amd/csrc/imsim_hip.hip -o var_libs/hist.so
   IMSIM_HIP_LIB=$PWD/var_libs/hist.so python3 tools/dbg/c5_tile_hist.py [n_ccd]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from imsim_amd import configs, catalog, _abi  # noqa: E402
from imsim_amd.engine import Renderer  # noqa: E402

n_ccd = int(sys.argv[1]) if len(sys.argv) > 1 else 40
scene = configs.BENCH_CONFIGS["c5"]["scene"]()
cat = configs._c5_catalog(n_ccd * 10000, scene, n_ccd=n_ccd)
phot = catalog.realize_fluxes(cat["nominal_flux"], scene.seed)
objects, _ = configs._c5_objects(cat, phot, scene)
r = Renderer(scene, "cuda:0")
step = configs._c5_step(r, objects, concurrent=4)
lib = _abi.load()
lib.ims_hist_read.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_ulonglong * 192)()
step(); torch.cuda.synchronize()
assert lib.ims_hist_read(buf, 1) == 0
t0 = time.perf_counter()
step(); torch.cuda.synchronize()
dt = time.perf_counter() - t0
assert lib.ims_hist_read(buf, 1) == 0
h = np.array(list(buf), dtype=np.float64).reshape(3, 64)
tiles, ticks, own = h
print(f"# C5, {n_ccd} CCDs, one step of {dt:.3f} s with the IMS_HIST build: listed tiles of k_update_list_j by charged cells in the 23 x 23 halo")
print(f"# total listed tiles {tiles.sum():.0f}, time inside update_listed_tile {ticks.sum() * 1e-5:.1f} ms (thread 0 of each; x4 wave slots)")
print("# halo cells   tiles    share   cum    time_ms  share   cum    us/tile  own-tile cells/tile")
ct = cs = 0.0
for b in range(64):
    if tiles[b] == 0:
        continue
    ct += tiles[b] / tiles.sum(); cs += ticks[b] / ticks.sum()
    print(f"{b if b < 63 else '63+':>10}  {tiles[b]:8.0f}  {tiles[b] / tiles.sum():6.3f} {ct:6.3f}  {ticks[b] * 1e-5:8.2f} {ticks[b] / ticks.sum():6.3f} {cs:6.3f}  "
          f"{ticks[b] * 1e-2 / tiles[b]:7.2f}  {own[b] / tiles[b]:6.2f}")
