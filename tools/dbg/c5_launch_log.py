"""Per launch of k_update_sparse_j in one C5 step (IMS_HIST build): listed tiles n, span from the first wavefront's start to the last one's
end, summed wavefront time -> how throughput (tiles per us, wavefronts in flight) depends on n.
   IMSIM_HIP_LIB=$PWD/var_libs/hist.so python3 tools/dbg/c5_launch_log.py [n_ccd]"""
import ctypes as C
import os
import sys

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
lib.ims_launch_log_read.argtypes = [C.c_void_p, C.c_int]
step(); torch.cuda.synchronize()
lib.ims_launch_seq(1)
step(); torch.cuda.synchronize()
n_l = min(lib.ims_launch_seq(0), 32768)
buf = (C.c_ulonglong * (5 * n_l))()
assert lib.ims_launch_log_read(buf, n_l) == 0
a = np.array(list(buf), dtype=np.float64).reshape(n_l, 5)
np.save(os.environ.get("R6_LOG_OUT", "/tmp/launch_log.npy"), a)
n, span, grid = a[:, 0], (a[:, 2] - a[:, 1]) * 1e-2, a[:, 4]
ok = (a[:, 2] > 0) & (span > 0)
n, span, grid = n[ok], span[ok], grid[ok]
print(f"# {n_l} launches of k_update_sparse_j in one step of {n_ccd} CCDs; span = workgroup 0's start .. the last workgroup's end (in-kernel clock)")
print("# listed tiles n   launches   mean grid   sum span ms   mean span us   tiles/us")
edges = [0, 1, 50, 100, 200, 400, 800, 1600, 3200, 6400, 12800, 25600, 1e9]
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (n >= lo) & (n < hi)
    if not m.any():
        continue
    print(f"{int(lo):>7} .. {int(min(hi, 1e7)):>8}  {m.sum():7d}  {grid[m].mean():8.0f}  {span[m].sum() * 1e-3:10.1f}  {span[m].mean():10.1f}  {n[m].sum() / span[m].sum():9.2f}")
print(f"# all: {len(n)} launches, sum span {span.sum() * 1e-3:.1f} ms, tiles {n.sum():.0f}")

# sampled launches: when do the workgroups of one launch start and how long do they live (wavefront 0 of each)
lib.ims_wg_log_read.argtypes = [C.c_void_p]
wbuf = (C.c_ulonglong * (64 * 1024 * 2))()
assert lib.ims_wg_log_read(wbuf) == 0
w = np.array(list(wbuf), dtype=np.float64).reshape(64, 1024, 2)
print("# sampled launches: listed tiles n, grid, then over the workgroups (wavefront 0): start offset from the first start [us] p50 / p90 / max, "
      "life [us] of those with a tile (workgroup < n / 4) p50 / p90 / max, of the empty ones p50 / max, last end - first start")
for k in range(min(64, n_l // 64)):
    seq = k * 64
    nk, gk = int(a[seq, 0]), int(a[seq, 4])
    g = min(gk, 1024)
    st, en = w[k, :g, 0], w[k, :g, 1]
    okk = (st > 0) & (en >= st)
    if not okk.any():
        continue
    t0 = st[okk].min()
    off = (st[okk] - t0) * 1e-2
    life = (en - st) * 1e-2
    busy = np.arange(g) < (nk + 3) // 4
    lb, le = life[okk & busy], life[okk & ~busy]
    f = lambda v, q: np.percentile(v, q) if len(v) else float("nan")
    print(f"launch {seq:5d}: n {nk:6d} grid {gk:5d} | start {f(off, 50):6.1f} {f(off, 90):6.1f} {off.max():6.1f} | busy life {f(lb, 50):6.1f} {f(lb, 90):6.1f} {lb.max() if len(lb) else 0:6.1f} "
          f"| empty life {f(le, 50):5.1f} {le.max() if len(le) else 0:6.1f} | span {(en[okk].max() - t0) * 1e-2:7.1f}")
