"""Where one brighter-fatter round of the brightest star goes: real-time stamps (10 ns ticks) written by thread 0 of the
middle workgroup of each of the three kernels in the LAST round of the chain.  Needs a -DIMS_PROBE build of the library:
   hipcc <flags of __graft_entry__> -DIMS_PROBE imsim_amd/csrc/imsim_hip.hip -o var_libs/probe.so
   IMSIM_HIP_LIB=$PWD/var_libs/probe.so python3 tools/dbg/round_probe.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from imsim_amd import configs, catalog, _abi
from imsim_amd.engine import Renderer

cfg = configs.BENCH_CONFIGS["c3"]
scene = cfg["scene"]()
cat = catalog.synthetic_catalog(100000, nx=scene.nx, ny=scene.ny)
phot = catalog.realize_fluxes(cat["nominal_flux"], scene.seed)
objects, _ = configs.c3_objects(cat, phot, scene)
top = objects[np.argsort(-objects["n_phot"])[:1]].copy()
# stop the chain after 100 full rounds so that the last round is a typical one (10 000 photons, update and refresh follow)
top["n_phot"] = 100 * 10000 + (5000 if len(sys.argv) < 2 else 0)
r = Renderer(scene)
step = r.prepared_lsst_image(top)
lib = _abi.load()
lib.ims_probe_read.argtypes = [C.c_void_p, C.c_void_p]
NAMES = {0: "acc: kernel entry", 1: "acc: object row in", 2: "acc: tile zeroed (barrier)", 3: "acc: photon loaded", 4: "acc: pixel found",
         5: "acc: barrier before flush", 6: "acc: flush issued", 8: "upd: kernel entry", 9: "upd: slot located", 10: "upd: halo staged",
         11: "upd: table ready", 12: "upd: points loaded", 13: "upd: window summed", 14: "upd: points stored", 16: "ref: kernel entry",
         17: "ref: tile flags read", 18: "ref: bounds written"}
for rep in range(4):
    r.image.zero_(); step(); torch.cuda.synchronize()
    out = (C.c_ulonglong * 32)()
    wg = (C.c_ulonglong * 512)()
    assert lib.ims_probe_read(out, wg) == 0
    if len(sys.argv) > 1:
        w = np.array(list(wg), dtype=np.int64).reshape(64, 8)[:40]
        t0 = w[:, 0].min()
        print(f"-- replay {rep}: accumulate of the last round, per workgroup (us from the first entry): entry, tile zeroed, photon in, own pixel tested, neighbours prefiltered (stale if none searched), pixel found, barrier, flush issued")
        for k in range(40):
            print(f"   wg {k:2d}  " + "  ".join(f"{(w[k, c] - t0) * 0.01:7.2f}" for c in range(8)))
        continue
    v = np.array(list(out), dtype=np.int64)
    # the last full round: accumulate of round 99 -> update -> refresh; the accumulate stamps are overwritten by round 100
    # (5 000 photons, no update behind it), so read accumulate from this tail round and update / refresh from round 99
    t0 = v[8]
    print(f"-- replay {rep}: update / refresh of round 99 (us from the update kernel's entry), then accumulate of round 100")
    for k in (8, 9, 10, 11, 12, 13, 14, 16, 17, 18, 0, 1, 2, 3, 4, 5, 6):
        print(f"   {NAMES[k]:32s} {(v[k] - t0) * 0.01:8.2f}")
