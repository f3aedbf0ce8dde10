"""Allocator behaviour of the focal-plane step: device allocations (hipMalloc calls) and reserved memory per CCD."""
import os, sys, time, gc
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from imsim_amd import configs, catalog
from imsim_amd.engine import Renderer

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
step = configs._c5_step(r, objects, concurrent=3)
for rep in range(3):
    s0 = torch.cuda.memory_stats()
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    s1 = torch.cuda.memory_stats()
    print(f"rep {rep}: {1e3 * dt / n_ccd:.1f} ms per CCD, device allocs {s1['num_device_alloc'] - s0['num_device_alloc']}, "
          f"device frees {s1['num_device_free'] - s0['num_device_free']}, reserved {s1['reserved_bytes.all.current'] / 2**30:.1f} GiB, "
          f"gc objects {len(gc.get_objects())}, gen2 collections {gc.get_stats()[2]['collections']}")
before = torch.cuda.memory_allocated() / 2**30
n = gc.collect()
print(f"gc.collect() freed {n} objects; allocated {before:.2f} -> {torch.cuda.memory_allocated() / 2**30:.2f} GiB")
big = [o for o in gc.get_objects() if isinstance(o, torch.Tensor) and o.is_cuda and o.numel() * o.element_size() > 50e6]
print("live big tensors:", sorted(((o.numel() * o.element_size()) >> 20 for o in big), reverse=True)[:40])
