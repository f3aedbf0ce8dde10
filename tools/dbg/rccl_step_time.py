"""C3 step (host enqueue and elapsed time) before an RCCL communicator exists in the process, with it, and after it is
destroyed (run under gpurun)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import socket
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), IMS_EXCHANGE_SINGLE_RANK="1")
import numpy as np, torch, torch.distributed as dist
from imsim_amd import configs, catalog, parallel
from imsim_amd.engine import Renderer
cfg = configs.BENCH_CONFIGS["c3"]
scene = cfg["scene"]()
cat = catalog.synthetic_catalog(100000, nx=scene.nx, ny=scene.ny)
phot = catalog.realize_fluxes(cat["nominal_flux"], scene.seed)
objects, _ = cfg["objects"](cat, phot, scene)
r = Renderer(scene)
step = cfg["make_step"](r, objects, 0, 1)
def loop(tag, reduce=False, n=6):
    for it in range(2 + n):
        if it == 2:
            torch.cuda.synchronize(); t0 = time.perf_counter(); host = 0.0
        h0 = time.perf_counter()
        r.image.zero_(); step()
        if reduce: parallel.reduce_image(r.image, 0, integer_counts=True)
        if it >= 2: host += time.perf_counter() - h0
    torch.cuda.synchronize()
    nthreads = len(os.listdir(f"/proc/{os.getpid()}/task"))
    print(f"{tag:34s} {1e3 * (time.perf_counter() - t0) / n:7.2f} ms per step, host enqueue {1e3 * host / n:6.2f} ms per step, {nthreads} threads", flush=True)
loop("before any process group")
if os.environ.get("PG_NO_DEVICE_ID") == "1":
    dist.init_process_group("nccl", rank=0, world_size=1)
else:
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
loop("process group initialised")
loop("  + image reduce in the step", reduce=True)
loop("  after the first collectives")
dist.destroy_process_group()
loop("process group destroyed")
