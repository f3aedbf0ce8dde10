"""What the CCD image reduce costs through RCCL with ONE rank (self-exchange; run under gpurun): per call, by GPU events and
by host time -- tells apart a slow single-rank copy kernel from a host synchronisation inside the collective."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import socket
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), IMS_EXCHANGE_SINGLE_RANK="1")
import torch, torch.distributed as dist
from imsim_amd import parallel
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
img = torch.randint(0, 1000, (4096, 4096), device="cuda").to(torch.float64)
for integer in (True, False):
    for _ in range(3):
        parallel.reduce_image(img, 0, integer_counts=integer)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(20):
        parallel.reduce_image(img, 0, integer_counts=integer)
    e1.record(); host = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"integer_counts={integer}: GPU {e0.elapsed_time(e1) / 20:.3f} ms per call, host enqueue {1e3 * host / 20:.3f} ms per call")
dist.destroy_process_group()
