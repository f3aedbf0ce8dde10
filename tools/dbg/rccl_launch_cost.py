"""Host cost of a kernel launch before and after an RCCL communicator exists in the process (run under gpurun)."""
import os, sys, time, threading
import socket
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
import torch, torch.distributed as dist
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from imsim_amd import _abi
x = torch.zeros(64, device="cuda")
streams = [torch.cuda.Stream() for _ in range(4)]
def probe(tag):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(4000):
        with torch.cuda.stream(streams[i & 3]):
            x.add_(1.0)
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    nthreads = len(os.listdir(f"/proc/{os.getpid()}/task"))
    print(f"{tag:28s} {1e6 * host / 4000:6.2f} us per launch (host), affinity {len(os.sched_getaffinity(0))} cpus, {nthreads} threads", flush=True)
probe("before")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
probe("after init (device_id)")
t = torch.ones(8, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
probe("after first collective")
dist.destroy_process_group()
probe("after destroy")
