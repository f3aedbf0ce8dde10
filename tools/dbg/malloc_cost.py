"""What a fresh process pays for device memory: torch.empty (= hipMalloc through the caching allocator) of several sizes, then
the first and the second fill of the block.  Run under gpurun: python tools/dbg/malloc_cost.py"""
import time
import torch

torch.cuda.init()
torch.zeros(1, device="cuda")
torch.cuda.synchronize()
for gib in (0.25, 1, 4, 16, 32, 64):
    n = int(gib * 2**30)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t = torch.empty(n, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    t.zero_()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    t.zero_()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(f"{gib:6.2f} GiB: alloc {1e3 * (t1 - t0):8.1f} ms ({1e3 * (t1 - t0) / gib:6.1f} ms/GiB), first fill {1e3 * (t2 - t1):7.1f} ms, "
          f"second fill {1e3 * (t3 - t2):7.1f} ms")
    del t
    torch.cuda.empty_cache()
# many small blocks
t0 = time.perf_counter()
blocks = [torch.empty(256 << 20, dtype=torch.uint8, device="cuda") for _ in range(64)]
torch.cuda.synchronize()
print(f"64 x 256 MiB: {1e3 * (time.perf_counter() - t0):.1f} ms")
del blocks
torch.cuda.empty_cache()
t0 = time.perf_counter()
pin = torch.empty(64 << 20, dtype=torch.uint8, pin_memory=True)
print(f"pinned 64 MiB: {1e3 * (time.perf_counter() - t0):.1f} ms")
