#!/usr/bin/env python
"""Host hand-over of one C3 step over PCIe: object table up (100 k rows x 256 B), float32 CCD image down (64 MiB),
pinned host memory.  bench.py's `value` excludes it (inputs resident in HBM); DESIGN.md quotes the inclusive rate."""
import time

import torch

up = torch.empty(100000 * 256, dtype=torch.uint8).pin_memory()
down = torch.empty(4096 * 4096, dtype=torch.float32).pin_memory()
d_up = torch.empty_like(up, device="cuda")
d_down = torch.zeros(4096 * 4096, dtype=torch.float32, device="cuda")
for _ in range(3):
    d_up.copy_(up, non_blocking=True)
    down.copy_(d_down, non_blocking=True)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    d_up.copy_(up, non_blocking=True)
torch.cuda.synchronize()
t_up = (time.perf_counter() - t0) / n
t0 = time.perf_counter()
for _ in range(n):
    down.copy_(d_down, non_blocking=True)
torch.cuda.synchronize()
t_down = (time.perf_counter() - t0) / n
print(f"object table up: {t_up * 1e3:.3f} ms ({up.numel() / t_up / 1e9:.1f} GB/s); image down: {t_down * 1e3:.3f} ms "
      f"({down.numel() * 4 / t_down / 1e9:.1f} GB/s); total {1e3 * (t_up + t_down):.3f} ms per step")
