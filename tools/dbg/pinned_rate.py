#!/usr/bin/env python
"""Device-to-host rate of a 64-MiB copy into page-locked buffers, buffer by buffer: a focal plane's last images were seen arriving at
1.25 ms OR at 5.3 / 6.7 / 9.5 ms each, the same buffers the same time in every step (round 6) -- which buffers are slow, and does the
CPU the allocating thread runs on decide it?   (under gpurun)  python tools/dbg/pinned_rate.py"""
import glob
import os
import time

import torch


def numa_of_gpu():
    out = []
    for p in glob.glob("/sys/class/drm/card*/device/numa_node"):
        try:
            out.append((p.split("/")[4], int(open(p).read())))
        except Exception:
            pass
    return out


def cpus_of_node(n):
    try:
        txt = open(f"/sys/devices/system/node/node{n}/cpulist").read().strip()
    except Exception:
        return None
    cpus = set()
    for part in txt.split(","):
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def rate(buf, src):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    buf.copy_(src, non_blocking=True)
    b.record()
    b.synchronize()
    return a.elapsed_time(b)


def main():
    print("GPU numa nodes:", numa_of_gpu(), " nodes:", sorted(os.path.basename(p) for p in glob.glob("/sys/devices/system/node/node*")))
    print("allowed CPUs:", len(os.sched_getaffinity(0)), " this thread on CPU", os.sched_getcpu() if hasattr(os, "sched_getcpu") else "?")
    src = torch.rand((4096, 4096), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    for label, prep in (("as the thread happens to run", None), ("thread confined to the GPU's node while it allocates", "gpu"), ("thread on the OTHER node", "other")):
        nodes = [n for _, n in numa_of_gpu() if n >= 0]
        keep = os.sched_getaffinity(0)
        if prep and nodes:
            want = cpus_of_node(nodes[0]) if prep == "gpu" else None
            if prep == "other":
                allnodes = sorted(int(os.path.basename(p)[4:]) for p in glob.glob("/sys/devices/system/node/node*"))
                others = [n for n in allnodes if n != nodes[0]]
                want = cpus_of_node(others[-1]) if others else None
            if want and (want & keep):
                os.sched_setaffinity(0, want & keep)
                time.sleep(0.01)
        bufs = []
        t0 = time.perf_counter()
        for _ in range(48):
            bufs.append(torch.empty((4096, 4096), dtype=torch.float32, pin_memory=True))
        t_alloc = (time.perf_counter() - t0) / 48
        os.sched_setaffinity(0, keep)
        ms = [min(rate(b, src) for _ in range(2)) for b in bufs]
        print(f"{label}: allocation {1e3 * t_alloc:.1f} ms per buffer; copy ms per buffer:", " ".join(f"{m:.1f}" for m in ms))
        del bufs
        torch.cuda.empty_cache()
        try:
            torch._C._host_emptyCache()
        except Exception:
            pass


if __name__ == "__main__":
    main()
