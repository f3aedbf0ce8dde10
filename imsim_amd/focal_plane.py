"""A visit's CCDs on the GPUs of one node (BASELINE config C5: 189 CCDs x 10 k sources, one CCD per stream).

The reference fans the CCDs of a visit out to worker processes (`output.nproc`, imsim/ccd.py:72-89); every
CCD is an independent `LSST_Image` build.  Here the CCDs are dealt round-robin to the ranks
(`parallel.shard_ccds`, one process per GPU, no exchange) and inside a rank up to `concurrent` CCDs (default 2: measured
10.5 ms per 10 k-source CCD against 11.4 with three in flight) are in
flight at once, each anchored to its own HIP stream: while the GPU works through the launch plan of one CCD
(which `Renderer.execute_plan` only enqueues), the host builds the object table and plan of the next.
Results do not depend on `concurrent`, on the rank count or on the order of the CCDs: every photon's random
stream is addressed by (CCD seed, object id, photon index).
"""
from . import parallel
from .engine import Renderer


_ANCHOR_STREAMS = {}


def render_focal_plane(ccds, build, device="cuda:0", rank=0, world=1, concurrent=2, nrecalc=None, sink=None):
    """ccds: sequence of CCD keys (detector numbers / names); build(key) -> (scene, objects) prepares one CCD
    on the host.  Returns {key: float32 image as a host array} for the CCDs this rank owns.

    The finished image of a CCD is copied into a page-locked host buffer on the CCD's own stream (one buffer per
    stream in flight, reused).  sink(key, image): optional consumer (the FITS writer in production) that is handed
    a VIEW of that buffer, valid until it returns; with a sink nothing is accumulated and {} is returned."""
    import torch
    dev = torch.device(device)
    mine = parallel.shard_ccds(list(ccds), rank, world)
    if concurrent < 1:
        raise ValueError("concurrent must be >= 1")
    # the anchor streams are kept per device for the life of the process: PyTorch's caching allocator files a freed block
    # under the stream it was allocated on, so fresh streams per call would miss the cache and hipMalloc every CCD's
    # gigabytes of sensor state again (measured: 13 -> 27 .. 34 ms per CCD for the calls that do)
    pool = _ANCHOR_STREAMS.setdefault(str(dev), [])
    while len(pool) < min(concurrent, max(len(mine), 1)):
        pool.append(torch.cuda.Stream(dev))
    streams = pool[:min(concurrent, max(len(mine), 1))]
    in_flight = [None] * len(streams)          # (key, renderer, host image, done event) per stream
    pinned = [None] * len(streams)
    out = {}

    def collect(slot):
        if in_flight[slot] is None:
            return
        key, renderer, host, done = in_flight[slot]
        done.synchronize()
        if sink is not None:
            sink(key, host.numpy())
        else:
            out[key] = host.numpy().copy()
        in_flight[slot] = None                 # drops the renderer: its HBM goes back to the caching allocator

    for k, key in enumerate(mine):
        slot = k % len(streams)
        collect(slot)                          # the CCD that used this stream before
        scene, objects = build(key)            # host work, overlaps with the CCDs still running on the GPU
        with torch.cuda.stream(streams[slot]):
            renderer = Renderer(scene, dev)
            renderer.render_lsst_image(objects, nrecalc=nrecalc)
            img = renderer.image_float()
            if pinned[slot] is None or pinned[slot].shape != img.shape:
                pinned[slot] = torch.empty(img.shape, dtype=img.dtype, pin_memory=True)
            pinned[slot].copy_(img, non_blocking=True)
            done = torch.cuda.Event()
            done.record(streams[slot])
        in_flight[slot] = (key, renderer, pinned[slot], done)
    for slot in range(len(streams)):
        collect(slot)
    return out
