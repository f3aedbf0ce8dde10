"""Multi-GPU: object sharding and the CCD image reduce.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on ROCm).  Objects are
independent in LSST_Image mode, so the only exchange is ONE sum-reduce of the fp32 CCD image onto
rank 0 (64 MiB over a ring: per-link bound, ~0.4 ms).  Because every photon's random stream is
addressed by (object id, photon index), the reduced image does not depend on the rank count."""
import numpy as np


def shard_objects(objects, rank, world):
    """Sort by photon count (descending) and deal round-robin to the ranks so photon counts
    balance; then order the shard by 256-pixel tiles so neighbouring workgroups touch nearby image
    lines and boundary state (XCD-local L2 reuse)."""
    order = np.argsort(-objects["n_phot"], kind="stable")
    mine = objects[order[rank::world]]
    tile = (mine["y0"] // 256).astype(np.int64) * 4096 + (mine["x0"] // 256).astype(np.int64)
    return mine[np.argsort(tile, kind="stable")]


def reduce_image(image, dst=0):
    """Sum the per-rank CCD images onto `dst` (no-op for a single process)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(image, dst=dst, op=dist.ReduceOp.SUM)
    return image
