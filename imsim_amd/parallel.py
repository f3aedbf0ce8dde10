"""Multi-GPU: object sharding and the CCD image reduce.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on ROCm).  Objects are
independent in LSST_Image mode, so the only exchange is ONE sum-reduce of the f64 CCD accumulation
image onto rank 0 (128 MiB over a ring: per-link bound, ~1 ms).  Photon-pooling mode shares the
sensor state between all objects: there the delta-charge image is all-reduced before every
pixel-boundary recalculation (`allreduce_delta`, once per photon batch).  Because every photon's
random stream is addressed by (object id, photon index) and unit fluxes make the sums exact, the
result does not depend on the rank count."""
import numpy as np

from . import tuning


def assign_ranks(n_phot, world, nrecalc=10000, round_photons=3.0e5):
    """Which rank renders which object.  Objects stay whole (in LSST_Image mode an object's brighter-fatter state is
    sequential in its own photons), so the balance is decided by a cost model measured on one MI355X:

        time of a rank ~ (longest brighter-fatter chain among its objects) + (its photons) / (photon rate)

    -- a bright object is a chain of ceil(n_phot / nrecalc) dependent rounds of ~70 us, worth `round_photons` photons of
    shooting each, and chains of different objects overlap while the photon work adds up.  Greedy, brightest first: every
    object goes to the rank whose modelled time grows least; the light tail is dealt in blocks to the least-loaded rank.
    The rank that owns the brightest star therefore gets little else: its chain is the floor of the strong scaling."""
    n_phot = np.asarray(n_phot, dtype=np.int64)
    owner = np.zeros(len(n_phot), dtype=np.int32)
    if world == 1:
        return owner
    order = np.argsort(-n_phot, kind="stable")
    rounds = np.where(n_phot > nrecalc, (n_phot + nrecalc - 1) // nrecalc, 0)
    chain = np.zeros(world)                       # longest chain of the rank, in photon equivalents
    bulk = np.zeros(world)                        # photons of the rank
    heavy = order[: min(len(order), 64 * world)]
    for i in heavy:
        c = float(rounds[i]) * round_photons
        t_new = np.maximum(chain, c) + bulk + float(n_phot[i])
        r = int(np.lexsort((bulk, t_new))[0])     # least resulting time, then least photons
        owner[i] = r
        chain[r] = max(chain[r], c)
        bulk[r] += float(n_phot[i])
    tail = order[len(heavy):]
    block = 64
    for a in range(0, len(tail), block):
        ids = tail[a:a + block]
        r = int(np.argmin(chain + bulk))
        owner[ids] = r
        bulk[r] += float(n_phot[ids].sum())
    return owner


def shard_objects(objects, rank, world):
    """This rank's share of the object table (balanced by photon count), ordered by 256-pixel
    tiles so neighbouring workgroups touch nearby image lines and boundary state (XCD-local L2
    reuse)."""
    mine = objects[assign_ranks(objects["n_phot"], world) == rank]
    tile = (mine["y0"] // 256).astype(np.int64) * 4096 + (mine["x0"] // 256).astype(np.int64)
    return mine[np.argsort(tile, kind="stable")]


def shard_ccds(dets, rank, world):
    """The CCDs this rank renders when a whole focal plane is spread over the GPUs of a node (C5: CCD i -> GPU
    i mod world; CCDs are independent, imsim/ccd.py:72-89)."""
    return [d for k, d in enumerate(dets) if k % world == rank]


def unit_flux_path(scene, objects=None):
    """True when every photon of a render carries exactly one electron: no flux-scaling operator in the chain
    (BandpassRatio multiplies the flux by a table value) and flux_per_photon == 1 for every object.  Only then are the CCD
    image and the delta-charge image integer counts, and only then may an exchange run on int32 copies."""
    from . import _abi
    for op in getattr(scene, "ops", None) or []:
        if int(op[0]) == _abi.IMS_OP_BANDPASS_RATIO:
            return False
    if objects is not None and len(objects) and not np.all(objects["flux_per_photon"] == 1.0):
        return False
    # FITS-stamp objects shot through an interpolant carry +- (integral |K|)^2 per photon (ims_image_tables_t)
    if getattr(scene, "image_profiles", None) and getattr(scene, "image_interpolant", "nearest") != "nearest":
        if objects is None or (len(objects) and np.any(objects["prof_table"] == _abi.IMS_PROF_IMAGE)):
            return False
    return True


def integer_counts_ok(image, world):
    """The check reduce_image(integer_counts=True) relies on, as ONE host synchronisation the caller places outside its timed
    region: every pixel a non-negative integer, and world x the largest pixel below 2^31 (so the SUM over the ranks cannot
    wrap either)."""
    import torch
    top = float(image.max())
    low = float(image.min())
    whole = bool(torch.equal(image, torch.floor(image)))
    return whole and low >= 0.0 and top * max(int(world), 1) < 2147483648.0


class LibraryComm:
    """The library's own RCCL communicator (ims_comm_init, include/imsim_hip.h): the exchanges of the path run inside
    libimsim_hip.so, torch.distributed (any backend, gloo included) only carries the 128-byte id from rank 0 to the others --
    so a host that is not Python runs the same exchanges.  One per process; `install` makes reduce_image / allreduce_delta use it."""

    def __init__(self, rank, world, device, broadcast=None):
        """broadcast(bytes or None) -> bytes: hands rank 0's id to every rank; default: torch.distributed.broadcast_object_list
        over the existing process group (world 1 needs none)"""
        import ctypes as C
        import torch
        from . import _abi
        self.lib = _abi.load()
        self.rank, self.world, self.device = int(rank), int(world), torch.device(device)
        ident = (C.c_char * 128)()
        if self.rank == 0:
            _abi.check(self.lib.ims_comm_unique_id(C.byref(ident)), "ims_comm_unique_id")
        raw = bytes(ident)
        if self.world > 1:
            if broadcast is None:
                import torch.distributed as dist

                def broadcast(b):
                    box = [b]
                    dist.broadcast_object_list(box, src=0)
                    return box[0]
            raw = broadcast(raw if self.rank == 0 else None)
        ident = (C.c_char * 128).from_buffer_copy(raw)
        handle = C.c_void_p()
        torch.cuda.set_device(self.device)
        _abi.check(self.lib.ims_comm_init(C.byref(ident), self.rank, self.world, C.byref(handle)), "ims_comm_init")
        self.handle = handle
        self._scratch = None

    def scratch(self, n):
        import torch
        if self._scratch is None or self._scratch.numel() < n:
            self._scratch = torch.empty(int(n), dtype=torch.int32, device=self.device)
        return self._scratch

    def _stream(self):
        import ctypes as C
        import torch
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def reduce_image(self, image, dst=0, integer_counts=False):
        from . import _abi
        n = image.numel()
        sc = self.scratch(n).data_ptr() if integer_counts else None
        _abi.check(self.lib.ims_reduce_image(self.handle, image.data_ptr(), sc, n, int(dst), 1 if integer_counts else 0, self._stream()),
                   "ims_reduce_image")
        return image

    def allreduce_delta(self, delta, integer_counts=False):
        from . import _abi
        n = delta.numel()
        sc = self.scratch(n).data_ptr() if integer_counts else None
        _abi.check(self.lib.ims_allreduce_delta(self.handle, delta.data_ptr(), sc, n, 1 if integer_counts else 0, self._stream()),
                   "ims_allreduce_delta")
        return delta

    def count_inexact(self, image):
        """number of values of this rank's image the int32 exchange would not carry exactly (0 = fine); one host sync"""
        import torch
        from . import _abi
        bad = torch.zeros(1, dtype=torch.int64, device=self.device)
        _abi.check(self.lib.ims_count_inexact(image.data_ptr(), image.numel(), self.world, bad.data_ptr(), self._stream()), "ims_count_inexact")
        return int(bad.item())

    def destroy(self):
        if getattr(self, "handle", None):
            self.lib.ims_comm_destroy(self.handle)
            self.handle = None


_LIBRARY_COMM = [None]


def install(comm):
    """make reduce_image / allreduce_delta run on a LibraryComm (None: back to torch.distributed)"""
    _LIBRARY_COMM[0] = comm


def _exchange_on(dist):
    """An exchange runs when a process group of more than one rank exists -- or of exactly one rank when
    IMS_EXCHANGE_SINGLE_RANK=1: the RCCL calls then execute as self-exchanges, which is how the one-GPU test box can run the
    nccl code path at all (tests/test_multi_gpu_hip.py, `bench.py` under a one-rank launcher)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or tuning.env("IMS_EXCHANGE_SINGLE_RANK", "0") == "1"


def exchanging(world):
    """Whether the delta-charge exchange of photon pooling runs for a job of `world` ranks: always for world > 1, and with one
    rank when IMS_EXCHANGE_SINGLE_RANK=1 asks for the collectives as self-exchanges (and a process group exists)."""
    if world > 1 or _LIBRARY_COMM[0] is not None:
        return True
    import torch.distributed as dist
    return tuning.env("IMS_EXCHANGE_SINGLE_RANK", "0") == "1" and dist.is_available() and dist.is_initialized()


def reduce_image(image, dst=0, integer_counts=False):
    """Sum the per-rank CCD images onto `dst` (no-op for a single process).

    integer_counts: the caller guarantees that every pixel holds a non-negative integer electron count and that the sum
    over the ranks stays below 2^31 (unit photon fluxes, the standard path: stamp.py:562-572 `poisson_flux=False`,
    n_photons=phot_flux; `unit_flux_path` says whether a scene qualifies, `integer_counts_ok` checks an image -- once,
    outside the timed region: nothing here synchronises with the host).  The exchange then runs on an int32 copy -- half the
    bytes of the f64 accumulation image on the per-link-bound xGMI ring -- and is still exact."""
    import torch
    import torch.distributed as dist
    if _LIBRARY_COMM[0] is not None:
        return _LIBRARY_COMM[0].reduce_image(image, dst, integer_counts)
    if _exchange_on(dist):
        buf = image.to(torch.int32) if integer_counts else image
        if dist.get_backend() == "gloo" and buf.is_cuda:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)       # gloo has no device reduce (dry runs on one GPU only)
        else:
            dist.reduce(buf, dst=dst, op=dist.ReduceOp.SUM)
        if integer_counts and dist.get_rank() == dst:
            image.copy_(buf)
    return image


def allreduce_delta(delta, integer_counts=False):
    """Sum the delta-charge image (charge accumulated since the last recalculation) over all ranks, in
    place: afterwards every rank holds the charge of ALL objects and runs the same
    updatePixelDistortions (imsim/photon_pooling.py:159 `recalc=(subbatch_num == 0)` semantics).
    integer_counts: as in reduce_image -- unit photon fluxes, the exchange runs on an int32 copy (half the bytes)."""
    import torch
    import torch.distributed as dist
    if _LIBRARY_COMM[0] is not None:
        return _LIBRARY_COMM[0].allreduce_delta(delta, integer_counts)
    if _exchange_on(dist):
        if integer_counts:
            buf = delta.to(torch.int32)
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
            delta.copy_(buf)
        else:
            dist.all_reduce(delta, op=dist.ReduceOp.SUM)
    return delta
