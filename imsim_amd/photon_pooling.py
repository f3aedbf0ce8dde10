"""Photon pooling: host logic of LSST_PhotonPoolingImageBuilder (imsim/photon_pooling.py) over the
GPU engine.

The batching helpers keep the reference's names and semantics (they are pure bookkeeping and are
tested against the expectations of the reference's tests/test_photon_pooling.py); `build_image`
replaces the per-object stamp loop + merge + per-op passes by three launches per sub-batch
(shoot into the pool, one pass applying every op, accumulate) with the whole CCD as ONE
brighter-fatter region recalculated at the first sub-batch of each batch.
"""
import dataclasses
import itertools
import os

import numpy as np

from . import parallel, tuning
from ._abi import OBJECT_DTYPE, IMS_OBJ_FAINT
from .stamp import ProcessingMode, ObjectInfo


def make_batches(objects, nbatch):
    """Split `objects` into nbatch consecutive batches, earlier batches taking the remainder
    (imsim/photon_pooling.py:228-247)."""
    base, extra = divmod(len(objects), nbatch)
    start = 0
    for i in range(nbatch):
        n = base + 1 if i < extra else base
        yield list(objects[start:start + n])
        start += n


def make_photon_batches(phot_objects, faint_objects, nbatch, rng=None):
    """Every bright object appears in all nbatch batches with the integer flux share
    (F(i+1))//nbatch - (F i)//nbatch; every faint object lands whole in one random batch
    (imsim/photon_pooling.py:279-313).  `rng`: callable returning a uniform deviate in [0,1)."""
    if not phot_objects and not faint_objects:
        return []
    batches = [[dataclasses.replace(obj, phot_flux=(obj.phot_flux * (i + 1)) // nbatch - (obj.phot_flux * i) // nbatch)
                for obj in phot_objects] for i in range(nbatch)]
    if rng is None:
        gen = np.random.default_rng(0)
        rng = gen.random
    for obj in faint_objects:
        batches[int(rng() * nbatch)].append(obj)
    return batches


def make_photon_subbatches(batch, nsubbatch):
    """Split a batch into nsubbatch nearly equal consecutive sub-batches (imsim/photon_pooling.py:316-331)."""
    per, extra = divmod(len(batch), nsubbatch)
    sizes = extra * [per + 1] + (nsubbatch - extra) * [per]
    edges = [0] + list(itertools.accumulate(sizes))
    return [batch[edges[i]:edges[i + 1]] for i in range(nsubbatch)]


def partition_objects(objects, nbatch):
    """(fft, phot, faint) lists; PHOT objects with fewer photons than batches are demoted to
    FAINT handling (imsim/photon_pooling.py:356-386)."""
    by_mode = {ProcessingMode.FFT: [], ProcessingMode.PHOT: [], ProcessingMode.FAINT: []}
    for obj in objects:
        mode = ProcessingMode.FAINT if (obj.phot_flux < nbatch and obj.mode == ProcessingMode.PHOT) else obj.mode
        by_mode[mode].append(obj)
    return by_mode[ProcessingMode.FFT], by_mode[ProcessingMode.PHOT], by_mode[ProcessingMode.FAINT]


class GalSimConfigValueError(ValueError):
    pass


def check_stamp_type(stamp_type):
    """LSST_PhotonPoolingImage requires stamp.type == LSST_Photons (imsim/photon_pooling.py:25-26)."""
    if stamp_type != "LSST_Photons":
        raise GalSimConfigValueError(f"Must use stamp.type = LSST_Photons with LSST_PhotonPoolingImage. ({stamp_type})")


def batch_plan(n_phot, modes, nbatch, seed):
    """The structure behind batch_shares: (phot_idx, F, nb, faint_idx, faint_where) -- the PHOT objects (in every batch with
    the share (F (i + 1)) // nb - (F i) // nb of their F photons), the number of batches, the FAINT objects and the batch
    each of them lands in."""
    n_phot = np.asarray(n_phot, dtype=np.int64)
    modes = np.asarray(modes)
    if modes.dtype == object:
        is_phot = np.fromiter((m == ProcessingMode.PHOT for m in modes), dtype=bool, count=len(modes))
        is_faint = np.fromiter((m == ProcessingMode.FAINT for m in modes), dtype=bool, count=len(modes))
    else:                                                         # integer codes: ProcessingMode values
        is_phot, is_faint = modes == int(ProcessingMode.PHOT.value), modes == int(ProcessingMode.FAINT.value)
    demoted = is_phot & (n_phot < nbatch)
    phot_idx = np.flatnonzero(is_phot & ~demoted)
    faint_idx = np.flatnonzero(is_faint | demoted)
    nb = max(min(nbatch, len(phot_idx)), 1)
    gen = np.random.default_rng([int(seed), 0xFA17])
    faint_where = (gen.random(len(faint_idx)) * nb).astype(np.int64)      # one uniform per faint object, in catalog order
    return phot_idx, n_phot[phot_idx], nb, faint_idx, faint_where


def batch_shares(n_phot, modes, nbatch, seed):
    """The photon batches of LSST_PhotonPoolingImageBuilder.buildImage (imsim/photon_pooling.py:116-140, :279-313, :356-386) as
    index arithmetic over the whole catalog: every PHOT object (at least nbatch photons) appears in every batch with the
    integer share (F (i + 1)) // nb - (F i) // nb of its photons, every FAINT object (and PHOT objects with fewer photons
    than batches) whole in one random batch.  Returns [(index into the catalog, first photon of the share within the
    object, photons of the share)] per batch, and the size of the smallest batch."""
    n_phot = np.asarray(n_phot, dtype=np.int64)
    phot_idx, F, nb, faint_idx, faint_where = batch_plan(n_phot, modes, nbatch, seed)
    out, smallest = [], None
    order = np.argsort(faint_where, kind="stable")
    bounds = np.searchsorted(faint_where[order], np.arange(nb + 1))
    for i in range(nb):
        lo, hi = (F * i) // nb, (F * (i + 1)) // nb
        fsel = faint_idx[order[bounds[i]:bounds[i + 1]]]
        index = np.concatenate([phot_idx, fsel])
        first = np.concatenate([lo, np.zeros(len(fsel), dtype=np.int64)])
        count = np.concatenate([hi - lo, n_phot[fsel]])
        out.append((index, first, count))
        smallest = len(index) if smallest is None else min(smallest, len(index))
    return out, smallest


def make_batch_tables(objects, modes, nbatch, seed):
    """batch_shares as object tables: ([(object table of the batch, row index into `objects`)], size of the smallest batch)."""
    shares, len_smallest = batch_shares(objects["n_phot"], modes, nbatch, seed)
    batch_tables = []
    for index, first, count in shares:
        table = objects[index].copy()
        table["phot_first"] = objects["phot_first"][index] + first
        table["n_phot"] = count
        batch_tables.append((table, index))
    return batch_tables, len_smallest


def _pool_fits(renderer, n_photons):
    """the HBM-resident pool of prepared_image / build_image: 32 B per photon, at most 80 % of the free memory"""
    if tuning.env("IMS_POOL_RESIDENT", "1") == "0" or not hasattr(renderer, "prepared_pooled_batches"):
        return False
    free = renderer.torch.cuda.mem_get_info(renderer.device)[0]
    fits = 32 * int(n_photons) < 0.8 * free
    # the resident form and the sub-batch loop issue different collectives: the ranks must take the same one
    import torch.distributed as dist
    if parallel._exchange_on(dist):
        flag = renderer.torch.tensor([1 if fits else 0], dtype=renderer.torch.int32,
                                     device=renderer.device if dist.get_backend() != "gloo" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        fits = bool(int(flag.item()))
    return fits


def prepared_image(renderer, objects, modes, nbatch=10, seed=0, rank=0, world=1, resident=None, realized=None,
                   after_batch=None, first_batch=0):
    """The same image as build_image, prepared for replay (bench.py).  The photons of a batch see frozen pixel boundaries,
    and a photon does not depend on the batch it lands in (its random stream is addressed by object and photon index), so
    neither the sub-batching (a memory bound of the reference) nor the order of shooting changes the result.

    resident (default: on unless IMS_POOL_RESIDENT=0 or the pool would not fit): ONE launch shoots the photons of ALL
    batches into an HBM-resident converted pool (32 B per photon: 49 GB for the 1.5e9 photons of C4), then every batch only
    runs the pixel search of its share (`Renderer.prepared_pooled_batches`) with the pixel-boundary recalculation between
    batches.  Otherwise every batch is one fused launch (shoot -> PSF -> ops -> sensor, ims_shoot_accumulate) over its own
    object table, in which an object of 190 photons is shot ten times in wavefronts a third full.

    world > 1: the batch tables are built from the FULL object table on every rank, a rank shoots the rows it owns,
    and before every recalculation the delta-charge image is all-reduced so that every rank applies the charge of ALL
    objects (build_image's multi-rank semantics; the tile marks are rank-local, so the update visits every tile).
    realized (resident form only): f64 device tensor over the rows of `objects` receiving base['realized_flux'];
    after_batch(i): called on the host after batch i has been enqueued (checkpoints); first_batch: batches before it are
    skipped (a resumed CCD; as in the reference it continues from fresh pixel boundaries).
    Returns a zero-argument callable."""
    if not isinstance(objects, np.ndarray):
        return _prepared_image_device(renderer, objects, modes, nbatch, seed, rank, world, realized, after_batch, first_batch)
    objects = np.ascontiguousarray(objects, dtype=OBJECT_DTYPE)
    tables, _ = make_batch_tables(objects, modes, nbatch, seed)
    sensor_on = renderer.scene.sensor is not None
    owner = parallel.assign_ranks(objects["n_phot"], world)
    tagged = sensor_on and world == 1
    mine = owner == rank if world > 1 else np.ones(len(objects), dtype=bool)
    in_batches = np.zeros(len(objects), dtype=bool)
    for table, index in tables:
        in_batches[index[table["n_phot"] > 0]] = True
    shot = np.flatnonzero(in_batches & mine)
    if resident is None:
        resident = _pool_fits(renderer, objects["n_phot"][shot].sum())
    if realized is not None and not resident:
        raise ValueError("prepared_image: realized fluxes need the resident form")

    def tidy(table):
        table["bf_state"] = 0
        table["flags"] &= ~IMS_OBJ_FAINT     # in pooling mode the ops and the sensor see every photon (photon_pooling.py:154-159)
        return table

    def spatial(table):
        # neighbours in the table share image lines and boundary state
        tile = (table["y0"] // 256).astype(np.int64) * 4096 + (table["x0"] // 256).astype(np.int64)
        return np.argsort(tile, kind="stable")

    shoot, launches = None, []
    r_rows = None
    delta_only = False
    if resident:
        shoot_table = tidy(objects[shot].copy())
        if world > 1 or tuning.env("IMS_POOL_SPATIAL", "1") != "0":
            order = spatial(shoot_table)
            shoot_table, shot = shoot_table[order], shot[order]
        row_of = np.full(len(objects), -1, dtype=np.int64)
        row_of[shot] = np.arange(len(shot))
        batches = []
        for i, (table, index) in enumerate(tables):
            keep = (table["n_phot"] > 0) & mine[index]
            rows = row_of[index[keep]]
            order = np.argsort(rows, kind="stable")            # the order of the shoot table (spatial with several ranks)
            rows = rows[order]
            first = (table["phot_first"][keep] - objects["phot_first"][index[keep]])[order]
            batches.append((rows, first.astype(np.int64), table["n_phot"][keep][order], (i % 255 + 1) if tagged else 0))
        r_rows = None
        if realized is not None:
            r_rows = renderer.torch.zeros(len(shoot_table), dtype=renderer.torch.float64, device=renderer.device)
        delta_only = _delta_only(renderer, sensor_on, world, parallel.unit_flux_path(renderer.scene, objects), after_batch)
        shoot, launches = renderer.prepared_pooled_batches(shoot_table, batches, realized=r_rows, delta_only=delta_only)
    else:
        for i, (table, index) in enumerate(tables):
            keep = (table["n_phot"] > 0) & mine[index]
            table = tidy(table[keep].copy())
            if world > 1:
                table = table[spatial(table)]
            launches.append(renderer.prepared(table, bf_tag=(i % 255 + 1) if tagged else 0))

    # int32 exchange of the delta-charge image only when every photon is exactly one electron (BandpassRatio and
    # flux_per_photon != 1 make fractional charge, which an integer copy would truncate)
    unit = parallel.unit_flux_path(renderer.scene, objects)

    def run():
        if sensor_on:
            renderer.init_boundaries(0, 1)                 # a new CCD starts from undistorted (+ tree ring) boundaries
        if shoot is not None:
            shoot()
        for i, launch in enumerate(launches):
            if i < first_batch:
                continue
            if sensor_on and i > first_batch:
                if parallel.exchanging(world):
                    parallel.allreduce_delta(renderer.delta_tensor(0), integer_counts=unit)
                renderer.update_distortions(0, 1, bf_tag=((i - 1) % 255 + 1) if tagged else 0, fold=delta_only)
            launch()
            if realized is not None and r_rows is not None:
                # batch by batch, so that a checkpoint taken after batch i carries the fluxes up to batch i
                realized.index_add_(0, shot_t, r_rows)
                r_rows.zero_()
            if after_batch is not None:
                after_batch(i)
        if delta_only:
            renderer.fold_delta()                          # the last batch's charge: no recalculation follows it
    shot_t = renderer.torch.from_numpy(shot).to(renderer.device) if realized is not None and shoot is not None else None
    if shoot is not None:
        run.photons, run.object_rows = shoot.photons, shoot.object_rows
        # the pool shoot is the dominant launch: 32 B per converted photon written + one 256-B row per object
        run.timed = {2: (getattr(shoot, "launches", 1), shoot.photons * 32 + shoot.object_rows * 256), 1: (0, 0)}
        run.timed_waves = {2: shoot.waves, 1: 0}
    else:
        run.photons = sum(l.photons for l in launches)
        run.object_rows = sum(l.object_rows for l in launches)
        run.timed = {1: (len(launches), sum(l.timed[1][1] for l in launches)), 2: (0, 0)}
        run.timed_waves = {1: sum(l.timed_waves[1] for l in launches), 2: 0}
    run.resident = bool(resident)
    run.keep = (shoot, launches)
    return run


def _delta_only(renderer, sensor_on, world, unit, after_batch):
    """Photon pooling with ONE atomic add per photon: a batch's launches deposit into the delta-charge image of slot 0 only
    (ims_render_params_t.track_static_delta 2) and the image takes the charge when the next recalculation consumes it -- Silicon's
    `target += delta` (Renderer.update_distortions(fold=True)), the last batch's by Renderer.fold_delta().  Exact for integer charge,
    so only where every photon is one electron; one rank (with several, the delta image every rank consumes is the all-reduced
    one); no checkpoint hook (it reads the image between the batches).  IMS_POOL_DELTA_ONLY=0: two atomic adds as before."""
    if not (sensor_on and world == 1 and unit and after_batch is None and tuning.flag("IMS_POOL_DELTA_ONLY")):
        return False
    sc = renderer.scene
    sl = renderer.bound._slots_host[0]
    return (int(sl["xmin"]), int(sl["ymin"]), int(sl["nx"]), int(sl["ny"])) == (sc.xmin, sc.ymin, sc.nx, sc.ny) and int(sl["offset"]) == 0


def _prepared_image_device(renderer, table, modes, nbatch, seed, rank, world, realized, after_batch, first_batch):
    """prepared_image for a device-resident object table (device_table.DeviceTable): the HBM-resident form only -- the batch
    shares are index arithmetic on the 16 bytes per object the table builder sent back, the shoot table and the per-batch
    tables are gathered on the device (ims_gather_rows with IMS_OBJ_FAINT cleared: in pooling mode the operators and the
    sensor see every photon).  Same image as the host-table form, bit for bit (tests/test_device_table.py)."""
    n_phot = table.n_phot
    phot_idx, F, nb, faint_idx, faint_where = batch_plan(n_phot, modes, nbatch, seed)
    sensor_on = renderer.scene.sensor is not None
    tagged = sensor_on and world == 1
    if world > 1:
        mine = parallel.assign_ranks(n_phot, world) == rank
        keep_p, keep_f = mine[phot_idx], mine[faint_idx]
        phot_idx, F, faint_idx, faint_where = phot_idx[keep_p], F[keep_p], faint_idx[keep_f], faint_where[keep_f]
    has = n_phot[faint_idx] > 0
    faint_idx, faint_where = faint_idx[has], faint_where[has]
    # per catalog object: 0 = not shot, 1 = PHOT, 2 + b = FAINT of batch b (one byte: the orderings below are radix sorts)
    role = np.zeros(table.n, dtype=np.uint8 if nb < 250 else np.int32)
    role[phot_idx] = 1
    role[faint_idx] = (2 + faint_where).astype(role.dtype)
    if world > 1 or tuning.env("IMS_POOL_SPATIAL", "1") != "0":
        # the shoot table by 256 x 256-pixel tiles of the CCD, catalog order inside a tile: a stable sort on a 16-bit key
        tile = ((table.y.astype(np.int64) >> 8).clip(0, 255) << 8 | (table.x.astype(np.int64) >> 8).clip(0, 255)).astype(np.uint16)
        order = np.argsort(tile, kind="stable")
        shot = order[role[order] != 0]
    else:
        shot = np.flatnonzero(role)
    shot = np.ascontiguousarray(shot, dtype=np.int64)
    n_shot = n_phot[shot]
    if not _pool_fits(renderer, int(n_shot.sum())):
        raise ValueError("photon pooling from a device table needs the HBM-resident pool (it does not fit)")
    role_s = role[shot]
    # The PHOT objects appear in every batch, in the order of the shoot table (neighbours share image lines and boundary
    # state): rows and photon counts once, the batch only changes the share.  The batch's few FAINT objects follow them.
    # which of the two pixel-search launches a share goes to (ims_accumulate_segments above IMS_POOL_SMALL_MAX photons, one
    # wavefront per object below) is decided ONCE from the object's mean share: the two give the same result, and a fixed
    # split makes every batch two contiguous runs of index arithmetic instead of masks over a million objects
    small_max = int(tuning.env("IMS_POOL_SMALL_MAX", "64"))
    big = (role_s == 1) & ((n_shot // nb) > small_max)
    rows_big = np.flatnonzero(big)
    rows_small = np.flatnonzero((role_s == 1) & ~big)
    F_big, F_small = n_shot[rows_big], n_shot[rows_small]
    rows_faint = np.flatnonzero(role_s >= 2)                      # ascending rows ...
    f_batch = role_s[rows_faint] - 2
    f_sorted = rows_faint[np.argsort(f_batch.astype(np.uint8 if nb < 250 else np.int64), kind="stable")]   # ... by batch, then by row
    f_bounds = np.searchsorted(np.sort(f_batch), np.arange(nb + 1))
    batches = []
    for i in range(nb):
        # PHOT objects: the share [F i // nb, F (i + 1) // nb) as a descriptor -- the renderer forms it on the device from F
        # (one upload per index array, not three arrays over a million objects per batch); the batch's FAINT objects, a few
        # thousand rows that differ from batch to batch, come as host arrays
        fr = f_sorted[f_bounds[i]:f_bounds[i + 1]]
        parts = [(rows_big, ("share", F_big, i, nb), None, False),
                 (rows_small, ("share", F_small, i, nb), None, True),
                 (fr, np.zeros(len(fr), dtype=np.int64), n_shot[fr], True)]
        batches.append(("parts", parts, (i % 255 + 1) if tagged else 0))
    r_rows = None
    if realized is not None:
        r_rows = renderer.torch.zeros(len(shot), dtype=renderer.torch.float64, device=renderer.device)
    unit = parallel.unit_flux_path(renderer.scene, None)
    delta_only = _delta_only(renderer, sensor_on, world, unit, after_batch)
    shoot, launches = renderer.prepared_pooled_batches((table, shot), batches, realized=r_rows, delta_only=delta_only)

    def run():
        if sensor_on:
            renderer.init_boundaries(0, 1)
        shoot()
        for i, launch in enumerate(launches):
            if i < first_batch:
                continue
            if sensor_on and i > first_batch:
                if parallel.exchanging(world):
                    parallel.allreduce_delta(renderer.delta_tensor(0), integer_counts=unit)
                renderer.update_distortions(0, 1, bf_tag=((i - 1) % 255 + 1) if tagged else 0, fold=delta_only)
            launch()
            if after_batch is not None:
                after_batch(i)
        if delta_only:
            renderer.fold_delta()
        if realized is not None:
            realized.index_add_(0, renderer.torch.from_numpy(shot).to(renderer.device), r_rows)
    run.photons, run.object_rows = shoot.photons, shoot.object_rows
    run.timed = {2: (getattr(shoot, "launches", 1), shoot.photons * 32 + shoot.object_rows * 256), 1: (0, 0)}
    run.timed_waves = {2: shoot.waves, 1: 0}
    run.resident = True
    run.keep = (shoot, launches)
    return run


def build_image(renderer, objects, modes, nbatch=10, nsubbatch=50, seed=0, realized=None, rank=0, world=1,
                checkpoint=None, chk_name="buildImage_photonpooling", nbatch_per_checkpoint=1):
    """LSST_PhotonPoolingImageBuilder.buildImage for the photon-shooting objects
    (imsim/photon_pooling.py:116-168).

    objects: OBJECT_DTYPE table (n_phot = phot_flux of the whole object); modes: ProcessingMode per row.
    Photon index ranges follow the integer flux split, so the union over batches is exactly the
    object's photon stream.  Returns the number of photons accumulated.

    world > 1 (one process per GPU): every rank builds the same batch tables from the FULL
    object table and shoots only the rows it owns (parallel.assign_ranks).  In pooling mode the sensor
    state is shared by all objects, so before every recalculation the delta-charge image accumulated
    since the last one is all-reduced (parallel.allreduce_delta) and every rank runs the identical
    updatePixelDistortions; the per-rank images are summed by the caller (parallel.reduce_image).  Unit
    fluxes make both sums exact, so the result equals the single-process one bit for bit.

    checkpoint (a checkpoint.Checkpointer): after every nbatch_per_checkpoint-th photon batch (and the last one) the image,
    the number of finished batches and the realized fluxes so far are saved under chk_name (with several ranks: chk_name +
    "_rank<r>", every rank its own partial image); a later call finds them, restores image and fluxes and skips those
    batches (imsim/photon_pooling.py:57-62, :129-136, :166-167, lsst_image.py:376-389 for the gate).  As in the reference
    the sensor state is not part of the record: a resumed CCD continues from fresh (tree-ring only) pixel boundaries."""
    objects = np.ascontiguousarray(objects, dtype=OBJECT_DTYPE)
    batch_tables, len_smallest = make_batch_tables(objects, modes, nbatch, seed)
    total = 0
    first_batch = 0
    if world > 1:
        chk_name = f"{chk_name}_rank{rank}"
    every = max(int(nbatch_per_checkpoint), 1)

    def record(i):
        """what is saved after batch i"""
        r = realized.cpu().numpy() if realized is not None and hasattr(realized, "cpu") else (None if realized is None else np.array(realized))
        return (renderer.image64_numpy(), i + 1, r)

    def due(i):
        return (i + 1) % every == 0 or i + 1 == len(batch_tables)

    if checkpoint is not None:
        saved = checkpoint.load(chk_name)
        if saved is not None:
            image, first_batch = saved[0], saved[1]
            renderer.set_image64(image)
            if realized is not None and len(saved) > 2 and saved[2] is not None:
                if hasattr(realized, "copy_"):
                    realized.copy_(renderer.torch.from_numpy(np.asarray(saved[2])).to(realized.device))
                else:
                    realized[...] = saved[2]
    if _pool_fits(renderer, sum(int(t["n_phot"].sum()) for t, _ in batch_tables)):
        # the HBM-resident form (prepared_image): all photons shot once, every batch only the pixel search of its share --
        # the sub-batches of the reference bound the memory of its photon arrays and do not change the image
        def save(i):
            if checkpoint is not None and due(i):
                checkpoint.save(chk_name, record(i))
        if first_batch < len(batch_tables):
            prepared_image(renderer, objects, modes, nbatch=nbatch, seed=seed, rank=rank, world=world, resident=True,
                           realized=realized, after_batch=save, first_batch=first_batch)()
        mine = parallel.assign_ranks(objects["n_phot"], world) == rank if world > 1 else None
        for i, (t, index) in enumerate(batch_tables):
            sel = t["n_phot"] > 0
            if mine is not None and i >= first_batch:
                sel &= mine[index]
            total += int(t["n_phot"][sel].sum())
        return total
    sensor_on = renderer.scene.sensor is not None
    unit = parallel.unit_flux_path(renderer.scene, objects)
    owner = parallel.assign_ranks(objects["n_phot"], world)
    nsub = max(min(nsubbatch, len_smallest or 1), 1)
    for i, (table, index) in enumerate(batch_tables):
        if i < first_batch:
            total += int(table["n_phot"].sum())
            continue
        if len(table) == 0:
            continue
        table["bf_state"] = 0
        table["flags"] &= ~IMS_OBJ_FAINT     # in pooling mode the ops and the sensor see every photon (photon_pooling.py:154-159)
        rows = np.arange(len(table))
        for s, sub in enumerate(make_photon_subbatches(list(rows), nsub)):
            if not sub:
                continue
            sub = np.asarray(sub)
            t = table[sub]
            keep = t["n_phot"] > 0
            t, idx = t[keep], index[sub][keep]
            if sensor_on and s == 0 and i > first_batch:
                # recalc=(subbatch_num == 0), resume afterwards; only tiles near the previous batch's charge move
                # (with several ranks the tile marks are rank-local, so every tile is visited)
                if parallel.exchanging(world):
                    parallel.allreduce_delta(renderer.delta_tensor(0), integer_counts=unit)      # as the resident form
                renderer.update_distortions(0, 1, bf_tag=((i - 1) % 255 + 1) if world == 1 else 0)
            if world > 1:
                mine = owner[idx] == rank
                t, idx = t[mine], idx[mine]
            if len(t) == 0:
                continue
            pool = renderer.shoot_photons(t)
            renderer.apply_ops(pool)
            tmp = None
            if realized is not None:
                tmp = renderer.torch.zeros(len(t), dtype=renderer.torch.float64, device=renderer.device)
            renderer.accumulate(pool, realized=tmp, bf_tag=(i % 255 + 1) if (sensor_on and world == 1) else 0)
            if realized is not None:
                realized.index_add_(0, renderer.torch.from_numpy(idx).to(renderer.device), tmp)
            total += int(t["n_phot"].sum())
        if checkpoint is not None and due(i):
            checkpoint.save(chk_name, record(i))
    return total
