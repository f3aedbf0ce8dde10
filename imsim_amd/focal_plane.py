"""A visit's CCDs on the GPUs of one node (BASELINE config C5: 189 CCDs x 10 k sources, one CCD per stream).

The reference fans the CCDs of a visit out to worker processes (`output.nproc`, imsim/ccd.py:72-89); every
CCD is an independent `LSST_Image` build.  Here the CCDs are dealt round-robin to the ranks
(`parallel.shard_ccds`, one process per GPU, no exchange) and inside a rank up to `concurrent` CCDs (default 3) are in
flight at once: while the GPU works through the launch plan of one CCD (which `Renderer.execute_plan` only enqueues), the
host builds the object table and plan of the next.  All CCDs of a device share FOUR streams by role
(`engine._focal_streams`: two for the long top chains, taken in turn, one for the wide launches, one for the middle and low
chain classes; HIP has four hardware queues, and any further stream shares one): 7.6 ms per 10 k-source CCD against
9.9 - 11.7 ms with one stream set for every renderer plus an anchor stream per CCD in flight (`IMS_FOCAL_STREAMS=0`).
Results do not depend on `concurrent`, on the rank count or on the order of the CCDs: every photon's random
stream is addressed by (CCD seed, object id, photon index).
"""
import os
import time

import numpy as np

from . import parallel, lsst_image
from . import tuning
from .engine import Renderer


_ANCHOR_STREAMS = {}
_HOST_IMAGES = {}           # (device, image shape) -> {"free": page-locked float32 buffers not in use, "made": how many exist}: kept for the
                            # life of the process -- page-locking 67 MB costs ~5 ms, and a step that runs short of buffers at its end (the
                            # last batches' images queue up behind PCIe) used to make twenty of them there: 1.30 instead of 1.15 s, one step in three
_ARENA_BYTES = {}           # device -> bytes of the sensor arena this process holds there (they count as usable when it is re-sized)


def wants_static_late(work):
    """the sky stage of a job draws on the pixel areas of slot 0 in the CCD's tail (lsst_image.sky_pixel_areas): such a CCD
    keeps a state of its own -- and has it made"""
    sky = getattr(work, "sky", None)
    return sky is not None and bool(sky.get("pixel_areas", True))


def _render_joint(mine, build, dev, nrecalc, sink, post, joint, chain_hint):
    """The CCDs in batches of `joint` whose top brighter-fatter chains advance in lockstep: ONE launch per kernel and round for
    the whole batch (engine.run_joint_plans / ims_plans_run_joint) instead of a chain per CCD on a stream of its own -- chains on
    different streams barely overlap on the device, while a launch that holds the same round of eight chains costs little more
    than the round of one.  Streams by role for all CCDs: `pre` (FFT objects, the regions' initial state, first pool slices),
    bulk (wide launches), mid (middle / low chain classes, then each CCD's tail: join, sky, float image, copy to the host),
    joint (the rounds).  Host order: the front of batch b + 1 is enqueued BEFORE the tails of batch b, so the shared streams
    work ahead while the joint chain of batch b runs.  chain_hint(key) -> a number that grows with the CCD's longest chain
    (its brightest photon-shot object): CCDs of similar chain length share a batch, the sum of the batches' longest chains falls."""
    import torch
    from .engine import _focal_streams, run_joint_plans, lazy_static_applies
    from . import _abi
    st_joint = _focal_streams(torch, dev, top_index=0)[0]
    pre, bulk, mid = _focal_streams(torch, dev, top_index=1)[:3]
    if pre is st_joint:
        raise RuntimeError("the joint focal plane needs two top streams (IMS_FOCAL_TOPS >= 2)")
    if tuning.env("IMS_FOCAL_PRE_PRIORITY", "0") == "0":
        # `pre` carries wide work (FFT draws, the regions' initial state, first pool slices): at the priority of the joint rounds it
        # competes with them for every wave slot -- at normal priority C5 takes 10.0 instead of 10.9 ms per CCD
        from .engine import _DEVICE_STREAMS
        pre = _DEVICE_STREAMS[("focal", str(dev))]["pre"]        # made (and first used) with the other role streams
    # IMS_FOCAL_FFT (here: mid; bulk; top = ahead of the plan on `pre`): the stream of a CCD's FFT-drawn objects, beside its plan.
    # Measured on C5 with the renderer's initialisation on `pre` (IMS_FOCAL_JOINT_INIT): mid 9.4, bulk 9.5, top 9.6 ms per CCD
    # (10.0 with the initialisation on the bulk stream)
    fft_on = {"bulk": bulk, "mid": mid}.get(tuning.env("IMS_FOCAL_FFT", "mid"))
    order = list(mine)
    if chain_hint is not None and tuning.env("IMS_NO_HINT", "0") != "1":
        order.sort(key=chain_hint, reverse=True)
    # Two batches of renderers are alive at a time (one running its joint rounds, one being enqueued): the batch is cut to what
    # the device's memory holds -- per CCD the static pixel-boundary state and the scratch regions (235 B per cell), the f64
    # image and its float copy, and ~1.5 GB for the photon pool and the FFT buffers of a bright CCD.  (C5 bench scene, 6 M scratch
    # cells: 6.9 GB per CCD, 16 per batch on 288 GB; config.Process's default of 24 M scratch cells: 11 GB, 10 per batch.)
    # IMS_FOCAL_ARENA (default 1): the pixel-boundary state of the CCDs in flight comes out of ONE arena per device
    # (engine.SensorArena: three rotating static regions, private cells leased by need) instead of 5.3 GB per renderer from the
    # allocator -- per CCD then ~0.3 GB of private cells, the f64 image and its float copy, and the pool / FFT buffers.
    def private_need(work, scene):
        """owner cells the private regions of a CCD's bright objects take (the sum of their stamps), capped by the scene's capacity"""
        ss = getattr(scene, "sensor", None)
        if ss is None:
            return 0
        table = work.objects if isinstance(work, lsst_image.CcdJob) else work
        if not isinstance(table, np.ndarray):
            return int(ss.scratch_cells)                   # a device table: its stamps are not on the host -- the full capacity
        nrec = getattr(work, "nrecalc", None) or (nrecalc if nrecalc is not None else ss.model.nrecalc)
        br = (table["n_phot"] > nrec) & ((table["flags"] & _abi.IMS_OBJ_FAINT) == 0) if nrec else np.zeros(len(table), bool)
        w = table["stamp_xmax"][br].astype(np.int64) - table["stamp_xmin"][br] + 2
        h = table["stamp_ymax"][br].astype(np.int64) - table["stamp_ymin"][br] + 2
        return int(min(int((w * h).sum()), int(ss.scratch_cells)))

    prebuilt = {}
    arena = None
    alive = max(int(tuning.env("IMS_FOCAL_ALIVE")), 2)            # batches alive at a time (the pipeline below)
    # IMS_FOCAL_LAZY_STATIC (default 1): a CCD's static pixel-boundary state (3.9 GB, 200 M evaluations of the tree-ring form)
    # is not made: only the fused launch of the ordinary objects reads it, for the 2 % of its photons near a pixel edge, which a
    # second launch finishes from the closed form (Renderer(lazy_static=True); C5 1.53 -> 1.33 s, same images)
    lazy_static = tuning.flag("IMS_FOCAL_LAZY_STATIC")
    if order:
        prebuilt[order[0]] = build(order[0])
        sc0, work0 = prebuilt[order[0]]
        ss0 = getattr(sc0, "sensor", None)
        use_arena = (tuning.env("IMS_FOCAL_ARENA", "1") != "0" and ss0 is not None and ss0.slots is not None and len(ss0.slots) == 1
                     and not sc0.track_static_delta)
        free, total = torch.cuda.mem_get_info(dev)
        usable = free + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)      # the allocator's cached blocks count
        if use_arena:
            from .engine import sensor_arena
            # cells per CCD the pool is sized for: IMS_FOCAL_ARENA_FACTOR (0.8) x what the first CCD needs (the one with the longest
            # chain when the caller gave a hint: more than the average CCD), at least a million, at most the scene's capacity; a
            # CCD that needs more gets more while the pool lasts (lease by need), a pool that runs dry collects the oldest batch
            # early or cuts the batch short (below)
            per_cells = int(min(max(float(tuning.env("IMS_FOCAL_ARENA_FACTOR")) * private_need(work0, sc0), 1.0e6), max(int(ss0.scratch_cells), 1)))
            per_ccd = per_cells * (ss0.owned_points() * 16 + 75) + sc0.nx * sc0.ny * 12 + 1.0e9
            # (with the static state not made, the leases' static region is an address nobody looks behind: one is enough)
            n_static = 1 if lazy_static else int(tuning.env("IMS_FOCAL_STATIC_REGIONS", "3"))
            static_bytes = n_static * ss0.total_cells() * (ss0.owned_points() * 16 + 75)
            have = _ARENA_BYTES.get(str(dev), 0)
            fit = int((0.85 * (usable + have) - static_bytes) / (alive * per_ccd))
            if fit < joint:
                joint = max(fit, 1)
            pool_cells = alive * joint * per_cells
            if tuning.env("IMS_FOCAL_ARENA_CELLS"):          # the private pool's size, as given (tests: a pool that runs dry)
                pool_cells = int(tuning.env("IMS_FOCAL_ARENA_CELLS"))
            arena = sensor_arena(torch, dev, ss0.owned_points(), ss0.total_cells(), n_static, pool_cells,
                                 exact=bool(tuning.env("IMS_FOCAL_ARENA_CELLS")))
            _ARENA_BYTES[str(dev)] = arena.nbytes()
        else:
            cells = ((sc0.nx + 1) * (sc0.ny + 1) + int(getattr(ss0, "scratch_cells", 0))) if ss0 is not None else 0
            per_ccd = cells * 235 + sc0.nx * sc0.ny * 12 + 1.5e9
            fit = int(0.85 * usable / (alive * per_ccd))
            if fit < joint:
                joint = max(fit, 1)
    render_focal_plane.last_joint_batch = joint
    render_focal_plane.last_arena_gib = arena.nbytes() / 2.0**30 if arena is not None else 0.0
    host_cap = max(int(tuning.env("IMS_FOCAL_PINNED")), 2)       # page-locked image buffers in flight at most
    inflight = []                                                # entries whose tails are enqueued and whose images are not handed on yet, oldest first
    out = {}
    host_s = [0.0]
    n_joint = [0]

    # IMS_FOCAL_AHEAD (default "pre:1"): the host enqueues the front of a CCD only when the front of the CCD `n` places before it
    # has run through on that stream -- deliberately: with everything of a batch queued at once the wide launches of eight CCDs
    # run side by side with the middle chains and the FFT draws, and all of them (and the joint rounds most of all) get slower
    # than they are one CCD after the other (measured on C5: 16.4 ms per CCD unthrottled, 13.3 with the pacing that synchronous
    # uploads used to impose by accident)
    ahead_on, _, ahead_n = tuning.env("IMS_FOCAL_AHEAD", "pre:1").partition(":")
    ahead_n = int(ahead_n or 0)
    fronts = []
    trace = [] if tuning.env("IMS_FOCAL_TRACE", "0") == "1" else None
    if trace is not None:
        t_base = torch.cuda.Event(enable_timing=True)
        t_base.record(bulk)
        h_base = time.perf_counter()

    def front(key, own_state=False):
        """everything of one CCD up to its deferred rounds; None when the arena cannot lease its private cells now (the CCD
        stays prebuilt for the retry).  own_state: the CCD's pixel-boundary state is not leased from the arena but its own (a CCD
        whose bright stamps need more private cells than the whole pool, with nothing alive to give any back: ADVICE r5)."""
        scene, work = prebuilt.pop(key) if key in prebuilt else build(key)
        lease = None
        ss = getattr(scene, "sensor", None)
        if (arena is not None and not own_state and ss is not None and not wants_static_late(work) and not scene.track_static_delta
                and (not lazy_static or lazy_static_applies(scene))          # (the lazy arena has ONE static region nobody may write)
                and ss.slots is not None and len(ss.slots) == 1 and ss.total_cells() == arena.static_cells
                and ss.owned_points() == arena.npo):
            # (a CCD of another geometry or sensor model than the arena was sized for keeps a state of its own)
            lease = arena.lease(private_need(work, scene))
            if lease is None:
                prebuilt[key] = (scene, work)
                return None
            if lazy_static:
                lease.static_waits = []                # nobody reads or writes the region: nothing to wait for
        if ahead_n > 0 and len(fronts) >= ahead_n:
            fronts[-ahead_n].synchronize()
        t_host = time.perf_counter()
        e = _front(key, scene, work, lease)
        ev = torch.cuda.Event()
        ev.record({"pre": pre, "bulk": bulk, "mid": mid}[ahead_on])
        fronts.append(ev)
        del fronts[:-8]
        if trace is not None:
            # IMS_FOCAL_TRACE=1: when did this CCD's work end on every stream (device clocks) and when was it enqueued (host)
            marks = {}
            for name, st in (("bulk", bulk), ("pre", pre), ("mid", mid)):
                marks[name] = torch.cuda.Event(enable_timing=True)
                marks[name].record(st)
            trace.append(dict(key=key, host0=t_host, host1=time.perf_counter(), marks=marks))
            e["trace"] = trace[-1]
        return e

    init_on = {"bulk": bulk, "mid": mid, "pre": pre}[tuning.env("IMS_FOCAL_JOINT_INIT", "pre")]

    def _front(key, scene, work, lease=None):
        with torch.cuda.stream(init_on):
            renderer = Renderer(scene, dev, stream_roles="focal", top_index=1, lease=lease,
                                lazy_static=lazy_static and not wants_static_late(work))
            if renderer.plan_streams[0] is not pre:
                renderer.plan_streams = (pre,) + tuple(renderer.plan_streams[1:])
                renderer.s_chain = pre
            ready = torch.cuda.Event()
            ready.record(init_on)
        with torch.cuda.stream(pre):
            pre.wait_event(ready)
            if isinstance(work, lsst_image.CcdJob):
                if work.nrecalc is None:
                    work.nrecalc = nrecalc
                fin = lsst_image.draw_job(renderer, work, defer=True, fft_stream=fft_on)
                plan = fin.plan
            else:
                plan = renderer.render_lsst_image(work, nrecalc=nrecalc, defer=True)
                fin = plan.join if plan is not None else (lambda: None)
                if plan is None:
                    # (the render could not be deferred and ran whole, joined into `pre`: the tail on `mid` must wait for it)
                    whole = torch.cuda.Event()
                    whole.record(pre)
                    fin = lambda whole=whole: torch.cuda.current_stream(dev).wait_event(whole)      # noqa: E731
        if lease is not None:
            # the static region's readers (the fused launch of the ordinary objects, on the stream of the wide launches) are
            # queued: the region may go to the next CCD behind them.  A plan that was not deferred ran whole on these streams too.
            lease.release_static({id(st): st for st in list(renderer.plan_streams) + [pre, init_on]}.values())
        return dict(key=key, renderer=renderer, fin=fin, plan=plan)

    def tail(entries):
        # shortest chain first: mid must not sit behind the batch's longest chain while the others' images wait
        for e in sorted(entries, key=lambda e: e["plan"].sizes.n_round_launches if e["plan"] is not None else 0):
            t_tail = time.perf_counter()
            host = host_image(tuple(e["renderer"].image.shape))       # (may hand an older image on first: not under `mid`)
            t_buf = time.perf_counter()
            with torch.cuda.stream(mid):
                e["fin"]()
                # (the image rounded straight into the page-locked buffer by one small launch instead of float copy + transfer was
                # measured twice in round 5: 1.91 against 1.78 s, 1.48 against 1.31 s; removed)
                host.copy_(e["renderer"].image_float(), non_blocking=True)
                done = torch.cuda.Event(enable_timing=trace is not None)
                done.record(mid)
            e["host"], e["done"] = host, done
            inflight.append(e)
            if trace is not None:
                e["trace"]["done"] = done
                e["trace"]["tail_host"] = (t_tail, t_buf, time.perf_counter())

    def host_image(shape):
        """a page-locked float32 buffer for one CCD image: a free one, a new one while fewer than IMS_FOCAL_PINNED exist, else the
        buffer of the oldest image in flight once that image has been handed on (PCIe is busy with it or behind it anyway)"""
        pool = _HOST_IMAGES.setdefault((str(dev), shape), {"free": [], "made": 0})
        while not pool["free"] and pool["made"] >= host_cap and inflight:
            collect_one(inflight[0])
        if pool["free"]:
            return pool["free"].pop()
        pool["made"] += 1
        return torch.empty(shape, dtype=torch.float32, pin_memory=True)

    def collect_one(e):
        if e.get("collected"):
            return
        e["done"].synchronize()
        if post is not None:
            post(e["key"], e["renderer"])
        if sink is not None:
            sink(e["key"], e["host"].numpy())
        else:
            out[e["key"]] = e["host"].numpy().copy()
        _HOST_IMAGES[(str(dev), tuple(e["host"].shape))]["free"].append(e["host"])
        e["renderer"].release_state()          # everything of this CCD has run (its image is on the host): its cells go back
        inflight.remove(e)
        e.clear()                              # drops the renderer: its HBM goes back to the caching allocator
        e["collected"] = True

    def collect(entries):
        for e in sorted(entries, key=lambda e: (e["plan"].sizes.n_round_launches if e["plan"] is not None else 0) if not e.get("collected") else -1):
            collect_one(e)

    # The pipeline.  Per batch: the fronts of its CCDs (this thread, paced) | the joint runs of its chain classes -- the middle and
    # low classes on `mid`, the top class on the joint stream: thousands of short launches, 5 .. 90 ms of host time per batch,
    # enqueued by a SECOND host thread (the library's calls release the interpreter lock) while this thread goes on with the
    # fronts of the next batch | its tails (join, sky, float image, copy to the host), enqueued once its rounds have RUN, so that
    # `mid` never sits in a join behind a chain that is still running | collected when the copies are through.  Nothing here
    # waits for the device except the pacing of the fronts and the bound on the batches alive (IMS_FOCAL_ALIVE, default 2:
    # memory).  Measured in round 5 (profiles/round5_c5_*): with everything enqueued by one thread and a wait for batch b - 1
    # before the fronts of batch b + 1, the joint stream and the stream of the wide launches took turns, ~100 ms each, and the
    # host spent 156 of 600 ms (64 CCDs) enqueueing rounds while the front streams idled.
    from concurrent.futures import ThreadPoolExecutor
    threaded = tuning.flag("IMS_FOCAL_JOINT_THREAD")
    worker = ThreadPoolExecutor(1) if threaded else None
    rounds = []                         # batches whose joint runs are being / have been enqueued: dict(entries, future, ran)
    awaiting = []                       # batches whose tails are enqueued, oldest first
    pending = list(order)

    # (The top classes of consecutive batches on TWO high-priority streams in turn -- two batches' rounds side by side -- were
    # measured in round 5: 1.61 - 1.65 against 1.54 s whatever shared a hardware queue with whatever; removed.)
    def joint_runs(entries):
        """both joint runs of a batch (worker thread); returns the events behind them"""
        torch.cuda.set_device(dev)
        left = [e["plan"] for e in entries if e["plan"] is not None and getattr(e["plan"], "deferred", 0)]
        # the middle and low chain classes of the batch jointly on the stream that carried them one CCD at a time, the top class
        # (the brightest stars) on the joint stream
        run_joint_plans(left, mid, 1, 3)
        ev_mid = torch.cuda.Event()
        ev_mid.record(mid)
        run_joint_plans(left, st_joint, 0, 1)
        ev_top = torch.cuda.Event(enable_timing=trace is not None)
        ev_top.record(st_joint)
        if trace is not None:
            for e in entries:
                e["trace"]["joint_end"] = ev_top
                e["trace"]["host_joint"] = time.perf_counter()
        return len(left), (ev_mid, ev_top)

    def start_rounds(entries):
        if worker is not None:
            rounds.append(dict(entries=entries, future=worker.submit(joint_runs, entries)))
        else:
            class _Done:
                def __init__(self, v):
                    self.v = v

                def done(self):
                    return True

                def result(self):
                    return self.v
            rounds.append(dict(entries=entries, future=_Done(joint_runs(entries))))

    def service(block_oldest=False, block_all=False):
        """move batches along without waiting: tails for batches whose rounds have run, collection of batches whose copies are
        through.  block_oldest: the oldest batch in its rounds gets its tails now whether its rounds have run or not (memory
        bound: its cells are needed); block_all: every batch (end of the step)."""
        first = True
        while rounds:
            b = rounds[0]
            force = block_all or (block_oldest and first)
            first = False
            if not (force or b["future"].done()):
                break
            n_left, evs = b["future"].result()
            if not force and not all(ev.query() for ev in evs):
                break
            n_joint[0] += n_left
            tail(b["entries"])
            awaiting.append(b["entries"])
            rounds.pop(0)
        while awaiting and all(e.get("collected") or e["done"].query() for e in awaiting[0]):
            collect(awaiting.pop(0))

    def alive_batches():
        return len(rounds) + len(awaiting)

    while pending:
        t0 = time.perf_counter()
        cur = []
        while pending and len(cur) < joint:
            service()
            e = front(pending[0])
            if e is None:
                # the arena's private pool is dry: wait for the cells of the oldest batch alive, else this batch ends here
                if rounds or awaiting:
                    if not awaiting:
                        service(block_oldest=True)         # (may collect the batch itself when its copies are through already)
                    if awaiting:
                        collect(awaiting.pop(0))
                    continue
                if cur:
                    break
                # nothing alive, nothing in this batch, and the pool still cannot hold this CCD's private regions: it is sized from
                # the FIRST CCD, and memory pressure may have cut the batch -- the CCD renders with a state of its own
                e = front(pending[0], own_state=True)
            pending.pop(0)
            cur.append(e)
        start_rounds(cur)
        host_s[0] += time.perf_counter() - t0
        # memory: at most `alive` batches (this one included) before the next one starts
        while alive_batches() > alive - 1:
            if not awaiting:
                service(block_oldest=True)
            else:
                collect(awaiting.pop(0))
    service(block_all=True)
    while awaiting:
        collect(awaiting.pop(0))
    if worker is not None:
        worker.shutdown(wait=True)
    if trace is not None:
        torch.cuda.synchronize()
        print("focal trace [ms since the start]: CCD, host enqueue begin / end, its work's end on bulk / pre / mid, its batch's joint rounds end, image on host")
        for t in trace:
            g = {k: t_base.elapsed_time(v) for k, v in t["marks"].items()}
            print(f"  CCD {t['key']:4d} host {1e3 * (t['host0'] - h_base):8.1f} {1e3 * (t['host1'] - h_base):8.1f} | bulk {g['bulk']:8.1f} pre {g['pre']:8.1f} "
                  f"mid {g['mid']:8.1f} | joint end {t_base.elapsed_time(t['joint_end']):8.1f} (enqueued {1e3 * (t['host_joint'] - h_base):8.1f}) "
                  f"done {t_base.elapsed_time(t['done']):8.1f} | tail enqueued {1e3 * (t['tail_host'][0] - h_base):8.1f} buffer {1e3 * (t['tail_host'][1] - t['tail_host'][0]):6.2f} "
                  f"rest {1e3 * (t['tail_host'][2] - t['tail_host'][1]):6.2f}")
    render_focal_plane.last_host_ms_per_ccd = 1e3 * host_s[0] / max(len(mine), 1)
    render_focal_plane.last_joint_plans = n_joint[0]          # CCDs whose top chain ran in joint launches
    return out


def reserve_host_images(device, shape, count=None):
    """Make the page-locked float32 buffers a focal plane's images travel through (up to IMS_FOCAL_PINNED of them) NOW, one after
    the other in a quiet moment, instead of one by one while the visit runs: the pages of a buffer made under load came out
    scattered often enough that a third of the steps ended with twenty image copies at 12 GB/s instead of 52 (round 6,
    tools/dbg/pinned_rate.py: made in a row, 48 of 48 buffers copy at full rate).  ~3 ms per 64-MiB buffer; kept for the life of the
    process.  Returns how many exist."""
    import torch
    dev = torch.device(device)
    shape = tuple(int(v) for v in shape)
    want = max(int(tuning.env("IMS_FOCAL_PINNED")), 2) if count is None else int(count)
    pool = _HOST_IMAGES.setdefault((str(dev), shape), {"free": [], "made": 0})
    while pool["made"] < want:
        pool["free"].append(torch.empty(shape, dtype=torch.float32, pin_memory=True))
        pool["made"] += 1
    return pool["made"]


def warm_fft(device="cuda:0"):
    """Start the hipFFT plans of the stream a focal plane's FFT draws run on (fft_draw.warm_up) in the background: call it as
    soon as the device is known -- before the catalogs are read -- so that the seconds a process's first plan costs are not paid
    in front of the visit's first CCD with a bright star.  Returns the thread (None when the joint path is off)."""
    import torch
    from . import fft_draw
    from .engine import _focal_streams
    dev = torch.device(device)
    if tuning.env("IMS_FOCAL_STREAMS") == "0" or int(tuning.env("IMS_FOCAL_JOINT")) <= 1 or not tuning.flag("IMS_FFT_WARM"):
        return None
    torch.cuda.set_device(dev)
    pre, bulk, mid = _focal_streams(torch, dev, top_index=1)[:3]
    on = {"bulk": bulk, "mid": mid}.get(tuning.env("IMS_FOCAL_FFT", "mid"), pre)
    return fft_draw.warm_up(dev, on)


def render_focal_plane(ccds, build, device="cuda:0", rank=0, world=1, concurrent=3, nrecalc=None, sink=None, post=None,
                       chain_hint=None):
    """ccds: sequence of CCD keys (detector numbers / names); build(key) -> (scene, work) prepares one CCD on the host, work
    being either the object table of a photon-shot CCD or an `lsst_image.CcdJob` -- the per-CCD build of the reference
    (imsim/lsst_image.py:276-395 under the fan-out of imsim/ccd.py:72-89): FFT-drawn objects first, then the launch plan of the
    photon-shot ones, then the optional sky.  Returns {key: float32 image as a host array} for the CCDs this rank owns.

    post(key, renderer): optional hook run once the CCD's work is through, before its renderer is dropped (readout and
    the FITS writers of config.Process work on the device image).

    The finished image of a CCD is copied into a page-locked host buffer on the CCD's own stream (one buffer per
    stream in flight, reused).  sink(key, image): optional consumer (the FITS writer in production) that is handed
    a VIEW of that buffer, valid until it returns; with a sink nothing is accumulated and {} is returned."""
    import torch
    dev = torch.device(device)
    mine = parallel.shard_ccds(list(ccds), rank, world)
    if concurrent < 1:
        raise ValueError("concurrent must be >= 1")
    # plan streams by role for all CCDs of the device: the long chains of two CCDs side by side (engine._focal_streams)
    roles = "focal" if tuning.env("IMS_FOCAL_STREAMS", "1") != "0" else "single"
    # IMS_FOCAL_JOINT (default 20; 0 / 1: off): the top chains of that many CCDs advance jointly (_render_joint)
    joint = int(tuning.env("IMS_FOCAL_JOINT"))
    heavy = False
    if roles == "focal" and mine:
        # Joint rounds pay where a CCD's chains are few objects wide (a focal plane of 10 k-source CCDs: 150 objects with rounds
        # of their own, 9.4 against 23 ms per CCD; 13.9 against 15.8 with four CCDs).  A CCD of 100 k sources has 1 500 of
        # them: its launches are wide already, and what counts is that the next CCD's host work overlaps its 25 ms on the GPU,
        # which the rolling window below does and a batch does not (three such CCDs: 60 against 77 ms per CCD).  The first
        # CCD's work decides (IMS_FOCAL_JOINT_MAX_BRIGHT objects with rounds of their own, default 600).
        first_key = max(mine, key=chain_hint) if chain_hint is not None and tuning.env("IMS_NO_HINT", "0") != "1" else mine[0]
        first = build(first_key)
        cache = {first_key: first}
        inner = build

        def build(key, inner=inner, cache=cache):               # noqa: F811 -- the first CCD is built once
            return cache.pop(key) if key in cache else inner(key)
        work = first[1]
        table = work.objects if isinstance(work, lsst_image.CcdJob) else work
        n_phot = np.asarray(table["n_phot"] if isinstance(table, np.ndarray) else getattr(table, "n_phot", np.zeros(0)))
        sensor = getattr(first[0], "sensor", None)
        nrec = nrecalc if nrecalc is not None else (getattr(work, "nrecalc", None) or (sensor.model.nrecalc if sensor is not None else 0))
        bright = int(np.count_nonzero(n_phot > nrec)) if nrec else 0
        heavy = bright > int(tuning.env("IMS_FOCAL_JOINT_MAX_BRIGHT", "600"))
        if not heavy and joint > 1 and tuning.env("IMS_NATIVE_PLAN", "1") != "0":
            torch.cuda.set_device(dev)
            # (the cyclic collector is paused while the visit renders: with the objects of 189 CCDs alive a generation-2 pass inside
            # the call is 80 ms of a 1.2-s visit, one call in three -- profiles/round6_c5_gc.log; reference counting frees what the
            # loop drops, and the collector runs again as soon as the call is over)
            import gc
            pause = gc.isenabled() and not tuning.flag("IMS_FOCAL_GC")
            if pause:
                gc.disable()
            try:
                with tuning.scoped(IMS_PHOTON_LDS=tuning.env("IMS_FOCAL_PHOTON_LDS")):
                    return _render_joint(mine, build, dev, nrecalc, sink, post, min(joint, 64), chain_hint)
            finally:
                if pause:
                    gc.enable()
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
        if post is not None:
            post(key, renderer)
        if sink is not None:
            sink(key, host.numpy())
        else:
            out[key] = host.numpy().copy()
        in_flight[slot] = None                 # drops the renderer: its HBM goes back to the caching allocator

    anchor_role = tuning.env("IMS_FOCAL_ANCHOR", "top") if roles == "focal" else ""
    # (pacing, below: for CCDs of the focal-plane kind only -- a CCD of 100 k sources keeps the GPU busy for 25 ms while the host
    # plans the next one, and waiting serialises the two: 70 against 59 ms per CCD)
    pace = not heavy and tuning.env("IMS_FOCAL_AHEAD", "pre:1") not in ("pre:0", "bulk:0", "mid:0", "0")
    host_s = [0.0]

    def enqueue(k, key, slot):
        """Everything of one CCD up to the event behind its image copy (host work + asynchronous launches)."""
        torch.cuda.set_device(dev)
        scene, work = build(key)               # host work, overlaps with the CCDs still running on the GPU
        anchor = init_on = copy_on = streams[slot]
        fft_on = None
        top_index = None
        if anchor_role:
            # no stream beside the four plan streams of the device: a fifth would share a hardware queue with one of them
            from .engine import _focal_streams
            top_index = k            # dealt over the device's top-chain streams (engine._focal_streams: index modulo their number)
            peek = _focal_streams(torch, dev, top_index=top_index)
            by_role = {"top": peek[0], "bulk": peek[1], "mid": peek[2]}
            anchor = by_role[anchor_role]
            # measured on 24 CCDs of C5 (tools/dbg/r3_c5.sh): the static state's initialisation beside the wide launches, the
            # image copy behind the middle chains: 7.2 ms per CCD with three CCDs in flight; everything on the top stream 8.0 - 8.7,
            # the copy on the stream of the wide launches 12 - 14
            init_on = by_role[tuning.env("IMS_FOCAL_INIT", "bulk")]
            copy_on = by_role[tuning.env("IMS_FOCAL_COPY", "mid")]
            # IMS_FOCAL_FFT=mid / bulk: the FFT-drawn objects beside the launch plan instead of ahead of it on the top-chain stream.
            # Measured on 24 CCDs of C5: 21.4 ms per CCD on the wide-launch stream, 32.9 on the middle stream, 21.6 ahead on the
            # top stream -- no gain worth a second ordering; default: ahead
            fft_on = by_role.get(tuning.env("IMS_FOCAL_FFT", "top"))
            if fft_on is anchor:
                fft_on = None
        with torch.cuda.stream(init_on):
            # (the static pixel-boundary state is not made where only the fused launch would read it: Renderer(lazy_static=True))
            renderer = Renderer(scene, dev, stream_roles=roles, top_index=top_index,
                                lazy_static=tuning.flag("IMS_FOCAL_LAZY_STATIC") and not wants_static_late(work))
            ready = torch.cuda.Event()
            ready.record(init_on)
        if pace:
            # the host goes on with this CCD's plan only when its renderer's initialisation has run -- i.e. when the stream of the
            # wide launches is through with the CCD before: with several CCDs' wide launches queued at once everything, the
            # latency-bound rounds most, gets slower (C5 with a chain per CCD: 4.8 s unpaced, 4.4 s paced -- what synchronous
            # uploads used to do by accident)
            ready.synchronize()
        with torch.cuda.stream(anchor):
            anchor.wait_event(ready)
            if isinstance(work, lsst_image.CcdJob):
                if work.nrecalc is None:
                    work.nrecalc = nrecalc
                lsst_image.draw_job(renderer, work, fft_stream=fft_on)
            else:
                renderer.render_lsst_image(work, nrecalc=nrecalc)
            through = torch.cuda.Event()
            through.record(anchor)
        with torch.cuda.stream(copy_on):
            copy_on.wait_event(through)
            img = renderer.image_float()
            if pinned[slot] is None or pinned[slot].shape != img.shape:
                pinned[slot] = torch.empty(img.shape, dtype=img.dtype, pin_memory=True)
            pinned[slot].copy_(img, non_blocking=True)
            done = torch.cuda.Event()
            done.record(copy_on)
        return key, renderer, pinned[slot], done

    # host threads (IMS_FOCAL_THREADS, default 1): the plan of one CCD is numpy and ctypes work that releases the interpreter
    # lock for most of its time (ims_run_plan enqueues ~550 launches per CCD in C), so two CCDs can be prepared side by
    # side and the library's event tables are locked for it -- measured on 24 CCDs of C5: 7.7 / 7.8 / 8.7 ms per CCD with
    # 1 / 2 / 3 threads: the host is not what a CCD waits for; results do not depend on it
    n_threads = max(1, min(int(tuning.env("IMS_FOCAL_THREADS", "1")), len(streams))) if roles == "focal" else 1
    if n_threads == 1:
        for k, key in enumerate(mine):
            slot = k % len(streams)
            collect(slot)                          # the CCD that used this stream before
            t0 = time.perf_counter()
            in_flight[slot] = enqueue(k, key, slot)
            host_s[0] += time.perf_counter() - t0
    else:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(n_threads) as pool_ex:
            futures = [None] * len(streams)
            for k, key in enumerate(mine):
                slot = k % len(streams)
                if futures[slot] is not None:
                    in_flight[slot] = futures[slot].result()
                    futures[slot] = None
                    collect(slot)
                futures[slot] = pool_ex.submit(enqueue, k, key, slot)
            for slot in range(len(streams)):
                if futures[slot] is not None:
                    in_flight[slot] = futures[slot].result()
    for slot in range(len(streams)):
        collect(slot)
    render_focal_plane.last_host_ms_per_ccd = 1e3 * host_s[0] / max(len(mine), 1)     # host time of the enqueues (one thread)
    return out
