"""Host engine: owns device memory (torch tensors = plumbing only) and drives libimsim_hip.so.

`Scene` is a pure-host description of everything one CCD needs (tables, PSF, op chain, optics,
sensor); `Renderer` uploads it once and then renders object tables into the CCD image by calling
the C-ABI.  There is no CPU fallback here: without the HIP library or a GPU this raises.
"""
import ctypes as C
import math
import os
import dataclasses
import threading
import weakref
from typing import List, Optional

import numpy as np

from . import _abi
from . import tuning
from ._abi import (OBJECT_DTYPE, BFSLOT_DTYPE, RenderParams, Photons, Sensor, Optics, Op, PsfComponent)


@dataclasses.dataclass
class SensorSetup:
    """Silicon sensor description (host).  slots[0] is the whole-CCD region (static tree-ring
    boundaries in LSST_Image mode, live brighter-fatter state in pooling mode); further slots are
    private regions of bright objects."""
    model: "object"                       # sensor.SiliconModel
    abs_wl: np.ndarray                    # nm, uniform grid
    abs_len: np.ndarray                   # micron
    tr_table: Optional[np.ndarray] = None  # f(r) on a uniform grid starting at 0
    tr_table2: Optional[np.ndarray] = None  # spline second derivatives (None -> linear interpolation)
    tr_dr: float = 3.0
    tr_center: tuple = (0.0, 0.0)
    slots: Optional[np.ndarray] = None    # BFSLOT_DTYPE array of the static regions (slot 0 = whole CCD)
    scratch_cells: int = 0                # owner-cell capacity for private regions of bright objects
    max_slots: int = 4096

    def owned_points(self):
        return 2 * self.model.num_vertices + 2

    def total_cells(self):
        s = self.slots
        return int(s["offset"][-1] + (int(s["nx"][-1]) + 1) * (int(s["ny"][-1]) + 1)) if len(s) else 0


def make_slots(regions):
    """regions: list of (xmin, ymin, nx, ny) -> BFSLOT_DTYPE array with packed owner-cell offsets."""
    out = np.zeros(len(regions), dtype=BFSLOT_DTYPE)
    off = 0
    for k, (xmin, ymin, nx, ny) in enumerate(regions):
        out[k] = (xmin, ymin, nx, ny, off)
        off += (nx + 1) * (ny + 1)
    return out


@dataclasses.dataclass
class Scene:
    nx: int
    ny: int
    xmin: int = 1
    ymin: int = 1
    seed: int = 0
    psf: List[tuple] = dataclasses.field(default_factory=list)     # (kind, table, p0, chrom_alpha, chrom_base)
    ops: List[tuple] = dataclasses.field(default_factory=list)     # (kind, table, [p0..p5])
    radial_r2: Optional[np.ndarray] = None
    radial_cdf: Optional[np.ndarray] = None
    image_profiles: Optional[list] = None        # 2-D arrays [ny][nx] sampled by IMS_PROF_IMAGE objects (FITS stamps)
    image_interpolant: str = "quintic"           # x_interpolant of those images: GalSim's default, or "nearest"
    sed_tables: Optional[np.ndarray] = None      # [n][n_pts] inverse CDFs uniform in u
    ratio_tables: Optional[np.ndarray] = None    # [n][n_pts] uniform in wavelength
    ratio_wl_min: float = 0.0
    ratio_wl_step: float = 1.0
    optics: Optional[Optics] = None
    sensor: Optional[SensorSetup] = None
    atm: Optional[object] = None                 # atm_psf.AtmosphericPSF (needed by IMS_PSF_SCREENS components)
    seg_size: int = 256
    track_static_delta: int = 0


def plan_bf_groups(objects, nrecalc, n_static, static_cells, scratch_cells, max_slots, max_photons=1 << 62):
    """LSST_Image mode: objects whose own charge triggers pixel-boundary recalculations
    (n_phot > nrecalc) need a private boundary region the size of their stamp.  Returns
    (normal_index, groups) with groups = [(index_array sorted by n_phot desc, slots_array)], each
    group fitting the scratch capacity."""
    n = objects["n_phot"]
    bright = (n > nrecalc) & ((objects["flags"] & _abi.IMS_OBJ_FAINT) == 0) if nrecalc > 0 else np.zeros(len(n), bool)
    normal = np.flatnonzero(~bright)
    idx = np.flatnonzero(bright)
    idx = idx[np.argsort(-n[idx], kind="stable")]
    groups = []
    sx = objects["stamp_xmin"][idx].astype(np.int64)
    sy = objects["stamp_ymin"][idx].astype(np.int64)
    nx = objects["stamp_xmax"][idx].astype(np.int64) - sx + 1
    ny = objects["stamp_ymax"][idx].astype(np.int64) - sy + 1
    cells = (nx + 1) * (ny + 1)
    nph = n[idx].astype(np.int64)
    start = 0
    while start < len(idx):
        # the longest run from `start` that fits the scratch cells, the slot table and the photon pool
        ccum = np.cumsum(cells[start:])
        pcum = np.cumsum(nph[start:])
        fit = (ccum <= scratch_cells) & (np.arange(len(ccum)) < max_slots - n_static)
        fit &= (pcum <= max_photons) | (np.arange(len(ccum)) == 0)
        k = start + (int(np.argmin(fit)) if not fit.all() else len(fit))
        if k == start:
            raise ValueError("brighter-fatter scratch capacity too small for one stamp; raise SensorSetup.scratch_cells")
        slots = np.zeros(k - start, dtype=BFSLOT_DTYPE)
        slots["xmin"], slots["ymin"], slots["nx"], slots["ny"] = sx[start:k], sy[start:k], nx[start:k], ny[start:k]
        slots["offset"] = static_cells + np.concatenate([[0], np.cumsum(cells[start:k])[:-1]])
        groups.append((idx[start:k], slots))
        start = k
    return normal, groups


class HostMem:
    """Pointer provider over numpy arrays (used by the test-side oracle binding, never by the product)."""
    def __init__(self):
        self.keep = []

    def write(self, handle, raw):
        flat = handle.view(np.uint8).reshape(-1)
        flat[:raw.size] = raw

    def put(self, arr, dtype=None):
        a = np.ascontiguousarray(arr, dtype=dtype)
        self.keep.append(a)
        return a, a.ctypes.data

    def put_struct(self, st):
        self.keep.append(st)
        return st, C.addressof(st)

    def zeros(self, n, dtype, uninitialised=0):
        return self.put(np.zeros(n, dtype=dtype))


class _AllEvents:
    """query() of several events at once"""
    def __init__(self, evs):
        self.evs = evs

    def query(self):
        return all(e.query() for e in self.evs)


class DeviceMem:
    """Pointer provider over torch device tensors.

    Small tables (scene tables, descriptors, the slot table, launch-plan arenas) live in ONE device arena per provider and
    are copied through a page-locked mirror of it: the copy is queued on the current stream and the host never waits for
    the GPU (a pageable source makes every hipMemcpy a synchronisation point -- 16 of them per CCD of a focal plane).
    Mirrors are recycled through a process-wide pool (allocating page-locked memory costs milliseconds)."""
    ARENA_BYTES = 16 << 20
    _MIRRORS = []                    # (page-locked tensor, event after the last copy that reads it)

    def __init__(self, device):
        import torch
        if not torch.cuda.is_available():
            raise _abi.ImsimHipError("no GPU visible: the imsim_amd render path is HIP-only (no CPU fallback)")
        self.torch = torch
        self.device = torch.device(device)
        self.keep = []
        self._dev = torch.empty(self.ARENA_BYTES, dtype=torch.uint8, device=self.device)
        self._pin = None
        for k, (t, ev) in enumerate(DeviceMem._MIRRORS):
            if ev.query():
                self._pin = DeviceMem._MIRRORS.pop(k)[0]
                break
        if self._pin is None:
            self._pin = torch.empty(self.ARENA_BYTES, dtype=torch.uint8, pin_memory=True)
        self._pin_np = self._pin.numpy()
        self._used = 0
        self._streams = {}               # every stream a copy out of the mirror was queued on
        self._slots = {}                 # write(): one fixed staging slot per handle, with the event of its last copy

    def _note_stream(self):
        st = self.torch.cuda.current_stream(self.device)
        self._streams[st.cuda_stream] = st
        return st

    def __del__(self):
        # the mirror goes back to the pool behind an event per stream that copied out of it (a renderer dropped while a copy is
        # still pending on a plan stream must not have its mirror overwritten under that copy)
        try:
            if self._pin is not None:
                evs = []
                for st in list(self._streams.values()) or [self.torch.cuda.current_stream(self.device)]:
                    ev = self.torch.cuda.Event()
                    ev.record(st)
                    evs.append(ev)
                DeviceMem._MIRRORS.append((self._pin, _AllEvents(evs)))
                self._pin = None
        except Exception:
            pass

    def _stage(self, raw):
        """raw bytes -> a page-locked copy (view of the mirror), or None when the mirror is full"""
        n = int(raw.size)
        off = (self._used + 255) & ~255
        if self._pin is None or off + n > self.ARENA_BYTES:
            return None, off
        self._pin_np[off:off + n] = raw
        self._used = off + n
        return self._pin[off:off + n], off

    def put(self, arr, dtype=None):
        a = np.ascontiguousarray(arr, dtype=dtype)
        raw = a.view(np.uint8).reshape(-1)
        src, off = self._stage(raw)
        if src is None:
            t = self.torch.from_numpy(raw).to(self.device)
            self.keep.append(t)
            return t, t.data_ptr()
        t = self._dev[off:off + raw.size]
        self._note_stream()
        t.copy_(src, non_blocking=True)
        return t, t.data_ptr()

    def put_struct(self, st):
        raw = np.frombuffer(bytes(st), dtype=np.uint8).copy()
        return self.put(raw)

    def zeros(self, n, dtype, uninitialised=0):
        """n zeroed elements; the first `uninitialised` of them are left as allocated (the caller overwrites every one of them
        before anything reads them: a fill of gigabytes that the next kernel would only write over)."""
        item = np.dtype(dtype).itemsize
        nbytes, skip = int(n) * item, min(int(uninitialised), int(n)) * item
        if skip == 0:
            t = self.torch.zeros(max(nbytes, 8), dtype=self.torch.uint8, device=self.device)
        else:
            t = self.torch.empty(max(nbytes, 8), dtype=self.torch.uint8, device=self.device)
            t[skip:].zero_()
        self.keep.append(t)
        return t, t.data_ptr()

    def write(self, handle, raw):
        """Stream-ordered update of a device table (queued on the current stream).  Every handle has ONE staging slot in the
        mirror, guarded by the event of the last copy out of it: a replay loop that rewrites the slot table a thousand times
        neither eats the mirror nor falls back to pageable (synchronising) copies."""
        key = handle.data_ptr()
        slot = self._slots.get(key)
        if slot is None or slot[1] < raw.size:
            src, off = self._stage(np.zeros(int(raw.size), dtype=np.uint8))
            if src is None:
                handle[:raw.size].copy_(self.torch.from_numpy(raw))
                return
            slot = self._slots[key] = [off, int(raw.size), None]
        off, cap, ev = slot
        if ev is not None:
            ev.synchronize()                           # the previous copy out of this slot (long done in practice)
        self._pin_np[off:off + raw.size] = raw
        st = self._note_stream()
        handle[:raw.size].copy_(self._pin[off:off + raw.size], non_blocking=True)
        ev = self.torch.cuda.Event()
        ev.record(st)
        slot[2] = ev


class SensorArena:
    """The pixel-boundary state of ALL the CCDs a device has in flight, in one set of arrays (boundary points, bounds lines,
    delta charge, tile flags, the `changed` bytes of the update kernels), allocated once per device and process and handed out
    as leases -- instead of 5.3 GB from the allocator for every renderer of a focal plane (imsim/ccd.py:72-89: 189 CCDs per
    visit; a fresh process paid 5.2 s of hipMalloc for 200 GiB of them).

    * `n_static` STATIC regions (slot 0 of a renderer: the whole CCD with its tree rings).  In LSST_Image mode only the fused
      launch of the ordinary objects reads it (objects above nrecalc own private regions), so a CCD holds its static region from
      its initialisation to the end of its front -- a couple of milliseconds, not the hundreds its longest chain takes -- and the
      regions rotate: `lease` hands out the next one with the events behind which its previous user's readers are through.
    * a pool of PRIVATE owner cells, leased by the size a CCD's bright objects need (the sum of their stamps: ~1.2 M cells for a
      10 k-source CCD of the focal-plane bench, where the fixed capacity was 6 M) and returned when the CCD is collected.

    All regions of a lease are addressed through the slot table's cell offsets (ims_bf_slot_t.offset) from the arena's base
    pointers, so kernels and planner see nothing new."""

    def __init__(self, torch, device, npo, static_cells, n_static, private_cells):
        self.torch, self.device = torch, torch.device(device)
        self.npo, self.static_cells, self.n_static = int(npo), int(static_cells), int(n_static)
        self.private_base = self.n_static * self.static_cells
        self.private_cells = int(private_cells)
        total = self.private_base + self.private_cells
        self.total_cells = total
        f64, u8 = torch.float64, torch.uint8
        self.boundary = torch.empty(total * self.npo * 2, dtype=f64, device=self.device)
        self.bounds = torch.empty(total * 8, dtype=f64, device=self.device)
        self.delta = torch.empty(total, dtype=f64, device=self.device)
        self.flags = torch.zeros(3 * total, dtype=u8, device=self.device)       # tile_charge | tile_changed | changed
        self.tile_charge, self.tile_changed, self.changed = self.flags[:total], self.flags[total:2 * total], self.flags[2 * total:]
        self._static_next = 0
        self._static_busy = [[] for _ in range(self.n_static)]     # events of the region's last readers
        self._free = [(self.private_base, self.private_cells)]     # (first cell, cells), ascending, coalesced
        self._lock = threading.Lock()

    def nbytes(self):
        return self.total_cells * (self.npo * 16 + 64 + 8 + 3)

    def lease(self, need_cells):
        """-> SensorLease, or None when the private pool cannot hold `need_cells` now (the caller collects a CCD and retries)"""
        need = max(int(need_cells), 1)
        with self._lock:
            for k, (first, count) in enumerate(self._free):
                if count >= need:
                    if count == need:
                        del self._free[k]
                    else:
                        self._free[k] = (first + need, count - need)
                    idx = self._static_next % self.n_static
                    self._static_next += 1
                    waits, self._static_busy[idx] = self._static_busy[idx], []
                    return SensorLease(self, idx, first, need, waits)
        return None

    def _give_back(self, first, count):
        with self._lock:
            self._free.append((first, count))
            self._free.sort()
            merged = []
            for a, n in self._free:
                if merged and merged[-1][0] + merged[-1][1] == a:
                    merged[-1] = (merged[-1][0], merged[-1][1] + n)
                else:
                    merged.append((a, n))
            self._free = merged


_SENSOR_ARENAS = {}


def sensor_arena(torch, device, npo, static_cells, n_static, private_cells, exact=False):
    """The device's SensorArena for this CCD geometry, created on first use and re-created only when a larger private pool (or
    more static regions) is asked for -- a focal plane after the first finds its state allocated.  exact: a pool of exactly
    `private_cells` (tests)."""
    key = (str(torch.device(device)), int(npo), int(static_cells))
    a = _SENSOR_ARENAS.get(key)
    if a is not None and a.n_static >= n_static and (a.private_cells == private_cells if exact else a.private_cells >= private_cells):
        return a
    if a is not None:
        torch.cuda.synchronize(device)
        _SENSOR_ARENAS.pop(key)
        del a
        torch.cuda.empty_cache()
    a = _SENSOR_ARENAS[key] = SensorArena(torch, device, npo, static_cells, n_static, private_cells)
    return a


class SensorLease:
    """One CCD's part of a SensorArena: static region `static_index` (from cell `static_offset`) and the private cells
    [private_offset, private_offset + private_cells)."""

    def __init__(self, arena, static_index, private_offset, private_cells, static_waits):
        self.arena, self.static_index = arena, static_index
        self.static_offset = static_index * arena.static_cells
        self.private_offset, self.private_cells = int(private_offset), int(private_cells)
        self.static_waits = static_waits              # the stream that initialises the static region waits for these first
        self._static_held, self._held = True, True

    def release_static(self, streams):
        """the static region may be rewritten once everything queued so far on `streams` has run (the CCD's front is enqueued)"""
        if not self._static_held:
            return
        t = self.arena.torch
        evs = []
        for st in streams:
            ev = t.cuda.Event()
            ev.record(st)
            evs.append(ev)
        with self.arena._lock:
            self.arena._static_busy[self.static_index].extend(evs)
        self._static_held = False

    def release(self, streams=None):
        """give the private cells back: the caller knows that everything of this CCD has run (focal_plane collects a CCD behind the
        event of its image copy), or names the streams whose queued work must be waited for first"""
        if not self._held:
            return
        if self._static_held:
            t = self.arena.torch
            self.release_static(streams if streams is not None else [t.cuda.current_stream(self.arena.device)])
        if streams is not None:
            for st in streams:
                st.synchronize()
        self._held = False
        self.arena._give_back(self.private_offset, self.private_cells)



# What the launch planner reads of an object, for tables whose 256-byte rows live on the device (device_table.DeviceTable):
# `row` is the index into the master table, the other fields carry the names they have in OBJECT_DTYPE.
SLIM_DTYPE = np.dtype([("row", "<i8"), ("phot_first", "<i8"), ("n_phot", "<i8"), ("flags", "<i4"), ("stamp_xmin", "<i4"),
                       ("stamp_xmax", "<i4"), ("stamp_ymin", "<i4"), ("stamp_ymax", "<i4"), ("bf_state", "<i4")])


def slim_view(table, select=None):
    """SLIM_DTYPE rows of a DeviceTable (all objects with photons, or the given indices)"""
    idx = np.flatnonzero(table.n_phot > 0) if select is None else np.asarray(select, dtype=np.int64)
    out = np.zeros(len(idx), dtype=SLIM_DTYPE)
    out["row"], out["n_phot"] = idx, table.n_phot[idx]
    out["flags"] = np.where(table.faint[idx], _abi.IMS_OBJ_FAINT, 0)
    st = table.stamp[idx]
    out["stamp_xmin"], out["stamp_xmax"], out["stamp_ymin"], out["stamp_ymax"] = st[:, 0], st[:, 1], st[:, 2], st[:, 3]
    return out


def segment_prefix(n_phot, seg_size):
    segs = (np.asarray(n_phot, dtype=np.int64) + seg_size - 1) // seg_size
    return np.concatenate([[0], np.cumsum(segs)]).astype(np.int64)


def _seg_ptr(pre_t):
    t = getattr(pre_t, "seg_object", None)
    return t.data_ptr() if t is not None else None


class _LibDerive:
    def __init__(self, lib):
        self.lib = lib

    def fill_derived_op(self, op_ref):
        _abi.check(self.lib.ims_fill_derived_op(op_ref), "ims_fill_derived_op")

    def fill_derived_medium(self, kind, c):
        _abi.check(self.lib.ims_fill_derived_medium(int(kind), c), "ims_fill_derived_medium")

    def fill_derived_struct(self, what, struct):
        """what: "optics" | "atmosphere" | "sensor" -- the launch-wide constants of include/imsim_hip.h marked "derived" """
        fn = getattr(self.lib, "ims_fill_derived_" + what)
        _abi.check(fn(C.byref(struct)), "ims_fill_derived_" + what)


def update_window_table(m):
    """ims_sensor_t.bf_dl: the Silicon model's vertex displacements re-ordered for updatePixelDistortions with qdist 3 --
    [dj + 3][di + 3][owned point][x, y] for the 8 x 8 source window dj, di = -3 .. 4 of an owner cell (the cell owns its
    bottom row of points incl. both corners and its left-edge points)."""
    nV = int(m.num_vertices)
    npo = 2 * nV + 2
    n = np.arange(npo)
    vtx = np.where(n <= nV + 1, n, 3 * nV + 4 + (nV - 1 - (n - nV - 2)))
    d = np.asarray(m.distortions, dtype=np.float64).reshape(m.nx, m.ny, 4 * nV + 4, 2)
    cx, cy = (m.nx - 1) // 2, (m.ny - 1) // 2
    out = np.zeros((8, 8, npo, 2))
    for a in range(8):
        for b in range(8):
            out[a, b] = d[b - 3 + cx, a - 3 + cy, vtx, :]
    return out


def optics_layout(opt):
    """ims_render_params_t.optics_layout of an _abi.Optics descriptor: (kind, shape) of every surface as a 4-bit code, first
    surface in the lowest nibble (code = 1 + 3 kc + shape; kc 0 mirror, 1 refracting, 2 detector / baffle; shape 0 plane,
    1 conic with R != 0, 2 conic with R != 0 and asphere terms).  0 when the descriptor cannot be coded (more than 15
    surfaces, asphere terms on a surface with R == 0): the kernels then loop over the surfaces."""
    if opt is None or opt.n_surfaces > 15:
        return 0
    code = 0
    for k in range(opt.n_surfaces):
        S = opt.surf[k]
        kc = {_abi.IMS_SURF_MIRROR: 0, _abi.IMS_SURF_REFRACT: 1}.get(S.kind, 2)
        if S.R == 0.0 and S.n_asphere == 0:
            shape = 0
        elif S.R != 0.0:
            shape = 2 if S.n_asphere > 0 else 1
        else:
            return 0
        code |= (1 + 3 * kc + shape) << (4 * k)
    return code


def lazy_static_applies(scene):
    """Renderer(lazy_static=True) can leave slot 0 without state for this scene: Silicon with 4 vertices per edge (the kernels that
    evaluate polygons from the closed form are written for it), ONE static slot that is not a live region (LSST_Image mode)"""
    ss = getattr(scene, "sensor", None)
    return bool(ss is not None and ss.slots is not None and len(ss.slots) == 1 and not scene.track_static_delta
                and ss.model.num_vertices == 4)


def treering_displacement_bound(ss):
    """Upper bound of |tree-ring shift| over the CCD [pixels]: on every table interval the interpolant is the chord plus
    the cubic-spline term ((a^3 - a) m0 + (b^3 - b) m1) h^2 / 6 with |a^3 - a| <= 2 / (3 sqrt 3); a few ulp on top."""
    if ss.tr_table is None:
        return 0.0
    v = np.abs(np.asarray(ss.tr_table, dtype=np.float64))
    bound = np.maximum(v[:-1], v[1:])
    if ss.tr_table2 is not None:
        m = np.abs(np.asarray(ss.tr_table2, dtype=np.float64))
        bound = bound + 0.3849002 * (m[:-1] + m[1:]) * float(ss.tr_dr) ** 2 / 6.0
    return float(bound.max()) * (1.0 + 1e-9) + 1e-12


def image_profile_cdf(img):
    """Cumulative distribution over the pixels of an image profile (row-major, negative pixels count as empty):
    w*h + 1 knots from 0 to 1 (ims_image_tables_t.cdf)."""
    v = np.clip(np.asarray(img, dtype=np.float64), 0.0, None).reshape(-1)
    tot = v.sum()
    if not tot > 0.0:
        raise ValueError("image profile has no positive pixel")
    cdf = np.concatenate([[0.0], np.cumsum(v) / tot])
    cdf[-1] = 1.0
    return cdf


def quintic_kernel(x):
    """GalSim's Quintic interpolant (Bernstein & Gruen 2014, the default x_interpolant of galsim.InterpolatedImage): the
    piecewise quintic on [-3, 3] that interpolates, is C1 and reproduces polynomials up to degree four."""
    x = np.abs(np.asarray(x, dtype=np.float64))
    out = np.zeros_like(x)
    a, b, c = x <= 1.0, (x > 1.0) & (x <= 2.0), (x > 2.0) & (x <= 3.0)
    out[a] = 1.0 + x[a] ** 3 * (-95.0 + 138.0 * x[a] - 55.0 * x[a] ** 2) / 12.0
    out[b] = (x[b] - 1.0) * (x[b] - 2.0) * (-138.0 + 348.0 * x[b] - 249.0 * x[b] ** 2 + 55.0 * x[b] ** 3) / 24.0
    out[c] = (x[c] - 2.0) * (x[c] - 3.0) ** 2 * (-54.0 + 50.0 * x[c] - 11.0 * x[c] ** 2) / 24.0
    return out


QUINTIC_NEGATIVE = (1.0, 2.0, (25.0 + math.sqrt(31.0)) / 11.0, 3.0)     # |x| intervals with K < 0 (GalSim: Interpolant.cpp, Quintic)


def interpolant_cdf(kernel=quintic_kernel, negative=QUINTIC_NEGATIVE, support=3.0, per_piece=128):
    """(kx, kcdf, norm) of ims_image_tables_t: the cumulative distribution of |K| -- what Interpolant::shoot of GalSim samples
    with a OneDimensionalDeviate -- at knots that divide every piece between two sign changes of K (and the integers) into
    per_piece equal intervals, so that no interval holds both signs and the mass of every interval is exact (Gauss-Legendre
    with 8 points per interval: exact for the piecewise quintic); norm = (integral |K|)^2."""
    cuts = sorted({0.0, support} | {float(v) for v in negative if 0.0 < v < support} | {float(k) for k in range(1, int(support))})
    half = np.concatenate([np.linspace(a, b, per_piece + 1)[:-1] for a, b in zip(cuts[:-1], cuts[1:])] + [[support]])
    kx = np.concatenate([-half[:0:-1], half])
    gx, gw = np.polynomial.legendre.leggauss(8)
    a, b = kx[:-1], kx[1:]
    xm = 0.5 * (a + b)[:, None] + 0.5 * (b - a)[:, None] * gx[None, :]
    mass = 0.5 * (b - a) * (np.abs(kernel(xm.reshape(-1))).reshape(xm.shape) * gw[None, :]).sum(axis=1)
    total = float(mass.sum())
    kcdf = np.concatenate([[0.0], np.cumsum(mass) / total])
    kcdf[-1] = 1.0
    return np.ascontiguousarray(kx), np.ascontiguousarray(kcdf), total * total


class _Arena:
    """Host staging of the many small tables of one launch plan (object rows, segment prefixes, pool offsets) for ONE
    upload: `add` returns the byte offset of an array, `patch` remembers a struct field that must receive its device
    address, `ref` hands out an object with the data_ptr() of a tensor; `upload` copies everything at once and fills
    the addresses in."""

    class Ref:
        def __init__(self, arena, off):
            self.arena, self.off = arena, off

        def data_ptr(self):
            return self.arena.base + self.off

    def __init__(self):
        self.parts, self.size, self.patches, self.base, self.tensor = [], 0, [], None, None

    def add(self, arr):
        a = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
        off = (self.size + 255) & ~255
        self.parts.append((off, a))
        self.size = off + a.size
        return off

    def patch(self, struct, field, off):
        self.patches.append((struct, field, off))

    def ref(self, arr):
        return _Arena.Ref(self, self.add(arr))

    def upload(self, mem):
        buf = np.zeros(max(self.size, 8), dtype=np.uint8)
        for off, a in self.parts:
            buf[off:off + a.size] = a
        self.tensor, self.base = mem.put(buf)
        for struct, field, off in self.patches:
            setattr(struct, field, self.base + off)
        self.parts = []
        return self.tensor


class PlanList(list):
    """a launch plan; `arena` keeps the device copy of its tables alive"""
    arena = None


class BoundScene:
    """A Scene whose tables live behind pointers of one memory provider; builds RenderParams."""

    def __init__(self, scene: Scene, mem, derive, lease=None):
        """derive: the library whose fill_derived_op / fill_derived_medium entry points complete the
        launch-wide derived fields (libimsim_hip.so for the product, the oracle for the checker).
        lease: a SensorLease -- the pixel-boundary state lives in the device's SensorArena instead of arrays of its own."""
        self.scene = scene
        self.derive = derive
        self.mem = mem
        self.lease = lease
        P = RenderParams()
        P.seed = scene.seed
        P.seg_size = scene.seg_size
        P.track_static_delta = scene.track_static_delta
        if len(scene.psf) > _abi.IMS_MAX_PSF:
            raise ValueError("too many PSF components")
        P.n_psf = len(scene.psf)
        for k, comp in enumerate(scene.psf):                # (kind, table, p0, chrom_alpha, chrom_base[, p1, p2])
            P.psf[k] = PsfComponent(*comp)
        if len(scene.ops) > _abi.IMS_MAX_OPS:
            raise ValueError("too many photon ops")
        P.n_ops = len(scene.ops)
        for k, (kind, table, p) in enumerate(scene.ops):
            pp = list(p) + [0.0] * (8 - len(p))
            P.ops[k] = Op(kind, table, (C.c_double * 8)(*pp))
            derive.fill_derived_op(C.byref(P.ops[k]))
        if scene.image_profiles:
            sizes, offs, cdfs, pos = [], [], [], 0
            for img in scene.image_profiles:
                cdf = image_profile_cdf(img)
                sizes.append((img.shape[1], img.shape[0]))
                offs.append(pos)
                cdfs.append(cdf)
                pos += len(cdf)
            P.images.n_images = len(sizes)
            _, P.images.size = mem.put(np.asarray(sizes), np.int32)
            _, P.images.offset = mem.put(np.asarray(offs), np.int64)
            _, P.images.cdf = mem.put(np.concatenate(cdfs), np.float64)
            if scene.image_interpolant == "quintic":
                kx, kcdf, norm = interpolant_cdf()
                P.images.interp, P.images.n_k, P.images.norm = 1, len(kx) - 1, norm
                P.images.neg = (C.c_double * 4)(*QUINTIC_NEGATIVE)
                _, P.images.kx = mem.put(kx, np.float64)
                _, P.images.kcdf = mem.put(kcdf, np.float64)
            elif scene.image_interpolant != "nearest":
                raise ValueError(f"image_interpolant {scene.image_interpolant!r}: 'quintic' or 'nearest'")
        if scene.radial_r2 is not None:
            r2 = np.atleast_2d(scene.radial_r2)
            P.radial.n_tables, P.radial.n_bins = r2.shape[0], r2.shape[1] - 1
            _, P.radial.r2 = mem.put(r2, np.float64)
            cdf = np.atleast_2d(scene.radial_cdf)
            _, P.radial.cdf = mem.put(cdf, np.float64)
            # guide table of the inverse-CDF bin search (ims_radial_tables_t.guide): last knot with cdf <= g / n_guide
            n_guide = 512
            grid = np.arange(n_guide + 1) / n_guide
            guide = np.stack([np.clip(np.searchsorted(row, grid, side="right") - 1, 0, len(row) - 2) for row in cdf])
            _, P.radial.guide = mem.put(guide, np.int32)
            P.radial.n_guide = n_guide
        if scene.sed_tables is not None:
            t = np.atleast_2d(scene.sed_tables)
            P.sed.n_tables, P.sed.n_pts = t.shape
            P.sed.arg_min, P.sed.arg_step = 0.0, 1.0 / (t.shape[1] - 1)
            _, P.sed.val = mem.put(t, np.float64)
        if scene.ratio_tables is not None:
            t = np.atleast_2d(scene.ratio_tables)
            P.ratio.n_tables, P.ratio.n_pts = t.shape
            P.ratio.arg_min, P.ratio.arg_step = scene.ratio_wl_min, scene.ratio_wl_step
            _, P.ratio.val = mem.put(t, np.float64)
        if scene.optics is not None:
            opt = type(scene.optics).from_buffer_copy(bytes(scene.optics))
            derive.fill_derived_medium(opt.in_medium_kind, opt.in_medium_c)
            for k in range(opt.n_surfaces):
                derive.fill_derived_medium(opt.surf[k].medium_kind, opt.surf[k].medium_c)
            derive.fill_derived_struct("optics", opt)
            _, P.optics = mem.put_struct(opt)
            P.optics_layout = optics_layout(opt)
        if scene.atm is not None:
            A = scene.atm.atmosphere_struct()
            scr = scene.atm.screens
            if isinstance(mem, DeviceMem):
                t = mem.torch
                if isinstance(scr, np.ndarray):
                    scr = getattr(scene.atm, "_screens_dev", None)
                    if scr is None or scr.device != mem.device:
                        scr = t.from_numpy(np.ascontiguousarray(scene.atm.screens, dtype=np.float32)).to(mem.device)
                        scene.atm._screens_dev = scr
                mem.keep.append(scr)
                A.screens = scr.data_ptr()
                # the 2 x 2 cells of the bilinear gradient as 16-byte items (ims_atmosphere_t.screen_quads): built once
                # per atmosphere and shared by every renderer that looks through it (4 x the screens: 6.4 GB for 6 x 8192^2)
                # (with the pre-pass of ims_screen_prepass, the default, the shooting kernels do not gather at all and the
                # pre-pass reads the plain screens: the four-fold table is only built when asked for)
                prepass_all = tuning.env("IMS_SCREEN_PREPASS", "0") == "1"
                if tuning.env("IMS_SCREEN_QUADS", "0" if prepass_all else "1") != "0":
                    quads = getattr(scene.atm, "_screen_quads", None)
                    if quads is None or quads.device != scr.device:
                        quads = t.stack([scr, t.roll(scr, -1, 2), t.roll(scr, -1, 1), t.roll(scr, (-1, -1), (1, 2))], dim=-1).contiguous()
                        scene.atm._screen_quads = quads
                    mem.keep.append(quads)
                    A.screen_quads = quads.data_ptr()
            elif isinstance(scr, np.ndarray):
                _, A.screens = mem.put(scr, np.float32)
            else:
                _, A.screens = mem.put(scr.cpu().numpy(), np.float32)
            derive.fill_derived_struct("atmosphere", A)
            self.atm_struct = A
            _, P.atm = mem.put_struct(A)
        self.sensor_host = None
        self.sensor_arrays = {}
        if scene.sensor is not None:
            self._bind_sensor(P, scene.sensor)
        P.nx, P.ny, P.xmin, P.ymin = scene.nx, scene.ny, scene.xmin, scene.ymin
        self.base_params = P

    def _bind_sensor(self, P, ss: SensorSetup):
        m = ss.model
        S = Sensor()
        S.kind = _abi.IMS_SENSOR_SILICON
        S.num_vertices, S.nx, S.ny, S.qdist = m.num_vertices, m.nx, m.ny, m.qdist
        S.num_elec, S.pixel_size, S.thickness, S.diff_step = m.num_elec, m.pixel_size, m.thickness, m.diff_step
        S.n_abs = len(ss.abs_len)
        S.abs_wl_min = float(ss.abs_wl[0])
        S.abs_wl_step = float(ss.abs_wl[1] - ss.abs_wl[0])
        _, S.abs_len = self.mem.put(ss.abs_len, np.float64)
        if ss.tr_table is not None:
            S.n_tr, S.tr_dr = len(ss.tr_table), ss.tr_dr
            S.tr_cx, S.tr_cy = float(ss.tr_center[0]), float(ss.tr_center[1])
            _, S.tr_table = self.mem.put(ss.tr_table, np.float64)
            if ss.tr_table2 is not None:
                _, S.tr_table2 = self.mem.put(ss.tr_table2, np.float64)
        else:
            S.n_tr, S.tr_dr = 0, 1.0
        _, S.distortions = self.mem.put(m.distortions, np.float64)
        if m.qdist == 3:
            _, S.bf_dl = self.mem.put(update_window_table(m), np.float64)
        _, S.emptypoly = self.mem.put(m.emptypoly, np.float64)
        slots = ss.slots if ss.slots is not None else make_slots([])
        self.n_static_slots = len(slots)
        # static_cells: the cell offset at which the private regions of bright objects begin (= the cells of the static slots
        # when the state is this renderer's own); scratch_cells: the owner cells available for them
        self.static_cells = ss.total_cells()
        self.scratch_cells = int(ss.scratch_cells)
        self.slot_capacity = max(len(slots), ss.max_slots)
        S.n_bf_slots = len(slots)
        lease = self.lease
        if lease is not None:
            if len(slots) != 1 or self.static_cells != lease.arena.static_cells or lease.arena.npo != ss.owned_points():
                raise ValueError("SensorLease: the arena was sized for another CCD geometry / sensor model")
            slots = slots.copy()
            slots["offset"] += lease.static_offset
            self.static_cells = lease.private_offset
            self.scratch_cells = lease.private_cells
        # LSST_Image mode never updates slot 0 (objects only distort their private regions): a rigorous bound of the
        # tree-ring displacement lets photons far from every pixel edge skip the boundary state (ims_sensor_t.pristine_margin)
        S.pristine_margin = -1.0
        if len(slots) > 0 and not self.scene.track_static_delta:
            S.pristine_margin = treering_displacement_bound(ss)
        cells = self.static_cells + self.scratch_cells
        npo = ss.owned_points()
        slots_host = np.zeros(self.slot_capacity, dtype=BFSLOT_DTYPE)
        slots_host[:len(slots)] = slots
        self._slots_buf, S.bf_slots = self.mem.put(slots_host.view(np.uint8))
        if lease is not None:
            # the arena's arrays: every region of this CCD is written by its init_boundaries (points, bounds line, zero delta)
            # before anything reads it; only the flag bytes of the leased cells are cleared (queued on the current stream)
            A = lease.arena
            for name in ("boundary", "bounds", "delta", "tile_charge", "tile_changed"):
                self.sensor_arrays[name] = getattr(A, name)
            S.bf_boundary, S.bf_bounds, S.bf_delta = A.boundary.data_ptr(), A.bounds.data_ptr(), A.delta.data_ptr()
            S.bf_tile_charge, S.bf_tile_changed = A.tile_charge.data_ptr(), A.tile_changed.data_ptr()
            for flags in (A.tile_charge, A.tile_changed, A.changed):
                flags[lease.private_offset:lease.private_offset + lease.private_cells].zero_()
        else:
            # (the points of the static slots -- 2.7 GB for a 4k x 4k CCD -- are written, every one of them, by the init_boundaries
            # that follows the construction of a renderer: no zero fill of that part)
            self.sensor_arrays["boundary"], S.bf_boundary = self.mem.zeros(cells * npo * 2, np.float64,
                                                                           uninitialised=self.static_cells * npo * 2)
            self.sensor_arrays["bounds"], S.bf_bounds = self.mem.zeros(cells * 8, np.float64)
            self.sensor_arrays["delta"], S.bf_delta = self.mem.zeros(cells, np.float64)
            # tile flags of the brighter-fatter rounds (one byte per owner cell, used at tile origins)
            self.sensor_arrays["tile_charge"], S.bf_tile_charge = self.mem.zeros(cells, np.uint8)
            self.sensor_arrays["tile_changed"], S.bf_tile_changed = self.mem.zeros(cells, np.uint8)
        self.derive.fill_derived_struct("sensor", S)
        # host copy whose slot table is a host pointer (sizes the launches)
        Sh = Sensor.from_buffer_copy(bytes(S))
        self._slots_host = slots_host
        Sh.bf_slots = slots_host.ctypes.data
        self.sensor_host = Sh
        self.sensor_struct = S
        self._sensor_buf, P.sensor = self.mem.put_struct(S)
        self.sensor_dev_ptr = P.sensor

    def set_private_slots(self, slots):
        """Replace the private (non-static) part of the slot table."""
        n0 = self.n_static_slots
        if n0 + len(slots) > self.slot_capacity:
            raise ValueError("too many brighter-fatter slots")
        self._slots_host[n0:n0 + len(slots)] = slots
        self.mem.write(self._slots_buf, self._slots_host.view(np.uint8).reshape(-1))
        self.sensor_struct.n_bf_slots = n0 + len(slots)
        self.sensor_host.n_bf_slots = n0 + len(slots)
        if isinstance(self._sensor_buf, Sensor):
            self._sensor_buf.n_bf_slots = n0 + len(slots)
        else:
            self.mem.write(self._sensor_buf, np.frombuffer(bytes(self.sensor_struct), dtype=np.uint8).copy())

    def params(self, objects_ptr, n_objects, seg_prefix_ptr, n_segments, image_ptr, realized_ptr=None, seg_object_ptr=None, lazy=False):
        """lazy: the launch is a fused LSST_Image render of a renderer whose slot 0 holds no state (Renderer.lazy_static).  The flag
        lives on the parameters of THOSE launches only -- never in base_params, where every pool / sensor entry point would inherit
        it and read the state that was not made."""
        P = RenderParams.from_buffer_copy(bytes(self.base_params))
        P.lazy_static = 1 if lazy else 0
        P.seg_object = seg_object_ptr
        P.objects, P.n_objects = objects_ptr, n_objects
        P.seg_prefix, P.n_segments = seg_prefix_ptr, n_segments
        P.image = image_ptr
        P.realized_flux = realized_ptr
        return P


class PhotonPool:
    """Device photon pool (the fields of galsim.PhotonArray) + the object offsets of the sub-batch."""
    FIELDS = ("x", "y", "flux", "dxdz", "dydz", "wavelength", "pupil_u", "pupil_v", "time")

    def __init__(self, torch, device, n, photon_offset_dev, objects_dev, n_objects, seg_prefix_dev, n_segments):
        self.n = int(n)
        self.t = {f: torch.empty(max(self.n, 1), dtype=torch.float64, device=device) for f in self.FIELDS}
        self.obj_index = torch.empty(max(self.n, 1), dtype=torch.int32, device=device)
        self.photon_offset_dev = photon_offset_dev
        self.objects_dev, self.n_objects = objects_dev, n_objects
        self.seg_prefix_dev, self.n_segments = seg_prefix_dev, n_segments

    def struct(self):
        ph = Photons()
        ph.n = self.n
        for f in self.FIELDS:
            setattr(ph, f, self.t[f].data_ptr())
        ph.obj_index = self.obj_index.data_ptr()
        return ph

    def to_host(self):
        out = {f: self.t[f][:self.n].cpu().numpy() for f in self.FIELDS}
        out["obj_index"] = self.obj_index[:self.n].cpu().numpy()
        return out


_DEVICE_STREAMS = {}


def _device_streams(torch, device):
    """The five streams of the launch plans (chain, bulk, chain1, chain2, chain3; the last only with three thresholds in
    IMS_CHAIN_CLASSES -- measured slower), ONE set per device and process, shared by every
    Renderer on that device.  HIP multiplexes its streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default):
    with a fresh set per renderer the chain of one CCD of a focal plane lands in the hardware queue of another CCD's wide
    launches and waits behind them (kernel trace of C5: one kernel in flight for two thirds of the time).  With one set,
    the plans of the CCDs in flight interleave role by role: the wide launches of the next CCD fill the GPU while the
    latency-bound chain of the previous one runs."""
    pr = [int(v) for v in tuning.env("IMS_STREAM_PRIORITIES", "-1,0,0,0,0").split(",")]   # chain, bulk, chain1, chain2, chain3
    pr = (pr + [0] * 5)[:5]
    # (a set per renderer, and two or three sets handed out in turn, were measured in round 2 and are gone: slower)
    key = (str(device), tuple(pr))
    st = _DEVICE_STREAMS.get(key)
    if st is None:
        st = _DEVICE_STREAMS[key] = tuple(torch.cuda.Stream(device, priority=p) for p in pr)
    return st


def _pool_stream(torch, device):
    """the stream of the overlapped pool shoot of photon-pooling mode (Renderer._overlapped_pool): one per device and process"""
    key = ("pool-shoot", str(device))
    with _STREAMS_LOCK:
        st = _DEVICE_STREAMS.get(key)
        if st is None:
            st = _DEVICE_STREAMS[key] = torch.cuda.Stream(device)
    return st


_STREAMS_LOCK = threading.Lock()
EVENT_BLOCK = 128                                 # library event numbers of one renderer (a plan uses one per pool slice: a few dozen)
_EVENT_BLOCKS = list(range(459, -1, -1))          # blocks below Renderer.PREPASS_EVENT (60 000)
_EVENT_BLOCKS_LOCK = threading.Lock()


def _take_event_block():
    with _EVENT_BLOCKS_LOCK:
        if not _EVENT_BLOCKS:
            raise _abi.ImsimHipError("more than 460 renderers alive: no block of library event numbers left (drop renderers that are done)")
        return _EVENT_BLOCKS.pop() * EVENT_BLOCK


def _give_event_block(first):
    with _EVENT_BLOCKS_LOCK:
        _EVENT_BLOCKS.append(int(first) // EVENT_BLOCK)


def _focal_streams(torch, device, peek=False, top_index=None):
    """Plan streams of a renderer that is one CCD of a focal plane (Renderer(..., stream_roles="focal")): four streams per
    device for ALL its renderers -- two for the long top chains, taken in turn, one for the wide launches, one for the
    middle and low chain classes -- in the order of Renderer.STREAMS.  A CCD of 10 k sources keeps the GPU busy for 3 - 4 ms
    but needs ~10 ms for the 100 - 200 dependent rounds of its brightest star; with one stream set for every renderer
    (_device_streams) those chains queue behind each other, with a set per renderer eight streams share four hardware queues
    and a chain sits behind another CCD's wide launches.  Here the top chains of two CCDs advance side by side on queues
    that hold nothing else."""
    key = ("focal", str(device))
    with _STREAMS_LOCK:                 # renderers of a focal plane may be created from several host threads
        st = _DEVICE_STREAMS.get(key)
        if st is None:
            # IMS_FOCAL_TOPS (default 2): streams for the long top chains; more than two only pays with more hardware queues
            # (GPU_MAX_HW_QUEUES) than HIP's default four
            n_top = max(1, int(tuning.env("IMS_FOCAL_TOPS", "2")))
            # ("pre": the joint path's stream of a CCD's FFT draws, initial states and first pool slices -- wide work at normal
            # priority, focal_plane._render_joint)
            st = _DEVICE_STREAMS[key] = {"next": 0, "top": [torch.cuda.Stream(device, priority=-1) for _ in range(n_top)],
                                         "bulk": torch.cuda.Stream(device), "mid": torch.cuda.Stream(device),
                                         "pre": torch.cuda.Stream(device)}
            order = tuning.env("IMS_FOCAL_TOUCH")
            if order:
                # the order in which the role streams are first used (HIP binds a stream to one of its hardware queues then:
                # queue = index of first use mod 4, EXPERIMENTS round 5)
                for name in order.split(","):
                    s = (torch.cuda.default_stream(device) if name == "null" else
                         st["top"][int(name[3:]) % n_top] if name.startswith("top") else st[name])
                    with torch.cuda.stream(s):
                        torch.zeros(1, device=device)
                    s.synchronize()
    n_top = len(st["top"])
    if top_index is not None:          # the caller deals the top streams itself (CCDs enqueued from several host threads)
        return (st["top"][top_index % n_top], st["bulk"], st["mid"], st["mid"], st["mid"])
    top = st["top"][st["next"] % n_top]
    if not peek:                       # peek: the set the NEXT renderer of the device will get
        st["next"] += 1
    return (top, st["bulk"], st["mid"], st["mid"], st["mid"])


_PINNED_ARENAS = []            # (page-locked tensor, events after the last run that copies out of it)


_PINNED_LOCK = threading.Lock()


def _pinned_arena(torch, nbytes):
    """A page-locked buffer of at least nbytes out of a process-wide pool (allocating one costs milliseconds); buffers come
    back through NativePlan.__del__ behind the events of the streams that may still copy out of them.  The smallest buffer
    that fits is taken (small uploads must not eat the plan arenas)."""
    with _PINNED_LOCK:
        best = None
        for k, (t, ev) in enumerate(_PINNED_ARENAS):
            if t.numel() >= nbytes and (best is None or t.numel() < _PINNED_ARENAS[best][0].numel()) and ev.query():
                best = k
        if best is not None:
            return _PINNED_ARENAS.pop(best)[0]
    return torch.empty(max(int(nbytes * 1.25), 1 << 16), dtype=torch.uint8, pin_memory=True)


def upload_async(torch, device, arr):
    """numpy array -> device tensor of the same dtype and shape, ASYNCHRONOUSLY on the current stream through a page-locked
    buffer of the process-wide pool.  `torch.from_numpy(a).to(device)` of pageable memory is a synchronous copy: the host waits
    until everything queued on the current stream before it is through -- on a stream shared by the CCDs of a focal plane that
    is the previous CCD's work (measured: 1.8 ms per call, 6 ms of a CCD's 12 ms of host time)."""
    a = np.ascontiguousarray(arr)
    if tuning.env("IMS_UPLOAD_SYNC", "0") == "1":          # the synchronous copy, for comparison
        return torch.from_numpy(a).to(device)
    raw = a.view(np.uint8).reshape(-1)
    n = int(raw.size)
    dev = torch.empty(max(n, 8), dtype=torch.uint8, device=device)
    if n:
        pin = _pinned_arena(torch, n)
        pin.numpy()[:n] = raw
        dev[:n].copy_(pin[:n], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        with _PINNED_LOCK:
            _PINNED_ARENAS.append((pin, ev))
    out = dev[:n]
    if a.dtype != np.uint8:
        out = out.view(getattr(torch, str(a.dtype)))
    return out.view(a.shape) if a.ndim != 1 else out


def plan_input(r, n_phot, stamp, faint, nrecalc=None, want_realized=False, coarse_slices=False):
    """ims_plan_input_t of one LSST_Image render on renderer r (the arrays must stay alive while the struct is used)"""
    ss = r.scene.sensor
    inp = _abi.PlanInput()
    inp.n = len(n_phot)
    inp.n_phot, inp.stamp, inp.faint = n_phot.ctypes.data, stamp.ctypes.data, faint.ctypes.data
    b = r.bound
    if ss is not None:
        inp.nrecalc = int(ss.model.nrecalc if nrecalc is None else nrecalc)
        inp.n_static_slots, inp.slot_capacity = b.n_static_slots, b.slot_capacity
        inp.static_cells, inp.scratch_cells = b.static_cells, int(b.scratch_cells)
    thresholds = list(r.chain_class_rounds)[:3]
    inp.n_class_rounds = len(thresholds)
    for k, v in enumerate(thresholds):
        inp.class_rounds[k] = int(v)
    inp.max_pool_photons = int(r.max_pool_photons)
    inp.seg_size, inp.want_realized = int(r.scene.seg_size), 1 if want_realized else 0
    inp.event_base, inp.use_tags = int(r._event_block), 1 if r.use_bf_tags else 0
    inp.coarse_slices = 1 if coarse_slices else 0
    return inp


def plan_sizes(r, objects, nrecalc=None):
    """ims_plan_sizes_t of the plan renderer r would build for an OBJECT_DTYPE host table -- host code only (launch counts,
    photons and rows per launch kind: what bench.py prices a focal plane's kernels with)"""
    objects = np.ascontiguousarray(objects, dtype=OBJECT_DTYPE)
    n_phot = np.ascontiguousarray(objects["n_phot"], dtype=np.int64)
    stamp = np.stack([objects["stamp_xmin"], objects["stamp_xmax"], objects["stamp_ymin"], objects["stamp_ymax"]],
                     axis=1).astype(np.int32) if len(objects) else np.zeros((0, 4), dtype=np.int32)
    faint = ((objects["flags"] & _abi.IMS_OBJ_FAINT) != 0).astype(np.uint8)
    inp = plan_input(r, n_phot, stamp, faint, nrecalc)
    handle, sizes = C.c_void_p(), _abi.PlanSizes()
    _abi.check(r.lib.ims_plan_lsst_image(C.byref(inp), C.byref(handle), C.byref(sizes)), "ims_plan_lsst_image")
    r.lib.ims_plan_destroy(handle)
    return sizes


class NativePlan:
    """The launch plan of one LSST_Image render, built (ims_plan_lsst_image), bound to memory (ims_plan_bind), uploaded
    (ims_plan_upload) and run (ims_plan_run) by the library.  torch provides the memory: a page-locked arena and its device copy,
    the gathered launch tables, the converted photon pool of the bright objects.  `objects`: an OBJECT_DTYPE host table (uploaded
    once as the master table) or a device_table.DeviceTable."""

    def __init__(self, renderer, objects, nrecalc=None, want_realized=False, coarse_slices=False):
        r = renderer
        self._renderer = weakref.ref(renderer)     # the renderer keeps its last plan: no cycle (its HBM must go when it is dropped)
        self._lib, self._torch, self._device, self._streams = r.lib, r.torch, r.device, tuple(r.plan_streams)
        t = r.torch
        ss = r.scene.sensor
        lib = r.lib
        if hasattr(objects, "rows") and hasattr(objects, "n_phot") and not isinstance(objects, np.ndarray):
            self.master = objects.rows                        # DeviceTable: the rows are on the device already
            n_phot = np.ascontiguousarray(objects.n_phot, dtype=np.int64)
            stamp = np.ascontiguousarray(objects.stamp, dtype=np.int32)
            faint = np.ascontiguousarray(objects.faint, dtype=np.uint8)
            self._keep_table = objects
        else:
            objects = np.ascontiguousarray(objects, dtype=OBJECT_DTYPE)
            self.master = upload_async(t, r.device, objects.view(np.uint8).reshape(-1)) if len(objects) else \
                t.zeros(OBJECT_DTYPE.itemsize, dtype=t.uint8, device=r.device)
            n_phot = np.ascontiguousarray(objects["n_phot"], dtype=np.int64)
            stamp = np.stack([objects["stamp_xmin"], objects["stamp_xmax"], objects["stamp_ymin"], objects["stamp_ymax"]],
                             axis=1).astype(np.int32) if len(objects) else np.zeros((0, 4), dtype=np.int32)
            faint = ((objects["flags"] & _abi.IMS_OBJ_FAINT) != 0).astype(np.uint8)
        self.n_master = len(n_phot)
        inp = plan_input(r, n_phot, stamp, faint, nrecalc, want_realized, coarse_slices)
        b = r.bound
        handle, sizes = C.c_void_p(), _abi.PlanSizes()
        _abi.check(lib.ims_plan_lsst_image(C.byref(inp), C.byref(handle), C.byref(sizes)), "ims_plan_lsst_image")
        if sizes.n_events > EVENT_BLOCK:
            lib.ims_plan_destroy(handle)
            raise ValueError(f"a plan may use at most {EVENT_BLOCK} library events (its renderer's block); this one needs {sizes.n_events}")
        self.handle, self.sizes = handle, sizes
        self.arena_pin = _pinned_arena(t, max(int(sizes.arena_bytes), 256))
        self.arena_dev = t.empty(max(int(sizes.arena_bytes), 256), dtype=t.uint8, device=r.device)
        self.rows = t.empty(int(sizes.rows_bytes), dtype=t.uint8, device=r.device)
        self.pool = t.empty(max(4 * int(sizes.pool_photons), 1), dtype=t.float64, device=r.device)
        self.realized = t.empty(max(int(sizes.realized_count), 1), dtype=t.float64, device=r.device) if sizes.realized_count else None
        self.P = b.params(None, 0, None, 0, r.image.data_ptr(), None, None, lazy=r.lazy_static)     # (ims_plan_bind keeps the flag on its RENDER launches only)
        _abi.check(lib.ims_plan_bind(handle, C.byref(self.P), self.arena_pin.data_ptr(), self.arena_dev.data_ptr(), self.rows.data_ptr(),
                                     self.master.data_ptr(), self.pool.data_ptr(),
                                     self.realized.data_ptr() if self.realized is not None else None), "ims_plan_bind")
        _abi.check(lib.ims_plan_upload(handle, r._stream()), "ims_plan_upload")
        if ss is not None and not hasattr(r, "_changed"):
            cells = b.static_cells + int(b.scratch_cells)
            r._changed = t.zeros(max(cells, 1), dtype=t.uint8, device=r.device)

    def run(self, defer=False):
        """enqueue the whole render on the renderer's plan streams, joined back into the current stream.
        defer: ims_plan_run_deferred -- no join yet (`join`), and the rounds of the top chain are left to `run_joint_plans` where
        they can run jointly with other CCDs' (self.deferred says whether any were left)"""
        r = self._renderer()
        if r is None:
            raise _abi.ImsimHipError("NativePlan.run: the renderer of this plan is gone")
        tuning.sync_library(r.lib)
        b = r.bound
        streams = r.plan_streams
        sarr = (C.c_void_p * len(streams))(*[st.cuda_stream for st in streams])
        has_sensor = r.scene.sensor is not None
        args = (self.handle, b.sensor_dev_ptr if has_sensor else None, C.byref(b.sensor_host) if has_sensor else None,
                b.sensor_struct.bf_slots if has_sensor else None, r._changed.data_ptr() if has_sensor else None, r._stream(), sarr,
                len(streams), 1 if r._plans_run else 0)
        if defer:
            left = C.c_int32(0)
            _abi.check(r.lib.ims_plan_run_deferred(*args, C.byref(left)), "ims_plan_run_deferred")
            self.deferred = int(left.value)        # chains whose rounds are left to run_joint_plans
            self._keep_renderer = r                # the library holds pointers into its bound scene until the joint run
        else:
            _abi.check(r.lib.ims_plan_run(*args), "ims_plan_run")
        r._plans_run += 1
        if has_sensor:
            b.sensor_struct.n_bf_slots = b.sensor_host.n_bf_slots          # the library moved the slot table on
            if isinstance(b._sensor_buf, Sensor):
                b._sensor_buf.n_bf_slots = b.sensor_host.n_bf_slots

    def join(self, stream=None):
        """after run(defer=True) (and run_joint_plans): `stream` (default: the current one) waits for everything of this plan"""
        st = stream if stream is not None else self._torch.cuda.current_stream(self._device)
        _abi.check(self._lib.ims_plan_join(self.handle, C.c_void_p(st.cuda_stream)), "ims_plan_join")
        self._keep_renderer = None

    def add_realized(self, realized):
        """realized[master row] += flux every object added to the image (base['realized_flux'], stamp.py:573)"""
        st = C.c_void_p(self._torch.cuda.current_stream(self._device).cuda_stream)
        _abi.check(self._lib.ims_plan_add_realized(self.handle, realized.data_ptr(), st), "ims_plan_add_realized")

    def __del__(self):
        try:
            t = self._torch
            if getattr(self, "arena_pin", None) is not None:
                evs = []
                for st in set(list(self._streams) + [t.cuda.current_stream(self._device)]):
                    ev = t.cuda.Event()
                    ev.record(st)
                    evs.append(ev)
                with _PINNED_LOCK:
                    _PINNED_ARENAS.append((self.arena_pin, _AllEvents(evs)))
                self.arena_pin = None
            if getattr(self, "handle", None):
                self._lib.ims_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def run_joint_plans(plans, stream, first_chain=0, n_chains=1):
    """The rounds that run(defer=True) left: the chain classes first_chain .. first_chain + n_chains - 1 (0 = the top class: the
    brightest stars) of the given plans in lockstep on `stream`, three launches per round for all of them (ims_plans_run_joint;
    at most 64 chains per call: longer lists go in several calls)."""
    plans = [p for p in plans if p is not None and getattr(p, "deferred", 0)]
    if not plans:
        return
    lib = plans[0]._lib
    tuning.sync_library(lib)
    per_call = max(1, 64 // max(int(n_chains), 1))
    for a in range(0, len(plans), per_call):
        part = plans[a:a + per_call]
        arr = (C.c_void_p * len(part))(*[p.handle.value for p in part])
        _abi.check(lib.ims_plans_run_joint(arr, len(part), C.c_void_p(stream.cuda_stream), int(first_chain), int(n_chains)),
                   "ims_plans_run_joint")


class Renderer:
    """One CCD on one GPU."""

    STREAMS = {"chain": 0, "bulk": 1, "chain1": 2, "chain2": 3, "chain3": 4}
    CHAIN_STREAMS = ("chain", "chain1", "chain2", "chain3")

    def __init__(self, scene: Scene, device="cuda:0", stream_roles="single", top_index=None, lease=None, lazy_static=None):
        """lease: a SensorLease of the device's SensorArena (focal planes): the pixel-boundary state of this CCD lives there; the
        current stream (which initialises the static region) first waits for the region's previous readers.
        lazy_static: the caller renders in LSST_Image mode and reads slot 0 through nothing but the fused launch of the ordinary
        objects -- the static state is then not made at all (ims_render_params_t.lazy_static: the ~2 % of the photons that would
        look at it are finished by a second launch from the tree-ring closed form; same bits).  Taken only where it applies
        (Silicon with 4 vertices per edge, slot 0 not a live region, one static slot)."""
        self.lib = _abi.load()
        tuning.sync_library(self.lib)            # the library's choice of kernel forms (ims_tuning_t) follows the environment
        self.mem = DeviceMem(device)
        self.torch = self.mem.torch
        self.device = self.mem.device
        self.torch.cuda.set_device(self.device)
        self.scene = scene
        self.lease = lease if scene.sensor is not None else None
        if self.lease is not None:
            here = self.torch.cuda.current_stream(self.device)
            for ev in self.lease.static_waits:
                here.wait_event(ev)
            self.lease.static_waits = []
        self.bound = BoundScene(scene, self.mem, _LibDerive(self.lib), lease=self.lease)
        if self.lease is not None:
            self._changed = self.lease.arena.changed
        # f64 accumulation image (exact for integer electron counts of any size, so the result does not
        # depend on the order of the atomics); image_numpy() rounds it to the float32 ImageF
        self.image = self.torch.zeros((scene.ny, scene.nx), dtype=self.torch.float64, device=self.device)
        # two side streams: the sequential brighter-fatter chain of the bright objects runs at high
        # priority while the wide single-launch work fills the CUs it leaves idle
        if stream_roles == "focal":
            self.plan_streams = _focal_streams(self.torch, self.device, top_index=top_index)
        else:
            self.plan_streams = _device_streams(self.torch, self.device)              # index = Renderer.STREAMS
        self._plans_run = 0
        # the library's record / wait events are process-wide and addressed by number: every renderer numbers its own from a
        # block of EVENT_BLOCK out of a free list (given back when the renderer is dropped), so that the plans of several CCDs may
        # be enqueued from different host threads at the same time and two renderers that are alive never share a number
        self._event_block = _take_event_block()
        self.s_chain, self.s_bulk, self.s_chain1, self.s_chain2, self.s_chain3 = self.plan_streams
        self.use_bf_tags = tuning.env("IMS_BF_TAGS", "0") != "0"
        # round-count thresholds that cut the bright objects into concurrent chains (plan_lsst_image)
        self.chain_class_rounds = tuple(int(v) for v in tuning.env("IMS_CHAIN_CLASSES", "40,6").split(",") if v)
        self.max_pool_photons = 300_000_000      # 32 B each
        ss = scene.sensor
        if lazy_static is None:
            lazy_static = tuning.flag("IMS_LAZY_STATIC")
        self.lazy_static = bool(lazy_static and lazy_static_applies(scene) and self.bound.sensor_struct.pristine_margin >= 0.0)
        if not self.lazy_static and ss is not None:
            self.init_boundaries(0, len(ss.slots))

    def _need_static(self, what):
        """Entry points that read the stored state of slot 0 (pooled accumulates of ordinary rows, pixel areas, whole-CCD updates)
        on a renderer that was made without it (lazy_static): the state is made now and the renderer is an ordinary one from here
        on.  A renderer whose state is leased from a focal plane's arena cannot do that -- the lazy arena's one static region
        belongs to nobody -- and raises."""
        if not self.lazy_static:
            return
        if self.lease is not None:
            raise RuntimeError(f"{what} reads the static pixel-boundary state, which this renderer (lazy_static, state leased from a "
                               "sensor arena) does not have: make it with lazy_static=False")
        self.lazy_static = False
        self.init_boundaries(0, len(self.scene.sensor.slots))

    def release_state(self, streams=None):
        """hand the leased pixel-boundary state back (SensorLease.release); a renderer without a lease has nothing to do"""
        if getattr(self, "lease", None) is not None:
            self.lease.release(streams)

    def __del__(self):
        try:
            if getattr(self, "lease", None) is not None and self.lease._held:
                self.lease.release(list(set(self.plan_streams)) + [self.torch.cuda.current_stream(self.device)])
        except Exception:
            pass
        block = getattr(self, "_event_block", None)
        if block is not None:
            self._event_block = None
            _give_event_block(block)

    # -- helpers --
    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    # what the pixel search of a brighter-fatter round needs of a photon: the `converted` pool format of
    # ims_photons_t (position at the conversion depth, flux, polygon shrink factor with the coin in its sign)
    POOL4 = ("x", "y", "flux", "dxdz")

    def _pool4(self, n):
        """A compact pool of converted photons (32 B each)."""
        t = {f: self.torch.empty(max(int(n), 1), dtype=self.torch.float64, device=self.device) for f in self.POOL4}
        ph = Photons()
        ph.n = int(n)
        ph.converted = 1
        for f in self.POOL4:
            setattr(ph, f, t[f].data_ptr())
        return ph, t

    def _upload_objects(self, objects):
        objects = np.ascontiguousarray(objects, dtype=OBJECT_DTYPE)
        obj_t = self.torch.from_numpy(objects.view(np.uint8).reshape(-1)).to(self.device)
        prefix = segment_prefix(objects["n_phot"], self.scene.seg_size)
        pre_t = self.mem.put(prefix)[0]
        # segment -> object map (saves every workgroup a binary search in the prefix)
        seg_obj = np.repeat(np.arange(len(objects), dtype=np.int32), np.diff(prefix))
        pre_t.seg_object = self.torch.from_numpy(seg_obj).to(self.device) if len(seg_obj) else None
        return objects, obj_t, prefix, pre_t

    def _gather_objects(self, master, index, first, count, segments=True):
        """A launch table gathered on the device from a DeviceTable: rows master[index] with phot_first += first, n_phot = count
        (None: the object's own), bf_state 0 and IMS_OBJ_FAINT cleared (photon pooling).  Returns (rows tensor, segment
        prefix on the host, its device tensor with .seg_object)."""
        t = self.torch
        if not (isinstance(index, np.ndarray) and index.dtype == np.int64 and index.flags.c_contiguous):
            index = np.ascontiguousarray(index, dtype=np.int64)
        n = len(index)
        n_phot = None if t.is_tensor(count) else (master.n_phot[index] if count is None else np.asarray(count)).astype(np.int64)
        dst = t.empty(max(n, 1) * OBJECT_DTYPE.itemsize, dtype=t.uint8, device=self.device)
        cache = getattr(master, "_index_tensors", None)
        if cache is None:
            cache = master._index_tensors = {}
        hit = cache.get(id(index))
        if hit is not None and hit[0] is index:
            idx_t = hit[1]                                       # an index array that recurs (the PHOT objects of every batch)
        else:
            idx_t = t.from_numpy(index).to(self.device)
            cache[id(index)] = (index, idx_t)
        on_device = t.is_tensor(count)            # shares formed on the device (prepared_pooled_batches): no host copy of them
        if on_device:
            first_t, count_t = first, count
        else:
            first_t = t.from_numpy(np.ascontiguousarray(first, dtype=np.int64)).to(self.device) if first is not None else None
            count_t = t.from_numpy(np.ascontiguousarray(n_phot)).to(self.device) if count is not None else None
        _abi.check(self.lib.ims_gather_rows(master.rows.data_ptr(), idx_t.data_ptr(), first_t.data_ptr() if first_t is not None else None,
                                            count_t.data_ptr() if count_t is not None else None, None, _abi.IMS_OBJ_FAINT,
                                            dst.data_ptr(), n, self._stream()), "ims_gather_rows")
        dst.keep = (idx_t, first_t, count_t)
        if not segments:                      # launches with one wavefront per object (ims_accumulate_small) need no segment table
            return dst, None, None
        if on_device:
            # segment table by the same arithmetic on the device; only the number of segments comes back (a launch parameter)
            segs = (count_t + (self.scene.seg_size - 1)) // self.scene.seg_size
            pre_t = t.zeros(n + 1, dtype=t.int64, device=self.device)
            t.cumsum(segs, 0, out=pre_t[1:])
            total = int(pre_t[-1].item())
            pre_t.seg_object = (t.repeat_interleave(t.arange(n, dtype=t.int32, device=self.device), segs, output_size=total)
                                if total else None)
            return dst, np.array([0, total], dtype=np.int64), pre_t
        prefix = segment_prefix(n_phot, self.scene.seg_size)
        pre_t = t.from_numpy(prefix).to(self.device)
        seg_obj = np.repeat(np.arange(n, dtype=np.int32), np.diff(prefix))
        pre_t.seg_object = t.from_numpy(seg_obj).to(self.device) if len(seg_obj) else None
        return dst, prefix, pre_t

    # -- fused path (LSST_Image / LSST_Silicon) --
    def render(self, objects, realized=None):
        """Shoot every object of the table and accumulate into self.image.  `realized`: optional
        float64 device tensor [n_objects] receiving base['realized_flux'] per object."""
        objects, obj_t, prefix, pre_t = self._upload_objects(objects)
        if len(objects) == 0:
            return
        P = self.bound.params(obj_t.data_ptr(), len(objects), pre_t.data_ptr(), int(prefix[-1]),
                              self.image.data_ptr(), realized.data_ptr() if realized is not None else None,
                              _seg_ptr(pre_t), lazy=self.lazy_static)
        _abi.check(self.lib.ims_shoot_accumulate(C.byref(P), self._stream()), "ims_shoot_accumulate")
        self._keep = (obj_t, pre_t)

    def plan_lsst_image(self, objects, nrecalc=None, want_realized=False):
        """Build the launch plan of LSST_Image + LSST_Silicon semantics (imsim/lsst_image.py:342-368,
        imsim/stamp.py:558-573): every object accumulates on its own stamp, so brighter-fatter only
        sees the object's own charge.  Objects below `nrecalc` photons never trigger a boundary
        update and share the static (tree-ring) CCD boundaries in ONE fused launch (bulk stream).
        Brighter objects get a private boundary region; the sensor-independent part of ALL their
        photons (shoot, PSF, op chain) is computed by one wide launch into a compact pool, and only
        the sensor step is advanced in rounds of `nrecalc` photons with one batched
        updatePixelDistortions between rounds (chain stream, high priority).  All object tables of
        the plan are uploaded here, so executing the plan touches no host data."""
        ss = self.scene.sensor
        # device_table.DeviceTable: the 256-byte rows stay on the device; the planner works on 40-byte views of them and every
        # launch table is gathered from the master table by ims_gather_rows (index, photon range, boundary slot)
        master = objects if hasattr(objects, "rows") and hasattr(objects, "n_phot") and not isinstance(objects, np.ndarray) else None
        if master is not None:
            objects = slim_view(master)
        else:
            objects = np.ascontiguousarray(objects, dtype=OBJECT_DTYPE)
        plan = PlanList()
        arena = plan.arena = _Arena()
        realized_parts, realized_refs = [], []
        gathers = []

        def finish():
            """ONE upload for every table of the plan (f-1: was ~1 000 separate copies), then the addresses"""
            t = arena.upload(self.mem)
            for off, n, tmp in realized_refs:
                realized_parts.append((t[off:off + 8 * n].view(self.torch.int64), tmp))
            base = arena.base
            for dst, o_idx, o_first, o_count, o_bf, n in gathers:
                _abi.check(self.lib.ims_gather_rows(master.rows.data_ptr(), base + o_idx, base + o_first, base + o_count, base + o_bf, 0,
                                                    dst.data_ptr(), n, self._stream()), "ims_gather_rows")
            plan.keep_rows = [g[0] for g in gathers] + ([master] if master is not None else [])
            return plan, realized_parts

        def upload(part, index, kind, extra=None):
            if master is None:
                part = np.ascontiguousarray(part, dtype=OBJECT_DTYPE)
            prefix = segment_prefix(part["n_phot"], self.scene.seg_size)
            seg_obj = np.repeat(np.arange(len(part), dtype=np.int32), np.diff(prefix))
            tmp = None
            if want_realized and kind != "shoot_pool":
                tmp = self.torch.zeros(len(part), dtype=self.torch.float64, device=self.device)
                where = np.asarray(index, dtype=np.int64)
                if master is not None:
                    where = objects["row"][where]                  # realized fluxes are indexed like the master table
                realized_refs.append((arena.add(where), len(part), tmp))
            P = self.bound.params(None, len(part), None, int(prefix[-1]),
                                  self.image.data_ptr(), tmp.data_ptr() if tmp is not None else None, None,
                                  lazy=self.lazy_static and kind == "render")
            if kind != "render" and getattr(self, "_prepass_event", None) is not None:
                P.screen_kick = None                  # the pre-pass covered the ordinary objects only: the bright ones gather in place
            if master is not None:
                dst = self.torch.empty(max(len(part), 1) * OBJECT_DTYPE.itemsize, dtype=self.torch.uint8, device=self.device)
                gathers.append((dst, arena.add(np.ascontiguousarray(part["row"])), arena.add(np.ascontiguousarray(part["phot_first"])),
                                arena.add(np.ascontiguousarray(part["n_phot"])), arena.add(np.ascontiguousarray(part["bf_state"])),
                                len(part)))
                P.objects = dst.data_ptr()
            else:
                arena.patch(P, "objects", arena.add(part))
            arena.patch(P, "seg_prefix", arena.add(prefix))
            if len(seg_obj):
                arena.patch(P, "seg_object", arena.add(seg_obj))
            return P, (tmp,)

        def add_render(part, index, stream="bulk"):
            P, keep = upload(part, index, "render")
            if getattr(self, "_prepass_event", None) is not None:
                plan.append(("wait", self._prepass_event, stream))       # the phase-screen pre-pass of these objects (side stream)
            plan.append(("render", P, keep, int(part["n_phot"].sum()), len(part), stream))

        if ss is None:
            if len(objects):
                add_render(objects, np.arange(len(objects)))
            return finish()
        if nrecalc is None:
            nrecalc = ss.model.nrecalc
        b = self.bound
        normal, groups = plan_bf_groups(objects, nrecalc, b.n_static_slots, b.static_cells, b.scratch_cells,
                                        b.slot_capacity, self.max_pool_photons)
        n_events = self._event_block
        render_done = False
        for idx, slots in groups:
            plan.append(("slots", slots))
            n0 = b.n_static_slots
            grp = objects[idx]                            # (a fancy-indexed copy) sorted by n_phot, brightest first
            grp["bf_state"] = n0 + np.arange(len(grp))
            total = grp["n_phot"].copy()
            offs = np.concatenate([[0], np.cumsum(total)]).astype(np.int64)
            pool, pool_t = self._pool4(offs[-1])
            n_rounds = (total + nrecalc - 1) // nrecalc
            # An object's rounds only depend on its OWN earlier rounds, so the few very bright objects
            # (hundreds of short, latency-bound rounds) must not wait for the many moderately bright
            # ones: the group is cut into up to three classes (four with a third threshold) by round count, each advancing on its
            # own chain stream, the longest chain first in every queue.
            cuts = [int(np.count_nonzero(n_rounds >= t)) for t in self.chain_class_rounds]
            bounds = [0] + [c for c in cuts if 0 < c < len(grp)] + [len(grp)]
            bounds = sorted(set(bounds))
            classes = [(bounds[k], bounds[k + 1]) for k in range(len(bounds) - 1)]
            # every class initialises the private regions of ITS objects on its own stream (the slots are in class
            # order): the longest chains start after ~0.2 ms instead of behind the initialisation of all regions
            for c, (ca, cb) in enumerate(classes):
                plan.append(("init", n0 + ca, cb - ca, self.CHAIN_STREAMS[c]))
            # 1. everything of the photons that does not depend on the sensor state goes into a compact
            #    pool, produced per class in slices of rounds: the first slice (round 0) on the class's
            #    chain stream so the chain can start at once, the later ones on the bulk stream (longest
            #    chain first); round r waits for the event of the slice that holds its photons.
            chains, bulk_items = [], []
            for c, (ca, cb) in enumerate(classes):
                cstream = self.CHAIN_STREAMS[c]
                ctot, cgrp, cidx, coffs = total[ca:cb], grp[ca:cb], idx[ca:cb], offs[ca:cb]
                rounds = int(n_rounds[ca])
                edges = [0] + [e for e in (1, 3, 8, 20, 60) if e < rounds] + [rounds]
                slice_of_round = np.zeros(rounds, dtype=np.int64)
                ev_base = n_events
                n_events += len(edges) - 1
                for k in range(len(edges) - 1):
                    ra, rb = edges[k], edges[k + 1]
                    slice_of_round[ra:rb] = k
                    lo = np.minimum(ctot, ra * nrecalc)
                    hi = np.minimum(ctot, rb * nrecalc)
                    act = hi > lo
                    part = cgrp[act].copy()
                    part["phot_first"] = cgrp["phot_first"][act] + lo[act]
                    part["n_phot"] = (hi - lo)[act]
                    offs_t = arena.ref((coffs + lo)[act])
                    P, keep = upload(part, cidx[act], "shoot_pool")
                    stream = cstream if k == 0 else "bulk"
                    item = ("shoot_pool", P, (keep, offs_t, pool_t), pool, offs_t, int(part["n_phot"].sum()),
                            len(part), stream)
                    if k == 0:
                        plan.append(item)
                    else:
                        bulk_items += [item, ("record", ev_base + k, stream)]
                chains.append(dict(stream=cstream, ca=ca, tot=ctot, grp=cgrp, idx=cidx, offs=coffs, rounds=rounds,
                                   slice_of_round=slice_of_round, ev_base=ev_base, waited=0, edges=edges))
            plan.extend(bulk_items)
            # The ordinary objects' fused launch is queued on the bulk stream right behind the pool slices, BEFORE the
            # ~1 300 short launches of the rounds: the host needs ~20 us per launch to enqueue those, and an item at
            # the end of the plan would leave the bulk stream idle until the host gets there.
            if len(normal) and not render_done:
                part = objects[normal]                    # fancy indexing copies
                part["bf_state"] = 0
                add_render(part, normal, "bulk")
                render_done = True
            # 2. the sequential part: rounds of nrecalc photons per object through the sensor (accumulate, then
            #    updatePixelDistortions for the objects that go on).  ONE plan item carries the chains of all classes:
            #    ims_run_plan derives every round's launches from the class table (objects sorted by photon count, so
            #    the objects of a round are a prefix of it) and enqueues the classes round-robin, one round each, so
            #    that all chains advance at the same pace.  No per-round tables, no per-round Python.
            descs = []
            for ch in chains:
                start_t = arena.ref(ch["offs"])
                P, keep = upload(ch["grp"], ch["idx"], "acc_pool")
                descs.append(dict(P=P, keep=(keep, start_t), pool=pool, start=start_t,
                                  n_phot=np.ascontiguousarray(ch["tot"], dtype=np.int64), first_slot=n0 + ch["ca"],
                                  stream=ch["stream"], rounds=ch["rounds"], edges=ch["edges"], ev_base=ch["ev_base"]))
            plan.append(("rounds", descs, int(nrecalc), 1 if self.use_bf_tags else 0))
            if n_events - self._event_block > EVENT_BLOCK or n_events >= self.PREPASS_EVENT:
                raise ValueError(f"a plan may use at most {EVENT_BLOCK} library events (its renderer's block)")
        if len(normal) and not render_done:
            part = objects[normal]
            part["bf_state"] = 0
            add_render(part, normal, "bulk")
        return finish()

    def _compile_plan(self, plan):
        """Turn a plan into ims_plan_item_t arrays (one per stretch between host-side slot-table
        writes) so that replaying it is ONE C call per stretch: the brighter-fatter chain is
        hundreds of short dependent launches and would otherwise be bound by the host launch rate."""
        b = self.bound
        if self.scene.sensor is not None and not hasattr(self, "_changed"):
            cells = b.static_cells + int(b.scratch_cells)
            self._changed = self.torch.zeros(max(cells, 1), dtype=self.torch.uint8, device=self.device)
        stretches, keep, cur = [], [], []
        cur_prefix = None

        def close():
            if cur:
                arr = (_abi.PlanItem * len(cur))(*cur)
                stretches.append(("run", arr, len(cur)))
                cur.clear()

        def tile_prefix_from(first):
            """prefix sum of the 16 x 16-cell tiles of the slots first, first + 1, ... (host array, device tensor)"""
            if cur_prefix is None:
                return self._tile_prefix(first)
            sl, cache = cur_prefix
            if first not in cache:
                part = sl[first - b.n_static_slots:]
                tiles = ((part["nx"].astype(np.int64) + 1 + 15) // 16) * ((part["ny"].astype(np.int64) + 1 + 15) // 16)
                prefix = np.concatenate([[0], np.cumsum(tiles)]).astype(np.int64)
                cache[first] = (prefix, self.mem.put(prefix)[0])
                keep.append(cache[first])
            return cache[first]

        for item in plan:
            kind = item[0]
            it = _abi.PlanItem()
            if kind == "render":
                it.kind, it.stream, it.params = _abi.IMS_PLAN_RENDER, self.STREAMS[item[5]], C.addressof(item[1])
            elif kind == "acc_pool":
                it.kind, it.params, it.pool, it.aux = _abi.IMS_PLAN_ACC_POOL, C.addressof(item[1]), C.addressof(item[3]), item[4].data_ptr()
                it.stream = self.STREAMS[item[7]]
            elif kind == "shoot_pool":
                it.kind, it.params, it.pool, it.aux = _abi.IMS_PLAN_SHOOT_POOL, C.addressof(item[1]), C.addressof(item[3]), item[4].data_ptr()
                it.stream = self.STREAMS[item[7]]
            elif kind == "rounds":
                descs, nrecalc, use_tags = item[1], item[2], item[3]
                arr = (_abi.Chain * len(descs))()
                for c, d in zip(arr, descs):
                    prefix, prefix_t = tile_prefix_from(d["first_slot"])
                    c.params, c.pool, c.pool_start = C.addressof(d["P"]), C.addressof(d["pool"]), d["start"].data_ptr()
                    c.n_phot, c.tile_prefix, c.tile_prefix_host = d["n_phot"].ctypes.data, prefix_t.data_ptr(), prefix.ctypes.data
                    c.n_objects, c.first_slot, c.stream, c.nrecalc = len(d["n_phot"]), d["first_slot"], self.STREAMS[d["stream"]], nrecalc
                    c.n_rounds, c.use_tags, c.ev_base = d["rounds"], use_tags, d["ev_base"]
                    edges = d["edges"][:-1]                      # slice starts; the last edge is the round count
                    if len(edges) > _abi.IMS_MAX_CHAIN_EDGES:
                        raise ValueError("too many pool slices for one chain")
                    c.n_edges = len(edges)
                    for k, e in enumerate(edges):
                        c.edges[k] = e
                    keep.append((prefix, prefix_t))
                keep.append(arr)
                it.kind, it.stream, it.n_slots, it.aux2 = _abi.IMS_PLAN_ROUNDS, 0, len(descs), C.addressof(arr)
            elif kind == "record":
                it.kind, it.n_slots, it.stream = _abi.IMS_PLAN_RECORD, item[1], self.STREAMS[item[2]]
            elif kind == "wait":
                it.kind, it.n_slots, it.stream = _abi.IMS_PLAN_WAIT, item[1], self.STREAMS[item[2]]
            elif kind == "slots":
                close()
                sl = item[1]
                cur_prefix = (sl, {})           # tile prefixes of this group's slot table, by first slot
                stretches.append(("slots", sl, 0))
                continue
            elif kind == "init":
                prefix, prefix_t = tile_prefix_from(item[1])
                it.kind, it.first_slot, it.n_slots = _abi.IMS_PLAN_INIT, item[1], item[2]
                it.aux, it.n_tiles = prefix_t.data_ptr(), int(prefix[item[2]])
                it.stream = self.STREAMS[item[3]]
            elif kind == "update":
                first, n = item[1], item[2]
                prefix, prefix_t = tile_prefix_from(first)
                it.kind, it.first_slot, it.n_slots = _abi.IMS_PLAN_UPDATE, first, n
                it.stream = self.STREAMS[item[3]]
                it.tag = item[4] if len(item) > 4 else 0
                it.aux, it.n_tiles = prefix_t.data_ptr(), int(prefix[n])
            cur.append(it)
        close()
        return stretches, keep

    def execute_plan(self, plan, compiled=None):
        """Run a launch plan.  Items tagged "bulk" go to the bulk stream, everything else (the
        brighter-fatter chain) to the high-priority chain stream; both are joined back into the
        caller's stream."""
        torch = self.torch
        if compiled is None:
            compiled = self._compile_plan(plan)
        stretches, _ = compiled
        main = torch.cuda.current_stream(self.device)
        ev0 = torch.cuda.Event()
        ev0.record(main)
        streams = self.plan_streams                                              # index = Renderer.STREAMS
        for st in set(streams):
            st.wait_event(ev0)
        b = self.bound
        sensor_dev = b.sensor_dev_ptr if self.scene.sensor is not None else None
        sensor_host = C.byref(b.sensor_host) if self.scene.sensor is not None else None
        changed = self._changed.data_ptr() if hasattr(self, "_changed") else None
        sarr = (C.c_void_p * len(streams))(*[st.cuda_stream for st in streams])
        own_work_queued = self._plans_run > 0
        for kind, payload, n in stretches:
            if kind == "slots":
                # the slot table is written by a copy on the chain stream: ordered behind everything queued so far on
                # every stream and ahead of everything that follows, on the GPU (no host synchronisation)
                # (the FIRST slot table of a renderer's FIRST plan has nothing of its own queued ahead of it -- what the
                # shared streams hold belongs to other renderers, other buffers: a CCD of a focal plane must not wait for the
                # wide launches of the previous one.  Every later table of the same plan -- a second brighter-fatter group --
                # is rewritten under this plan's own launches and waits for all of them)
                for st in (set(streams) if own_work_queued else ()):
                    if st is not self.s_chain:
                        ev = torch.cuda.Event()
                        ev.record(st)
                        self.s_chain.wait_event(ev)
                with torch.cuda.stream(self.s_chain):
                    b.set_private_slots(payload)
                    ev = torch.cuda.Event()
                    ev.record(self.s_chain)
                for st in set(streams):
                    if st is not self.s_chain:
                        st.wait_event(ev)
                continue
            _abi.check(self.lib.ims_run_plan(payload, n, sensor_dev, sensor_host, changed, sarr, len(streams)), "ims_run_plan")
            own_work_queued = True
        self._plans_run += 1
        for st in set(streams):
            ev = torch.cuda.Event()
            ev.record(st)
            main.wait_event(ev)

    # -- the native planner (ims_plan_*): plan, bind, upload and run of one LSST_Image render inside the library --
    def native_plan_ok(self, objects):
        """The library's planner covers the default path; the options it does not know run the numpy planner: the phase-screen
        pre-pass, and IMS_NATIVE_PLAN=0 (which keeps the numpy planner as the checker it is in the tests)."""
        if tuning.env("IMS_NATIVE_PLAN", "1") == "0":
            return False
        if tuning.env("IMS_SCREEN_PREPASS", "0") != "0" and self.scene.atm is not None:
            return False
        return True

    def native_plan(self, objects, nrecalc=None, want_realized=False, coarse_slices=False):
        return NativePlan(self, objects, nrecalc, want_realized, coarse_slices)

    def touch_streams(self):
        """Run a trivial operation on every plan stream and wait for it: the streams then hold their hardware queues before
        anything else of the process (an RCCL communicator, say) brings streams of its own into use."""
        t = self.torch
        # IMS_STREAM_TOUCH: the order of first use (indices into Renderer.STREAMS; HIP binds a stream to a hardware queue then)
        order = [int(v) for v in tuning.env("IMS_STREAM_TOUCH").split(",") if v != ""]
        for k in order + [k for k in range(len(self.plan_streams)) if k not in order]:
            if k >= len(self.plan_streams):
                continue
            with t.cuda.stream(self.plan_streams[k]):
                t.zeros(1, device=self.device).add_(1.0)
            self.plan_streams[k].synchronize()
        t.cuda.synchronize(self.device)

    # -- phase-screen pre-pass (ims_screen_prepass): the gathers of all photons of a render, in cache-friendly order --
    PREPASS_EVENT = 60000          # library event (ims_run_plan RECORD / WAIT) that says "the pre-pass is through"

    def screen_prepass(self, objects, nrecalc=None):
        """For scenes with a phase-screen PSF (AtmosphericPSF): returns (objects with screen_base set, run) where run()
        enqueues the pre-pass and must precede every execution of a plan built from those objects while
        `self.bound.base_params.screen_kick` was set (plan_lsst_image copies it); (objects, None) when there is nothing to do.
        IMS_SCREEN_PREPASS: 0 (default) -- off: every photon gathers where it is made (measured fastest: C3b 37.2 ms against
        41.1 / 41.3 ms); 1 -- every photon of the render, ahead of everything (HBM traffic of the shooting
        kernels down to their algorithmic bytes); 2 -- the ORDINARY objects only (one fused launch, whose in-place gathers
        are the one memory-bound launch of the path), on a side stream beside the pool shoots of the bright objects.
        Host tables only."""
        mode = tuning.env("IMS_SCREEN_PREPASS", "0")
        comp = next((k for k, c in enumerate(self.scene.psf) if int(c[0]) == _abi.IMS_PSF_SCREENS), None)
        self._prepass_event = None
        if (comp is None or self.scene.atm is None or mode == "0" or not isinstance(objects, np.ndarray) or len(objects) == 0):
            return objects, None
        objects = np.ascontiguousarray(objects, dtype=OBJECT_DTYPE).copy()
        covered = np.arange(len(objects))
        ss = self.scene.sensor
        if mode == "2":
            if ss is None:
                mode = "1"
            else:
                b = self.bound
                covered, _ = plan_bf_groups(objects, ss.model.nrecalc if nrecalc is None else nrecalc, b.n_static_slots,
                                            b.static_cells, b.scratch_cells, b.slot_capacity, self.max_pool_photons)
        sub = objects[covered]
        n_phot = sub["n_phot"].astype(np.int64)
        cum = np.concatenate([[0], np.cumsum(n_phot)]).astype(np.int64)
        total = int(cum[-1])
        # an entry of the sort packs the photon's index in its object's stream (phot_first + j) into 32 bits
        if (total == 0 or int(n_phot.max()) >= 2 ** 31 or len(sub) >= 2 ** 31
                or int((sub["phot_first"].astype(np.int64) + n_phot).max()) >= 2 ** 32 or int(sub["phot_first"].min()) < 0):
            return objects, None
        sub["screen_base"] = cum[:-1] - sub["phot_first"]
        objects["screen_base"][covered] = sub["screen_base"]
        # eight slices of the (spatially sorted) table with about equal photon counts, one per XCD
        cuts = np.searchsorted(cum, total * np.arange(1, 8) / 8.0, side="left")
        first = np.concatenate([[0], np.clip(cuts, 0, len(sub)), [len(sub)]]).astype(np.int64)
        first = np.maximum.accumulate(first)
        per_slice = cum[first[1:]] - cum[first[:-1]]
        n_buckets = int(tuning.env("IMS_SCREEN_BUCKETS", "128"))
        t = self.torch
        _, obj_t, prefix, pre_t = self._upload_objects(sub)
        P = self.bound.params(obj_t.data_ptr(), len(sub), pre_t.data_ptr(), int(prefix[-1]), self.image.data_ptr(), None,
                              _seg_ptr(pre_t))
        entries = t.empty(total, dtype=t.int64, device=self.device)
        kick = t.empty(2 * total, dtype=t.float64, device=self.device)
        scratch = t.zeros(8 * n_buckets + 16 + 4 * len(sub), dtype=t.int64, device=self.device)
        self.bound.base_params.screen_kick = kick.data_ptr()
        first = np.concatenate([first, prefix[first]]).astype(np.int64)      # objects, then segments
        keep = (obj_t, pre_t, entries, kick, scratch, first, P)
        side = mode == "2"
        if side:
            self._prepass_event = self.PREPASS_EVENT
            rec = (_abi.PlanItem * 1)()
            rec[0].kind, rec[0].stream, rec[0].n_slots = _abi.IMS_PLAN_RECORD, self.STREAMS["chain2"], self.PREPASS_EVENT

        def run():
            def go():
                _abi.check(self.lib.ims_screen_prepass(C.byref(P), comp, n_buckets, first.ctypes.data, int(per_slice.max()),
                                                       scratch.data_ptr(), entries.data_ptr(), kick.data_ptr(), self._stream()),
                           "ims_screen_prepass")
            if not side:
                return go()
            # on the third chain stream, ordered behind whatever the caller's stream holds; the fused launch of the plan waits
            # for the library event recorded behind it
            main = t.cuda.current_stream(self.device)
            self.s_chain2.wait_stream(main)
            with t.cuda.stream(self.s_chain2):
                go()
            streams = self.plan_streams
            sarr = (C.c_void_p * len(streams))(*[st.cuda_stream for st in streams])
            _abi.check(self.lib.ims_run_plan(rec, 1, None, None, None, sarr, len(streams)), "ims_run_plan")
        run.keep = keep
        run.photons = total
        run.side = side
        return objects, run

    def render_lsst_image(self, objects, nrecalc=None, realized=None, defer=False):
        """defer (focal_plane's joint mode): the plan is enqueued without its join and, where possible, without the rounds of its
        top chain (NativePlan.run(defer=True)); returns the plan -- the caller runs engine.run_joint_plans over the CCDs of a
        batch, then plan.join() and, for `realized`, plan.add_realized.  None: this render cannot be deferred and ran whole."""
        if self.native_plan_ok(objects):
            plan = self.native_plan(objects, nrecalc, want_realized=realized is not None,
                                    coarse_slices=defer and tuning.flag("IMS_FOCAL_COARSE_SLICES"))
            self._keep_plan = plan
            if defer:
                plan.run(defer=True)
                return plan
            plan.run()
            if realized is not None:
                plan.add_realized(realized)
            return
        objects, prepass = self.screen_prepass(objects, nrecalc)
        try:
            plan, parts = self.plan_lsst_image(objects, nrecalc, want_realized=realized is not None)
        finally:
            self.bound.base_params.screen_kick = None
            self._prepass_event = None
        if prepass is not None:
            prepass()
            plan.prepass = prepass
        self.execute_plan(plan)
        if realized is not None:
            for index, tmp in parts:
                realized.index_add_(0, index, tmp)
        self._keep_plan = plan

    def prepared_lsst_image(self, objects, nrecalc=None):
        """Upload everything once; returns a callable replaying the whole LSST_Image render (used
        by bench.py: the timed region starts with all inputs resident in HBM)."""
        if self.native_plan_ok(objects):
            plan = self.native_plan(objects, nrecalc)
            z = plan.sizes

            def launch():
                plan.run()
            launch.plan = plan
            launch.prepass = None
            launch.photons = int(z.render_photons + z.shoot_photons)
            launch.object_rows = int(z.render_rows + z.shoot_rows + z.chain_rows)
            launch.pool_photons = int(z.shoot_photons)
            has_screens = self.scene.atm is not None and any(int(c[0]) == _abi.IMS_PSF_SCREENS for c in self.scene.psf)
            scr = 96 if has_screens else 0
            # 4: the pixel search of the brighter-fatter rounds -- every pooled photon once: its 32-byte pool record read, the
            # 16-byte read-modify-write of the f64 image (the bounds line and polygon points it gathers are sensor state, not
            # counted: they are what the counter traffic shows on top)
            launch.timed = {1: (int(z.n_render_launches), int(z.render_photons * (16 + scr) + z.render_rows * 256)),
                            2: (int(z.n_shoot_launches), int(z.shoot_photons * (32 + scr) + z.shoot_rows * 256)),
                            4: (int(z.n_round_launches), int(z.shoot_photons * 48))}
            launch.timed_waves = {1: 4 * int(z.render_segments), 2: 4 * int(z.shoot_segments), 4: 4 * int((z.shoot_photons + 255) // 256)}
            return launch
        objects, prepass = self.screen_prepass(objects, nrecalc)
        try:
            plan, _ = self.plan_lsst_image(objects, nrecalc)
        finally:
            self.bound.base_params.screen_kick = None
            self._prepass_event = None
        n_groups = sum(1 for it in plan if it[0] == "slots")
        if n_groups == 1:
            # the slot table never changes between replays: set it once, outside the timed region
            for it in plan:
                if it[0] == "slots":
                    self.bound.set_private_slots(it[1])
            arena = plan.arena
            plan = PlanList(it for it in plan if it[0] != "slots")
            plan.arena = arena

        compiled = self._compile_plan(plan)

        def launch():
            if prepass is not None:
                prepass()
            self.execute_plan(plan, compiled)
        launch.plan = plan
        launch.prepass = prepass
        launch.photons = sum(it[3] for it in plan if it[0] == "render") + sum(it[5] for it in plan if it[0] == "shoot_pool")
        launch.object_rows = sum(it[4] for it in plan if it[0] == "render") + sum(it[6] for it in plan if it[0] in ("shoot_pool", "acc_pool", "chain"))
        launch.pool_photons = sum(it[5] for it in plan if it[0] == "shoot_pool")
        # the two photon-pipeline kernels the library can time (ims_enable_timing): launches per replay and
        # their algorithmic bytes.  Fused render: f64 image RMW (16 B/photon); pool shoot: the four f64
        # fields of a converted photon it writes (32 B/photon); both + one 256-B object row per object
        # With the 6-layer AtmosphericPSF every photon also reads 6 x 4 fp32 screen samples (+ 96 B) where it is
        # made, or its 16-byte gradient sum where the pre-pass gathered it
        has_screens = self.scene.atm is not None and any(int(c[0]) == _abi.IMS_PSF_SCREENS for c in self.scene.psf)
        scr_render = 0 if not has_screens else (16 if prepass is not None else 96)
        scr_pool = 0 if not has_screens else (16 if (prepass is not None and not prepass.side) else 96)
        launch.timed = {
            1: (sum(1 for it in plan if it[0] == "render"),
                sum(it[3] * (16 + scr_render) + it[4] * 256 for it in plan if it[0] == "render")),
            2: (sum(1 for it in plan if it[0] == "shoot_pool"),
                sum(it[5] * (32 + scr_pool) + it[6] * 256 for it in plan if it[0] == "shoot_pool")),
        }
        # wavefronts per replay of the two kernels (4 per 256-thread segment, as SQ_WAVES counts them)
        launch.timed_waves = {1: sum(4 * int(it[1].n_segments) for it in plan if it[0] == "render"),
                              2: sum(4 * int(it[1].n_segments) for it in plan if it[0] == "shoot_pool")}
        return launch

    def prepared(self, objects, bf_tag=0):
        """Upload an object table once; returns a zero-argument callable that launches the fused
        kernel (used by bench.py so that the timed region has its inputs resident in HBM)."""
        objects, obj_t, prefix, pre_t = self._upload_objects(objects)
        P = self.bound.params(obj_t.data_ptr(), len(objects), pre_t.data_ptr(), int(prefix[-1]),
                              self.image.data_ptr(), None, _seg_ptr(pre_t), lazy=self.lazy_static)
        P.bf_tag = bf_tag
        ref = C.byref(P)
        keep = (obj_t, pre_t, P)

        def launch():
            _abi.check(self.lib.ims_shoot_accumulate(ref, self._stream()), "ims_shoot_accumulate")
        launch.keep = keep
        launch.photons = int(objects["n_phot"].sum())
        launch.object_rows = len(objects)
        launch.timed = {1: (1, launch.photons * 16 + launch.object_rows * 256), 2: (0, 0)}
        launch.timed_waves = {1: 4 * int(prefix[-1]), 2: 0}
        return launch

    # -- pooled path (LSST_PhotonPoolingImage / LSST_Photons) --
    def shoot_photons(self, objects):
        objects, obj_t, prefix, pre_t = self._upload_objects(objects)
        offs = np.concatenate([[0], np.cumsum(objects["n_phot"])]).astype(np.int64)
        off_t = self.torch.from_numpy(offs).to(self.device)
        pool = PhotonPool(self.torch, self.device, offs[-1], off_t, obj_t, len(objects), pre_t, int(prefix[-1]))
        P = self.bound.params(obj_t.data_ptr(), len(objects), pre_t.data_ptr(), int(prefix[-1]), self.image.data_ptr(),
                              None, _seg_ptr(pre_t))
        ph = pool.struct()
        _abi.check(self.lib.ims_shoot_photons(C.byref(P), off_t.data_ptr(), C.byref(ph), self._stream()), "ims_shoot_photons")
        return pool

    def shoot_ops_photons(self, objects, converted=False):
        """ims_shoot_ops_photons: shoot + PSF + the whole op chain in one launch, into a pool.  converted=True also
        runs the boundary-independent half of the sensor step and stores the `converted` format of ims_photons_t
        (what plan_lsst_image produces for the brighter-fatter chains)."""
        objects, obj_t, prefix, pre_t = self._upload_objects(objects)
        offs = np.concatenate([[0], np.cumsum(objects["n_phot"])]).astype(np.int64)
        off_t = self.torch.from_numpy(offs).to(self.device)
        pool = PhotonPool(self.torch, self.device, offs[-1], off_t, obj_t, len(objects), pre_t, int(prefix[-1]))
        P = self.bound.params(obj_t.data_ptr(), len(objects), pre_t.data_ptr(), int(prefix[-1]), self.image.data_ptr(),
                              None, _seg_ptr(pre_t))
        ph = pool.struct()
        ph.converted = 1 if converted else 0
        _abi.check(self.lib.ims_shoot_ops_photons(C.byref(P), off_t.data_ptr(), C.byref(ph), self._stream()), "ims_shoot_ops_photons")
        pool.converted = bool(converted)
        return pool

    def _num_vertices(self):
        return int(self.scene.sensor.model.num_vertices) if self.scene.sensor is not None else 0

    def prepared_pooled_batches(self, shoot_table, batches, realized=None, delta_only=False):
        """Photon-pooling mode with the pool resident in HBM: ONE launch shoots every photon of every batch (shoot, PSF, op
        chain, conversion depth and diffusion: `ims_shoot_ops_photons` into a converted pool, 32 B per photon -- 49 GB for
        the 1.5e9 photons of C4, which is what 288 GB of HBM are for), then every batch only runs the pixel search of ITS
        share of the photons against the sensor state of its turn (`ims_accumulate_segments`).  The photons do not depend on
        the batch they land in (streams are addressed by object and photon index), so this is the image of one fused launch
        per batch; what it saves is shooting objects of a few dozen photons per batch in mostly empty wavefronts.

        shoot_table: OBJECT_DTYPE rows with the FULL photon counts; batches: [(rows, first, count, bf_tag)] -- indices into
        shoot_table, first photon of the batch's share within the object, its photon count.  realized: optional f64 device
        tensor over the rows of shoot_table receiving the flux every object added to the image.  Returns
        (shoot, [accumulate]), zero-argument callables."""
        master = None
        if isinstance(shoot_table, tuple):
            # (DeviceTable, index): the rows are gathered on the device; the host works on photon counts alone
            master, shot = shoot_table
            shot = np.ascontiguousarray(shot, dtype=np.int64)
            shoot_n = master.n_phot[shot].astype(np.int64)
            obj_t, prefix, pre_t = self._gather_objects(master, shot, None, None)
            shoot_table = None
        else:
            shoot_table, obj_t, prefix, pre_t = self._upload_objects(shoot_table)
            shoot_n = shoot_table["n_phot"].astype(np.int64)
        n_shoot = len(shoot_n)
        base = np.concatenate([[0], np.cumsum(shoot_n)]).astype(np.int64)
        base_t = self.torch.from_numpy(base).to(self.device)
        pool, pool_t = self._pool4(base[-1])
        P = self.bound.params(obj_t.data_ptr(), n_shoot, pre_t.data_ptr(), int(prefix[-1]), self.image.data_ptr(),
                              None, _seg_ptr(pre_t))
        nv = self._num_vertices()

        def shoot():
            _abi.check(self.lib.ims_shoot_ops_photons(C.byref(P), base_t.data_ptr(), C.byref(pool), self._stream()),
                       "ims_shoot_ops_photons")
        shoot.keep = (obj_t, pre_t, base_t, pool_t, P, pool)
        shoot.photons = int(base[-1])
        shoot.object_rows = n_shoot
        shoot.waves = 4 * int(prefix[-1])
        shoot.launches = 1
        launches = []
        small_max = int(tuning.env("IMS_POOL_SMALL_MAX", "64"))    # shares up to a wavefront: one wavefront per object
        # IMS_POOL_OVERLAP=1 (default 0): the pool is shot BATCH BY BATCH on a stream of its own, ahead of the batches' pixel
        # searches and recalculations on the caller's stream -- the shoot is f64-VALU bound, the search and the whole-CCD update /
        # refresh wait on memory.  Only objects whose share of a batch fills wavefronts are shot by share (n // batches >
        # IMS_POOL_SMALL_MAX: 89 % of C4's photons); the many objects of a few dozen photons are shot whole, ahead of everything.
        # Measured (round 5): C4 244 against 236 ms, same image -- the eleven shoots take 148 instead of 136 ms and the searches
        # beside them exactly as much longer as they overlap: two WIDE kernels share the wave slots (capping the photon kernels
        # at three or two workgroups per CU: 241 / 289 ms).  What overlaps on this device is small latency-bound work beside wide
        # work (the rounds of C3 / C5), not wide beside wide.
        overlap = tuning.flag("IMS_POOL_OVERLAP") and len(batches) > 1
        by_share = None
        if overlap:
            by_share = self._pool_shares(batches, shoot_n, small_max)
            overlap = by_share is not None
        base_cache = {}
        for batch in batches:
            if isinstance(batch[0], str) and batch[0] == "parts":
                # pre-split by the caller: [(rows, first, count, small)] -- no masks over the whole batch
                _, parts_in, bf_tag = batch
                todo = [(np.asarray(rw), fi if isinstance(fi, tuple) else np.asarray(fi), None if co is None else np.asarray(co),
                         self.lib.ims_accumulate_small if sm else self.lib.ims_accumulate_segments)
                        for rw, fi, co, sm in parts_in if len(rw)]
            else:
                rows, first, count, bf_tag = batch
                rows, first, count = np.asarray(rows), np.asarray(first), np.asarray(count)
                todo = [(rows[sel], first[sel], count[sel], entry)
                        for sel, entry in ((count > small_max, self.lib.ims_accumulate_segments), (count <= small_max, self.lib.ims_accumulate_small))
                        if sel.any()]
            calls = []
            for rows_s, first_s, count_s, entry in todo:
                # base[rows] / shot[rows] of an index array that recurs in every batch (the PHOT objects) are formed once
                key = id(rows_s)
                if key not in base_cache:
                    base_cache[key] = (rows_s, base[rows_s], shot[rows_s] if master is not None else None)
                _, base_rows, shot_rows = base_cache[key]
                start_t = None
                if master is not None:
                    small = entry is self.lib.ims_accumulate_small
                    if isinstance(first_s, tuple):
                        # ("share", F, i, nb): the share [F i // nb, F (i + 1) // nb) of every object's F photons -- formed on the
                        # device from F (uploaded once per index array), as are the pool offsets and the segment table
                        _, F_s, bi, nbat = first_s
                        dkey = ("dev", id(rows_s))
                        if dkey not in base_cache:
                            base_cache[dkey] = (self.torch.from_numpy(np.ascontiguousarray(F_s, dtype=np.int64)).to(self.device),
                                                self.torch.from_numpy(np.ascontiguousarray(base_rows, dtype=np.int64)).to(self.device), F_s)
                        F_t, base_rows_t, _ = base_cache[dkey]
                        lo_t = (F_t * bi) // nbat
                        count_t = (F_t * (bi + 1)) // nbat - lo_t
                        start_t = base_rows_t + lo_t
                        part_t, bprefix, bpre_t = self._gather_objects(master, shot_rows, None, count_t, segments=not small)
                    else:
                        part_t, bprefix, bpre_t = self._gather_objects(master, shot_rows, None, count_s, segments=not small)
                    n_part = len(rows_s)
                    if small:
                        bprefix, bpre_t = np.zeros(1, dtype=np.int64), None
                else:
                    part = shoot_table[rows_s].copy()
                    part["n_phot"] = count_s
                    part, part_t, bprefix, bpre_t = self._upload_objects(part)
                    n_part = len(part)
                if start_t is None:
                    start_t = self.torch.from_numpy(np.ascontiguousarray(base_rows + first_s, dtype=np.int64)).to(self.device)
                tmp = rows_t = None
                if realized is not None:
                    tmp = self.torch.zeros(n_part, dtype=self.torch.float64, device=self.device)
                    rows_t = self.torch.from_numpy(np.ascontiguousarray(rows_s, dtype=np.int64)).to(self.device)
                Pb = self.bound.params(part_t.data_ptr(), n_part, bpre_t.data_ptr() if bpre_t is not None else None, int(bprefix[-1]),
                                       self.image.data_ptr(), tmp.data_ptr() if tmp is not None else None,
                                       _seg_ptr(bpre_t) if bpre_t is not None else None)
                Pb.bf_tag = bf_tag
                if delta_only:
                    Pb.track_static_delta = 2          # one atomic add per photon: the image is fed from the delta image (update_distortions(fold=True))
                calls.append((entry, Pb, start_t, (part_t, bpre_t, tmp, rows_t)))

            def accumulate(calls=calls):
                for entry, Pb, start_t, (_, _, tmp, rows_t) in calls:
                    if tmp is not None:
                        tmp.zero_()
                    _abi.check(entry(C.byref(Pb), C.byref(pool), start_t.data_ptr(), nv, self._stream()), "ims_accumulate_segments / _small")
                    if tmp is not None:
                        realized.index_add_(0, rows_t, tmp)
            accumulate.keep = calls
            launches.append(accumulate)
        if overlap:
            shoot, launches = self._overlapped_pool(shoot, launches, by_share, master, shot if master is not None else None, shoot_table,
                                                    base, pool, pool_t)
        return shoot, launches

    def _pool_shares(self, batches, shoot_n, small_max):
        """For the overlapped pool shoot: (rows shot whole and ahead, [(rows, first photon, photons) of the objects shot by share,
        per batch]), or None when the batches do not cover the by-share objects' photons exactly once (then nothing overlaps)."""
        nb = len(batches)
        big = (shoot_n // nb) > small_max
        covered = np.zeros(len(shoot_n), dtype=np.int64)
        shares = []
        for k, batch in enumerate(batches):
            pieces = []
            if isinstance(batch[0], str) and batch[0] == "parts":
                for rw, fi, co, _sm in batch[1]:
                    rw = np.asarray(rw)
                    if isinstance(fi, tuple):
                        _, F_s, bi, nbat = fi
                        F_s = np.asarray(F_s, dtype=np.int64)
                        lo = (F_s * bi) // nbat
                        fi, co = lo, (F_s * (bi + 1)) // nbat - lo
                    pieces.append((rw, np.asarray(fi, dtype=np.int64), np.asarray(co, dtype=np.int64)))
            else:
                pieces.append((np.asarray(batch[0]), np.asarray(batch[1], dtype=np.int64), np.asarray(batch[2], dtype=np.int64)))
            rows = np.concatenate([p[0] for p in pieces]) if pieces else np.zeros(0, dtype=np.int64)
            first = np.concatenate([p[1] for p in pieces]) if pieces else np.zeros(0, dtype=np.int64)
            count = np.concatenate([p[2] for p in pieces]) if pieces else np.zeros(0, dtype=np.int64)
            sel = big[rows] & (count > 0)
            order = np.argsort(rows[sel], kind="stable")                 # the order of the shoot table
            r, f, c = rows[sel][order], first[sel][order], count[sel][order]
            np.add.at(covered, r, c)
            shares.append((np.ascontiguousarray(r, dtype=np.int64), f, c))
        if not np.array_equal(covered[big], shoot_n[big]) or not big.any():
            return None
        return np.flatnonzero(~big & (shoot_n > 0)), shares

    def _overlapped_pool(self, shoot_all, launches, by_share, master, shot, shoot_table, base, pool, pool_t):
        """shoot(): the whole-object rows, then the shares of batch 0, 1, ... on the pool stream, an event behind each;
        accumulate k waits for the events it needs.  Same pool contents as the one launch: a photon's place in the pool and its
        random streams are functions of (object, photon index)."""
        t = self.torch
        rows_up, shares = by_share
        side = _pool_stream(t, self.device)

        def table_of(rows, first, count):
            if master is not None:
                obj_t, prefix, pre_t = self._gather_objects(master, shot[rows], first, count)
            else:
                part = shoot_table[rows].copy()
                if first is not None:
                    part["phot_first"] = part["phot_first"] + first
                    part["n_phot"] = count
                _, obj_t, prefix, pre_t = self._upload_objects(part)
            start = base[rows] + (first if first is not None else 0)
            start_t = t.from_numpy(np.ascontiguousarray(start, dtype=np.int64)).to(self.device)
            Ps = self.bound.params(obj_t.data_ptr(), len(rows), pre_t.data_ptr(), int(prefix[-1]), self.image.data_ptr(), None, _seg_ptr(pre_t))
            return Ps, start_t, (obj_t, pre_t), int(prefix[-1])
        pieces = []
        if len(rows_up):
            pieces.append(table_of(rows_up, None, None))
        n_up = len(pieces)
        for r, f, c in shares:
            pieces.append(table_of(r, f, c) if len(r) else None)
        events = [t.cuda.Event() for _ in pieces]

        def shoot():
            main = t.cuda.current_stream(self.device)
            side.wait_stream(main)                 # the tables are uploaded, and the last replay's searches are through with the pool
            with t.cuda.stream(side):
                for piece, ev in zip(pieces, events):
                    if piece is not None:
                        _abi.check(self.lib.ims_shoot_ops_photons(C.byref(piece[0]), piece[1].data_ptr(), C.byref(pool), self._stream()),
                                   "ims_shoot_ops_photons")
                    ev.record(side)
        shoot.keep = (shoot_all.keep, pieces, pool_t)
        shoot.photons, shoot.object_rows = shoot_all.photons, shoot_all.object_rows
        shoot.waves = 4 * sum(p[3] for p in pieces if p is not None)
        shoot.launches = sum(1 for p in pieces if p is not None)
        out = []
        for k, acc in enumerate(launches):
            def accumulate(k=k, acc=acc):
                main = t.cuda.current_stream(self.device)
                for ev in events[:n_up] + [events[n_up + k]]:
                    main.wait_event(ev)
                acc()
            accumulate.keep = acc
            out.append(accumulate)
        return shoot, out

    def accumulate_segments(self, pool, realized=None, small=False):
        """ims_accumulate_segments on a converted pool (segment-mapped: one workgroup per 256 photons of one object);
        small=True: ims_accumulate_small (one wavefront per object row, any photon count)."""
        self._need_static("accumulate_segments")
        P = self._pool_params(pool, realized)
        P.seg_object = _seg_ptr(pool.seg_prefix_dev)
        ph = pool.struct()
        ph.converted = 1 if getattr(pool, "converted", False) else 0
        entry = self.lib.ims_accumulate_small if small else self.lib.ims_accumulate_segments
        _abi.check(entry(C.byref(P), C.byref(ph), pool.photon_offset_dev.data_ptr(), self._num_vertices(), self._stream()),
                   "ims_accumulate_small" if small else "ims_accumulate_segments")

    def _pool_params(self, pool, realized=None):
        return self.bound.params(pool.objects_dev.data_ptr(), pool.n_objects, pool.seg_prefix_dev.data_ptr(),
                                 pool.n_segments, self.image.data_ptr(),
                                 realized.data_ptr() if realized is not None else None)

    def apply_ops(self, pool):
        P = self._pool_params(pool)
        ph = pool.struct()
        _abi.check(self.lib.ims_apply_ops(C.byref(P), pool.photon_offset_dev.data_ptr(), C.byref(ph), self._stream()), "ims_apply_ops")

    def accumulate(self, pool, realized=None, want_pixel_index=False, bf_tag=0):
        """bf_tag (1..255): mark the 16x16 tiles that receive delta charge so that the next
        update_distortions(..., bf_tag=same) only visits tiles within reach of that charge."""
        self._need_static("accumulate")
        P = self._pool_params(pool, realized)
        P.bf_tag = bf_tag
        ph = pool.struct()
        pix = None
        if want_pixel_index:
            pix = self.torch.empty(max(pool.n, 1), dtype=self.torch.int32, device=self.device)
        _abi.check(self.lib.ims_accumulate(C.byref(P), pool.photon_offset_dev.data_ptr(), C.byref(ph),
                                           pix.data_ptr() if pix is not None else None, self._stream()), "ims_accumulate")
        return pix[:pool.n] if pix is not None else None

    # -- sensor state --
    def init_boundaries(self, first_slot, n_slots, stream=None):
        prefix, prefix_t = self._tile_prefix(first_slot)
        _abi.check(self.lib.ims_sensor_init_boundaries(self.bound.sensor_dev_ptr, C.byref(self.bound.sensor_host),
                                                       first_slot, n_slots, prefix_t.data_ptr(), int(prefix[n_slots]),
                                                       stream if stream is not None else self._stream()),
                   "ims_sensor_init_boundaries")

    def _tile_prefix(self, first_slot):
        """Device prefix sum of 16x16 owner-cell tiles over slots [first_slot, n_bf_slots) (cached
        until the slot table changes)."""
        b = self.bound
        sl = b._slots_host[first_slot:b.sensor_host.n_bf_slots]
        table_key = b._slots_host[:b.sensor_host.n_bf_slots].tobytes()
        cache = getattr(self, "_tile_cache", None)
        if cache is None or cache[0] != table_key:
            cache = self._tile_cache = (table_key, {})
        if first_slot not in cache[1]:
            tiles = ((sl["nx"].astype(np.int64) + 1 + 15) // 16) * ((sl["ny"].astype(np.int64) + 1 + 15) // 16)
            prefix = np.concatenate([[0], np.cumsum(tiles)]).astype(np.int64)
            cache[1][first_slot] = (prefix, self.mem.put(prefix)[0])
        return cache[1][first_slot]

    def delta_tensor(self, slot=0):
        """f64 device view of a slot's delta-charge image ((nx+1)*(ny+1) owner cells)"""
        if slot == 0:
            self._need_static("delta_tensor(0)")
        sl = self.bound._slots_host[slot]
        n = (int(sl["nx"]) + 1) * (int(sl["ny"]) + 1)
        off = int(sl["offset"])
        return self.bound.sensor_arrays["delta"].view(self.torch.float64)[off:off + n]

    def update_distortions(self, first_slot, n_slots, stream=None, bf_tag=0, fold=False):
        """fold (slot 0 alone, photon pooling): Silicon's `target += delta` -- the delta charge the recalculation consumes is added
        to self.image (the batch's launches deposited into the delta image only: ims_render_params_t.track_static_delta 2)."""
        if first_slot == 0:
            self._need_static("update_distortions(0, ..)")
        if not hasattr(self, "_changed"):
            cells = self.bound.static_cells + int(self.bound.scratch_cells)
            self._changed = self.torch.zeros(max(cells, 1), dtype=self.torch.uint8, device=self.device)
        prefix, prefix_t = self._tile_prefix(first_slot)
        if fold:
            if first_slot != 0 or n_slots != 1:
                raise ValueError("update_distortions(fold=True) is the recalculation of slot 0 alone")
            _abi.check(self.lib.ims_sensor_update_distortions_fold(self.bound.sensor_dev_ptr, C.byref(self.bound.sensor_host),
                                                                   prefix_t.data_ptr(), int(prefix[1]), self._changed.data_ptr(), bf_tag,
                                                                   self.image.data_ptr(), self.scene.nx, self.scene.ny,
                                                                   stream if stream is not None else self._stream()),
                       "ims_sensor_update_distortions_fold")
            return
        _abi.check(self.lib.ims_sensor_update_distortions(self.bound.sensor_dev_ptr, C.byref(self.bound.sensor_host),
                                                          first_slot, n_slots, prefix_t.data_ptr(), int(prefix[n_slots]),
                                                          self._changed.data_ptr(), bf_tag,
                                                          stream if stream is not None else self._stream()),
                   "ims_sensor_update_distortions")

    def fold_delta(self, stream=None):
        """ims_sensor_fold_delta: the delta charge of slot 0 that no recalculation has consumed yet into self.image (the end of a
        pooled render whose launches deposited into the delta image only)."""
        _abi.check(self.lib.ims_sensor_fold_delta(self.bound.sensor_dev_ptr, C.byref(self.bound.sensor_host), 0, self.image.data_ptr(),
                                                  self.scene.nx, self.scene.ny, stream if stream is not None else self._stream()),
                   "ims_sensor_fold_delta")

    def image_float(self):
        """float32 device tensor of the CCD image (what the reference's ImageF holds)"""
        out = self.torch.empty((self.scene.ny, self.scene.nx), dtype=self.torch.float32, device=self.device)
        _abi.check(self.lib.ims_image_to_float(self.image.data_ptr(), out.data_ptr(), out.numel(), self._stream()),
                   "ims_image_to_float")
        return out

    def image_numpy(self):
        return self.image_float().cpu().numpy()

    def image64_numpy(self):
        """the f64 accumulation image on the host (checkpoint records)"""
        return self.image.cpu().numpy()

    def set_image64(self, arr):
        self.image.copy_(self.torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64)))

    def synchronize(self):
        self.torch.cuda.synchronize(self.device)
