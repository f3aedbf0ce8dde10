"""ORACLE -- TEST INFRASTRUCTURE ONLY.

Python binding of oracle/liboracle.so (the CPU restatement).  Imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never by anything under imsim_amd/.
It reuses the product's pure-host struct builders (imsim_amd.engine.BoundScene over HostMem) so
both sides are fed byte-identical parameter blocks.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from imsim_amd import _abi
from imsim_amd.engine import BoundScene, HostMem, segment_prefix, plan_bf_groups
from imsim_amd._abi import OBJECT_DTYPE, Photons

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
_lib = None

PHOTON_FIELDS = ("x", "y", "flux", "dxdz", "dydz", "wavelength", "pupil_u", "pupil_v", "time")


def build():
    subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        lib = C.CDLL(_LIB)
        for k, st in enumerate(_abi.STRUCTS):
            assert lib.orc_struct_size(k) == C.sizeof(st), st.__name__
        lib.orc_render_objects.argtypes = [C.POINTER(_abi.RenderParams), C.c_int64, C.c_void_p, C.c_void_p]
        lib.orc_shoot_pool.argtypes = [C.POINTER(_abi.RenderParams), C.c_void_p, C.POINTER(Photons)]
        lib.orc_apply_op.argtypes = [C.POINTER(_abi.RenderParams), C.c_int, C.POINTER(Photons), C.c_void_p]
        lib.orc_accumulate_range.argtypes = [C.POINTER(_abi.RenderParams), C.POINTER(Photons), C.c_void_p,
                                             C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.orc_sensor_init_boundaries.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.orc_sensor_update_distortions.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.orc_test_math.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64]
        lib.orc_fft_kspace_fill.argtypes = [C.POINTER(_abi.FftParams), C.c_void_p, C.c_int64, C.c_void_p]
        lib.orc_fft_finish.argtypes = [C.POINTER(_abi.FftParams), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.orc_test_poisson.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_uint64, C.c_int64]
        lib.orc_fft_spikes.argtypes = [C.POINTER(_abi.FftParams), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        lib.orc_test_stencil.argtypes = [C.POINTER(_abi.Spikes), C.c_int, C.c_void_p]
        lib.orc_test_gauss.argtypes = [C.c_uint64, C.c_int64, C.c_int64, C.c_int64, C.c_uint32, C.c_void_p]
        lib.orc_sensor_pixel_areas.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        lib.orc_flat_add.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_uint64, C.c_int64, C.c_int32, C.c_int32,
                                     C.c_void_p, C.c_void_p]
        lib.orc_readout_bleed.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_int32, C.c_void_p]
        lib.orc_readout_segments.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(_abi.Readout), C.c_void_p, C.c_void_p]
        lib.orc_readout_cte.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(_abi.Readout), C.c_void_p, C.c_int32, C.c_int32]
        lib.orc_readout_finish.argtypes = [C.c_void_p, C.POINTER(_abi.Readout), C.c_uint64, C.c_void_p]
        lib.orc_build_object_table.argtypes = [C.POINTER(_abi.Catalog), C.c_void_p, C.c_void_p, C.c_void_p]
        lib.orc_fill_derived_op.argtypes = [C.c_void_p]
        lib.orc_fill_derived_medium.argtypes = [C.c_int32, C.POINTER(C.c_double)]
        _lib = lib
    return _lib


def math_probe(which, x):
    """Evaluate the spec's elementary function `which` (see ims_test_math) on the CPU."""
    lib = load()
    x = np.ascontiguousarray(x, dtype=np.float64)
    n = x.size // 2 if which == 12 else x.size          # 12: (w0, w1) pairs in, one deviate out
    m = 2 if which in (2, 4, 10) else 1
    out = np.empty(n * m)
    lib.orc_test_math(which, x.ctypes.data, out.ctypes.data, n)
    return out


def gauss_probe(seed, obj, n, slot):
    lib = load()
    out = np.empty(2 * n)
    lib.orc_test_gauss(seed, obj, 0, n, slot, out.ctypes.data)
    return out


class HostPool:
    def __init__(self, n):
        self.n = int(n)
        self.a = {f: np.zeros(max(self.n, 1)) for f in PHOTON_FIELDS}
        self.obj_index = np.zeros(max(self.n, 1), dtype=np.int32)

    def struct(self):
        ph = Photons()
        ph.n = self.n
        for f in PHOTON_FIELDS:
            setattr(ph, f, self.a[f].ctypes.data)
        ph.obj_index = self.obj_index.ctypes.data
        return ph

    def to_host(self):
        out = {f: self.a[f][:self.n].copy() for f in PHOTON_FIELDS}
        out["obj_index"] = self.obj_index[:self.n].copy()
        return out


class _OracleDerive:
    """the oracle's own restatement of ims_fill_derived_op / ims_fill_derived_medium"""

    def __init__(self, lib):
        self.lib = lib

    def fill_derived_op(self, op_ref):
        self.lib.orc_fill_derived_op(op_ref)

    def fill_derived_medium(self, kind, c):
        self.lib.orc_fill_derived_medium(int(kind), c)

    def fill_derived_struct(self, what, struct):
        """The oracle reads none of the derived optics / atmosphere / sensor constants: it forms every product, square
        and reciprocal in place, which is what checks the host-side values the kernels consume."""


class OracleScene:
    """CPU counterpart of imsim_amd.engine.Renderer."""

    def __init__(self, scene):
        self.lib = load()
        self.scene = scene
        self.mem = HostMem()
        self.bound = BoundScene(scene, self.mem, _OracleDerive(self.lib))
        self.image64 = np.zeros((scene.ny, scene.nx), dtype=np.float64)      # f64 accumulation, as on the GPU
        if scene.sensor is not None:
            self.init_boundaries(0, len(scene.sensor.slots))

    @property
    def image(self):
        """the float32 CCD image (galsim.ImageF) rounded from the f64 accumulation"""
        return self.image64.astype(np.float32)

    def image64_numpy(self):
        return self.image64.copy()

    def set_image64(self, arr):
        self.image64[...] = arr

    def _objects(self, objects):
        objects = np.ascontiguousarray(objects, dtype=OBJECT_DTYPE)
        prefix = segment_prefix(objects["n_phot"], self.scene.seg_size)
        return objects, prefix

    def render(self, objects, nrecalc=0, realized=None):
        objects, prefix = self._objects(objects)
        P = self.bound.params(objects.ctypes.data, len(objects), prefix.ctypes.data, int(prefix[-1]),
                              self.image64.ctypes.data)
        rc = self.lib.orc_render_objects(C.byref(P), int(nrecalc), self.image64.ctypes.data,
                                         realized.ctypes.data if realized is not None else None)
        assert rc == 0

    def render_lsst_image(self, objects, nrecalc=None, realized=None):
        """CPU counterpart of Renderer.render_lsst_image: the reference's order, one object at a
        time, recalculating that object's pixel boundaries every `nrecalc` photons."""
        ss = self.scene.sensor
        objects = np.ascontiguousarray(objects, dtype=OBJECT_DTYPE)
        if ss is None:
            return self.render(objects, 0, realized)
        if nrecalc is None:
            nrecalc = ss.model.nrecalc
        b = self.bound
        normal, groups = plan_bf_groups(objects, nrecalc, b.n_static_slots, b.static_cells, ss.scratch_cells,
                                        b.slot_capacity)

        def run(part, index):
            tmp = np.zeros(len(part)) if realized is not None else None
            self.render(part, nrecalc, tmp)
            if realized is not None:
                np.add.at(realized, index, tmp)

        if len(normal):
            part = objects[normal].copy()
            part["bf_state"] = 0
            run(part, normal)
        for idx, slots in groups:
            b.set_private_slots(slots)
            n0 = b.n_static_slots
            self.init_boundaries(n0, len(slots))
            part = objects[idx].copy()
            part["bf_state"] = n0 + np.arange(len(idx))
            run(part, idx)

    def shoot_pool(self, objects):
        objects, prefix = self._objects(objects)
        offs = np.concatenate([[0], np.cumsum(objects["n_phot"])]).astype(np.int64)
        pool = HostPool(offs[-1])
        pool.objects, pool.prefix, pool.offs = objects, prefix, offs
        P = self._pool_params(pool)
        ph = pool.struct()
        self.lib.orc_shoot_pool(C.byref(P), offs.ctypes.data, C.byref(ph))
        return pool

    def _pool_params(self, pool, image_ptr=None):
        return self.bound.params(pool.objects.ctypes.data, len(pool.objects), pool.prefix.ctypes.data,
                                 int(pool.prefix[-1]), image_ptr)

    def apply_ops(self, pool):
        P = self._pool_params(pool)
        ph = pool.struct()
        for k in range(len(self.scene.ops)):
            self.lib.orc_apply_op(C.byref(P), k, C.byref(ph), pool.offs.ctypes.data)

    def accumulate(self, pool, realized=None, want_pixel_index=False, bf_tag=0):
        img = np.zeros((self.scene.ny, self.scene.nx), dtype=np.float64)
        P = self._pool_params(pool, img.ctypes.data)
        ph = pool.struct()
        pix = np.empty(max(pool.n, 1), dtype=np.int32) if want_pixel_index else None
        self.lib.orc_accumulate_range(C.byref(P), C.byref(ph), pool.offs.ctypes.data, 0, pool.n, img.ctypes.data,
                                      realized.ctypes.data if realized is not None else None,
                                      pix.ctypes.data if pix is not None else None)
        self.image64 += img
        return pix[:pool.n] if pix is not None else None

    def init_boundaries(self, first_slot, n_slots):
        self.lib.orc_sensor_init_boundaries(self.bound.sensor_dev_ptr, first_slot, n_slots)

    # the names Renderer uses, so that host logic written against the engine (imsim_amd.photon_pooling.build_image)
    # can be exercised on CPU-only machines with the oracle standing in (tests/test_multi_gpu_gloo.py)
    def shoot_photons(self, objects):
        return self.shoot_pool(objects)

    def delta_tensor(self, slot=0):
        import torch
        sl = self.bound._slots_host[slot]
        n = (int(sl["nx"]) + 1) * (int(sl["ny"]) + 1)
        off = int(sl["offset"])
        return torch.from_numpy(self.bound.sensor_arrays["delta"].view(np.float64))[off:off + n]

    def update_distortions(self, first_slot, n_slots, bf_tag=0):
        self.lib.orc_sensor_update_distortions(self.bound.sensor_dev_ptr, first_slot, n_slots)

    def build_flat(self, counts_per_pixel, max_counts_per_iter, seed=0, base=None):
        """CPU counterpart of imsim_amd.flat.LSST_FlatBuilder.build_image (imsim/flat.py:133-268, area branch)"""
        import math
        sc = self.scene
        niter = int(math.ceil(counts_per_pixel / max_counts_per_iter))
        counts_per_iter = counts_per_pixel / niter
        n = sc.nx * sc.ny
        level = counts_per_iter
        b = None
        if base is not None:
            b = np.ascontiguousarray(base, dtype=np.float64)
            level = counts_per_iter / float(b.mean())
        silicon = sc.sensor is not None
        area = np.empty(n, dtype=np.float64)
        for it in range(niter):
            if silicon:
                acc = np.zeros(1, dtype=np.int64)
                self.lib.orc_sensor_pixel_areas(self.bound.sensor_dev_ptr, 0, area.ctypes.data, acc.ctypes.data)
                mean_area = float(int(acc[0])) / float(n) * 2.0 ** -32
                delta = self.bound.sensor_arrays["delta"].view(np.float64)
                self.lib.orc_flat_add(area.ctypes.data, b.ctypes.data if b is not None else None, level, 1.0 / mean_area,
                                      seed, it, sc.nx, sc.ny, self.image64.ctypes.data, delta.ctypes.data)
                if it + 1 < niter:
                    self.update_distortions(0, 1)
            else:
                self.lib.orc_flat_add(None, b.ctypes.data if b is not None else None, level, 1.0, seed, it, sc.nx, sc.ny,
                                      self.image64.ctypes.data, None)
        b = int(getattr(sc, "flat_buffer", 0))
        return self.image64[b:sc.ny - b, b:sc.nx - b] if b else self.image64

    def sensor_array(self, name):
        return self.bound.sensor_arrays[name].view(np.float64)


def poisson_probe(mean, seed=1, obj_id=0):
    lib = load()
    mean = np.ascontiguousarray(mean, dtype=np.float64)
    out = np.empty_like(mean)
    lib.orc_test_poisson(mean.ctypes.data, out.ctypes.data, mean.size, seed, obj_id)
    return out


class OracleFft:
    """CPU counterpart of imsim_amd.fft_draw.FftDrawer (numpy.fft for the transform)."""

    def __init__(self, scene, kpsf, sersic_indices=(1.0, 4.0), add_noise=True, diffraction_fft=None, wavelength=622.2,
                 extra_ktables=()):
        from imsim_amd import fft_draw, tables
        self.lib = load()
        self.scene = scene
        self.keep = []
        tabs = [tables.sersic_ktable(n) for n in sersic_indices]
        more = [tables.sersic_ktable(n)[1] for n in (getattr(scene, "sersic_extra_n", ()) or ())]
        extra_ktables = list(extra_ktables) + more

        def put(a):
            a = np.ascontiguousarray(a, dtype=np.float64)
            self.keep.append(a)
            return a, a.ctypes.data
        self.P, _ = fft_draw.fft_params(scene, kpsf, np.stack([t[1] for t in tabs] + list(extra_ktables)),
                                        float(tabs[0][0][1] - tabs[0][0][0]),
                                        scene.seed, add_noise, put)
        self.image = np.zeros((scene.ny, scene.nx), dtype=np.float64)
        fft_draw.set_spikes(self.P, diffraction_fft, wavelength)

    def spikes(self, rows, rbuf):
        rows = np.ascontiguousarray(rows, dtype=_abi.FFT_OBJECT_DTYPE)
        rbuf = np.ascontiguousarray(rbuf, dtype=np.float64)
        out = np.empty_like(rbuf)
        self.lib.orc_fft_spikes(C.byref(self.P), rows.ctypes.data, len(rows), rbuf.ctypes.data, out.ctypes.data)
        return out

    def fill(self, rows):
        rows = np.ascontiguousarray(rows, dtype=_abi.FFT_OBJECT_DTYPE)
        nfft = rows["nfft"].astype(np.int64)
        kbuf = np.zeros(int(np.sum(nfft * (nfft // 2 + 1))), dtype=np.complex128)
        self.lib.orc_fft_kspace_fill(C.byref(self.P), rows.ctypes.data, len(rows), kbuf.ctypes.data)
        return kbuf

    def inverse(self, rows, kbuf):
        out = []
        for o in rows:
            n = int(o["nfft"])
            nh = n // 2 + 1
            spec = kbuf[int(o["k_offset"]):int(o["k_offset"]) + n * nh].reshape(n, nh)
            out.append(np.fft.irfft2(spec, s=(n, n)).ravel())
        return np.concatenate(out)

    def finish(self, rows, rbuf, realized=None):
        rows = np.ascontiguousarray(rows, dtype=_abi.FFT_OBJECT_DTYPE)
        rbuf = np.ascontiguousarray(rbuf, dtype=np.float64)
        self.lib.orc_fft_finish(C.byref(self.P), rows.ctypes.data, len(rows), rbuf.ctypes.data, self.image.ctypes.data,
                                realized.ctypes.data if realized is not None else None)


# ---------------------------------------------------------------------------------------------
# CCD readout (oracle/orc_readout.c)
# ---------------------------------------------------------------------------------------------
def bleed_eimage(eimage, full_well, midline_stop=True):
    """bleed_trails.bleed_eimage on a float64 [ny][nx] array (returns a new array)."""
    lib = load()
    img = np.ascontiguousarray(eimage, dtype=np.float64).copy()
    ny, nx = img.shape
    flags = np.zeros(nx * ny, dtype=np.uint8)
    lib.orc_readout_bleed(img.ctypes.data, nx, ny, float(full_well), int(bool(midline_stop)), flags.ctypes.data)
    return img


def readout_chain(eimage, ro, full_well, midline_stop, dark_level, dark_stream, seed, pcte_band, scte_band, stages=None):
    """The steps of CcdReadout.build_amp_images on host arrays; `ro` is the _abi.Readout descriptor the product
    built.  Returns the int32 segments [n_amps][raw_h][raw_w]; `stages` (a dict) receives the intermediate arrays."""
    lib = load()
    img = bleed_eimage(eimage, full_well, midline_stop)
    ny, nx = img.shape
    if stages is not None:
        stages["bled"] = img.copy()
    lib.orc_flat_add(None, None, float(dark_level), 1.0, int(seed), int(dark_stream), nx, ny, img.ctypes.data, None)
    if stages is not None:
        stages["dark"] = img.copy()
    shape = (ro.n_amps, ro.raw_h, ro.raw_w)
    a = np.zeros(shape, dtype=np.float32)
    b = np.zeros(shape, dtype=np.float32)
    scratch = np.zeros(ro.n_amps * ro.seg_w * ro.seg_h, dtype=np.float32)
    lib.orc_readout_segments(img.ctypes.data, nx, ny, C.byref(ro), a.ctypes.data, scratch.ctypes.data)
    if stages is not None:
        stages["segments"] = a.copy()
    for band, axis in ((pcte_band, 0), (scte_band, 1)):
        if band is None:
            continue
        bd = np.ascontiguousarray(band, dtype=np.float64)
        lib.orc_readout_cte(a.ctypes.data, b.ctypes.data, C.byref(ro), bd.ctypes.data, bd.shape[1], axis)
        a, b = b, a
    if stages is not None:
        stages["cte"] = a.copy()
    out = np.zeros(shape, dtype=np.int32)
    lib.orc_readout_finish(a.ctypes.data, C.byref(ro), int(seed), out.ctypes.data)
    return out


def build_object_table(scene, cat, visit, phot_flux=None, **kw):
    """CPU restatement of ims_build_object_table (oracle/orc_catalog.c): (rows, meta) for a catalog dict."""
    from imsim_amd import device_table
    lib = load()
    keep = []

    def ptr_of(name, a, dt):
        a = np.ascontiguousarray(a, dtype=dt)
        keep.append(a)
        return a.ctypes.data
    sersic_index = getattr(scene, "sersic_index", None)
    cols = device_table.catalog_columns(cat, phot_flux, sersic_index, kw.pop("stamp_size", None))
    st = device_table.fill_catalog_struct(cols, ptr_of, scene.seed, visit, sersic_index=sersic_index, **kw)
    n = len(cat["x"])
    rows = np.zeros(n, dtype=OBJECT_DTYPE)
    meta = np.zeros(n, dtype=_abi.META_DTYPE)
    optics = scene.optics
    assert lib.orc_build_object_table(C.byref(st), C.addressof(optics), rows.ctypes.data, meta.ctypes.data) == 0
    return rows, meta
