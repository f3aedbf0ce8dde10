"""The reference's per-call object contracts over the GPU engine, for callers that use imSim's classes directly (as its own
tests do) instead of the batch entry points:

  * `PhotonOp.applyTo(photon_array, local_wcs=None, rng=None)` -- imsim/photon_ops.py:81 (RubinOptics), :304 (RubinDiffraction),
    :192 (RubinDiffractionOptics), :520 (BandpassRatio), and GalSim's TimeSampler, PupilAnnulusSampler, PhotonDCR, FocusDepth,
    Refraction as configured at config/imsim-config.yaml:281-320: the photon array's fields go up, ONE launch of `ims_apply_ops`
    applies the operator, the fields come back in place;
  * `sensor.accumulate(photons, image, orig_center=None, resume=False, recalc=True)` -- GalSim's SiliconSensor as called at
    imsim/stamp.py:558-573 and imsim/photon_pooling.py:195-225 -- over `ims_accumulate` / `ims_sensor_update_distortions`;
  * `LSST_SiliconBuilder.draw(prof, image, method, offset, config, base, logger)` -- imsim/stamp.py:411: one object through
    `ims_shoot_accumulate` (or the FFT branch) into the caller's stamp image.

These are convenience mirrors: every call uploads and downloads its arrays, so they cost milliseconds where the batch path
(engine.Renderer, lsst_image.draw_job) costs microseconds per object.  Random streams: GalSim hands an `rng`; here a deviate
is a function of (seed, object id, photon index), so `rng` may be an int seed, an object with `.raw()` (GalSim's BaseDeviate)
or None (seed 0); photon k of the array is photon index k of object `obj_id`.
"""
import math

import numpy as np

from . import _abi
from ._abi import OBJECT_DTYPE

FIELDS = ("x", "y", "flux", "dxdz", "dydz", "wavelength", "pupil_u", "pupil_v", "time")


class PhotonArray:
    """The fields of galsim.PhotonArray as float64 numpy arrays (imsim/photon_ops.py reads and writes them by these names)."""

    def __init__(self, n, **fields):
        self._n = int(n)
        for f in FIELDS:
            v = fields.get(f)
            setattr(self, f, np.zeros(self._n) if v is None else np.ascontiguousarray(v, dtype=np.float64).copy())
        self._pupil = "pupil_u" in fields
        self._times = "time" in fields

    def size(self):
        return self._n

    def __len__(self):
        return self._n

    def hasAllocatedPupil(self):
        return self._pupil

    def hasAllocatedTimes(self):
        return self._times

    def hasAllocatedAngles(self):
        return True

    def hasAllocatedWavelengths(self):
        return True


def seed_of(rng):
    if rng is None:
        return 0
    if isinstance(rng, (int, np.integer)):
        return int(rng)
    if hasattr(rng, "raw"):
        return int(rng.raw())
    raise TypeError("rng: an int seed, an object with .raw() (galsim.BaseDeviate), or None")


def _one_row(n, obj_id, nx, ny, xmin, ymin, center=(0.0, 0.0), winv=None, dcr=(0.0, 0.0, 1.0)):
    o = np.zeros(1, dtype=OBJECT_DTYPE)
    o["obj_id"], o["n_phot"], o["flux_per_photon"] = obj_id, n, 1.0
    o["x0"], o["y0"] = center
    o["jac"] = (1.0, 0.0, 0.0, 1.0)
    o["winv"] = winv if winv is not None else (5.0, 0.0, 0.0, 5.0)
    o["dcr_tanz"], o["dcr_sinp"], o["dcr_cosp"] = dcr
    o["prof_table"] = _abi.IMS_PROF_POINT
    o["sed_table"] = -1
    o["stamp_xmin"], o["stamp_xmax"], o["stamp_ymin"], o["stamp_ymax"] = xmin, xmin + nx - 1, ymin, ymin + ny - 1
    return o


def _pool_from(renderer, photons, objects):
    """engine.PhotonPool holding the caller's arrays (one object row owns all photons)"""
    from .engine import PhotonPool
    t = renderer.torch
    n = len(photons.x)
    objects, obj_t, prefix, pre_t = renderer._upload_objects(objects)
    off_t = t.from_numpy(np.array([0, n], dtype=np.int64)).to(renderer.device)
    pool = PhotonPool(t, renderer.device, n, off_t, obj_t, 1, pre_t, int(prefix[-1]))
    for f in FIELDS:
        pool.t[f][:n].copy_(t.from_numpy(np.ascontiguousarray(getattr(photons, f), dtype=np.float64)))
    pool.obj_index.zero_()
    return pool


def _write_back(pool, photons, fields=FIELDS):
    n = len(photons.x)
    for f in fields:
        getattr(photons, f)[...] = pool.t[f][:n].cpu().numpy()


class DevicePhotonOp:
    """One registered photon operator (or a chain of them) on the GPU.

    ops: engine.Scene.ops tuples (config.build_photon_ops); optics: _abi.Optics for the Rubin operators (telescope, the image
    WCS and icrf_to_field, spider geometry); ratio: (tables, wl_min, wl_step) of BandpassRatio."""

    def __init__(self, ops, optics=None, ratio=None, nx=4096, ny=4096, device="cuda:0", stamp_center=None, obj_id=0,
                 dcr=(0.0, 0.0, 1.0)):
        from .engine import Scene
        self.scene = Scene(nx=nx, ny=ny, seed=0, psf=[], ops=list(ops), optics=optics)
        if ratio is not None:
            self.scene.ratio_tables, self.scene.ratio_wl_min, self.scene.ratio_wl_step = ratio
        self.device, self.stamp_center, self.obj_id, self.dcr = device, stamp_center, int(obj_id), dcr
        self._renderer = None
        self.needs_pupil = any(int(op[0]) in (_abi.IMS_OP_RUBIN_OPTICS, _abi.IMS_OP_RUBIN_DIFFRACTION, _abi.IMS_OP_RUBIN_DIFFRACTION_OPTICS)
                               for op in ops)

    def renderer(self):
        if self._renderer is None:
            from .engine import Renderer
            self._renderer = Renderer(self.scene, self.device)
        return self._renderer

    def applyTo(self, photon_array, local_wcs=None, rng=None):
        """In place on x, y, flux, dxdz, dydz, pupil_u, pupil_v, time (imsim/photon_ops.py:81-127).  local_wcs: None, or the
        inverse local jacobian [pixels / arcsec] as (dxdu, dxdv, dydu, dydv) (what PhotonDCR converts its shift with)."""
        if self.needs_pupil:                                        # imsim/photon_ops.py:139-140, :327-328
            assert photon_array.hasAllocatedPupil()
            assert photon_array.hasAllocatedTimes()
        r = self.renderer()
        n = len(photon_array.x)
        if n == 0:
            return
        r.bound.base_params.seed = seed_of(rng)
        sc = self.scene
        row = _one_row(n, self.obj_id, sc.nx, sc.ny, sc.xmin, sc.ymin, self.stamp_center or (0.0, 0.0), local_wcs, self.dcr)
        pool = _pool_from(r, photon_array, row)
        r.apply_ops(pool)
        r.synchronize()
        _write_back(pool, photon_array, ("x", "y", "flux", "dxdz", "dydz", "pupil_u", "pupil_v", "time"))


def _named_op(name):
    """class RubinOptics / RubinDiffraction / RubinDiffractionOptics / BandpassRatio with the reference's constructor surface
    reduced to what this engine needs"""
    kind = {"RubinOptics": _abi.IMS_OP_RUBIN_OPTICS, "RubinDiffraction": _abi.IMS_OP_RUBIN_DIFFRACTION,
            "RubinDiffractionOptics": _abi.IMS_OP_RUBIN_DIFFRACTION_OPTICS}[name]

    class Op(DevicePhotonOp):
        __doc__ = f"imsim.photon_ops.{name} (imsim/photon_ops.py): optics = telescope + WCS pair + pointing geometry"

        def __init__(self, optics, shift_photons=False, stamp_center=None, disable_field_rotation=False, **kw):
            DevicePhotonOp.__init__(self, [(kind, 0, [1.0 if shift_photons else 0.0, 1.0 if disable_field_rotation else 0.0])],
                                    optics=optics, stamp_center=stamp_center, **kw)
    Op.__name__ = Op.__qualname__ = name
    return Op


RubinOptics = _named_op("RubinOptics")
RubinDiffraction = _named_op("RubinDiffraction")
RubinDiffractionOptics = _named_op("RubinDiffractionOptics")


class BandpassRatio(DevicePhotonOp):
    """imsim.photon_ops.BandpassRatio (imsim/photon_ops.py:506-533): flux *= (target / initial)(wavelength)"""

    def __init__(self, target_bandpass, initial_bandpass, **kw):
        DevicePhotonOp.__init__(self, [(_abi.IMS_OP_BANDPASS_RATIO, 0, [])], ratio=target_bandpass.ratio_table(initial_bandpass), **kw)


class SiliconSensor:
    """galsim.SiliconSensor as imSim uses it (config/imsim-config.yaml:230-235; imsim/stamp.py:558-573;
    imsim/photon_pooling.py:195-225): `accumulate(photons, image, orig_center=None, resume=False, recalc=True)` lands the
    photons in `image` (a float numpy array [ny][nx] whose pixel (0, 0) is image pixel (xmin, ymin)) through the distorted
    pixel boundaries; with resume the charge of the previous calls keeps shaping them -- recalculated when `recalc` is set,
    as GalSim does for the first sub-batch of a pooled batch.  Returns the flux added."""

    def __init__(self, setup, nx, ny, xmin=1, ymin=1, device="cuda:0", has_angles=True, obj_id=0):
        """setup: engine.SensorSetup for the image region (configs.silicon_setup); has_angles: the photons' dxdz / dydz are
        meaningful (they went through a ray-tracing operator)"""
        from .engine import Scene
        # accumulate only asks "did the photons go through a ray trace" (then dxdz / dydz carry their inclination into the silicon):
        # the operator is named, with a telescope descriptor as the library's argument check wants one, and never applied
        ops, optics = [], None
        if has_angles:
            from . import configs
            ops, optics = [(_abi.IMS_OP_RUBIN_OPTICS, 0, [0.0, 0.0])], configs.rubin_optics_struct(nx, ny)
        self.scene = Scene(nx=nx, ny=ny, xmin=xmin, ymin=ymin, seed=0, psf=[], ops=ops, sensor=setup, optics=optics)
        self.scene.track_static_delta = 1
        self.device, self.obj_id = device, int(obj_id)
        self._renderer = None
        self._seed = 0
        self._calls = 0

    def renderer(self):
        if self._renderer is None:
            from .engine import Renderer
            self._renderer = Renderer(self.scene, self.device)
        return self._renderer

    def updateRNG(self, rng):
        self._seed = seed_of(rng)

    def accumulate(self, photons, image, orig_center=None, resume=False, recalc=True, photon_first=0):
        r = self.renderer()
        t = r.torch
        sc = self.scene
        image = np.asarray(image)
        if image.shape != (sc.ny, sc.nx):
            raise ValueError(f"image must be [{sc.ny}][{sc.nx}]")
        if not resume:
            r.init_boundaries(0, len(sc.sensor.slots))             # a fresh exposure: undistorted (+ tree ring) pixels
            r.delta_tensor(0).zero_()
        elif recalc and self._calls:
            r.update_distortions(0, 1)                              # the charge since the last recalculation shapes the pixels
        self._calls += 1
        n = len(photons.x)
        if n == 0:
            return 0.0
        r.bound.base_params.seed = self._seed
        row = _one_row(n, self.obj_id, sc.nx, sc.ny, sc.xmin, sc.ymin)
        row["phot_first"] = int(photon_first)
        pool = _pool_from(r, photons, row)
        r.image.zero_()
        real = t.zeros(1, dtype=t.float64, device=r.device)
        r.accumulate(pool, realized=real)
        r.synchronize()
        add = r.image.cpu().numpy()
        if image.dtype in (np.float32, np.float64):
            image += add.astype(image.dtype)
        else:                                                       # other dtypes through a float64 copy (photon_pooling.py:213-225)
            image[...] = (image.astype(np.float64) + add).astype(image.dtype)
        return float(real.item())
