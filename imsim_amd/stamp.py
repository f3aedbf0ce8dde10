"""Stamp layer: the per-object decisions of LSST_SiliconBuilder / LSST_PhotonsBuilder
(imsim/stamp.py) expressed over whole object tables.

The reference decides one object at a time inside GalSim's config machinery; here the same rules
are applied in bulk (numpy) to build the rows the GPU kernels consume:
  * phot_flux ~ Poisson(nominal_flux), skip if 0                      stamp.py:190-202
  * faint: nominal_flux < max_flux_simple -> no photon ops, no sensor stamp.py:435-465, 555-556
  * FFT only if nominal_flux >= 1e6 and fft_sb_thresh set             stamp.py:275-277
"""
import dataclasses
from enum import Enum, auto

import numpy as np


class ProcessingMode(Enum):
    FFT = auto()
    PHOT = auto()
    FAINT = auto()


@dataclasses.dataclass
class ObjectInfo:
    """Quantities of one object needed to pick its rendering mode (imsim/stamp.py:23-33)."""
    index: int
    phot_flux: float
    mode: ProcessingMode


def classify(nominal_flux, max_flux_simple=100.0, fft_sb_thresh=0.0, max_sb=None):
    """Vectorised build_obj mode decision (imsim/stamp.py:85-91, :275-277; psf_utils.py:212):
    FFT when the object is bright enough AND its peak surface brightness exceeds fft_sb_thresh,
    FAINT below max_flux_simple, PHOT otherwise.  Returns an array of ProcessingMode."""
    nominal_flux = np.asarray(nominal_flux, dtype=np.float64)
    mode = np.full(nominal_flux.shape, ProcessingMode.PHOT, dtype=object)
    mode[nominal_flux < max_flux_simple] = ProcessingMode.FAINT
    if fft_sb_thresh:
        cand = (nominal_flux >= 1.0e6) & (nominal_flux >= fft_sb_thresh)
        if max_sb is not None:
            cand &= np.asarray(max_sb) > fft_sb_thresh
        mode[cand] = ProcessingMode.FFT
    return mode


def object_infos(phot_flux, modes):
    return [ObjectInfo(i, f, m) for i, (f, m) in enumerate(zip(phot_flux, modes))]


class SkipThisObject(Exception):
    """galsim.config.SkipThisObject: the object contributes nothing (imsim/stamp.py:155-156, :199-202)"""


class StampImage:
    """What `draw` needs of a galsim.Image: `array` ([ny][nx], float32 or float64) and the 1-based inclusive `bounds`
    (xmin, xmax, ymin, ymax); `added_flux` is set by draw (image.added_flux, stamp.py:573)."""

    def __init__(self, xmin, xmax, ymin, ymax, dtype=np.float32):
        self.bounds = (int(xmin), int(xmax), int(ymin), int(ymax))
        self.array = np.zeros((self.bounds[3] - self.bounds[2] + 1, self.bounds[1] - self.bounds[0] + 1), dtype=dtype)
        self.added_flux = 0.0


class LSST_SiliconBuilder:
    """`stamp.type: LSST_Silicon` one object at a time -- the reference's StampBuilder contract (imsim/stamp.py:95-575):
    setup -> buildPSF -> getDrawMethod -> draw, with the `base` side channel (nominal_flux, phot_flux, fft_flux, realized_flux).
    The object is a one-row catalog (the columns of catalog.synthetic_catalog / instcat.to_catalog) in base['_object']; the
    render itself is the batch engine on a scene cut down to the stamp: same kernels, same photon streams (keyed by seed and
    object id), so a stamp drawn here equals the object's contribution to a CCD rendered by lsst_image.draw_job bit for bit.
    Every draw builds a small renderer: milliseconds per object -- the batch path is the fast one."""

    def __init__(self, scene, make_objects, kpsf=None, fwhm_total=0.8, fft_sb_thresh=0.0, max_flux_simple=100.0,
                 diffraction_fft=None, wavelength=622.2, extra_ktables=(), nrecalc=None, device="cuda:0"):
        self.scene, self.make_objects = scene, make_objects
        self.kpsf, self.fwhm_total, self.fft_sb_thresh, self.max_flux_simple = kpsf, fwhm_total, fft_sb_thresh, max_flux_simple
        self.diffraction_fft, self.wavelength, self.extra_ktables = diffraction_fft, wavelength, tuple(extra_ktables)
        self.nrecalc, self.device = nrecalc, device

    def setup(self, config, base, xsize=0, ysize=0, ignore=(), logger=None):
        """-> (xsize, ysize, image_pos, world_pos) (imsim/stamp.py:109-249): Poisson realisation of the flux unless
        base['phot_flux'] is given, SkipThisObject for 0 photons, stamp size from get_stamp_size unless given"""
        from . import catalog
        obj = base["_object"]
        nominal = float(np.asarray(obj["nominal_flux"]).reshape(-1)[0])
        phot = base.get("phot_flux")
        if phot is None:
            phot = int(catalog.realize_fluxes(np.array([nominal]), int(base.get("seed", self.scene.seed)))[0])
        if phot <= 0:
            raise SkipThisObject("no photons")
        base["nominal_flux"], base["phot_flux"], base["fft_flux"], base["realized_flux"] = nominal, float(phot), 0.0, 0.0
        cat = {k: (np.atleast_1d(v) if not isinstance(v, (dict, list, tuple, str)) else v) for k, v in obj.items()}
        rows, sizes = self.make_objects(cat, np.array([phot], dtype=np.int64))
        if xsize and ysize:                                            # a given stamp size wins (stamp.py:205-207)
            icx, icy = int(np.floor(rows["x0"][0] + 0.5)), int(np.floor(rows["y0"][0] + 0.5))
            rows["stamp_xmin"], rows["stamp_xmax"] = icx - xsize // 2, icx - xsize // 2 + xsize - 1
            rows["stamp_ymin"], rows["stamp_ymax"] = icy - ysize // 2, icy - ysize // 2 + ysize - 1
        faint = nominal < self.max_flux_simple
        from ._abi import IMS_OBJ_FAINT
        rows["flags"] = np.where(faint, rows["flags"] | IMS_OBJ_FAINT, rows["flags"] & ~IMS_OBJ_FAINT)
        self._rows, self._cat = rows, cat
        xs = int(rows["stamp_xmax"][0] - rows["stamp_xmin"][0] + 1)
        ys = int(rows["stamp_ymax"][0] - rows["stamp_ymin"][0] + 1)
        return xs, ys, (float(rows["x0"][0]), float(rows["y0"][0])), None

    def buildPSF(self, config, base, gsparams=None, logger=None):
        """the FFT-or-photons decision of stamp.py:251-310 (get_fft_psf_maybe): returns the k-space PSF when the object goes
        down the FFT branch, else None; writes base['fft_flux'] and zeroes base['phot_flux'] like :304-305"""
        from . import fft_draw, lsst_image
        cat = self._cat
        nominal = np.array([base["nominal_flux"]])
        is_fft = lsst_image.LSST_ImageBuilderBase._use_fft(cat, nominal, self.fwhm_total, self.fft_sb_thresh, self.kpsf, self.extra_ktables)
        self.use_fft = bool(is_fft[0]) and int(np.asarray(cat["kind"])[0]) < 3
        if self.use_fft:
            base["fft_flux"], base["phot_flux"] = base["nominal_flux"], 0.0
        return self.kpsf if self.use_fft else None

    def getDrawMethod(self, config, base, logger=None):
        """stamp.py:312-336: an explicit stamp.draw_method wins, `auto` follows buildPSF"""
        method = (config or {}).get("draw_method", "auto")
        if method not in ("auto", "fft", "phot"):
            raise ValueError("Invalid draw_method: %s" % method)
        if method == "auto":
            method = "fft" if getattr(self, "use_fft", False) else "phot"
        return method

    def draw(self, prof, image, method, offset, config, base, logger=None):
        """stamp.py:411-575: the object into `image` (add_to_image semantics), returns the image.  prof / offset are carried
        by base['_object'] and the rows of setup; `image.bounds` need not be the stamp's own: the overlap is drawn."""
        import copy
        from . import fft_draw
        from .engine import Renderer, make_slots
        rows = self._rows
        xmin, xmax, ymin, ymax = image.bounds
        nx, ny = xmax - xmin + 1, ymax - ymin + 1
        sc = copy.copy(self.scene)
        sc.nx, sc.ny, sc.xmin, sc.ymin = nx, ny, xmin, ymin
        if sc.sensor is not None:
            sc.sensor = copy.copy(sc.sensor)
            sc.sensor.slots = make_slots([(xmin, ymin, nx, ny)])
            size = int(rows["stamp_xmax"][0] - rows["stamp_xmin"][0] + 2) * int(rows["stamp_ymax"][0] - rows["stamp_ymin"][0] + 2)
            sc.sensor.scratch_cells = size
            sc.sensor.max_slots = 4
        r = Renderer(sc, self.device)
        t = r.torch
        real = t.zeros(1, dtype=t.float64, device=r.device)
        if method == "fft":
            if self.kpsf is None:
                raise ValueError("FFT drawing needs the k-space PSF description")
            flux = np.array([base["fft_flux"] or base["nominal_flux"]])
            frows, _ = fft_draw.build_fft_objects(rows, flux, fft_draw.profile_ktable_ids(sc, rows["prof_table"], len(self.extra_ktables)))
            fft_draw.FftDrawer(r, self.kpsf, add_noise=True, diffraction_fft=self.diffraction_fft, wavelength=self.wavelength,
                               extra_ktables=self.extra_ktables).draw(frows, realized=real)
        else:
            r.render_lsst_image(rows, nrecalc=self.nrecalc, realized=real)
        r.synchronize()
        add = r.image.cpu().numpy()
        image.array += add.astype(image.array.dtype)
        image.added_flux = base["realized_flux"] = float(real.item())
        return image
