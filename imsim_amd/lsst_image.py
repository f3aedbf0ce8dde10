"""Image layer: LSST_ImageBuilder / LSST_PhotonPoolingImageBuilder over the GPU engine
(imsim/lsst_image.py:15-126, :276-395; imsim/photon_pooling.py:29-174).

`setup` keeps the reference's parameter surface (required det_name; optional size, xsize, ysize,
dtype, apply_sky_gradient, apply_fringing, boresight, camera, nbatch, nsubbatch, nbatch_fft,
nbatch_per_checkpoint with the same defaults); `build_image` is the draw loop: every object is
classified FFT / PHOT / FAINT by the reference's rules and rendered into the CCD image.  Sky,
noise, checkpointing and FITS output are out of scope.
"""
import dataclasses
from typing import Optional

import numpy as np

from . import catalog, fft_draw, photon_pooling, stamp
from ._abi import IMS_OBJ_FAINT

IMAGE_REQ = {"det_name": str}
IMAGE_OPT = {"size": int, "xsize": int, "ysize": int, "dtype": None, "apply_sky_gradient": bool,
             "apply_fringing": bool, "boresight": None, "camera": str, "nbatch": int, "nsubbatch": int,
             "nbatch_fft": int, "nbatch_per_checkpoint": int}
IMAGE_IGNORE = ["image_pos", "world_pos", "stamp_size", "stamp_xsize", "stamp_ysize", "nobjects", "type", "random_seed",
                "bandpass", "wcs", "noise", "sky_level", "sensor", "use_flux_sky_areas", "nproc"]
# detector bounding boxes the reference gets from lsst.obs.lsst (imsim/camera.py); E2V 4096x4004, ITL 4072x4000
NOISE_STREAM = 1 << 20            # iteration id space of the sky-noise Poisson streams (flat iterations use 0..niter)
DETECTOR_SIZE = {"E2V": (4096, 4004), "ITL": (4072, 4000)}


class GalSimConfigError(ValueError):
    pass


def get_all_params(config, req, opt, ignore=()):
    """galsim.config.GetAllParams semantics: required keys must be present, unknown keys raise."""
    out = {}
    for k in req:
        if k not in config:
            raise GalSimConfigError(f"Attribute {k} is required")
    for k, v in config.items():
        if k in req or k in opt:
            out[k] = v
        elif k not in ignore and not k.startswith("_"):
            raise GalSimConfigError(f"Unexpected attribute {k} found")
    return out


class LSST_ImageBuilderBase:
    def setup(self, config, det_type="E2V"):
        """Parse the image field (imsim/lsst_image.py:49-126).  Returns (xsize, ysize)."""
        params = get_all_params(config, IMAGE_REQ, IMAGE_OPT, IMAGE_IGNORE)
        size = params.get("size", 0)
        xsize, ysize = params.get("xsize", size), params.get("ysize", size)
        if xsize == 0 or ysize == 0:
            xsize, ysize = DETECTOR_SIZE[det_type]
        self.det_name = params["det_name"]
        self.camera_name = params.get("camera", "LsstCamSim")
        self.apply_fringing = params.get("apply_fringing", False)
        if self.apply_fringing and "boresight" not in params:
            raise GalSimConfigError("Boresight is missing in image config dict. This is required for fringing.")
        self.nbatch = params.get("nbatch", 10)
        self.nbatch_per_checkpoint = params.get("nbatch_per_checkpoint", 1)
        self.nsubbatch = params.get("nsubbatch", 50)
        self.nbatch_fft = params.get("nbatch_fft", 1)
        # image.nobjects: "" (how the reference's tests switch it off, tests/test_lsst_image.py:36) or absent = every object
        nobj = config.get("nobjects")
        self.nobjects = None if nobj is None or (isinstance(nobj, str) and nobj.strip() == "") else int(nobj)
        return xsize, ysize


    @staticmethod
    def _use_fft(sub, nominal, fwhm_total, fft_sb_thresh, kpsf, extra_ktables):
        """FFT or photons for every object (stamp.py:275-277 + get_fft_psf_maybe, psf_utils.py:195-212): GalSim's max_sb
        of Convolve(object, FFT-mode PSF) from the analytic peaks of the components."""
        from . import tables
        kw = {}
        if kpsf:
            q_step = tables.KTABLE_QMAX / (tables.KTABLE_NPTS - 1)
            kw["psf_peaks"] = fft_draw.kpsf_peak_per_flux(kpsf, list(extra_ktables), q_step)
        if sub.get("sersic_n") is not None:
            kw["sersic_n"] = sub["sersic_n"]
        if sub.get("mu") is not None:
            kw["jac_det"] = sub["mu"]                        # shear preserves area; the lens magnifies it by mu
        return fft_draw.use_fft(nominal, sub["kind"], sub["hlr"], fwhm_total, fft_sb_thresh, **kw)

    def sky_pixel_areas(self, renderer, use_flux=False):
        """Pixel areas for drawing the sky on a Silicon sensor (`sensor.calculate_pixel_areas(image, use_flux=...)`, which
        GalSim's sky image is multiplied by; `image.use_flux_sky_areas`, config/imsim-config.yaml:222-228): the polygon areas
        of the CCD's boundary state -- tree rings only by default, or (use_flux) distorted in ONE step by the flux already in
        the image, after which the state is put back (LSST_Image keeps slot 0 static).  Device tensor [ny][nx], or None
        without a Silicon sensor."""
        from . import _abi
        sc = renderer.scene
        if sc.sensor is None:
            return None
        torch = renderer.torch
        b = renderer.bound
        renderer._need_static("sky_pixel_areas")
        sl = b._slots_host[0]
        if (int(sl["xmin"]), int(sl["ymin"]), int(sl["nx"]), int(sl["ny"])) != (sc.xmin, sc.ymin, sc.nx, sc.ny) or int(sl["offset"]) != 0:
            raise ValueError("sky_pixel_areas expects slot 0 to be the whole image")
        if use_flux:
            # the charge of the image as the delta-charge image of slot 0 ((nx + 1) x (ny + 1) owner cells), one recalculation
            delta = renderer.delta_tensor(0).view(sc.ny + 1, sc.nx + 1)
            delta.zero_()
            delta[:sc.ny, :sc.nx] = renderer.image
            renderer.update_distortions(0, 1)
        area = torch.empty(sc.nx * sc.ny, dtype=torch.float64, device=renderer.device)
        acc = torch.zeros(1, dtype=torch.int64, device=renderer.device)
        _abi.check(renderer.lib.ims_sensor_pixel_areas(b.sensor_dev_ptr, _abi.C.byref(b.sensor_host), 0, area.data_ptr(),
                                                       acc.data_ptr(), renderer._stream()), "ims_sensor_pixel_areas")
        if use_flux:
            renderer.init_boundaries(0, 1)                  # back to the undistorted (+ tree ring) state the objects were drawn on
        return area.view(sc.ny, sc.nx)

    def add_noise(self, renderer, sky_level, pixel_scale=0.2, sky_gradient=None, multiplier=None, seed=0, stream_id=0,
                  pixel_areas=None):
        """addNoise (imsim/lsst_image.py:128-200): sky = sky_level [photons/arcsec^2] x pixel area, optionally
        times a linear sky gradient (a, b, c): factor = a + b x + c y in 0-based pixel indices (SkyGradient,
        sky_model.py) and a per-pixel multiplier map [ny][nx] (vignetting x fringing); the image receives a
        Poisson deviate of that expectation per pixel (the objects already carry their own shot noise).  The
        Rubin sky-brightness model that supplies sky_level in the reference is out of scope: pass the number.
        pixel_areas: device tensor [ny][nx] from sky_pixel_areas (a Silicon sensor's pixels collect sky in proportion to
        their area), folded into the multiplier."""
        from . import _abi
        torch = renderer.torch
        sc = renderer.scene
        base_t = None
        if sky_gradient is not None or multiplier is not None or pixel_areas is not None:
            m = torch.ones((sc.ny, sc.nx), dtype=torch.float64, device=renderer.device)
            if pixel_areas is not None:
                m = m * pixel_areas
            if sky_gradient is not None:
                a, b, c = (float(v) for v in sky_gradient)
                xx = torch.arange(sc.nx, dtype=torch.float64, device=renderer.device)[None, :]
                yy = torch.arange(sc.ny, dtype=torch.float64, device=renderer.device)[:, None]
                m = m * (a + b * xx + c * yy)
            if multiplier is not None:
                m = m * torch.as_tensor(np.ascontiguousarray(multiplier, dtype=np.float64), device=renderer.device)
            base_t = m.contiguous()
        level = float(sky_level) * pixel_scale * pixel_scale
        _abi.check(renderer.lib.ims_flat_add(None, base_t.data_ptr() if base_t is not None else None, level, 1.0, int(seed),
                                             NOISE_STREAM + int(stream_id), sc.nx, sc.ny, renderer.image.data_ptr(), None,
                                             renderer._stream()), "ims_flat_add")
        self._noise_keep = base_t
        return renderer.image


@dataclasses.dataclass
class CcdJob:
    """Everything of ONE CCD's `LSST_Image` build that is decided on the host, ahead of the GPU: the rows of the
    photon-shot objects (FAINT flagged), the FFT-drawn ones as FFT rows with what their drawer needs, the optional sky
    stage.  `draw_job` only enqueues -- so a focal plane (focal_plane.render_focal_plane) can prepare the next CCD while
    the GPU works through this one, exactly as it does for a bare object table."""
    objects: "object"                                  # OBJECT_DTYPE rows of the photon-shot objects (may be empty)
    nrecalc: Optional[int] = None
    fft_rows: Optional[np.ndarray] = None              # FFT_OBJECT_DTYPE rows sorted by FFT size (fft_draw.build_fft_objects)
    kpsf: Optional[list] = None                        # k-space PSF of the FFT drawer
    extra_ktables: tuple = ()
    diffraction_fft: "object" = None
    wavelength: float = 622.2
    sky: Optional[dict] = None                         # keyword arguments of add_noise (+ use_flux_sky_areas)
    phot_index: Optional[np.ndarray] = None            # positions of `objects` / of `fft_rows` among the kept catalog rows
    fft_index: Optional[np.ndarray] = None
    n_kept: int = 0
    want_realized: bool = False                        # draw_job allocates `realized` (f64 device tensor over the kept rows) itself
    realized: "object" = None
    host: dict = dataclasses.field(default_factory=dict)   # what build_image needs for the truth record

    @property
    def n_fft(self):
        return 0 if self.fft_rows is None else len(self.fft_rows)


def draw_job(renderer, job, realized=None, fft_stream=None, defer=False):
    """The draw loop of one CCD on the device (imsim/lsst_image.py:342-368 over imsim/stamp.py:411-575), enqueue only: the
    FFT objects first (k-space fill, inverse transforms, spikes, Poisson noise, stamp -> CCD add), then the launch plan of
    the photon-shot ones, then the optional sky.  realized: f64 device tensor over the kept catalog rows.

    fft_stream: a side stream for the FFT objects.  Their stamps carry Poisson noise, so every value added to the f64 CCD image
    is an integer like the photons' and the image does not depend on the order of the additions: the FFT branch (milliseconds
    for a 4096^2 stamp with its spike stencil) may then run BESIDE the launch plan instead of ahead of it on the stream that
    carries a CCD's longest brighter-fatter chain (focal_plane.render_focal_plane); the sky stage waits for both.

    defer (focal_plane's joint mode): the launch plan is enqueued without its join and without the rounds of its top chain;
    returns finish() -- to be called under the stream that carries the CCD's tail, once engine.run_joint_plans has run the
    batch's plans (finish.plan; None if there was nothing to defer): it joins the plan, adds the realized fluxes and the sky."""
    from .engine import upload_async
    torch = renderer.torch
    if realized is None and job.want_realized:
        realized = job.realized = torch.zeros(job.n_kept, dtype=torch.float64, device=renderer.device)
    fft_done = None
    if job.n_fft:
        if job.kpsf is None:
            raise GalSimConfigError("FFT drawing needs the k-space PSF description")
        main = torch.cuda.current_stream(renderer.device)
        side = fft_stream if (fft_stream is not None and fft_stream != main) else None
        # tables, buffers and uploads under the caller's stream (whose cached blocks fit: a buffer allocated under another stream
        # comes out of hipMalloc, a device-wide synchronisation); only the launches go to the side stream
        drawer = fft_draw.FftDrawer(renderer, job.kpsf, add_noise=True, diffraction_fft=job.diffraction_fft,
                                    wavelength=job.wavelength, extra_ktables=job.extra_ktables)
        r_fft = torch.zeros(job.n_fft, dtype=torch.float64, device=renderer.device) if realized is not None else None
        state = drawer._upload(job.fft_rows)
        if side is not None:
            side.wait_stream(main)                                # the renderer's scene tables, its zeroed image, the uploads above
        with torch.cuda.stream(side if side is not None else main):
            drawer._run(state, r_fft)
            if side is not None:
                fft_done = torch.cuda.Event()
                fft_done.record(side)
        renderer._keep_fft = drawer._last + (drawer._keep,)     # the buffers, not the drawer (which refers back to the renderer)
    plan = None
    r_ph = None
    if len(job.objects):
        r_ph = torch.zeros(len(job.objects), dtype=torch.float64, device=renderer.device) if realized is not None else None
        plan = renderer.render_lsst_image(job.objects, nrecalc=job.nrecalc, realized=r_ph, defer=defer)
    front = torch.cuda.current_stream(renderer.device)
    front_done = None
    if defer:
        front_done = torch.cuda.Event()                      # the FFT objects and uploads of the front stream
        front_done.record(front)

    def finish():
        here = torch.cuda.current_stream(renderer.device)
        if plan is not None:
            plan.join()
            if r_ph is not None:
                plan.add_realized(r_ph)
        if front_done is not None and here != front:
            here.wait_event(front_done)
        if len(job.objects) and realized is not None:
            realized.index_add_(0, upload_async(torch, renderer.device, np.ascontiguousarray(job.phot_index, dtype=np.int64)), r_ph)
        if fft_done is not None:
            here.wait_event(fft_done)
        if job.n_fft and realized is not None:
            realized.index_add_(0, upload_async(torch, renderer.device, np.ascontiguousarray(job.fft_index, dtype=np.int64)), r_fft)
        if job.sky is not None:
            kw = dict(job.sky)
            base = LSST_ImageBuilderBase()
            areas = base.sky_pixel_areas(renderer, use_flux=bool(kw.pop("use_flux_sky_areas", False))) if kw.pop("pixel_areas", True) else None
            base.add_noise(renderer, kw.pop("sky_level"), pixel_areas=areas, **kw)
            renderer._keep_sky = base
        return renderer.image
    if defer:
        finish.plan = plan
        return finish
    return finish()


class LSST_ImageBuilder(LSST_ImageBuilderBase):
    """`image.type: LSST_Image` with `stamp.type: LSST_Silicon`."""

    def prepare(self, scene, cat, phot_flux, make_objects, fft_sb_thresh=0.0, max_flux_simple=100.0,
                draw_method="auto", kpsf=None, fwhm_total=0.8, diffraction_fft=None, wavelength=622.2,
                nrecalc=None, extra_ktables=(), vignetting=None, sky=None):
        """Host half of the draw loop: which objects exist (SkipThisObject for phot_flux == 0, stamp.py:199-202), FFT /
        photons / faint for each (stamp.py:275-336), the rows of both kinds.  Nothing here touches the GPU."""
        n_all = len(cat["x"])
        if self.nobjects is not None:
            n_all = min(n_all, int(self.nobjects))
        sel = np.arange(n_all)
        sub = {k: (v[sel] if isinstance(v, np.ndarray) and len(v) >= n_all else v) for k, v in cat.items()}
        phot = np.asarray(phot_flux)[sel]
        nominal = sub["nominal_flux"]
        # mode decision (stamp.py:275-336)
        if draw_method == "fft":
            is_fft = np.ones(n_all, dtype=bool)
        elif draw_method == "phot":
            is_fft = np.zeros(n_all, dtype=bool)
        else:
            is_fft = self._use_fft(sub, nominal, fwhm_total, fft_sb_thresh, kpsf, extra_ktables)
        is_fft &= np.asarray(sub["kind"]) < 3                  # knots and streaks have no k-space form here: always photons
        objects, sizes = make_objects(sub, np.where(phot > 0, phot, 0))
        keep = np.flatnonzero(phot > 0)                       # SkipThisObject for phot_flux == 0 (stamp.py:199-202)
        fft_rows = is_fft[keep]
        faint = nominal[keep] < max_flux_simple
        objects["flags"] = np.where(faint, objects["flags"] | IMS_OBJ_FAINT, objects["flags"] & ~IMS_OBJ_FAINT)
        fft_flux = np.zeros(len(objects))
        job = CcdJob(objects=objects[np.flatnonzero(~fft_rows)], nrecalc=nrecalc, kpsf=kpsf, extra_ktables=tuple(extra_ktables),
                     diffraction_fft=diffraction_fft, wavelength=wavelength, sky=sky, phot_index=np.flatnonzero(~fft_rows),
                     n_kept=len(objects))
        if fft_rows.any():
            if kpsf is None:
                raise GalSimConfigError("FFT drawing needs the k-space PSF description")
            fobj = objects[fft_rows]
            fflux = nominal[keep][fft_rows]
            if vignetting is not None:
                # FFT-drawn objects do not pass the ray trace: the empirical vignetting function scales their flux
                # (get_fft_psf_maybe, imsim/psf_utils.py:220-233); base['fft_flux'] carries the scaled value
                fflux = fflux * vignetting.at_pixel(self.det_name, fobj["x0"], fobj["y0"], scene.nx, scene.ny)
            tables_needed = fft_draw.profile_ktable_ids(scene, fobj["prof_table"], len(extra_ktables))
            job.fft_rows, order = fft_draw.build_fft_objects(fobj, fflux, tables_needed)
            job.fft_index = np.flatnonzero(fft_rows)[order]
            fft_flux[np.flatnonzero(fft_rows)] = fflux
        job.host = dict(sel=sel, keep=keep, sub=sub, nominal=nominal, phot=phot, fft_rows=fft_rows, faint=faint, fft_flux=fft_flux)
        return job

    def build_image(self, renderer, cat, phot_flux, make_objects, fft_sb_thresh=0.0, max_flux_simple=100.0,
                    draw_method="auto", kpsf=None, fwhm_total=0.8, diffraction_fft=None, wavelength=622.2,
                    nrecalc=None, truth=None, extra_ktables=(), vignetting=None):
        """The draw loop (imsim/lsst_image.py:342-368 + imsim/stamp.py:411-575).

        cat / phot_flux: catalog dict and Poisson-realised fluxes; make_objects(cat, phot) builds the
        OBJECT_DTYPE rows.  truth: optional dict receiving nominal_flux / phot_flux / fft_flux /
        realized_flux per object (the base[...] side channel, stamp.py:193-196, :304-305, :525, :573)."""
        job = self.prepare(renderer.scene, cat, phot_flux, make_objects, fft_sb_thresh=fft_sb_thresh, max_flux_simple=max_flux_simple,
                           draw_method=draw_method, kpsf=kpsf, fwhm_total=fwhm_total, diffraction_fft=diffraction_fft,
                           wavelength=wavelength, nrecalc=nrecalc, extra_ktables=extra_ktables, vignetting=vignetting)
        realized = renderer.torch.zeros(job.n_kept, dtype=renderer.torch.float64, device=renderer.device)
        draw_job(renderer, job, realized=realized)
        if truth is not None:
            fill_truth(truth, job, realized.cpu().numpy())
        return renderer.image


def fill_truth(truth, job, realized):
    """the base[...] side channel of one CCD (stamp.py:193-196, :304-305, :525, :573) from a drawn job"""
    h = job.host
    keep, fft_rows = h["keep"], h["fft_rows"]
    truth["index"] = h["sel"][keep]
    truth["x"], truth["y"] = h["sub"]["x"][keep], h["sub"]["y"][keep]
    truth["nominal_flux"] = h["nominal"][keep]
    truth["phot_flux"] = np.where(fft_rows, 0.0, h["phot"][keep])           # stamp.py:305
    truth["fft_flux"] = h["fft_flux"]
    truth["realized_flux"] = realized
    truth["mode"] = np.where(fft_rows, "fft", np.where(h["faint"], "faint", "phot"))
    return truth


class LSST_PhotonPoolingImageBuilder(LSST_ImageBuilderBase):
    """`image.type: LSST_PhotonPoolingImage`, requires `stamp.type: LSST_Photons`."""

    def setup(self, config, stamp_type, det_type="E2V"):
        photon_pooling.check_stamp_type(stamp_type)
        return super().setup(config, det_type)

    def build_image(self, renderer, cat, phot_flux, make_objects, max_flux_simple=100.0, seed=0, truth=None, fft_sb_thresh=0.0,
                    kpsf=None, fwhm_total=0.8, diffraction_fft=None, wavelength=622.2, extra_ktables=(), vignetting=None,
                    checkpoint=None):
        """LSST_PhotonPoolingImageBuilder.buildImage (imsim/photon_pooling.py:29-174): the fluxes of all objects are known
        up front, so the objects are partitioned into FFT / photon-shooting / faint; the FFT objects are drawn FIRST
        (:84-114, in nbatch_fft batches over objects), then the photon batches run through the pooled path."""
        torch = renderer.torch
        n_all = len(cat["x"]) if self.nobjects is None else min(len(cat["x"]), int(self.nobjects))
        sub = {k: (v[:n_all] if isinstance(v, np.ndarray) and len(v) >= n_all else v) for k, v in cat.items()}
        phot = np.asarray(phot_flux)[:n_all]
        nominal = np.asarray(sub["nominal_flux"])
        objects, _ = make_objects(sub, phot)
        keep = np.flatnonzero(phot > 0)
        is_fft = self._use_fft(sub, nominal, fwhm_total, fft_sb_thresh, kpsf, extra_ktables) & (np.asarray(sub["kind"]) < 3)
        fft_rows = is_fft[keep]
        modes = stamp.classify(nominal[keep], max_flux_simple)
        modes[fft_rows] = stamp.ProcessingMode.FFT
        realized = torch.zeros(len(objects), dtype=torch.float64, device=renderer.device)
        fft_flux = np.zeros(len(objects))
        chk_name = "buildImage_photonpooling_" + str(self.det_name)
        # a checkpoint of this CCD: its image already holds the FFT objects, and their fluxes sit in a record of their own
        resumed_fft = checkpoint.load(chk_name + "_fft") if checkpoint is not None else None
        if resumed_fft is not None and checkpoint.load(chk_name) is not None:
            fft_flux[...] = resumed_fft[0]
            realized.copy_(torch.from_numpy(np.asarray(resumed_fft[1])).to(renderer.device))
        elif fft_rows.any():
            if kpsf is None:
                raise GalSimConfigError("FFT drawing needs the k-space PSF description")
            idx_fft = np.flatnonzero(fft_rows)
            drawer = fft_draw.FftDrawer(renderer, kpsf, add_noise=True, diffraction_fft=diffraction_fft, wavelength=wavelength,
                                        extra_ktables=extra_ktables)
            nb = max(min(self.nbatch_fft, len(idx_fft)), 1)
            for batch in photon_pooling.make_batches(list(idx_fft), nb):
                if not batch:
                    continue
                batch = np.asarray(batch)
                fobj = objects[batch]
                fflux = nominal[keep][batch]
                if vignetting is not None:
                    sc = renderer.scene
                    fflux = fflux * vignetting.at_pixel(self.det_name, fobj["x0"], fobj["y0"], sc.nx, sc.ny)
                rows, order = fft_draw.build_fft_objects(
                    fobj, fflux, fft_draw.profile_ktable_ids(renderer.scene, fobj["prof_table"], len(extra_ktables)))
                r_fft = torch.zeros(len(rows), dtype=torch.float64, device=renderer.device)
                drawer.draw(rows, realized=r_fft)
                realized.index_add_(0, torch.from_numpy(batch[order]).to(renderer.device), r_fft)
                fft_flux[batch] = fflux
            if checkpoint is not None:
                checkpoint.save(chk_name + "_fft", (fft_flux.copy(), realized.cpu().numpy()))
        pidx = np.flatnonzero(~fft_rows)
        if len(pidx):
            r_ph = torch.zeros(len(pidx), dtype=torch.float64, device=renderer.device)
            photon_pooling.build_image(renderer, objects[pidx], modes[pidx], nbatch=self.nbatch, nsubbatch=self.nsubbatch, seed=seed,
                                       realized=r_ph, checkpoint=checkpoint, chk_name=chk_name,
                                       nbatch_per_checkpoint=self.nbatch_per_checkpoint)
            realized.index_add_(0, torch.from_numpy(pidx).to(renderer.device), r_ph)
        if truth is not None:
            truth["index"] = keep
            truth["x"], truth["y"] = sub["x"][keep], sub["y"][keep]
            truth["nominal_flux"] = nominal[keep]
            truth["phot_flux"] = np.where(fft_rows, 0.0, phot[keep])
            truth["fft_flux"] = fft_flux
            truth["incident_flux"] = realized.cpu().numpy()
            truth["mode"] = np.array([m.name.lower() for m in modes])
        return renderer.image
