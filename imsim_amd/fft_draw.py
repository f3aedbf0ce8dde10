"""FFT branch of LSST_SiliconBuilder.draw (imsim/stamp.py:482-525) over the GPU engine.

Which objects take this branch is decided as in the reference (imsim/stamp.py:275-277 and
imsim/psf_utils.py:152-239): nominal_flux >= 1e6, fft_sb_thresh set and exceeded by half the peak
surface brightness of the object convolved with the FFT-mode PSF.  The FFT-mode PSF replaces
PhaseScreenPSF by VonKarman and SecondKick by Airy (make_fft_psf, psf_utils.py:94-149); with the
Kolmogorov + Gaussian PSF it is the PSF itself.
"""
import ctypes as C
import math

import numpy as np

from . import _abi, tables, tuning
from ._abi import FFT_OBJECT_DTYPE, FftParams, KPsf

KOLMOGOROV_K0 = 2.992934 * 0.9758634299       # k0 * fwhm for exp(-(k/k0)^(5/3))  [rad/arcsec * arcsec]


def next_fft_size(n, minimum=32):
    m = minimum
    while m < n:
        m *= 2
    return m


def kolmogorov_gaussian_kpsf(fwhm_atm, fwhm_sys):
    """k-space PSF list for Convolve(Kolmogorov(fwhm), Gaussian(fwhm)) (imsim/atmPSF.py:534-536)."""
    return [(_abi.IMS_KPSF_KOLMOGOROV, 0, KOLMOGOROV_K0 / fwhm_atm),
            (_abi.IMS_KPSF_GAUSSIAN, 0, fwhm_sys / 2.3548200450309493)]


ARCSEC_PER_RAD = 206264.80624709636


def vonkarman_structure_function(r, r0, L0):
    """Phase structure function of von Karman turbulence (galsim.VonKarman):
    D(r) = 0.17166 (L0/r0)^(5/3) [Gamma(5/6) / 2^(1/6) - (2 pi r / L0)^(5/6) K_5/6(2 pi r / L0)],
    -> 6.8839 (r/r0)^(5/3) for r << L0."""
    from scipy import special
    r = np.asarray(r, dtype=np.float64)
    x = 2.0 * math.pi * r / L0
    with np.errstate(invalid="ignore", divide="ignore"):
        t = np.where(x > 0.0, x ** (5.0 / 6.0) * special.kv(5.0 / 6.0, np.where(x > 0.0, x, 1.0)), 1.005634917998590172)
    return 0.1716613621245709486 * (L0 / r0) ** (5.0 / 3.0) * (1.005634917998590172 - t)


def vonkarman_ktable(lam, r0_500, L0, qmax=tables.KTABLE_QMAX, npts=tables.KTABLE_NPTS):
    """k-table of galsim.VonKarman(lam, r0_500, L0) -- what make_fft_psf puts in place of a PhaseScreenPSF
    (imsim/psf_utils.py:117-121).  Returns (table over the common q grid, p0) with MTF(k) = table(k * p0),
    k in rad/arcsec: the baseline is r = lam k / (2 pi) [m per radian of angle]; the grid spans 4 r0."""
    r0 = r0_500 * (lam / 500.0) ** 1.2
    q = np.linspace(0.0, qmax, npts)
    unit = 4.0 * r0 / qmax                                   # metres of baseline per unit q
    T = np.exp(-0.5 * vonkarman_structure_function(q * unit, r0, L0))
    p0 = lam * 1.0e-9 * ARCSEC_PER_RAD / (2.0 * math.pi) / unit
    return T, p0


def airy_ktable(lam, diam, obscuration, qmax=tables.KTABLE_QMAX, npts=tables.KTABLE_NPTS):
    """k-table of galsim.Airy(lam, diam, obscuration) -- make_fft_psf's stand-in for the SecondKick
    (psf_utils.py:112-115): the autocorrelation of the annular pupil, zero beyond the baseline `diam`."""
    from .atm_psf import annulus_mtf
    q = np.linspace(0.0, qmax, npts)
    unit = 1.02 * diam / qmax
    T = annulus_mtf(q * unit, diam, obscuration)
    p0 = lam * 1.0e-9 * ARCSEC_PER_RAD / (2.0 * math.pi) / unit
    return T, p0


def atmospheric_fft_kpsf(atm, wavelength, first_table, fwhm_sys=None):
    """make_fft_psf (imsim/psf_utils.py:94-149) for an AtmosphericPSF evaluated at `wavelength` [nm]:
    PhaseScreenPSF -> VonKarman(lam, r0_500_effective, L0), SecondKick -> Airy(lam, diam, obscuration), the
    optional Gaussian optics term unchanged.  first_table: index the first extra k-table will get (after the
    profile tables).  Returns (kpsf list for FftDrawer, list of extra k-tables)."""
    kpsf, extra = [], []
    T, p0 = vonkarman_ktable(wavelength, atm.r0_500, atm.L0)
    kpsf.append((_abi.IMS_KPSF_TABLE, first_table + len(extra), p0))
    extra.append(T)
    if atm.second_kick is not None:
        T, p0 = airy_ktable(wavelength, atm.diam, atm.obscuration)
        kpsf.append((_abi.IMS_KPSF_TABLE, first_table + len(extra), p0))
        extra.append(T)
    if fwhm_sys:
        kpsf.append((_abi.IMS_KPSF_GAUSSIAN, 0, fwhm_sys / 2.3548200450309493))
    return kpsf, extra


def kpsf_peak_per_flux(kpsf, ktables=None, q_step=None, n_base=2):
    """Central surface brightness per unit flux [1/arcsec^2] of every k-space PSF component: I(0) = int T(k) k dk / 2 pi.
    Gaussian 1 / (2 pi sigma^2); Kolmogorov (3/5) Gamma(6/5) k0^2 / (2 pi); a k-table by quadrature of its samples."""
    out = []
    for kind, table, p0 in kpsf:
        if kind == _abi.IMS_KPSF_GAUSSIAN:
            out.append(1.0 / (2.0 * math.pi * p0 * p0))
        elif kind == _abi.IMS_KPSF_KOLMOGOROV:
            out.append(0.6 * math.gamma(1.2) * p0 * p0 / (2.0 * math.pi))
        else:
            T = np.asarray(ktables[table - n_base], dtype=np.float64)
            q = np.arange(len(T)) * q_step
            out.append(float(np.trapezoid(T * q, q)) / (2.0 * math.pi * p0 * p0))
    return out


def max_surface_brightness(flux, kind, hlr, fwhm_total=None, pixel_scale=0.2, sersic_n=None, jac_det=None, psf_peaks=None):
    """`Convolve(gal_achrom, fft_psf).withFlux(F).max_sb / 2 * pixel_scale^2` [photons/pixel], the quantity
    get_fft_psf_maybe compares with fft_sb_thresh (imsim/psf_utils.py:201-212).  GalSim's max_sb of a convolution is the
    estimate that is exact for Gaussians: F / sum_i (1 / peak_i) with peak_i the central surface brightness per unit
    flux of component i (a DeltaFunction contributes nothing).  Sersic: b^2n / (2 pi n Gamma(2n) hlr^2 |det J|).
    psf_peaks: kpsf_peak_per_flux of the FFT-mode PSF; without it a single Gaussian of FWHM fwhm_total stands in."""
    from scipy import special
    flux = np.asarray(flux, dtype=np.float64)
    kind = np.asarray(kind)
    hlr = np.asarray(hlr, dtype=np.float64)
    if psf_peaks is None:
        sigma = fwhm_total / 2.3548200450309493
        psf_peaks = [1.0 / (2.0 * math.pi * sigma * sigma)]
    inv = np.full(flux.shape, sum(1.0 / p for p in psf_peaks))
    n = np.where(kind == 1, 1.0, np.where(kind == 2, 4.0, 0.0))
    if sersic_n is not None:
        sn = np.asarray(sersic_n, dtype=np.float64)
        n = np.where(((kind == 1) | (kind == 2)) & (sn > 0), sn, n)
    gal = n > 0
    if gal.any():
        ng = n[gal]
        b = special.gammaincinv(2.0 * ng, 0.5)
        peak = np.exp(2.0 * ng * np.log(b) - special.gammaln(2.0 * ng)) / (2.0 * math.pi * ng * hlr[gal] ** 2)
        if jac_det is not None:
            peak = peak / np.abs(np.asarray(jac_det, dtype=np.float64)[gal])
        inv[gal] = inv[gal] + 1.0 / peak
    other = (~gal) & (kind != 0)        # knots, streaks, images: never FFT-drawn here; a broad stand-in keeps them photon-shot
    inv[other] = np.inf
    return flux / inv / 2.0 * pixel_scale ** 2


def use_fft(nominal_flux, kind, hlr, fwhm_total, fft_sb_thresh, **kw):
    """The FFT-vs-phot decision of LSST_SiliconBuilder.buildPSF (stamp.py:275-308)."""
    nominal_flux = np.asarray(nominal_flux, dtype=np.float64)
    if not fft_sb_thresh:
        return np.zeros(nominal_flux.shape, dtype=bool)
    cand = (nominal_flux >= 1.0e6) & (nominal_flux >= fft_sb_thresh)
    return cand & (max_surface_brightness(nominal_flux, kind, hlr, fwhm_total, **kw) > fft_sb_thresh)


def profile_ktable_ids(scene, prof_table, n_extra_ktables=0, n_base=2):
    """k-table id of every object from its radial table id: tables 0 / 1 (n = 1 / 4) keep their ids, the scene's
    further Sersic indices sit behind the n_extra_ktables PSF tables in the order of scene.sersic_extra_n;
    everything else (points) is -1."""
    prof_table = np.asarray(prof_table)
    out = np.where((prof_table == 0) | (prof_table == 1), prof_table, -1).astype(np.int32)
    index = getattr(scene, "sersic_index", None) or {}
    for j, n in enumerate(getattr(scene, "sersic_extra_n", ()) or ()):
        out[prof_table == index[n]] = n_base + n_extra_ktables + j
    return out


def build_fft_objects(objects, fft_flux, prof_ktable, pixel_scale=0.2):
    """OBJECT_DTYPE rows (geometry as for photon shooting) -> FFT_OBJECT_DTYPE rows, grouped by FFT
    size.  The profile affine is expressed along the pixel axes: jac' = s * winv * jac."""
    n = len(objects)
    out = np.zeros(n, dtype=FFT_OBJECT_DTYPE)
    size = objects["stamp_xmax"] - objects["stamp_xmin"] + 1
    nfft = np.array([next_fft_size(int(s)) for s in size], dtype=np.int32)
    order = np.argsort(nfft, kind="stable")
    objects, fft_flux, prof_ktable, size, nfft = objects[order], np.asarray(fft_flux)[order], np.asarray(prof_ktable)[order], size[order], nfft[order]
    out["obj_id"] = objects["obj_id"]
    out["flux"] = fft_flux
    out["nfft"] = nfft
    pad = (nfft - size) // 2
    out["x0"] = objects["stamp_xmin"] - pad
    out["y0"] = objects["stamp_ymin"] - pad
    out["cx"] = objects["x0"] - out["x0"]
    out["cy"] = objects["y0"] - out["y0"]
    out["prof_ktable"] = prof_ktable
    out["prof_scale"] = objects["prof_scale"]
    w, j = objects["winv"] * pixel_scale, objects["jac"]
    out["jac"] = np.stack([w[:, 0] * j[:, 0] + w[:, 1] * j[:, 2], w[:, 0] * j[:, 1] + w[:, 1] * j[:, 3],
                           w[:, 2] * j[:, 0] + w[:, 3] * j[:, 2], w[:, 2] * j[:, 1] + w[:, 3] * j[:, 3]], axis=1)
    for f in ("stamp_xmin", "stamp_xmax", "stamp_ymin", "stamp_ymax"):
        out[f] = objects[f]
    nh = nfft.astype(np.int64) // 2 + 1
    out["k_offset"] = np.concatenate([[0], np.cumsum(nfft.astype(np.int64) * nh)])[:-1]
    out["r_offset"] = np.concatenate([[0], np.cumsum(nfft.astype(np.int64) ** 2)])[:-1]
    return out, order


def set_spikes(P, diffraction_fft, wavelength, renderer=None):
    """Fill the spikes block of an FftParams from a DiffractionFFT config (None = disabled)."""
    from . import diffraction_fft as dfft
    if diffraction_fft is None or not diffraction_fft.enabled:
        P.spikes.enabled = 0
        return None
    k = diffraction_fft.constants(wavelength)
    S = P.spikes
    S.enabled, S.cutoff, S.threshold = 1, k.cutoff, float(diffraction_fft.brightness_threshold)
    S.cos0, S.sin0, S.a_lo, S.d_alpha, S.scale, S.r0, S.norm = k.cos0, k.sin0, k.a_lo, k.d_alpha, k.scale, dfft.SPIKE_R0, k.norm
    S.tab_row = S.tab_col = S.tab_val = None
    if renderer is not None and tuning.flag("IMS_SPIKE_TABLE"):
        tab = spike_table(renderer, diffraction_fft, wavelength, S)
        S.tab_row, S.tab_col, S.tab_val = tab[0].data_ptr(), tab[1].data_ptr(), tab[2].data_ptr()
    return k


def spike_table(renderer, diffraction_fft, wavelength, S):
    """ims_spikes_t.tab_*: the non-zero entries of the normalised spike stencil on the renderer's device, made once per (visit's
    DiffractionFFT, wavelength, device) by two launches of ims_fft_spike_table and kept with the DiffractionFFT instance that every CCD
    of a visit shares -- the reference makes the whole stencil array once per visit (imsim/stamp.py:36-68)."""
    torch = renderer.torch
    cache = diffraction_fft.__dict__.setdefault("_device_tables", {})
    key = (str(renderer.device), float(wavelength), float(S.norm), int(S.cutoff))
    if key not in cache:
        lib = renderer.lib
        rows = 2 * int(S.cutoff) + 1
        with torch.cuda.device(renderer.device):
            st = C.c_void_p(torch.cuda.current_stream(renderer.device).cuda_stream)
            count = torch.zeros(rows, dtype=torch.int32, device=renderer.device)
            _abi.check(lib.ims_fft_spike_table(C.byref(S), None, count.data_ptr(), None, None, st), "ims_fft_spike_table")
            ptr = torch.zeros(rows + 1, dtype=torch.int32, device=renderer.device)
            ptr[1:] = torch.cumsum(count, 0).to(torch.int32)
            n = int(ptr[-1].item())
            col = torch.empty(max(n, 1), dtype=torch.int32, device=renderer.device)
            val = torch.empty(max(n, 1), dtype=torch.float64, device=renderer.device)
            _abi.check(lib.ims_fft_spike_table(C.byref(S), ptr.data_ptr(), None, col.data_ptr(), val.data_ptr(), st), "ims_fft_spike_table")
            torch.cuda.current_stream(renderer.device).synchronize()
        cache[key] = (ptr, col, val, n)
    return cache[key]


def psf_mtf(kpsf, ktables, q_step, k):
    """Product of the PSF components' MTFs at radial frequency k [rad/arcsec] (host mirror of kspace_at's PSF factor)"""
    amp = 1.0
    for kind, table, p0 in kpsf:
        if kind == _abi.IMS_KPSF_GAUSSIAN:
            amp *= math.exp(-0.5 * p0 * p0 * k * k)
        elif kind == _abi.IMS_KPSF_KOLMOGOROV:
            amp *= math.exp(-(k / p0) ** (5.0 / 3.0))
        else:
            t = np.atleast_2d(ktables)[table]
            f = k * p0 / q_step
            amp *= 0.0 if f >= len(t) - 1 else float(np.interp(f, np.arange(len(t)), t))
    return abs(amp)


def alias_order(kpsf, ktables, q_step, pixel_scale=0.2, tol=1.0e-4):
    """How many aliases of the sampling frequency the k-space fill has to fold in (ims_fft_params_t.n_alias): 0 when the
    PSF's MTF is below `tol` at the Nyquist frequency (any seeing-limited PSF), else the smallest m whose first omitted
    alias, at (2m + 1) x Nyquist, is below it.  GalSim reaches the same end by drawing on a k-grid out to maxk and
    wrapping it onto the image's grid."""
    k_nyq = math.pi / pixel_scale
    for m in range(0, 4):
        if psf_mtf(kpsf, ktables, q_step, (2 * m + 1) * k_nyq) < tol:
            return m
    return 4


def fft_params(scene, kpsf, ktables, q_step, seed, add_noise=True, mem_put=None):
    P = FftParams()
    P.seed = int(seed)
    P.pixel_scale = 0.2
    P.n_kpsf = len(kpsf)
    for k, (kind, table, p0) in enumerate(kpsf):
        P.kpsf[k] = KPsf(kind, table, p0)
    P.add_noise = 1 if add_noise else 0
    t = np.atleast_2d(np.ascontiguousarray(ktables, dtype=np.float64))
    P.ktables.n_tables, P.ktables.n_pts = t.shape
    P.ktables.arg_min, P.ktables.arg_step = 0.0, float(q_step)
    keep, P.ktables.val = mem_put(t)
    P.nx, P.ny, P.xmin, P.ymin = scene.nx, scene.ny, scene.xmin, scene.ymin
    P.n_alias = alias_order(kpsf, t, q_step, P.pixel_scale) if kpsf else 0
    return P, keep


_WARM = {}
_WARM_ATEXIT = []
import threading as _threading
_WARM_LOCK = _threading.Lock()


def raw_inverse():
    """Whether FftDrawer leaves the 1 / N^2 of the inverse transform to the kernel that reads the real-space buffer next
    (ims_fft_inverse_raw + ims_fft_params_t.rbuf_raw; IMS_FFT_RAW, default on; never with the torch front end)."""
    return tuning.env("IMS_FFT_TORCH", "0") == "0" and tuning.env("IMS_FFT_RAW", "1") != "0"


def image_from_rbuf(fft_objects, rbuf, raw=None):
    """The real-space IMAGES of a draw from the buffer FftDrawer.draw returned (a numpy array): with the raw inverse every object's
    grid still lacks its 1 / (nfft * nfft) -- applied here by the same product the kernels form on reading (a checker's helper:
    the oracle's spike and finish steps take images)."""
    out = np.array(rbuf, dtype=np.float64, copy=True)
    if raw_inverse() if raw is None else raw:
        at = 0
        for n in np.asarray(fft_objects["nfft"], dtype=np.int64):
            out[at:at + n * n] *= 1.0 / (float(n) * float(n))
            at += int(n * n)
    return out


def warm_up(device, stream, sizes=(1024, 2048, 4096, 512)):
    """Make the library's hipFFT plans of `stream` for the grid sizes a visit's FFT-drawn objects take, on a BACKGROUND thread
    (returns it; join() before timing anything).  A process's first plan costs seconds (hipFFT / rocFFT start up and compile
    their kernels at run time: 2.4 s + 0.5 s for the next new size on a fresh MI355X box, round 5), which a visit otherwise
    pays in front of its first CCD with a bright star; called while the host still reads catalogs it costs nothing.  Plans
    are kept per stream: `stream` must be the one the draws will run on (focal plane: engine._focal_streams' middle stream)."""
    import threading
    import torch
    key = (str(device), int(stream.cuda_stream))
    if key in _WARM:
        return _WARM[key]
    lib = _abi.load()

    codes = {}

    def one(n):
        torch.cuda.set_device(device)
        rc = lib.ims_fft_warm(int(n), C.c_void_p(stream.cuda_stream))
        codes[int(n)] = (rc, lib.ims_last_error().decode() if rc else "")          # (ims_last_error is thread-local: read here)

    class _All:
        """the threads of one warm-up (a size each: the run-time compilations run side by side).  join() raises when a plan
        could not be made (no libhipfft, a failed plan): the first draw would otherwise find out inside a timed step."""
        def __init__(self, threads):
            self.threads = threads
            self.codes = codes

        def join(self):
            for t in self.threads:
                t.join()
            bad = {n: c for n, c in codes.items() if c[0] != 0}
            if bad:
                raise RuntimeError("hipFFT warm-up failed: " + "; ".join(f"size {n}: {c[1]} ({c[0]})" for n, c in sorted(bad.items())))
    with _WARM_LOCK:
        if key in _WARM:
            return _WARM[key]
        threads = [threading.Thread(target=one, args=(n,), name=f"ims-fft-warm-{n}", daemon=True) for n in sizes]
        for t in threads:
            t.start()
        _WARM[key] = _All(threads)
        if not _WARM_ATEXIT:
            # (daemon threads inside hipfftPlanMany's run-time compilation when the interpreter exits: wait for them first)
            import atexit
            atexit.register(lambda: [t.join(timeout=30.0) for w in list(_WARM.values()) for t in w.threads])
            _WARM_ATEXIT.append(True)
    return _WARM[key]


class FftDrawer:
    """Batched FFT rendering into a Renderer's CCD image."""

    def __init__(self, renderer, kpsf, sersic_indices=(1.0, 4.0), add_noise=True, diffraction_fft=None, wavelength=622.2,
                 extra_ktables=()):
        """extra_ktables: radial k-tables on the common q grid appended after the profile tables (the
        VonKarman / Airy tables of atmospheric_fft_kpsf, addressed by IMS_KPSF_TABLE components).  The k-tables of the
        scene's further Sersic indices (Scene.sersic_extra_n, configs.add_sersic_tables) follow behind those:
        profile_ktable_ids maps an object's radial table id to its k-table id."""
        self.r = renderer
        self.torch = renderer.torch
        extra_n = tuple(getattr(renderer.scene, "sersic_extra_n", ()) or ())
        tabs = [tables.sersic_ktable(n) for n in sersic_indices]
        more = [tables.sersic_ktable(n)[1] for n in extra_n]
        self.q_step = float(tabs[0][0][1] - tabs[0][0][0])
        self.P, self._keep = fft_params(renderer.scene, kpsf, np.stack([t[1] for t in tabs] + list(extra_ktables) + more), self.q_step,
                                        renderer.scene.seed, add_noise, lambda a: renderer.mem.put(a, np.float64))
        self.P.image = renderer.image.data_ptr()
        set_spikes(self.P, diffraction_fft, wavelength, renderer)

    def draw(self, fft_objects, realized=None):
        """fft_objects: FFT_OBJECT_DTYPE rows sorted by nfft (build_fft_objects).  `realized`:
        optional float64 device tensor [n]."""
        if len(fft_objects) == 0:
            return
        return self._run(self._upload(fft_objects), realized)

    def prepared(self, fft_objects):
        """Upload the rows and allocate the k-space / real-space buffers once; returns a zero-argument callable that
        draws the batch (bench.py: the timed region starts with its inputs resident in HBM)."""
        state = self._upload(fft_objects)

        def launch():
            self._run(state, None)
        nfft = state[0]["nfft"].astype(np.int64)
        launch.objects = len(nfft)
        launch.pixels = int(np.sum(nfft * nfft))
        launch.kspace_elements = int(np.sum(nfft * (nfft // 2 + 1)))
        return launch

    def prepared_chunked(self, fft_objects, max_pixels=1 << 30):
        """`prepared` for a table whose buffers do not fit at once (thousands of bright stars on 1024^2 .. 4096^2 grids: 24 B per grid
        point): the rows -- grouped by FFT size, as build_fft_objects leaves them -- are cut into consecutive chunks of at most
        max_pixels grid points that share ONE set of k-space / real-space / spike buffers and are drawn one after the other on the
        current stream.  Same launches per chunk as `prepared`; returns a zero-argument callable."""
        torch, r = self.torch, self.r
        rows = np.ascontiguousarray(fft_objects, dtype=FFT_OBJECT_DTYPE)
        nfft = rows["nfft"].astype(np.int64)
        px = nfft * nfft
        cuts, acc = [0], 0
        for k in range(len(rows)):
            if acc and acc + px[k] > max_pixels:
                cuts.append(k)
                acc = 0
            acc += int(px[k])
        cuts.append(len(rows))
        spans = [(a, b) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
        k_need = max(int(np.sum(nfft[a:b] * (nfft[a:b] // 2 + 1))) for a, b in spans)
        r_need = max(int(px[a:b].sum()) for a, b in spans)
        n_need = max(b - a for a, b in spans)
        kbuf = torch.empty(k_need, dtype=torch.complex128, device=r.device)
        rbuf = torch.empty(r_need, dtype=torch.float64, device=r.device)
        spike = (torch.empty(r_need, dtype=torch.float64, device=r.device), torch.empty(4 * n_need, dtype=torch.int32, device=r.device)) \
            if self.P.spikes.enabled else None
        self._spike_list = self._make_spike_list(r_need) if self.P.spikes.enabled else None
        from .engine import upload_async
        states = []
        for a, b in spans:
            part = rows[a:b].copy()
            nf = nfft[a:b]
            kpre = np.concatenate([[0], np.cumsum(nf * (nf // 2 + 1))]).astype(np.int64)
            rpre = np.concatenate([[0], np.cumsum(nf * nf)]).astype(np.int64)
            part["k_offset"], part["r_offset"] = kpre[:-1], rpre[:-1]
            obj_t = upload_async(torch, r.device, part.view(np.uint8).reshape(-1))
            states.append((part, obj_t, nf, kpre, rpre, upload_async(torch, r.device, kpre), upload_async(torch, r.device, rpre),
                           kbuf[:int(kpre[-1])], rbuf[:int(rpre[-1])]))

        def launch():
            for st in states:
                if spike is not None:
                    self._spike_bufs = (spike[0][:int(st[4][-1])], spike[1][:4 * len(st[0])])
                self._run(st, None)
        launch.objects = len(rows)
        launch.chunks = len(states)
        launch.pixels = int(px.sum())
        launch.kspace_elements = int(np.sum(nfft * (nfft // 2 + 1)))
        launch.keep = (states, kbuf, rbuf, spike)
        return launch

    def _upload(self, fft_objects):
        torch, r = self.torch, self.r
        rows = np.ascontiguousarray(fft_objects, dtype=FFT_OBJECT_DTYPE)
        from .engine import upload_async
        obj_t = upload_async(torch, r.device, rows.view(np.uint8).reshape(-1))
        nfft = rows["nfft"].astype(np.int64)
        kpre = np.concatenate([[0], np.cumsum(nfft * (nfft // 2 + 1))]).astype(np.int64)
        rpre = np.concatenate([[0], np.cumsum(nfft * nfft)]).astype(np.int64)
        kpre_t, rpre_t = upload_async(torch, r.device, kpre), upload_async(torch, r.device, rpre)
        kbuf = torch.empty(int(kpre[-1]), dtype=torch.complex128, device=r.device)
        rbuf = torch.empty(int(rpre[-1]), dtype=torch.float64, device=r.device)
        # every buffer of the draw is allocated HERE (under the caller's current stream), so that _run only launches: a run on a
        # side stream then never allocates there
        self._spike_bufs = ((torch.empty_like(rbuf), torch.empty(4 * len(rows), dtype=torch.int32, device=r.device))
                            if self.P.spikes.enabled else None)
        self._spike_list = self._make_spike_list(int(rpre[-1])) if self.P.spikes.enabled else None
        return rows, obj_t, nfft, kpre, rpre, kpre_t, rpre_t, kbuf, rbuf

    def _make_spike_list(self, n_pix):
        """the list of the listed spike step (ims_fft_spikes_listed): n_pix / 8 entries (the arms of a bright star's cross are ~3 % of
        a 4096^2 stamp) and its counter; None with IMS_FFT_SPIKE_LIST=0 (the one-launch form)"""
        if tuning.env("IMS_FFT_SPIKE_LIST", "1") == "0":
            return None
        # (small stamps are mostly arms: a draw of up to 2^20 pixels gets a list that cannot run over)
        cap = max(int(tuning.env("IMS_FFT_SPIKE_LIST_CAP", "0")) or max(n_pix // 8, min(n_pix, 1 << 20)), 64)
        torch, r = self.torch, self.r
        return (torch.empty(cap, dtype=torch.int64, device=r.device), torch.zeros(2, dtype=torch.int32, device=r.device))

    def _run(self, state, realized):
        torch, r = self.torch, self.r
        rows, obj_t, nfft, kpre, rpre, kpre_t, rpre_t, kbuf, rbuf = state
        n = len(rows)
        P = self.P
        P.realized_flux = realized.data_ptr() if realized is not None else None
        st = r._stream()
        _abi.check(r.lib.ims_fft_kspace_fill(C.byref(P), obj_t.data_ptr(), n, kpre_t.data_ptr(), int(kpre[-1]),
                                             kbuf.data_ptr(), st), "ims_fft_kspace_fill")
        # batched inverse real 2-D FFTs, one batch per FFT size (plain library transform: rocFFT through hipFFT); IMS_FFT_TORCH=1
        # takes torch.fft instead -- the same library behind another front end, kept as the checker
        import os
        use_torch = tuning.env("IMS_FFT_TORCH", "0") != "0"
        # the 1 / N^2 of the inverse is applied by whoever reads the real-space buffer next (ims_fft_params_t.rbuf_raw): the same
        # product without a pass of its own over the branch's largest buffer (IMS_FFT_RAW=0: the scaling pass, the same bits)
        raw = raw_inverse()
        inverse = r.lib.ims_fft_inverse_raw if raw else r.lib.ims_fft_inverse
        kspace = kbuf
        if getattr(self, "keep_kspace", False):
            kspace = kbuf.clone()                     # hipFFT's complex-to-real transform uses its input as work space
        for size in np.unique(nfft):
            sel = np.flatnonzero(nfft == size)
            a, b = int(sel[0]), int(sel[-1]) + 1
            nh = int(size) // 2 + 1
            if use_torch:
                spec = kbuf[int(kpre[a]):int(kpre[b])].view(b - a, int(size), nh)
                # the transform writes straight into the real-space buffer (no staging copy)
                torch.fft.irfft2(spec, s=(int(size), int(size)), norm="backward",
                                 out=rbuf[int(rpre[a]):int(rpre[b])].view(b - a, int(size), int(size)))
            else:
                # the library's own hipFFT plans (ims_fft_inverse): nothing of the branch needs a Python-side transform
                _abi.check(inverse(kbuf.data_ptr() + 16 * int(kpre[a]), rbuf.data_ptr() + 8 * int(rpre[a]), int(size), b - a, st),
                           "ims_fft_inverse")
        final = rbuf
        bbox = None
        if P.spikes.enabled:
            # DiffractionFFT.apply between the clip and the noise (stamp.py:519-522)
            final, bbox = self._spike_bufs if getattr(self, "_spike_bufs", None) is not None else (
                torch.empty_like(rbuf), torch.empty(4 * n, dtype=torch.int32, device=r.device))
            P.rbuf_raw = int(raw)
            lst = getattr(self, "_spike_list", None)
            if lst is not None:
                _abi.check(r.lib.ims_fft_spikes_listed(C.byref(P), obj_t.data_ptr(), n, rpre_t.data_ptr(), int(rpre[-1]),
                                                       rbuf.data_ptr(), final.data_ptr(), bbox.data_ptr(), lst[0].data_ptr(),
                                                       lst[0].numel(), lst[1].data_ptr(), st), "ims_fft_spikes_listed")
            else:
                _abi.check(r.lib.ims_fft_spikes(C.byref(P), obj_t.data_ptr(), n, rpre_t.data_ptr(), int(rpre[-1]),
                                                rbuf.data_ptr(), final.data_ptr(), bbox.data_ptr(), st), "ims_fft_spikes")
            raw = False                         # the spike step writes the image itself
        P.rbuf_raw = int(raw)
        try:
            _abi.check(r.lib.ims_fft_finish(C.byref(P), obj_t.data_ptr(), n, rpre_t.data_ptr(), int(rpre[-1]),
                                            final.data_ptr(), st), "ims_fft_finish")
        finally:
            P.rbuf_raw = 0
        # EVERY device buffer the queued launches touch stays referenced from _last (a caller that runs them on a side stream keeps
        # _last until they are through): the saturated-region boxes too -- freed at the end of this call, their block went back to
        # the allocator and to the next CCD's uploads while k_fft_bbox / k_fft_spikes were still queued (round 5: the rows of the
        # next CCD's k-space fill overwritten, a fault or a hang once the side stream ran behind)
        self._last = (kbuf, rbuf, final, obj_t, kpre_t, rpre_t, bbox, getattr(self, "_spike_list", None))
        return kspace, rbuf
