"""Synthetic instance catalog and vectorised object-table construction.

This is the host-side stand-in for the per-object Python of LSST_SiliconBuilder.setup
(imsim/stamp.py:109-249): flux realisation, stamp size, faint/phot classification and the
per-object geometry the kernels need, done in bulk with numpy instead of once per object.
"""
import math

import numpy as np

from . import tables
from ._abi import OBJECT_DTYPE, IMS_OBJ_FAINT, IMS_PROF_POINT, IMS_PROF_BOX, IMS_PROF_KNOTS, IMS_PROF_IMAGE

PIXEL_SCALE = 0.2          # arcsec / pixel (LSST_SiliconBuilder._pixel_scale, stamp.py:102)
NMAX = 4096                # stamp.py:106
TINY_FLUX = 10             # stamp.py:105
FT_DEFAULT = 5.0e-3        # galsim.GSParams().folding_threshold
STEPK_MIN_HLR = 5.0        # galsim.GSParams().stepk_minimum_hlr
SERSIC_N = (1.0, 4.0)
KIND_KNOTS, KIND_STREAK, KIND_IMAGE = 3, 4, 5
FLAT_OBJECT_ID = 0x7E00000000          # object ids of the photon-flat iterations (imsim_amd.flat)


def synthetic_catalog(n, seed=20261001, nx=4096, ny=4096, mag_min=16.0, mag_max=27.0):
    """The seeded synthetic catalog of SURVEY.md 8(d): uniform positions, dN/dm ~ 10^(0.35 m),
    nominal_flux = 30 s * 10^(0.4 (28.13 - m)) e-, 50 % point / 30 % Sersic n=1 / 20 % Sersic n=4
    (the mix of examples/example_instance_catalog.txt), hlr log-uniform 0.05-1", q~U(0.2,1),
    PA~U(0,180)."""
    rng = np.random.default_rng(seed)
    x = rng.uniform(0.5, nx + 0.5, n)
    y = rng.uniform(0.5, ny + 0.5, n)
    a = 0.35 * math.log(10.0)
    u = rng.uniform(0.0, 1.0, n)
    mag = np.log(np.exp(a * mag_min) + u * (np.exp(a * mag_max) - np.exp(a * mag_min))) / a
    flux = 30.0 * 10.0 ** (0.4 * (28.13 - mag))
    t = rng.uniform(0.0, 1.0, n)
    kind = np.where(t < 0.5, 0, np.where(t < 0.8, 1, 2)).astype(np.int32)   # 0 point, 1 sersic n=1, 2 sersic n=4
    hlr = np.exp(rng.uniform(math.log(0.05), math.log(1.0), n))
    q = rng.uniform(0.2, 1.0, n)
    pa = rng.uniform(0.0, 180.0, n)
    # flat-in-photons SED over the tabulated r band: the object evaluated at the effective wavelength carries
    # nominal_flux / (integral of the throughput) photons per nm (what the stamp-size surface-brightness test sees)
    wl, thr = tables.synthetic_r_band()
    return dict(x=x, y=y, mag=mag, nominal_flux=flux, kind=kind, hlr=hlr, q=q, pa=pa,
                obj_id=np.arange(n, dtype=np.int64), sb_flux=flux / float(np.trapezoid(thr, wl)))


def realize_fluxes(nominal_flux, seed):
    """phot_flux ~ Poisson(nominal_flux) per object (stamp.py:190)."""
    rng = np.random.default_rng([int(seed), 0x5151])
    return rng.poisson(nominal_flux).astype(np.int64)


def shear_matrix(q, beta_deg):
    """galsim.Shear(q=, beta=) as the area-preserving 2x2 used by GSObject._shear (instcat.py:519-520)."""
    g = (1.0 - q) / (1.0 + q)
    b = np.deg2rad(beta_deg)
    g1, g2 = g * np.cos(2 * b), g * np.sin(2 * b)
    f = 1.0 / np.sqrt(1.0 - g * g)
    return np.stack([f * (1 + g1), f * g2, f * g2, f * (1 - g1)], axis=-1)


def lens_matrix(g1, g2, mu):
    """GSObject._lens(g1, g2, mu): shear by the reduced shear then magnify (instcat.py:521-522)."""
    g = np.sqrt(g1 * g1 + g2 * g2)
    f = np.sqrt(mu) / np.sqrt(1.0 - g * g)
    return np.stack([f * (1 + g1), f * g2, f * g2, f * (1 - g1)], axis=-1)


def _mat2(a, b):
    return np.stack([a[..., 0] * b[..., 0] + a[..., 1] * b[..., 2], a[..., 0] * b[..., 1] + a[..., 1] * b[..., 3],
                     a[..., 2] * b[..., 0] + a[..., 3] * b[..., 2], a[..., 2] * b[..., 1] + a[..., 3] * b[..., 3]], axis=-1)


# ---------------- stamp sizes (imsim/stamp_utils.py restated, vectorised) ----------------
def _good_size(stepk, pixel_scale=PIXEL_SCALE):
    """GSObject.getGoodImageSize: N = ceil(2 pi / (stepk * scale)) rounded up to even."""
    n = np.ceil(2.0 * np.pi / (stepk * pixel_scale)).astype(np.int64)
    return 2 * ((n + 1) // 2)


def gaussian_stepk(sigma, ft):
    r = np.maximum(np.sqrt(-2.0 * np.log(ft)), STEPK_MIN_HLR * 1.1774100225154747)
    return np.pi / (r * sigma)


def _radius_enclosing(table, frac):
    r2, cdf = table
    return np.sqrt(np.interp(frac, cdf, r2))


def kolmogorov_stepk(fwhm, ft):
    """pi / R with R the radius enclosing (1 - ft) of the flux, at least stepk_minimum_hlr half-light radii."""
    tab = tables.kolmogorov_table()
    hlr = _radius_enclosing(tab, 0.5)
    # beyond the tabulated radius use the r^(-5/3) tail: 1 - F = ft
    r_tab = _radius_enclosing(tab, np.minimum(1.0 - ft * (1.0 - tables.SHOOT_ACCURACY), 1.0))
    r_last = math.sqrt(tab[0][-1])
    r_tail = r_last * (tables.SHOOT_ACCURACY / np.maximum(ft, 1e-300)) ** 0.6
    r = np.where(ft < tables.SHOOT_ACCURACY, r_tail, r_tab)
    r = np.maximum(r, STEPK_MIN_HLR * hlr)
    return np.pi / (r * fwhm)


def kolmogorov_gaussian_fwhm(airmass=1.2, raw_seeing=0.7, band="r"):
    """make_kolmogorov_and_gaussian_psf / BuildKolmogorovPSF (psf_utils.py:42-91, atmPSF.py:524-536)."""
    wlen_eff = dict(u=365.49, g=480.03, r=622.20, i=754.06, z=868.21, y=991.66)[band]
    fwhm_atm = raw_seeing * (wlen_eff / 500.0) ** -0.3 * airmass ** 0.6
    fwhm_sys = math.sqrt(0.25 ** 2 + 0.3 ** 2 + 0.08 ** 2) * airmass ** 0.6
    return fwhm_atm, fwhm_sys


def star_stamp_size(nominal_flux, noise_var, airmass=1.2, raw_seeing=0.7, band="r", nmax=NMAX):
    """get_star_stamp_size (stamp_utils.py:79-155)."""
    nominal_flux = np.asarray(nominal_flux, dtype=np.float64)
    with np.errstate(divide="ignore"):               # a source of zero flux: ft = inf takes the default threshold
        ft = noise_var / nominal_flux
    use_default = (ft >= FT_DEFAULT) | (ft == 0)
    ft = np.where(use_default, FT_DEFAULT, np.exp(np.floor(np.log(np.where(use_default, 1.0, ft)))))
    fwhm_atm, fwhm_sys = kolmogorov_gaussian_fwhm(airmass, raw_seeing, band)
    sk = kolmogorov_stepk(fwhm_atm, ft)
    sg = gaussian_stepk(fwhm_sys / 2.3548200450309493, ft)
    stepk = 1.0 / np.sqrt(1.0 / sk ** 2 + 1.0 / sg ** 2)
    return np.minimum(_good_size(stepk), nmax)


def _sersic_norm(n):
    """I(0) hlr^2 / flux of a Sersic profile: b^(2n) / (2 pi n Gamma(2n))"""
    from scipy import special
    b = special.gammaincinv(2.0 * n, 0.5)
    return b, b ** (2.0 * n) / (2.0 * np.pi * n * special.gamma(2.0 * n))


def sersic_xvalue(n, hlr, flux, jac, x, y):
    """GSObject.xValue of Sersic(n, hlr).transform(jac).withFlux(flux) at sky offsets (x, y) [arcsec]:
    flux * I0(|J^-1 x|) / |det J| with I0(r) = norm / hlr^2 exp(-b (r/hlr)^(1/n))."""
    a, b_, c, d = jac[..., 0], jac[..., 1], jac[..., 2], jac[..., 3]
    det = a * d - b_ * c
    u = (d * x - b_ * y) / det
    v = (-c * x + a * y) / det
    bn, norm = _sersic_norm(n)
    r = np.hypot(u, v) / hlr
    return flux * norm / (hlr * hlr * np.abs(det)) * np.exp(-bn * r ** (1.0 / n))


def _phot_stamp_size1(n0, xvalue, keep_sb_level, nmax, pixel_scale=PIXEL_SCALE, factor=1.1):
    """get_good_phot_stamp_size1 (stamp_utils.py:293-354), vectorised: grow N by 10 % until the surface
    brightness on the edges and corners of the square is below keep_sb_level, cap at Nmax, then shrink
    while the next smaller square still is (not below 64).  xvalue(h) -> max of the 8 edge/corner values."""
    N = np.asarray(n0, dtype=np.float64).copy()
    keep = np.broadcast_to(np.asarray(keep_sb_level, dtype=np.float64), N.shape)
    active = N < nmax
    for _ in range(200):
        if not active.any():
            break
        mv = xvalue(N / 2.0 * pixel_scale)
        stop = mv < keep
        grow = active & ~stop
        N = np.where(grow, N * factor, N)
        active = grow & (N < nmax)
    N = np.minimum(N, nmax)
    active = N >= 64 * factor
    for _ in range(200):
        if not active.any():
            break
        mv = xvalue(N / (2.0 * factor) * pixel_scale)
        shrink = active & ~(mv > keep)
        N = np.where(shrink, N / factor, N)
        active = shrink & (N >= 64 * factor)
    return N.astype(np.int64)


def _edge_max(fn, h):
    """max over the 4 edge midpoints and 4 corners of the square of half-width h (stamp_utils.py:327-331)"""
    # the eight points in ONE call ([8][n] arrays broadcast against the per-object parameters): same elementwise arithmetic,
    # an eighth of the numpy calls (the surface-brightness loop of the bright galaxies was 4 of the 6 ms of a device table build)
    z = 0 * h
    xs = np.stack([h, -h, z, z, h, h, -h, -h])
    ys = np.stack([z, z, h, -h, h, -h, h, -h])
    return np.max(fn(xs, ys), axis=0)


def double_gaussian_xvalue(x, y, fwhm1=0.6, fwhm2=0.12, wgt1=1.0, wgt2=0.1):
    """make_double_gaussian (psf_utils.py:8-39), unit flux"""
    s1, s2 = fwhm1 / 2.355, fwhm2 / 2.355
    r2 = x * x + y * y
    g1 = np.exp(-0.5 * r2 / (s1 * s1)) / (2.0 * np.pi * s1 * s1)
    g2 = np.exp(-0.5 * r2 / (s2 * s2)) / (2.0 * np.pi * s2 * s2)
    return (wgt1 * g1 + wgt2 * g2) / (wgt1 + wgt2)


def _sersic_n_of(kind, sersic_n=None):
    """Sersic index per object: the catalog's quantised `sersic_n` where given, else 1 / 4 for kinds 1 / 2"""
    n = np.where(kind == 1, 1.0, np.where(kind == 2, 4.0, 0.0))
    if sersic_n is not None:
        sn = np.asarray(sersic_n, dtype=np.float64)
        n = np.where(((kind == 1) | (kind == 2)) & (sn > 0), np.round(sn * 20.0) / 20.0, n)
    return n


def gal_stamp_size(kind, hlr, max_scale, nmax=NMAX, jac=None, nominal_flux=None, noise_var=800.0, sb_flux=None, sersic_n=None):
    """get_gal_stamp_size (stamp_utils.py:158-220).  First the GoodImageSize of the object convolved with the
    DoubleGaussian proxy PSF; for bright objects (more than 10 photons per stamp pixel on average) or stamps
    beyond Nmax the size follows from a surface-brightness limit of sqrt(noise_var)/8 via
    get_good_phot_stamp_size (object and proxy PSF grown separately, added in quadrature), relaxed to 3x
    that limit when it exceeds Nmax.  sb_flux: the flux obj_achrom carries in the reference (the object
    evaluated at the effective wavelength, i.e. photons per nm); defaults to nominal_flux."""
    sizes = np.zeros(len(hlr), dtype=np.int64)
    own = np.zeros(len(hlr), dtype=np.int64)
    dg_stepk = min(gaussian_stepk(0.6 / 2.355, FT_DEFAULT), gaussian_stepk(0.12 / 2.355, FT_DEFAULT))
    n_obj = _sersic_n_of(kind, sersic_n)
    n_distinct = [float(n) for n in np.unique(n_obj) if n > 0]
    for n in n_distinct:
        sel = n_obj == n
        r = max(_radius_enclosing(tables.sersic_table(n), 1.0 - FT_DEFAULT), STEPK_MIN_HLR)
        stepk_gal = np.pi / (r * hlr[sel] * max_scale[sel])
        stepk = 1.0 / np.sqrt(1.0 / stepk_gal ** 2 + 1.0 / dg_stepk ** 2)
        sizes[sel] = _good_size(stepk)
        own[sel] = _good_size(stepk_gal)
    if jac is None or nominal_flux is None:
        return np.minimum(sizes, nmax)
    nominal_flux = np.asarray(nominal_flux, dtype=np.float64)
    sb_flux = nominal_flux if sb_flux is None else np.asarray(sb_flux, dtype=np.float64)
    bright = (nominal_flux > 10.0 * sizes.astype(np.float64) ** 2) | (sizes > nmax)
    if bright.any():
        keep = math.sqrt(noise_var) / 8.0
        psf_n0 = np.array([_good_size(dg_stepk)])
        idx = np.flatnonzero(bright)

        def phot_size(level):
            out = np.zeros(len(idx), dtype=np.int64)
            psf_size = _phot_stamp_size1(psf_n0, lambda h: _edge_max(double_gaussian_xvalue, h), level, nmax)[0]
            for n in n_distinct:
                m = n_obj[idx] == n
                if not m.any():
                    continue
                ii = idx[m]
                fn = lambda x, y, ii=ii, n=n: sersic_xvalue(n, hlr[ii], sb_flux[ii], jac[ii], x, y)
                gal = _phot_stamp_size1(own[ii], lambda h: _edge_max(fn, h), level, nmax)
                out[m] = np.sqrt(gal.astype(np.float64) ** 2 + float(psf_size) ** 2).astype(np.int64)
            return out
        sz = phot_size(keep)
        huge = sz > nmax
        if huge.any():
            sz3 = phot_size(3.0 * keep)
            sz = np.where(huge, np.minimum(sz3, nmax), sz)
        sizes[idx] = sz
    return np.minimum(sizes, nmax)


def build_object_table(cat, phot_flux, noise_var=800.0, sed_table=0, max_flux_simple=100.0,
                       winv=(1.0 / PIXEL_SCALE, 0.0, 0.0, 1.0 / PIXEL_SCALE), airmass=1.2, raw_seeing=0.7,
                       band="r", image_bounds=None, dcr=(0.0, 0.0, 1.0), stamp_size=None, sersic_index=None):
    """Vectorised LSST_SiliconBuilder.setup for a whole catalog -> OBJECT_DTYPE rows.

    Objects with phot_flux == 0 are dropped (SkipThisObject, stamp.py:199-202).
    sersic_index: {n: radial table id} of the scene (configs.add_sersic_tables) when the catalog carries Sersic
    indices other than 1 and 4 in cat["sersic_n"]; a catalog index with no table raises."""
    n = len(cat["x"])
    kind = cat["kind"]
    nominal = cat["nominal_flux"]
    keep = phot_flux > 0
    obj = np.zeros(n, dtype=OBJECT_DTYPE)
    obj["obj_id"] = cat["obj_id"]
    obj["phot_first"] = 0
    obj["n_phot"] = phot_flux
    obj["x0"], obj["y0"] = cat["x"], cat["y"]
    obj["flux_per_photon"] = 1.0
    # kinds: 0 point, 1 / 2 Sersic n = 1 / 4 (radial tables 0 / 1), 3 RandomKnots, 4 streak (Box)
    knots, streak, image = kind == KIND_KNOTS, kind == KIND_STREAK, kind == KIND_IMAGE
    obj["prof_table"] = np.where(kind == 0, IMS_PROF_POINT, np.where(knots, IMS_PROF_KNOTS, np.where(streak, IMS_PROF_BOX,
                                 np.where(image, IMS_PROF_IMAGE, kind - 1))))
    sersic_n = cat.get("sersic_n") if isinstance(cat, dict) else None
    n_obj = _sersic_n_of(kind, sersic_n)
    odd = (n_obj > 0) & (n_obj != 1.0) & (n_obj != 4.0)
    if odd.any():
        index = sersic_index or {}
        missing = sorted(set(np.round(n_obj[odd], 2).tolist()) - set(index))
        if missing:
            raise ValueError(f"no radial table for Sersic index {missing}: call configs.add_sersic_tables(scene, cat['sersic_n'])")
        obj["prof_table"][odd] = [index[round(float(n), 2)] for n in n_obj[odd]]
    obj["prof_scale"] = np.where(kind == 0, 0.0, cat["hlr"])
    if knots.any():
        obj["prof_scale"][knots] = cat["hlr"][knots] / 1.1774100225154747      # Gaussian sigma of the knots' parent profile
        obj["prof_aux"][knots] = cat["n_knots"][knots]
    if streak.any():
        obj["prof_scale"][streak] = cat["box_length"][streak]
        obj["prof_aux"][streak] = cat["box_width"][streak]
    if image.any():                                             # InterpolatedImage(file, scale=pixel_scale), instcat.py:552-561
        obj["prof_scale"][image] = cat["image_scale"][image]
        obj["prof_aux"][image] = cat["image_index"][image]
    beta = 90.0 - cat["pa"]                                     # flip_g2 convention, instcat.py:503-508
    jac = shear_matrix(cat["q"], beta)
    if "g1" in cat:
        jac = _mat2(lens_matrix(cat["g1"], cat["g2"], cat["mu"]), jac)
    jac[kind == 0] = (1.0, 0.0, 0.0, 1.0)
    if streak.any():                                            # Box(length, width).rotate(position_angle), instcat.py:494-496
        t = np.deg2rad(cat["pa"][streak])
        jac[streak] = np.stack([np.cos(t), -np.sin(t), np.sin(t), np.cos(t)], axis=-1)
    if image.any():                                             # obj.rotate(-theta), then the lens (instcat.py:557-561)
        t = -np.deg2rad(cat["pa"][image])
        rot = np.stack([np.cos(t), -np.sin(t), np.sin(t), np.cos(t)], axis=-1)
        jac[image] = _mat2(lens_matrix(cat["g1"][image], cat["g2"][image], cat["mu"][image]), rot) if "g1" in cat else rot
    obj["jac"] = jac
    obj["winv"] = np.broadcast_to(np.asarray(winv, dtype=np.float64), (n, 4))
    obj["dcr_tanz"], obj["dcr_sinp"], obj["dcr_cosp"] = dcr
    obj["sed_table"] = cat["sed_table"] if isinstance(cat, dict) and cat.get("sed_table") is not None else sed_table
    obj["sed_wave"] = 0.0
    obj["flags"] = np.where(nominal < max_flux_simple, IMS_OBJ_FAINT, 0)
    obj["bf_state"] = 0
    # stamp size: given -> 32 for tiny flux -> get_stamp_size (stamp.py:205-232)
    if stamp_size is None:
        size = np.zeros(n, dtype=np.int64)
        star = kind == 0
        if star.any():
            size[star] = star_stamp_size(nominal[star], noise_var, airmass, raw_seeing, band)
        gal = ~star
        if gal.any():
            # largest singular value of the profile affine scales the real-space extent
            a, b, c, d = jac[gal, 0], jac[gal, 1], jac[gal, 2], jac[gal, 3]
            s1 = a * a + b * b + c * c + d * d
            s2 = np.sqrt(np.maximum((a * a + b * b - c * c - d * d) ** 2 + 4 * (a * c + b * d) ** 2, 0.0))
            max_scale = np.sqrt(0.5 * (s1 + s2))
            sb = cat["sb_flux"][gal] if "sb_flux" in cat else None
            size[gal] = gal_stamp_size(kind[gal], cat["hlr"][gal], max_scale, jac=jac[gal], nominal_flux=nominal[gal],
                                       noise_var=noise_var, sb_flux=sb, sersic_n=n_obj[gal])
            # knots: GoodImageSize of the parent Gaussian; streaks: of the box (stepk = pi / max(length, width)),
            # both convolved with the proxy PSF (first branch of get_gal_stamp_size)
            dg_stepk = min(gaussian_stepk(0.6 / 2.355, FT_DEFAULT), gaussian_stepk(0.12 / 2.355, FT_DEFAULT))
            gk, gs = knots[gal], streak[gal]
            if gk.any():
                sk = gaussian_stepk(obj["prof_scale"][gal][gk] * max_scale[gk], FT_DEFAULT)
                size[np.flatnonzero(gal)[gk]] = np.minimum(_good_size(1.0 / np.sqrt(1.0 / sk ** 2 + 1.0 / dg_stepk ** 2)), NMAX)
            if gs.any():
                sk = np.pi / np.maximum(obj["prof_scale"][gal][gs], obj["prof_aux"][gal][gs])
                size[np.flatnonzero(gal)[gs]] = np.minimum(_good_size(1.0 / np.sqrt(1.0 / sk ** 2 + 1.0 / dg_stepk ** 2)), NMAX)
            gi = image[gal]
            if gi.any():
                # the stamp of an image profile: stepk = pi / (largest extent of the image on the sky)
                sk = np.pi / (cat["image_extent"][gal][gi] * max_scale[gi])
                size[np.flatnonzero(gal)[gi]] = np.minimum(_good_size(1.0 / np.sqrt(1.0 / sk ** 2 + 1.0 / dg_stepk ** 2)), NMAX)
        size[nominal < TINY_FLUX] = 32
    else:
        size = np.broadcast_to(np.asarray(stamp_size, dtype=np.int64), (n,)).copy()
    # stamp bounds centred on the integer pixel nearest image_pos (GalSim locateStamp for even sizes)
    icx = np.floor(cat["x"] + 0.5).astype(np.int64)
    icy = np.floor(cat["y"] + 0.5).astype(np.int64)
    obj["stamp_xmin"] = icx - size // 2
    obj["stamp_xmax"] = icx - size // 2 + size - 1
    obj["stamp_ymin"] = icy - size // 2
    obj["stamp_ymax"] = icy - size // 2 + size - 1
    return obj[keep], size[keep]
