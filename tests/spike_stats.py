"""The reference's spike estimators (restated from /root/reference/tests/test_diffraction_fft.py:515-607) and the one bright
star of its stored ray-tracing statistics (create_test_config, :35-141) as a scene of this build.  Shared by
tests/test_spike_pins_gpu.py and tools/spike_sweep.py."""
import math
import os

import numpy as np
from scipy import stats

from imsim_amd import _abi, configs, catalog, tables

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fft-diffraction")
XSIZE = YSIZE = 3500          # tests/test_diffraction_fft.py:294-295
STAMP = 1000
ROTTELPOS, ALT, AZ = 20.0, 88.0, 73.7707957
R_MIN = 5.0
R_OUTER, R_INNER = 4.18, 2.55  # PupilAnnulusSampler of the reference's config (:85)
N_PHOT = 6_000_000             # the reference's photon count is not on file (Vega through LSST_r.dat); see tools/spike_sweep.py

# the reference's own tolerances where it compares an image with the stored statistics (:383-420)
TOL = {"c": 2.0, "angle_deg": 1.0, "angle_stddev_deg": 2.0, "slope": 0.1, "intercept": 0.5}


def stored(exptime):
    return np.load(os.path.join(GOLD, f"raytrace_diffraction_values_{int(exptime)}_exptime.npz"))


def center_of_brightness(image):
    return np.array([np.sum(image * np.arange(image.shape[0])[:, None]), np.sum(image * np.arange(image.shape[1]))]) / np.sum(image)


def folded_spike_angle(image, x_center, y_center, r_min):
    x, y = np.mgrid[0:image.shape[0], 0:image.shape[1]]
    r = np.hypot(y - y_center, x - x_center)
    m = r > r_min
    alpha = np.arctan2(y[m] - y_center, x[m] - x_center) % (np.pi / 2.0)
    w = image[m] / np.sum(image[m])
    xm, ym = np.sum(np.cos(4 * alpha) * w), np.sum(np.sin(4 * alpha) * w)
    R = math.hypot(xm, ym)
    return math.atan2(ym, xm) / 4, math.sqrt(-2 * math.log(R)) / 4


def radial_brightness_asymptotics(image, x_center, y_center, r_min=R_MIN, num_bins=25):
    x, y = np.mgrid[0:image.shape[0], 0:image.shape[1]]
    r = np.hypot(y - y_center, x - x_center)
    r_max = np.max(r[image > 0.0])
    b, r = image[r <= r_max], r[r <= r_max]
    bins = np.geomspace(r_min, np.max(r), num=num_bins)
    dist, _ = np.histogram(r, bins=bins, weights=b)
    dist = dist / (np.diff(bins) * np.sum(b))
    reg = stats.linregress(np.log((bins[1:] + bins[:-1]) / 2.0), np.log(dist))
    return reg.slope, reg.intercept, reg.stderr, reg.intercept_stderr


def sed_table(kind):
    """Inverse-CDF wavelength table of the star: 'r-flat' = flat in photons over the stand-in r band (what the bench scenes
    use), 'mono' = 577.6 nm (diffraction_fft.WAVELENGTH), 'vega-r' = a 9 600 K black body in photons through the stand-in r
    band (the reference draws vega.txt through LSST_r.dat; neither table is in /root/reference)."""
    wl, thr = tables.synthetic_r_band()
    if kind == "r-flat":
        return tables.inverse_cdf_table(wl, thr)[None, :]
    if kind == "mono":
        w = np.array([577.6 - 1e-6, 577.6 + 1e-6])
        return tables.inverse_cdf_table(w, np.ones(2))[None, :]
    if kind == "vega-r":
        lam = wl * 1e-9
        planck_photons = 1.0 / (lam ** 4 * (np.exp(6.62607015e-34 * 2.99792458e8 / (lam * 1.380649e-23 * 9600.0)) - 1.0))
        return tables.inverse_cdf_table(wl, thr * planck_photons)[None, :]
    raise ValueError(kind)


def scene(exptime, r_outer=R_OUTER, r_inner=R_INNER, sed="r-flat"):
    optics = configs.rubin_optics_struct(XSIZE, YSIZE, rottelpos=ROTTELPOS, altitude=ALT, azimuth=AZ)
    sc = configs.scene_c2(nx=XSIZE, ny=YSIZE)
    sc.optics = optics
    sc.psf = [(_abi.IMS_PSF_GAUSSIAN, 0, 0.3 / 2.3548200450309493, 0.0, 1.0)]
    sc.ops = [(_abi.IMS_OP_TIME_SAMPLER, 0, [0.0, exptime]), (_abi.IMS_OP_PUPIL_ANNULUS_SAMPLER, 0, [r_outer, r_inner]),
              (_abi.IMS_OP_RUBIN_DIFFRACTION_OPTICS, 0, [1.0, 1.0 if exptime == 0.0 else 0.0])]
    if sed != "r-flat":
        sc.sed_tables = sed_table(sed)
    return sc


def star(sc, ref_c, n_phot):
    """the star at the stored centre: c = (row, column) array indices -> 1-based image coordinates"""
    cat = catalog.synthetic_catalog(1, nx=XSIZE, ny=YSIZE)
    cat["x"][:], cat["y"][:] = ref_c[1] + 1.0, ref_c[0] + 1.0
    cat["kind"][:] = 0
    cat["nominal_flux"][:] = float(n_phot)
    objects, _ = configs.c3_objects(cat, np.array([n_phot]), sc)
    icx, icy = int(math.floor(cat["x"][0] + 0.5)), int(math.floor(cat["y"][0] + 0.5))
    objects["stamp_xmin"], objects["stamp_xmax"] = icx - STAMP // 2, icx - STAMP // 2 + STAMP - 1
    objects["stamp_ymin"], objects["stamp_ymax"] = icy - STAMP // 2, icy - STAMP // 2 + STAMP - 1
    return objects


def render(sc, ref, n_phot=N_PHOT):
    from imsim_amd.engine import Renderer
    r = Renderer(sc)
    r.render(star(sc, ref["c"], n_phot))
    r.synchronize()
    img = r.image.cpu().numpy()
    del r
    return img


def image_stats(img):
    """the five stored statistics (+ the two standard errors) of one image, angles in degrees"""
    c = center_of_brightness(img)
    angle, angle_std = folded_spike_angle(img, c[0], c[1], r_min=10.0)
    slope, intercept, slope_err, intercept_err = radial_brightness_asymptotics(img, c[0], c[1])
    return {"c": c, "angle_deg": math.degrees(angle), "angle_stddev_deg": math.degrees(angle_std), "slope": slope,
            "intercept": intercept, "slope_stderr": slope_err, "intercept_stderr": intercept_err}


def stored_stats(ref):
    return {"c": np.asarray(ref["c"]), "angle_deg": math.degrees(float(ref["angle"])),
            "angle_stddev_deg": math.degrees(float(ref["angle_stddev"])), "slope": float(ref["slope"]),
            "intercept": float(ref["intercept"])}


def within(got, want):
    """which of the five stored statistics `got` meets under the reference's tolerances"""
    return {k: bool(np.all(np.abs(np.asarray(got[k]) - np.asarray(want[k])) <= TOL[k])) for k in TOL}
