"""Scene builders for the BASELINE.json measurement configs."""
import numpy as np

from . import _abi, tables, catalog
from .engine import Scene


def standard_tables():
    """Radial tables: 0 = Sersic n=1, 1 = Sersic n=4, 2 = Kolmogorov (units of FWHM)."""
    tabs = [tables.sersic_table(1.0), tables.sersic_table(4.0), tables.kolmogorov_table()]
    return tables.stack_radial(tabs)


def r_band_sed_table():
    wl, thr = tables.synthetic_r_band()
    return tables.inverse_cdf_table(wl, thr)[None, :], tables.effective_wavelength(wl, thr)


def scene_c2(nx=4096, ny=4096, seed=398414, airmass=1.2, raw_seeing=0.75):
    """C2: photon shooting, Kolmogorov-equivalent Gaussian atmospheric PSF (+ Gaussian system
    term), no photon ops, no sensor (image.sensor: "")."""
    r2, cdf = standard_tables()
    sed, _ = r_band_sed_table()
    fwhm_atm, fwhm_sys = catalog.kolmogorov_gaussian_fwhm(airmass, raw_seeing, "r")
    s = 1.0 / 2.3548200450309493
    psf = [(_abi.IMS_PSF_GAUSSIAN, 0, fwhm_atm * s, 0.0, 1.0),
           (_abi.IMS_PSF_GAUSSIAN, 0, fwhm_sys * s, 0.0, 1.0)]
    return Scene(nx=nx, ny=ny, seed=seed, psf=psf, ops=[], radial_r2=r2, radial_cdf=cdf, sed_tables=sed)


def _c2_objects(cat, phot):
    return catalog.build_object_table(cat, phot)


BENCH_CONFIGS = {
    "c2": dict(
        n_objects=10000,
        workload="C2: 10k-source synthetic instcat, photon_shooting, Gaussian atmPSF, Silicon sensor off, 4096x4096 CCD",
        scene=scene_c2,
        objects=_c2_objects,
        make_step=lambda renderer, objects: renderer.prepared(objects),
        bytes_per_photon=8,
        kernel="k_shoot_accumulate",
        cpu_sample=10000,
        cpu_scene=lambda scene: scene,
        cpu_step=lambda orc, sample: orc.render(sample),
    ),
}
