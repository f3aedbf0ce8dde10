"""Host-side table builders for the photon-shooting kernels (numpy/scipy, run once per visit).

Radial profile tables replace GalSim's per-profile photon shooters (SBSersic / SBKolmogorov /
SBExponential `shoot`, call sites imsim/instcat.py:484-561, imsim/atmPSF.py:534-536): a circular
profile is stored as (r^2, enclosed flux) knots and sampled by inverse CDF with uniform surface
density inside each annulus, so the centre (F ~ r^2) is exact and the error is that of a
piecewise-constant surface-brightness profile on a logarithmic radial grid.
"""
import functools
import os

import numpy as np
from scipy import special, integrate

SHOOT_ACCURACY = 1.0e-5   # GalSim GSParams.shoot_accuracy default: flux fraction allowed outside the sampled radius
N_BINS = 512


def _finish(r, F, n_bins=N_BINS):
    """Resample monotone (r, F) onto n_bins+1 knots and normalise F to [0, 1]."""
    r = np.asarray(r, dtype=np.float64)
    F = np.asarray(F, dtype=np.float64)
    F = np.maximum.accumulate(F)
    F = (F - F[0]) / (F[-1] - F[0])
    return r * r, F


@functools.lru_cache(maxsize=None)
def sersic_table(n, n_bins=N_BINS):
    """Radial table of a Sersic profile of index n, radius in units of the half-light radius.

    Enclosed flux F(r) = P(2n, b r^(1/n)) with P the regularised lower incomplete gamma function
    and b defined by F(1) = 1/2.  Sampled out to the radius leaving SHOOT_ACCURACY of the flux.
    """
    n = float(n)
    b = special.gammaincinv(2.0 * n, 0.5)
    x_max = special.gammaincinv(2.0 * n, 1.0 - SHOOT_ACCURACY)
    r_max = (x_max / b) ** n
    x_min = special.gammaincinv(2.0 * n, 1.0e-7)
    r_min = (x_min / b) ** n
    r = np.concatenate([[0.0], np.geomspace(r_min, r_max, n_bins)])
    F = special.gammainc(2.0 * n, b * r ** (1.0 / n))
    F[0] = 0.0
    return _finish(r, F, n_bins)


_KOLM_CACHE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "kolmogorov_table_%d.npz")


@functools.lru_cache(maxsize=None)
def kolmogorov_table(n_bins=N_BINS):
    """Radial table of the Kolmogorov PSF, radius in units of its FWHM (cached on disk: the
    quadrature takes ~10 s).  See _compute_kolmogorov_table."""
    path = _KOLM_CACHE % n_bins
    if os.path.exists(path):
        with np.load(path) as z:
            return z["r2"], z["cdf"]
    r2, cdf = _compute_kolmogorov_table(n_bins)
    try:
        np.savez(path, r2=r2, cdf=cdf)
    except OSError:
        pass
    return r2, cdf


def _compute_kolmogorov_table(n_bins=N_BINS):
    """MTF T(k) = exp(-k^(5/3)); enclosed flux F(r) = r int_0^inf T(k) J1(k r) dk, joined to the
    r^(-5/3) asymptote of the tail.  (galsim.Kolmogorov as used by imsim/atmPSF.py:534.)
    """
    k = np.linspace(0.0, 12.0, 240001)
    T = np.exp(-k ** (5.0 / 3.0))
    r_num = np.concatenate([[0.0], np.geomspace(1.0e-3, 60.0, 600)])
    F_num = np.empty_like(r_num)
    I_num = np.empty_like(r_num)
    for i, r in enumerate(r_num):
        F_num[i] = r * integrate.simpson(T * special.j1(k * r), x=k)
        I_num[i] = integrate.simpson(T * special.j0(k * r) * k, x=k) / (2.0 * np.pi)
    # FWHM of I(r)
    half = 0.5 * I_num[0]
    j = np.argmax(I_num < half)
    r_half = np.interp(half, [I_num[j], I_num[j - 1]], [r_num[j], r_num[j - 1]])
    fwhm = 2.0 * r_half
    # tail: 1 - F(r) = C r^(-5/3), C from the small-k expansion T ~ 1 - k^(5/3)
    A = -(2.0 ** (8.0 / 3.0)) * special.gamma(11.0 / 6.0) / special.gamma(-5.0 / 6.0) / (2.0 * np.pi)
    Ctail = 2.0 * np.pi * A * 3.0 / 5.0
    r_join = 30.0
    r_max = (Ctail / SHOOT_ACCURACY) ** 0.6
    r = np.concatenate([[0.0], np.geomspace(2.0e-3, r_max, n_bins)])
    F = np.where(r <= r_join, np.interp(r, r_num, F_num), 1.0 - Ctail * np.maximum(r, 1e-30) ** (-5.0 / 3.0))
    # continuity at the joint: rescale the numeric part to meet the asymptote
    F_join_num = np.interp(r_join, r_num, F_num)
    F_join_asym = 1.0 - Ctail * r_join ** (-5.0 / 3.0)
    F = np.where(r <= r_join, F * (F_join_asym / F_join_num), F)
    F[0] = 0.0
    return _finish(r / fwhm, F, n_bins)


def stack_radial(tables):
    """Stack [(r2, cdf), ...] into the two contiguous arrays the ABI wants."""
    r2 = np.ascontiguousarray(np.stack([t[0] for t in tables]), dtype=np.float64)
    cdf = np.ascontiguousarray(np.stack([t[1] for t in tables]), dtype=np.float64)
    return r2, cdf


def inverse_cdf_table(x, pdf, n_pts=2049):
    """Inverse CDF of a tabulated density, uniform in u in [0,1] (WavelengthSampler restated:
    wavelengths are drawn from SED(lambda) x bandpass(lambda))."""
    x = np.asarray(x, dtype=np.float64)
    pdf = np.maximum(np.asarray(pdf, dtype=np.float64), 0.0)
    cdf = np.concatenate([[0.0], np.cumsum(0.5 * (pdf[1:] + pdf[:-1]) * np.diff(x))])
    cdf /= cdf[-1]
    u = np.linspace(0.0, 1.0, n_pts)
    keep = np.concatenate([[True], np.diff(cdf) > 0])
    return np.interp(u, cdf[keep], x[keep])


def synthetic_r_band(n=361):
    """A smooth stand-in for the LSST r-band total throughput (the real table comes from
    rubin_sim throughputs, imsim/bandpass.py:62-227, which is external data)."""
    wl = np.linspace(520.0, 720.0, n)
    rise = 0.5 * (1.0 + np.tanh((wl - 552.0) / 4.0))
    fall = 0.5 * (1.0 - np.tanh((wl - 691.0) / 4.0))
    return wl, 0.55 * rise * fall * (1.0 - 0.0004 * (wl - 620.0))


class Bandpass:
    """Tabulated throughput on a wavelength grid [nm] -- the little of galsim.Bandpass the path needs: a call,
    scaling by a number (`$bandpass*0.8`, tests/test_photon_ops.py:781), the ratio of two bandpasses
    (BandpassRatio, imsim/photon_ops.py:506-533) and the effective wavelength."""

    def __init__(self, wl, thr):
        self.wl = np.asarray(wl, dtype=np.float64)
        self.thr = np.asarray(thr, dtype=np.float64)
        self.blue_limit, self.red_limit = float(self.wl[0]), float(self.wl[-1])

    def __call__(self, wave):
        return np.interp(wave, self.wl, self.thr, left=0.0, right=0.0)

    @property
    def effective_wavelength(self):
        return effective_wavelength(self.wl, self.thr)

    def __mul__(self, other):
        if isinstance(other, Bandpass):
            return Bandpass(self.wl, self.thr * other(self.wl))
        return Bandpass(self.wl, self.thr * float(other))
    __rmul__ = __mul__

    def __truediv__(self, other):
        if isinstance(other, Bandpass):
            lo, hi = max(self.blue_limit, other.blue_limit), min(self.red_limit, other.red_limit)
            wl = self.wl[(self.wl >= lo) & (self.wl <= hi)]
            den = other(wl)
            return Bandpass(wl, np.divide(self(wl), den, out=np.zeros_like(wl), where=den > 0.0))
        return Bandpass(self.wl, self.thr / float(other))

    def ratio_table(self, initial, n_pts=1025):
        """(table, wl_min, wl_step) of self / initial on a uniform grid over the common wavelength range: the
        ims_lin_tables_t row an IMS_OP_BANDPASS_RATIO op looks its photons' wavelengths up in."""
        r = self / initial
        grid = np.linspace(r.blue_limit, r.red_limit, n_pts)
        return r(grid), float(grid[0]), float(grid[1] - grid[0])


def effective_wavelength(wl, thr):
    return float(np.trapezoid(wl * thr, wl) / np.trapezoid(thr, wl))


def silicon_abs_length_table(temperature=173.0, wl_min=255.0, wl_max=1450.0, step=5.0):
    """Absorption length of silicon [micron] vs wavelength [nm] from the Rajkanan, Singh & Shewchun
    (1979) fit -- a stand-in for GalSim's share/sensors/abs_length.dat (external data)."""
    wl = np.arange(wl_min, wl_max + 0.5 * step, step)
    E = 1239.84193 / wl
    kT = 8.617333e-5 * temperature
    beta, gamma = 7.021e-4, 1108.0
    Eg0 = (1.1557, 2.5)
    Egd0 = 3.2
    Ep = (1.827e-2, 5.773e-2)
    Cc = (5.5, 4.0)
    Aa = (3.231e2, 7.237e3)
    Ad = 1.052e6
    shift = beta * temperature ** 2 / (temperature + gamma)
    alpha = np.zeros_like(E)
    for i in range(2):
        for j in range(2):
            Eg = Eg0[j] - shift
            t1 = np.clip(E - Eg + Ep[i], 0.0, None) ** 2 / (np.exp(Ep[i] / kT) - 1.0)
            t2 = np.clip(E - Eg - Ep[i], 0.0, None) ** 2 / (1.0 - np.exp(-Ep[i] / kT))
            alpha += Cc[i] * Aa[j] * (t1 + t2)
    alpha += Ad * np.sqrt(np.clip(E - (Egd0 - shift), 0.0, None))
    alpha = np.maximum(alpha, 1.0e-6)      # cm^-1
    return wl, 1.0e4 / alpha               # micron


# ---------------- k-space tables for the FFT branch ----------------
KTABLE_QMAX = 160.0
KTABLE_NPTS = 2049
_KT_CACHE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "sersic_ktable_n%s.npz")


@functools.lru_cache(maxsize=None)
def sersic_ktable(n, qmax=KTABLE_QMAX, npts=KTABLE_NPTS):
    """Hankel transform of the Sersic profile, F(q) with q = k * half_light_radius, F(0) = 1, on a
    uniform q grid (GalSim SBSersic::kValue restated: there a per-n lookup table as well).
    With x = r^(1/n):  F(q) = int exp(-b x) J0(q x^n) n x^(2n-1) dx / Gamma(2n) * b^(2n)."""
    n = float(n)
    path = _KT_CACHE % (("%g" % n).replace(".", "p"))
    if os.path.exists(path):
        with np.load(path) as z:
            if z["q"][-1] == qmax and len(z["q"]) == npts:
                return z["q"], z["F"]
    b = special.gammaincinv(2.0 * n, 0.5)
    q = np.linspace(0.0, qmax, npts)
    x = np.linspace(0.0, 40.0 / b, 120001)
    w = np.exp(-b * x) * n * x ** (2.0 * n - 1.0)
    norm = integrate.simpson(w, x=x)
    xn = x ** n
    F = np.empty_like(q)
    for a in range(0, npts, 64):
        F[a:a + 64] = integrate.simpson(w[None, :] * special.j0(q[a:a + 64, None] * xn[None, :]), x=x, axis=1) / norm
    try:
        np.savez(path, q=q, F=F)
    except OSError:
        pass
    return q, F


def gaussian_max_sb(flux, sigma):
    return flux / (2.0 * np.pi * sigma * sigma)
