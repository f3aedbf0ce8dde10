"""Atmospheric PSF: host-side mirror of imsim/atmPSF.py (AtmosphericPSF :83-336, AtmLoader :339-426).

What the reference builds once per visit -- six frozen-flow von Karman phase screens
(`galsim.Atmosphere(...).instantiate(kmax=kcrit/r0)`, :164-191) and the `SecondKick` for the
turbulence above kcrit (:195-202) -- is built here as plain arrays; shooting photons through
them (the per-photon wavefront-gradient gather + chromatic dilation + second kick) runs in the
HIP kernels (`IMS_PSF_SCREENS`, csrc/ims_photon.h::screen_gradient).

GalSim's screen synthesis is restated from its published algorithm (white Gaussian noise
filtered by the square root of the von Karman spectrum 0.00058 r0^(-5/3) (f^2 + 1/L0^2)^(-11/6)
waves^2 m^2 at 500 nm); the reference's random streams (boost mt19937) are not reproducible, so
layer parameters come from a numpy Generator seeded like `AtmLoader` does (random_seed + 271828,
:406-417).
"""
import math

import numpy as np
from scipy import special
from scipy.optimize import bisect

from . import _abi, tables

ARCSEC = 180.0 / math.pi * 3600.0
WLEN_EFF = dict(u=365.49, g=480.03, r=622.20, i=754.06, z=868.21, y=991.66)   # atmPSF.py:125


def kolmogorov_fwhm(r0_500, wavelength):
    """FWHM [arcsec] of galsim.Kolmogorov(r0_500=, lam=): 0.9758634299 lam / r0."""
    r0 = r0_500 * (wavelength / 500.0) ** 1.2
    return 0.9758634299 * wavelength * 1.0e-9 / r0 * ARCSEC


def vk_seeing(r0_500, wavelength, L0):
    """von Karman FWHM from the Tokovinin (2002) fitting formula (atmPSF.py:217-226)."""
    kolm = kolmogorov_fwhm(r0_500, wavelength)
    r0 = r0_500 * (wavelength / 500.0) ** (6.0 / 5)
    arg = 1.0 - 2.183 * (r0 / L0) ** 0.356
    return kolm * (math.sqrt(arg) if arg > 0.0 else 0.0)


def r0_500_for_seeing(wavelength, L0, target_seeing):
    """r0_500 giving the target von Karman seeing (atmPSF.py:232-242)."""
    r0_max = min(1.0, L0 * (1.0 / 2.183) ** (-0.356) * (wavelength / 500.0) ** (6.0 / 5))
    return bisect(lambda r: vk_seeing(r, wavelength, L0) - target_seeing, 0.01, r0_max)


def von_karman_screen(npix, scale, r0_500, L0, rng, kmax=np.inf, xp=np):
    """One phase screen [nm of optical path] (galsim AtmosphericScreen restated): white Gaussian
    noise filtered in Fourier space by psi(f) = sqrt(0.00058) r0^(-5/6) (f^2 + L0^-2)^(-11/12)
    * npix / screen_size * 500 (so that the screen has the von Karman structure function: variance per
    mode = PSD * df^2), low-passed at 2 pi |f| <= kmax (the part above kmax is
    the second kick).  `xp` = numpy, or torch for an on-device build of the 8192^2 screens."""
    size = npix * scale
    if xp is np:
        fx = np.fft.fftfreq(npix, scale)
        fsq = fx[None, :] ** 2 + fx[:, None] ** 2
        with np.errstate(divide="ignore"):
            psi = (math.sqrt(0.00058) * r0_500 ** (-5.0 / 6.0) * (fsq + 1.0 / L0 ** 2) ** (-11.0 / 12.0)
                   * npix / size * 500.0)
        psi[0, 0] = 0.0
        if np.isfinite(kmax):
            psi[(2 * np.pi) ** 2 * fsq > kmax ** 2] = 0.0
        noise = rng.standard_normal((npix, npix))
        return np.ascontiguousarray(np.fft.ifft2(np.fft.fft2(noise) * psi).real)
    torch = xp
    dev = rng.device
    fx = torch.fft.fftfreq(npix, scale, dtype=torch.float64, device=dev)
    fsq = fx[None, :] ** 2 + fx[:, None] ** 2
    psi = (math.sqrt(0.00058) * r0_500 ** (-5.0 / 6.0) * (fsq + 1.0 / L0 ** 2) ** (-11.0 / 12.0)
           * npix / size * 500.0)
    psi[0, 0] = 0.0
    if np.isfinite(kmax):
        psi[(2 * math.pi) ** 2 * fsq > kmax ** 2] = 0.0
    noise = torch.randn((npix, npix), dtype=torch.float64, device=dev, generator=rng)
    return torch.fft.ifft2(torch.fft.fft2(noise) * psi).real.contiguous()


# ---------------- second kick ----------------
def _circle_overlap(r1, r2, d):
    d = np.asarray(d, dtype=np.float64)
    out = np.zeros_like(d)
    inside = d <= abs(r1 - r2)
    out[inside] = math.pi * min(r1, r2) ** 2
    mid = (~inside) & (d < r1 + r2)
    dm = d[mid]
    a1 = np.arccos(np.clip((dm ** 2 + r1 ** 2 - r2 ** 2) / (2 * dm * r1), -1, 1))
    a2 = np.arccos(np.clip((dm ** 2 + r2 ** 2 - r1 ** 2) / (2 * dm * r2), -1, 1))
    out[mid] = r1 ** 2 * a1 + r2 ** 2 * a2 - 0.5 * np.sqrt(np.clip((-dm + r1 + r2) * (dm + r1 - r2) * (dm - r1 + r2) * (dm + r1 + r2), 0, None))
    return out


def annulus_mtf(rho, diam, obscuration):
    R, e = diam / 2.0, obscuration
    a = _circle_overlap(R, R, rho) + _circle_overlap(e * R, e * R, rho) - 2.0 * _circle_overlap(R, e * R, rho)
    return a / (math.pi * R * R * (1.0 - e * e))


def second_kick_table(lam, r0, diam, obscuration, kcrit, n_bins=tables.N_BINS, theta_max=30.0):
    """Radial table [arcsec] of galsim.SecondKick(lam, r0, diam, obscuration, kcrit)
    (atmPSF.py:195-202): the expectation of the turbulence above kcrit/r0 convolved with the
    annular-aperture diffraction pattern.  MTF(rho) = MTF_annulus(rho) exp(-D_hk(rho)/2) with D_hk
    the Kolmogorov structure function restricted to spatial frequencies above kcrit / (2 pi r0)."""
    rho = np.linspace(0.0, diam, 40001)
    kc = kcrit / r0 / (2.0 * math.pi)                    # cycles / m
    # low-k part of the structure function, by quadrature in kappa
    kap = np.linspace(0.0, kc, 2001)[1:]
    w = np.gradient(np.concatenate([[0.0], kap]))[1:]
    x = 2.0 * math.pi * kap[None, :] * rho[::40, None]
    d_low_c = 2.0 * 0.0228 * 2.0 * math.pi * r0 ** (-5.0 / 3.0) * np.sum(kap[None, :] ** (-8.0 / 3.0) * (1.0 - special.j0(x)) * w[None, :], axis=1)
    d_low = np.interp(rho, rho[::40], d_low_c)
    d_hk = np.clip(6.8839 * (rho / r0) ** (5.0 / 3.0) - d_low, 0.0, None)
    T = annulus_mtf(rho, diam, obscuration) * np.exp(-0.5 * d_hk)
    theta = np.concatenate([[0.0], np.geomspace(1.0e-4, theta_max, n_bins)])      # arcsec
    th_rad = theta / ARCSEC
    lam_m = lam * 1.0e-9
    I = np.empty_like(theta)
    for i, th in enumerate(th_rad):
        I[i] = np.trapezoid(T * special.j0(2.0 * math.pi * rho * th / lam_m) * rho, rho)
    I = np.clip(I, 0.0, None)
    F = np.concatenate([[0.0], np.cumsum(0.5 * (I[1:] * theta[1:] + I[:-1] * theta[:-1]) * np.diff(theta))])
    F = np.maximum.accumulate(F)
    return theta * theta, F / F[-1]


class AtmosphericPSF:
    """Mirror of imsim.atmPSF.AtmosphericPSF: same constructor arguments and derived quantities
    (targetFWHM :128, six layers with Ellerbroek altitudes/weights, truncated log-normal L0,
    r0_500 solved for the target seeing, speeds <= 20 m/s, isotropic directions :244-296)."""

    def __init__(self, airmass, rawSeeing, band, boresight=None, seed=0, t0=0.0, exptime=30.0, kcrit=0.2,
                 screen_size=819.2, screen_scale=0.1, exponent=-0.3, no2k=False, device=None):
        self.airmass, self.rawSeeing, self.boresight = airmass, rawSeeing, boresight
        self.wlen_eff = WLEN_EFF[band]
        self.targetFWHM = rawSeeing * airmass ** 0.6 * (self.wlen_eff / 500.0) ** (-0.3)
        self.t0, self.exptime, self.exponent, self.kcrit = t0, exptime, exponent, kcrit
        self.screen_scale = screen_scale
        self.npix = int(round(screen_size / screen_scale))
        self.screen_size = self.npix * screen_scale
        self.diam, self.obscuration = 8.36, 0.61                                  # atmPSF.py:168
        rng = np.random.default_rng(int(seed) + 271828)                           # AtmLoader, atmPSF.py:415
        kw = self._get_atm_kwargs(rng)
        self.__dict__.update(kw)
        self.r0 = self.r0_500 * (self.wlen_eff / 500.0) ** (6.0 / 5)
        self.kmax = kcrit / self.r0
        self.screens = self._build_screens(rng, device)
        self.second_kick = None if no2k else second_kick_table(self.wlen_eff, self.r0, self.diam, self.obscuration, kcrit)

    def _get_atm_kwargs(self, rng):
        altitudes = [0.2, 2.58, 5.16, 7.73, 12.89, 15.46]                         # km, ground layer raised (:249-252)
        weights = np.array([0.652, 0.172, 0.055, 0.025, 0.074, 0.022])
        weights = np.abs(weights * (1.0 + 0.1 * rng.standard_normal(6)))
        weights = np.clip(weights, 0.01, 0.8)
        weights /= weights.sum()
        L0 = 0.0
        while L0 < 10.0 or L0 > 100.0:
            L0 = math.exp(rng.standard_normal() * 0.6 + math.log(25.0))
        r0_500 = r0_500_for_seeing(self.wlen_eff, L0, self.targetFWHM)
        speeds = rng.uniform(0.0, 20.0, 6)
        directions = rng.uniform(0.0, 2.0 * math.pi, 6)
        return dict(altitudes=np.array(altitudes), r0_weights=weights, L0=L0, r0_500=r0_500, speeds=speeds,
                    directions=directions)

    def _build_screens(self, rng, device):
        # galsim.Atmosphere: per-layer r0_500 = r0_500 * weight^(-3/5)
        r0s = self.r0_500 * self.r0_weights ** (-3.0 / 5.0)
        # the kernels gather fp32 samples (192 -> 96 B per photon; 1e-7 relative on ~1e3 nm of path is far below
        # anything a photon can notice): the screens are synthesised in f64 and stored as fp32
        if device is None:
            return np.stack([von_karman_screen(self.npix, self.screen_scale, r0, self.L0, rng, self.kmax).astype(np.float32)
                             for r0 in r0s])
        import torch
        gen = torch.Generator(device=device)
        gen.manual_seed(int(rng.integers(1 << 62)))
        return torch.stack([von_karman_screen(self.npix, self.screen_scale, float(r0), self.L0, gen, self.kmax, xp=torch).to(torch.float32)
                            for r0 in r0s])

    # -- what the kernels consume --
    def atmosphere_struct(self):
        """_abi.Atmosphere without the screens pointer (filled when bound to a memory provider)."""
        A = _abi.Atmosphere()
        A.n_layers, A.npix, A.scale = len(self.altitudes), self.npix, self.screen_scale
        A.x0 = -0.5 * self.npix * self.screen_scale
        A.t0, A.exptime = self.t0, self.exptime
        A.aper_r_outer, A.aper_r_inner = self.diam / 2.0, self.diam * self.obscuration / 2.0
        for l in range(A.n_layers):
            A.vx[l] = float(self.speeds[l] * math.cos(self.directions[l]))
            A.vy[l] = float(self.speeds[l] * math.sin(self.directions[l]))
            A.alt[l] = float(self.altitudes[l] * 1000.0)
        return A

    def psf_components(self, second_kick_table_id):
        """PSF component tuples for Scene.psf: ChromaticAtmosphere(PhaseScreenPSF, alpha=exponent,
        base_wavelength=wlen_eff) then the achromatic second kick (atmPSF.py:305-322)."""
        comps = [(_abi.IMS_PSF_SCREENS, 0, 1.0e-9 * ARCSEC, self.exponent, self.wlen_eff)]
        if self.second_kick is not None:
            comps.append((_abi.IMS_PSF_RADIAL, second_kick_table_id, 1.0, 0.0, 1.0))
        return comps
