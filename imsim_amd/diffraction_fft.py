"""Diffraction spikes for FFT-drawn (saturated) objects: host side of the `stamp.diffraction_fft`
option (imsim/stamp.py:36-68 DiffractionFFT, applied at :520-521; algorithm in
imsim/diffraction_fft.py:78-208).

The spike stencil -- a Lorentzian cross of angular width d_alpha (the field rotation during the
exposure), 4-fold symmetric, laid over the bounding box of the pixels above the brightness
threshold -- is evaluated ANALYTICALLY per pixel offset on the GPU (csrc/ims_fft.h::spike_stencil)
instead of being tabulated on an 8001 x 8001 grid; only its normalisation (the sum over the full
(2 cutoff + 1)^2 grid, imsim/diffraction_fft.py:120-123) is computed here, once per visit.
"""
import dataclasses
import math

import numpy as np

from . import diffraction

# Lorentzian spike profile rho(r) = 2 / (R0 pi) / (1 + (r/R0)^2), asymptotically A / r^2 with A
# fitted to photon-shooting data (imsim/diffraction_fft.py:5-15)
SPIKE_A = 0.0706052627908828
SPIKE_R0 = 0.5 * SPIKE_A * math.pi
SPIKE_WAVELENGTH = 577.6
RUBIN_LATITUDE = math.radians(-30.244633)      # lsst.obs.lsst SIMONYI_LOCATION.lat (stamp.py:8, :45)


@dataclasses.dataclass
class SpikeConstants:
    cos0: float          # cos / sin of alpha - d_alpha / 2 (axis of the anti-aliased cross)
    sin0: float
    a_lo: float          # alpha - d_alpha: lower angular edge of the spike
    d_alpha: float
    scale: float         # SPIKE_WAVELENGTH / wavelength
    cutoff: int
    norm: float = 1.0


def field_rotation_angle(latitude, azimuth, altitude, exptime):
    """Field rotation over the exposure (imsim/diffraction_fft.py:148-154 via
    imsim/diffraction.py:318-351)."""
    ez0 = diffraction.zenith_direction(latitude)
    ef = diffraction.e_equatorial(latitude, altitude, azimuth)
    w = diffraction.OMEGA_EARTH * exptime
    ez = np.array([math.cos(latitude) * math.cos(w), math.cos(latitude) * math.sin(w), math.sin(latitude)])
    eh, eh0 = np.cross(ef, ez), np.cross(ef, ez0)
    nrm = np.linalg.norm(eh) * np.linalg.norm(eh0)
    return math.atan2(float(ez @ eh0) / nrm, float(eh @ eh0) / nrm)


def stencil(a, b, k: SpikeConstants):
    """Un-normalised spike stencil at integer offsets (a along image rows, b along columns):
    the value prepare_psf_field_rotation puts at psf[w + a, h + b] before normalising."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    xr = k.cos0 * a + k.sin0 * b
    yr = -k.sin0 * a + k.cos0 * b
    val = np.maximum(0.0, 1.0 - np.minimum(np.abs(xr), np.abs(yr)))
    dth = np.mod(np.arctan2(b, a) - k.a_lo, math.pi / 2.0)
    val = np.where(dth <= k.d_alpha, 1.0, val)
    r = np.hypot(a, b)
    prof = (2.0 / math.pi) * (np.arctan((r + 0.5) * k.scale / SPIKE_R0) - np.arctan((r - 0.5) * k.scale / SPIKE_R0))
    val = val * prof / np.maximum(r * k.d_alpha, 1.0)
    return np.where((a == 0) & (b == 0), 2.0 * val, val)


def stencil_norm(k: SpikeConstants, chunk=256):
    """Sum of the stencil over the full (2 cutoff + 1)^2 grid, in row chunks."""
    rng = np.arange(-k.cutoff, k.cutoff + 1, dtype=np.float64)
    total = 0.0
    for s in range(0, len(rng), chunk):
        total += float(np.sum(stencil(rng[s:s + chunk, None], rng[None, :], k)))
    return total


@dataclasses.dataclass
class DiffractionFFT:
    """Config subsection enabling diffraction spikes in FFT mode (imsim/stamp.py:36-68); angles in
    radians."""
    exptime: float
    azimuth: float
    altitude: float
    rotTelPos: float
    spike_length_cutoff: float = 4000
    brightness_threshold: float = 1.0e5
    latitude: float = RUBIN_LATITUDE
    enabled: bool = True

    def constants(self, wavelength):
        """Spike geometry and stencil normalisation at `wavelength`: once per visit and wavelength (the normalisation is a
        sum over the (2 cutoff + 1)^2 grid -- seconds of host time; every CCD of a focal plane shares the instance)."""
        cache = self.__dict__.setdefault("_constants", {})
        key = (float(wavelength), self.exptime, self.azimuth, self.altitude, self.rotTelPos, self.spike_length_cutoff, self.latitude)
        if key not in cache:
            cache[key] = self._constants_at(wavelength)
        return cache[key]

    def _constants_at(self, wavelength):
        d_alpha = field_rotation_angle(self.latitude, self.azimuth, self.altitude, self.exptime)
        alpha = math.pi / 4.0 - self.rotTelPos                       # imsim/diffraction_fft.py:155
        k = SpikeConstants(math.cos(alpha - d_alpha / 2.0), math.sin(alpha - d_alpha / 2.0), alpha - d_alpha, d_alpha,
                           SPIKE_WAVELENGTH / wavelength, int(self.spike_length_cutoff))
        k.norm = stencil_norm(k)
        return k
