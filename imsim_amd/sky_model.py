"""Sky-background inputs of the step right after the draw loop (LSST_ImageBuilder.addNoise, imsim/lsst_image.py:128-200):
the planar sky gradient and the fringing map that multiply the sky level per pixel (with the vignetting map of
imsim_amd/vignetting.py) before lsst_image.add_noise draws the Poisson sky on the GPU.

The sky-brightness model itself (rubin_sim.skybrightness, imsim/sky_model.py:14-85) is external data and code: a sky
level -- a number, or any callable (ra, dec) -> photons/arcsec^2 -- is passed in.
"""
import math
import os

import numpy as np
from scipy import interpolate

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


class SkyGradient:
    """Plane through the sky levels at the CCD centre, the lower-left and the lower-right corner; the call returns the
    level relative to the centre at 0-based pixel indices (imsim/sky_model.py:88-118)."""

    def __init__(self, sky_level, pix_to_world, center_pix, image_xsize):
        """sky_level(ra, dec) -> photons/arcsec^2; pix_to_world(x, y) -> (ra, dec); center_pix: image position of
        the world centre."""
        cx, cy = center_pix
        self.sky_level_center = float(sky_level(*pix_to_world(cx, cy)))
        M = np.array([[cx, cy, 1.0], [0.0, 0.0, 1.0], [float(image_xsize), 0.0, 1.0]])
        z = np.array([self.sky_level_center, float(sky_level(*pix_to_world(0.0, 0.0))),
                      float(sky_level(*pix_to_world(float(image_xsize), 0.0)))])
        self.a, self.b, self.c = np.linalg.inv(M) @ z

    def __call__(self, x, y):
        return (self.a * x + self.b * y + self.c) / self.sky_level_center

    def coefficients(self):
        """(a, b, c) of lsst_image.add_noise's `sky_gradient`: factor = a + b x + c y"""
        return (self.c / self.sky_level_center, self.a / self.sky_level_center, self.b / self.sky_level_center)


def project_gnomonic(center, coord):
    """galsim.CelestialCoord.project (gnomonic): tangent-plane position (u, v) [rad] of `coord` seen from `center`, both
    (ra, dec) in radians; +v is north, +u is WEST (GalSim's sign convention)."""
    ra0, dec0 = center
    ra, dec = coord
    cosc = math.sin(dec0) * math.sin(dec) + math.cos(dec0) * math.cos(dec) * math.cos(ra - ra0)
    u = -math.cos(dec) * math.sin(ra - ra0) / cosc
    v = (math.cos(dec0) * math.sin(dec) - math.sin(dec0) * math.cos(dec) * math.cos(ra - ra0)) / cosc
    return u, v


class CCD_Fringing:
    """Normalised fringing map of one CCD (imsim/sky_model.py:121-243): a fractal height field, fixed per sensor by
    `seed` (the hash of the sensor's serial number, lsst_image.py:184-186), turned into a cos() interference pattern
    of relative amplitude `amplitude` x the OH sky-line level at the CCD's place in the field of view.  The height field
    draws from numpy's PCG64 stream (the reference draws from GalSim's BaseDeviate): same statistics, other numbers."""

    def __init__(self, true_center, boresight, seed, spatial_vary=True, data_dir=None):
        """true_center, boresight: (ra, dec) in radians"""
        self.true_center = true_center
        self.boresight = boresight
        self.seed = int(seed)
        self.spatial_vary = spatial_vary
        self.data_dir = data_dir or DATA_DIR

    def generate_heightfield(self, fractal_dimension=2.5, n=4096):
        """Spectral synthesis of the silicon thickness map behind the fringes (imsim/sky_model.py:152-171 does the same
        thing): a Gaussian random field whose Fourier amplitude falls as |k|^(2 p), p = -(H + 1) / 1.2 with the Hurst
        exponent H = 3 - D of a surface of fractal dimension D, tapered by exp(-|k|^2 / k0^2) with k0 the n/64-th
        frequency of the grid, no mean.  Every mode gets a normal amplitude and a uniform phase; the complex inverse
        transform is returned (the caller takes the real part)."""
        hurst = 3.0 - fractal_dimension
        p = -(hurst + 1.0) / 1.2
        f = np.fft.fftfreq(n)
        f2 = f * f
        k2 = f2[:, None] + f2[None, :]
        k2[0, 0] = 1.0                                   # the mean: removed below
        envelope = np.exp(p * np.log(k2) - k2 / f[n // 64] ** 2)
        envelope[0, 0] = 0.0
        rng = np.random.default_rng(self.seed)
        angle = (2.0 * np.pi) * rng.uniform(size=(n, n))
        strength = envelope * rng.normal(size=(n, n))
        return np.fft.ifft2(strength * np.cos(angle) + 1j * (strength * np.sin(angle)))

    def simulate_fringes(self, amp=0.002, n_side=4096):
        n, n1, nwaves_rms = 1.2, 1.5, 10.0
        X = self.generate_heightfield(n, n_side)
        X *= nwaves_rms / np.std(X.real)
        return amp * np.cos(2 * n1 * X.real)

    def fringe_variation_level(self):
        """OH sky-line level at the CCD relative to the centre of the field (skyline_var.fits, :203-219)"""
        if not self.spatial_vary:
            return 1
        from . import fits_io
        hdr, z = next((h, d) for h, d in fits_io.read_fits(os.path.join(self.data_dir, "fringing_data", "skyline_var.fits"))
                      if d is not None)
        z = np.asarray(z, dtype=np.float64)
        nx, ny = z.shape
        x = np.linspace(float(hdr["XMIN"]), float(hdr["XMAX"]), nx)
        y = np.linspace(float(hdr["YMIN"]), float(hdr["YMAX"]), ny)
        interp = interpolate.RectBivariateSpline(x, y, z)
        dx, dy = project_gnomonic(self.boresight, self.true_center)
        return float(interp(math.degrees(dx), math.degrees(dy))[0, 0] / interp(0, 0)[0, 0])

    def calculate_fringe_amplitude(self, x, y, amplitude=0.002, n_side=4096):
        level = self.fringe_variation_level()
        fringe_im = self.simulate_fringes(amp=amplitude * level, n_side=n_side)
        if (np.all(fringe_im) != True) or (True in np.isnan(fringe_im)):          # noqa: E712 -- the reference's check
            raise ValueError(" 0 or nan value in the fringe map!")
        fringe_im += 1
        xx = np.arange(fringe_im.shape[-1])
        yy = np.arange(fringe_im.shape[0])
        interp_func = interpolate.RegularGridInterpolator((xx, yy), fringe_im.T)
        return interp_func((x, y))


def sensor_seed(serial_number):
    """`int(sha256(serial).hexdigest(), 16) & 0xFFFFFFFF` (imsim/lsst_image.py:184-186)"""
    import hashlib
    return int(hashlib.sha256(serial_number.encode("UTF-8")).hexdigest(), 16) & 0xFFFFFFFF
