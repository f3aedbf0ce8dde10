"""FFT branch, CPU side: the oracle's k-space fill / Poisson deviate and the host decision logic."""
import math

import numpy as np
import pytest

from imsim_amd import _abi, configs, catalog, fft_draw, tables
from oracle import orc_loader


def _one_object(flux=2.0e6, kind=0, hlr=0.5, nx=256):
    scene = configs.scene_c2(nx=nx, ny=nx)
    cat = dict(x=np.array([100.3]), y=np.array([120.7]), mag=np.array([15.0]), nominal_flux=np.array([flux]),
               kind=np.array([kind]), hlr=np.array([hlr]), q=np.array([0.6]), pa=np.array([30.0]), obj_id=np.array([7]))
    objects, _ = catalog.build_object_table(cat, np.array([int(flux)]), stamp_size=64)
    return scene, objects


def test_poisson_deviate_moments():
    for mean in (0.3, 3.0, 9.5, 10.5, 47.0, 1.0e3, 2.0e6):
        k = orc_loader.poisson_probe(np.full(200000, mean), seed=11, obj_id=3)
        assert np.all(k >= 0) and np.all(k == np.floor(k))
        assert abs(k.mean() - mean) < 5 * np.sqrt(mean / len(k)) + 1e-12
        assert abs(k.var() / mean - 1.0) < 0.03
    assert np.all(orc_loader.poisson_probe(np.zeros(10)) == 0)


def test_fft_image_conserves_flux_and_is_centred():
    scene, objects = _one_object()
    fwhm_atm, fwhm_sys = catalog.kolmogorov_gaussian_fwhm()
    orc = orc_loader.OracleFft(scene, fft_draw.kolmogorov_gaussian_kpsf(fwhm_atm, fwhm_sys), add_noise=False)
    rows, _ = fft_draw.build_fft_objects(objects, objects["n_phot"].astype(float), objects["prof_table"])
    kbuf = orc.fill(rows)
    assert kbuf[0].real == rows["flux"][0] and kbuf[0].imag == 0.0
    img = orc.inverse(rows, kbuf).reshape(64, 64)
    assert abs(img.sum() / rows["flux"][0] - 1.0) < 1e-9            # DC term = flux
    yy, xx = np.mgrid[0:64, 0:64]
    cx, cy = (img * xx).sum() / img.sum(), (img * yy).sum() / img.sum()
    assert abs(cx - rows["cx"][0]) < 0.02 and abs(cy - rows["cy"][0]) < 0.02
    assert img.min() > -1e-6 * img.max()
    real = np.zeros(1)
    orc.finish(rows, img.ravel(), real)
    assert abs(real[0] / rows["flux"][0] - 1.0) < 1e-3
    assert abs(orc.image.sum() - real[0]) < 1e-6 * real[0]


def test_sersic_ktable_matches_exponential_closed_form():
    from scipy import special
    q, F = tables.sersic_ktable(1.0)
    b = special.gammaincinv(2.0, 0.5)
    np.testing.assert_allclose(F, (1 + (q / b) ** 2) ** -1.5, atol=1e-10)
    q4, F4 = tables.sersic_ktable(4.0)
    assert F4[0] == 1.0 and np.all(np.diff(F4[:500]) < 0)


def test_fft_decision_follows_the_reference_rules():
    """stamp.py:275-277: phot unless flux >= 1e6, fft_sb_thresh set, and the peak SB exceeds it."""
    flux = np.array([5.0e5, 2.0e6, 2.0e6, 5.0e7])
    kind = np.array([0, 0, 2, 2])
    hlr = np.array([0.0, 0.0, 3.0, 0.2])
    assert not fft_draw.use_fft(flux, kind, hlr, 0.8, 0.0).any()
    got = fft_draw.use_fft(flux, kind, hlr, 0.8, 2.0e5)
    assert list(got) == [False, False, False, True] or list(got) == [False, True, False, True]


# ---------------- diffraction spikes of FFT-drawn objects ----------------
import ctypes as C
import os

from imsim_amd import diffraction_fft as dfft

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "diffraction_fft_golden.npz")


def test_spike_stencil_matches_reference():
    """prepare_psf_field_rotation (imsim/diffraction_fft.py:78-123) for three (wavelength, alpha,
    d_alpha) cases: host numpy stencil and the oracle's C stencil against the reference's output."""
    g = np.load(GOLD)
    lib = orc_loader.load()
    for k in range(3):
        w, lam, alpha, dalpha = g[f"psf{k}_args"]
        w = int(w)
        kc = dfft.SpikeConstants(np.cos(alpha - dalpha / 2), np.sin(alpha - dalpha / 2), alpha - dalpha, dalpha,
                                 dfft.SPIKE_WAVELENGTH / lam, w)
        rng = np.arange(-w, w + 1)
        mine = dfft.stencil(rng[:, None], rng[None, :], kc)
        np.testing.assert_allclose(mine / mine.sum(), g[f"psf{k}"], rtol=0, atol=2e-15)
        S = _abi.Spikes(1, w, 1e5, kc.cos0, kc.sin0, kc.a_lo, kc.d_alpha, kc.scale, dfft.SPIKE_R0, 1.0)
        out = np.empty((2 * w + 1, 2 * w + 1))
        lib.orc_test_stencil(C.byref(S), w, out.ctypes.data)
        np.testing.assert_allclose(out / out.sum(), g[f"psf{k}"], rtol=0, atol=2e-15)


def test_apply_diffraction_psf_matches_reference():
    """apply_diffraction_psf (imsim/diffraction_fft.py:126-208) on a 40x44 image with a saturated
    core: field-rotation angle, saturated bounding box, box (x) stencil, cropping."""
    g = np.load(GOLD)
    lam, rot, exptime, lat, az, alt, thr, cutoff = g["apply_args"]
    cfg = dfft.DiffractionFFT(exptime=exptime, azimuth=az, altitude=alt, rotTelPos=rot, spike_length_cutoff=cutoff,
                              brightness_threshold=thr, latitude=lat)
    img = g["apply_in"]
    ny, nx = img.shape
    from imsim_amd.engine import Scene
    scene = Scene(nx=nx, ny=ny, seed=1)
    orc = orc_loader.OracleFft(scene, [], add_noise=False, diffraction_fft=cfg, wavelength=lam)
    rows = np.zeros(1, dtype=_abi.FFT_OBJECT_DTYPE)
    n = 64
    rows["nfft"], rows["x0"], rows["y0"] = n, 1, 1
    rows["stamp_xmin"], rows["stamp_xmax"], rows["stamp_ymin"], rows["stamp_ymax"] = 1, nx, 1, ny
    grid = np.zeros((n, n))
    grid[:ny, :nx] = img
    out = orc.spikes(rows, grid.ravel()).reshape(n, n)[:ny, :nx]
    np.testing.assert_allclose(out, g["apply_out"], rtol=1e-12, atol=1e-9)
    assert abs(out.sum() / img.sum() - 1) < 0.05      # flux is moved into the spikes, not created


@pytest.mark.parametrize("rot", [0.0, 0.3, 0.698, math.pi / 4, 1.3, math.pi / 2, 2.1, 3.0])
@pytest.mark.parametrize("d_alpha", [2.2e-3, -2.2e-3, 0.04, -0.04, 0.0])
def test_stencil_early_exit_returns_exactly_the_zeros_of_the_unshortcut_stencil(rot, d_alpha):
    """The spike stencil of the oracle (and, in the same words, of the kernel: csrc/ims_fft.h) returns before its arctangents
    for offsets it knows to be exact zeros -- further than a pixel from both arms and outside the wedge the field rotation
    sweeps.  Checked here against the numpy stencil that has no such shortcut (imsim_amd.diffraction_fft.stencil, itself
    pinned by vectors generated from the reference's prepare_psf_field_rotation): over the whole (2 w + 1)^2 grid, for arms
    along the axes, the diagonals and in between, and both signs of the swept angle, the two have the SAME set of zeros and
    agree to rounding elsewhere."""
    w = 150
    alpha = math.pi / 4.0 - rot
    k = dfft.SpikeConstants(math.cos(alpha - d_alpha / 2.0), math.sin(alpha - d_alpha / 2.0), alpha - d_alpha, d_alpha, 1.1, w)
    rng = np.arange(-w, w + 1)
    want = dfft.stencil(rng[:, None], rng[None, :], k)
    S = _abi.Spikes(1, w, 1e5, k.cos0, k.sin0, k.a_lo, k.d_alpha, k.scale, dfft.SPIKE_R0, 1.0)
    got = np.empty((2 * w + 1, 2 * w + 1))
    orc_loader.load().orc_test_stencil(C.byref(S), w, got.ctypes.data)
    assert np.array_equal(got == 0.0, want == 0.0), "the early exit and the full expression disagree on which offsets are zero"
    assert np.count_nonzero(want) > 4 * w and np.count_nonzero(want == 0.0) > 0.5 * want.size
    np.testing.assert_allclose(got, want, rtol=0, atol=4e-16 * want.max())


@pytest.mark.parametrize("rot,box", [(0.698, 9), (0.0, 13), (math.pi / 4, 5), (math.pi / 2, 11), (2.1, 17)])
def test_spike_convolution_visits_exactly_the_nonzero_terms(rot, box):
    """The box (x) stencil sum of apply_diffraction_psf (imsim/diffraction_fft.py:176-208) as the oracle forms it -- whole
    sums skipped far from every arm, of a source row only the columns near the two arms through the target pixel -- equals
    the plain double sum over the whole box with the oracle's own stencil values, term by term in the same order: bit for
    bit, for arms along the axes, along the diagonals and in between."""
    n, w = 96, 60
    cfg = dfft.DiffractionFFT(exptime=30.0, azimuth=math.radians(114.39), altitude=math.radians(53.16), rotTelPos=rot,
                              spike_length_cutoff=w, brightness_threshold=1e5)
    from imsim_amd.engine import Scene
    orc = orc_loader.OracleFft(Scene(nx=n, ny=n, seed=1), [], add_noise=False, diffraction_fft=cfg, wavelength=622.0)
    rows = np.zeros(1, dtype=_abi.FFT_OBJECT_DTYPE)
    rows["nfft"], rows["x0"], rows["y0"] = n, 1, 1
    rows["stamp_xmin"], rows["stamp_xmax"], rows["stamp_ymin"], rows["stamp_ymax"] = 1, n, 1, n
    rng = np.random.default_rng(int(rot * 1000) + box)
    grid = rng.random((n, n)) * 10.0
    cy, cx = n // 2 + 3, n // 2 - 5
    r0, r1, c0, c1 = cy - box // 2, cy + box // 2, cx - box // 2, cx + box // 2
    grid[r0:r1 + 1, c0:c1 + 1] = 2e5 * (1 + rng.random((box, box)))
    out = orc.spikes(rows, grid.ravel()).reshape(n, n)
    S = orc.P.spikes
    st = np.empty((2 * w + 1, 2 * w + 1))
    orc_loader.load().orc_test_stencil(C.byref(S), w, st.ctypes.data)
    want = grid.copy()
    want[r0:r1 + 1, c0:c1 + 1] = 0.0                               # the saturated box is replaced by its spread
    for iy in range(n):
        for ix in range(n):
            acc = 0.0
            for ry in range(r0, r1 + 1):
                a = iy - ry
                if abs(a) > w:
                    continue
                for rx in range(c0, c1 + 1):
                    b = ix - rx
                    if abs(b) > w or st[a + w, b + w] == 0.0:
                        continue
                    acc = acc + st[a + w, b + w] / S.norm * grid[ry, rx]
            want[iy, ix] = want[iy, ix] + acc
    assert np.array_equal(out.view(np.uint64), want.view(np.uint64))
    assert np.count_nonzero(out != grid) > 4 * box * box           # the cross reaches well beyond the box


# ---------------------------------------------------------------------------------------------
# make_fft_psf stand-ins: VonKarman for the PhaseScreenPSF, Airy for the SecondKick (psf_utils.py:94-149)
# ---------------------------------------------------------------------------------------------
def _profile_fwhm(T, p0, q_step, theta_max=4.0):
    """FWHM [arcsec] of the real-space profile whose MTF is T(k p0), by Hankel transform"""
    from scipy import special
    q = np.arange(len(T)) * q_step
    k = q / p0                                           # rad / arcsec
    theta = np.linspace(0.0, theta_max, 4001)
    I = np.array([np.trapezoid(T * special.j0(k * t) * k, k) for t in theta])
    half = I[0] / 2.0
    i = np.argmax(I < half)
    return 2.0 * np.interp(half, [I[i], I[i - 1]], [theta[i], theta[i - 1]])


def test_vonkarman_ktable_limits_and_seeing():
    from imsim_amd import atm_psf, tables
    lam, r0_500 = 622.2, 0.16
    r0 = r0_500 * (lam / 500.0) ** 1.2
    q_step = tables.KTABLE_QMAX / (tables.KTABLE_NPTS - 1)
    # very large outer scale: the Kolmogorov MTF exp(-3.44 (r/r0)^(5/3)) (the limit is approached as
    # (r/L0)^(1/3), i.e. slowly)
    T, p0 = fft_draw.vonkarman_ktable(lam, r0_500, 1.0e7)
    r = np.arange(len(T)) * q_step * 4.0 * r0 / tables.KTABLE_QMAX
    np.testing.assert_allclose(T, np.exp(-0.5 * 6.8839 * (r / r0) ** (5.0 / 3.0)), atol=2e-3)
    assert T[0] == 1.0 and T[-1] < 1e-12
    # finite outer scale: the FWHM follows the Tokovinin formula imSim inverts (atmPSF.py:_vkSeeing)
    for L0 in (10.0, 25.0, 60.0):
        T, p0 = fft_draw.vonkarman_ktable(lam, r0_500, L0)
        fwhm = _profile_fwhm(T, p0, q_step)
        np.testing.assert_allclose(fwhm, atm_psf.vk_seeing(r0_500, lam, L0), rtol=0.03)


def test_airy_ktable_is_the_pupil_autocorrelation():
    from imsim_amd import tables
    q_step = tables.KTABLE_QMAX / (tables.KTABLE_NPTS - 1)
    lam, diam = 622.2, 8.36
    # unobscured: closed form (2/pi)(acos v - v sqrt(1 - v^2))
    T, p0 = fft_draw.airy_ktable(lam, diam, 0.0)
    v = np.clip(np.arange(len(T)) * q_step * 1.02 / tables.KTABLE_QMAX, 0, 1)
    np.testing.assert_allclose(T, 2 / np.pi * (np.arccos(v) - v * np.sqrt(1 - v * v)), atol=1e-12)
    # obscured: autocorrelation of a sampled annulus
    T, p0 = fft_draw.airy_ktable(lam, diam, 0.61)
    n, half = 1024, 2.2 * diam
    y, x = (np.mgrid[0:n, 0:n] - n / 2) * (half * 2 / n)
    rr = np.hypot(x, y)
    pupil = ((rr <= diam / 2) & (rr >= 0.61 * diam / 2)).astype(float)
    ac = np.fft.fftshift(np.fft.ifft2(np.abs(np.fft.fft2(pupil)) ** 2).real)
    ac /= ac.max()
    shifts = np.arange(0, n // 2) * (half * 2 / n)
    num = ac[n // 2, n // 2:]
    tab = np.interp(shifts, np.arange(len(T)) * q_step * 1.02 * diam / tables.KTABLE_QMAX, T)
    np.testing.assert_allclose(num, tab, atol=6e-3)
    assert T[0] == 1.0 and np.all(T[-30:] == 0.0)
    # cutoff at k = 2 pi D / lam
    k_cut = 2 * np.pi * diam / (lam * 1e-9) / fft_draw.ARCSEC_PER_RAD
    assert abs(tables.KTABLE_QMAX / 1.02 / p0 / k_cut - 1) < 1e-12


def test_max_sb_is_galsims_convolution_estimate():
    """get_fft_psf_maybe (imsim/psf_utils.py:201-212) thresholds `Convolve(gal, fft_psf).withFlux(F).max_sb / 2 * scale^2`.
    GalSim's max_sb of a convolution is F / sum_i(flux_i / max_sb_i): exact for Gaussians, where the variances add."""
    import math
    from imsim_amd import _abi
    s1, s2, F = 0.3, 0.4, 5.0e6
    kpsf = [(_abi.IMS_KPSF_GAUSSIAN, 0, s1), (_abi.IMS_KPSF_GAUSSIAN, 0, s2)]
    peaks = fft_draw.kpsf_peak_per_flux(kpsf)
    got = fft_draw.max_surface_brightness(np.array([F]), np.array([0]), np.array([0.0]), psf_peaks=peaks)[0]
    np.testing.assert_allclose(got, F / (2 * math.pi * (s1 * s1 + s2 * s2)) / 2 * 0.04, rtol=1e-13)
    # Kolmogorov: central intensity (3/5) Gamma(6/5) k0^2 / 2 pi, against the quadrature of exp(-(k/k0)^(5/3))
    k0 = 7.3
    k = np.linspace(0.0, 30 * k0, 400001)
    num = np.trapezoid(np.exp(-(k / k0) ** (5.0 / 3.0)) * k, k) / (2 * math.pi)
    np.testing.assert_allclose(fft_draw.kpsf_peak_per_flux([(_abi.IMS_KPSF_KOLMOGOROV, 0, k0)])[0], num, rtol=1e-6)
    # a tabulated MTF: the table of a Gaussian gives the Gaussian's peak
    q = np.linspace(0.0, tables.KTABLE_QMAX, tables.KTABLE_NPTS)
    p0 = tables.KTABLE_QMAX / (9.0 / s1)
    tab = np.exp(-0.5 * (q / p0 * s1) ** 2)
    got = fft_draw.kpsf_peak_per_flux([(_abi.IMS_KPSF_TABLE, 2, p0)], [tab], q[1] - q[0])[0]
    np.testing.assert_allclose(got, 1 / (2 * math.pi * s1 * s1), rtol=1e-4)
    # a Sersic galaxy adds flux / peak of its own: n = 1 peak is b^2 / (2 pi hlr^2); magnification dilutes it
    from scipy import special
    b = special.gammaincinv(2.0, 0.5)
    hlr = 0.5
    sb = fft_draw.max_surface_brightness(np.array([F, F]), np.array([1, 1]), np.array([hlr, hlr]), psf_peaks=peaks,
                                         jac_det=np.array([1.0, 2.0]))
    want = F / (2 * math.pi * (s1 * s1 + s2 * s2) + 2 * math.pi * hlr * hlr / b ** 2) / 2 * 0.04
    np.testing.assert_allclose(sb[0], want, rtol=1e-12)
    assert sb[1] < sb[0]
    # the index matters: a de Vaucouleurs profile is far peakier than an exponential of the same size
    sb14 = fft_draw.max_surface_brightness(np.array([F, F]), np.array([1, 2]), np.array([hlr, hlr]), psf_peaks=peaks,
                                           sersic_n=np.array([1.0, 4.0]))
    assert sb14[1] > sb14[0]


def test_vignetting_spline_and_pixel_radii():
    """imsim/vignetting.py: radial B-spline normalised at the focal-plane centre; tests/test_vignetting.py checks that the
    per-pixel map and the value at a position agree at the CCD corners."""
    from imsim_amd.vignetting import Vignetting, detector_center_mm
    v = Vignetting("LSSTCam_vignetting_data.json")
    assert abs(v.apply_to_radii(0.0) - 1.0) < 1e-15
    r = np.linspace(0.0, 350.0, 50)
    f = v.apply_to_radii(r)
    assert f[0] == 1.0 and f[-1] < 0.8 and np.all(f <= 1.0 + 1e-3)             # throughput falls towards the edge of the field
    assert detector_center_mm("R22_S11") == (0.0, 0.0) and detector_center_mm("R30_S11") == (127.0, -254.0)
    for det in ("R22_S11", "R30_S11", "R01_S20"):
        img = v(det, 4096, 4004)
        assert img.shape == (4004, 4096)
        for (cx, cy) in ((0, 0), (4095, 0), (0, 4003), (4095, 4003)):
            np.testing.assert_almost_equal(v.at_pixel(det, cx + 1.0, cy + 1.0, 4096, 4004), img[cy, cx])
    assert abs(v("R22_S11", 4096, 4004).mean() - 1.0) < 5e-3 and v("R01_S20", 4096, 4004).mean() < v("R22_S11", 4096, 4004).mean()
    import pytest
    with pytest.raises(OSError):
        Vignetting("no_such_file.json")
    assert Vignetting("LSSTComCamSim_vignetting_data.json").apply_to_radii(0.0) == 1.0


def test_airy_ktable_has_galsims_fwhm_and_half_light_radius():
    """galsim.Airy, make_fft_psf's stand-in for the second kick (imsim/psf_utils.py:112-115): GalSim documents
    fwhm = 1.028993 lam / D and half_light_radius = 0.5348321 lam / D for an unobscured aperture.  The k-table the fill
    kernel multiplies in (autocorrelation of the pupil over baselines, `fft_draw.airy_ktable`) Hankel-transforms to a
    profile with that FWHM and that half-light radius."""
    from scipy import special, integrate
    lam, diam = 700.0, 8.36
    T, _ = fft_draw.airy_ktable(lam, diam, 0.0)
    from imsim_amd import tables
    b = np.linspace(0.0, 1.02 * diam, len(T))                        # baselines of the table [m]
    k = 2.0 * np.pi * b / (lam * 1.0e-9)                             # rad^-1
    unit = lam * 1.0e-9 / diam                                       # lam / D [rad]
    theta = np.linspace(0.0, 3.0, 3001) * unit
    prof = np.array([integrate.simpson(T * special.j0(k * t) * k, x=k) for t in theta])
    half = np.interp(0.5 * prof[0], prof[:1300][::-1], theta[:1300][::-1])       # the main lobe is monotonic to 1.22 lam / D
    assert abs(2.0 * half / unit / 1.028993 - 1.0) < 2.0e-3
    enclosed = np.array([t * integrate.simpson(T * special.j1(k * t), x=k) for t in theta])      # F(theta) = theta int T J1(k theta) dk
    hlr = np.interp(0.5, enclosed[:1300], theta[:1300])
    assert abs(hlr / unit / 0.5348321 - 1.0) < 5.0e-3


def test_image_from_rbuf_applies_each_grids_own_normalisation(monkeypatch):
    """fft_draw.image_from_rbuf (the checker's view of FftDrawer's real-space buffer): with the raw inverse (IMS_FFT_RAW, the default)
    every object's grid is multiplied by 1 / (nfft * nfft) -- its own, by the product the kernels form on reading; with the scaling
    pass or the torch front end the buffer already holds the images."""
    from imsim_amd import fft_draw
    rows = np.zeros(3, dtype=fft_draw.FFT_OBJECT_DTYPE)
    rows["nfft"] = (4, 6, 8)
    rng = np.random.default_rng(5)
    buf = rng.normal(size=16 + 36 + 64) * 1e6
    monkeypatch.delenv("IMS_FFT_RAW", raising=False)
    monkeypatch.delenv("IMS_FFT_TORCH", raising=False)
    assert fft_draw.raw_inverse()
    got = fft_draw.image_from_rbuf(rows, buf)
    want = np.concatenate([buf[:16] * (1.0 / 16.0), buf[16:52] * (1.0 / 36.0), buf[52:] * (1.0 / 64.0)])
    assert np.array_equal(got, want) and got is not buf
    monkeypatch.setenv("IMS_FFT_RAW", "0")
    assert not fft_draw.raw_inverse() and np.array_equal(fft_draw.image_from_rbuf(rows, buf), buf)
    monkeypatch.setenv("IMS_FFT_RAW", "1")
    monkeypatch.setenv("IMS_FFT_TORCH", "1")
    assert not fft_draw.raw_inverse()
    assert np.array_equal(fft_draw.image_from_rbuf(rows, buf, raw=True), want)
