"""AtmosphericPSF host side: the reference's r0_500 inversion test (tests/test_psf.py:228-246) and
physical sanity of the synthesised screens / second kick."""
import numpy as np

from imsim_amd import atm_psf


def test_r0_500_inversion_matches_target_seeing():
    """Inversion of the Tokovinin fitting formula (tests/test_psf.py:228-246, atol 1e-3)."""
    rs = np.random.RandomState(57721)
    for _ in range(10):
        airmass = rs.uniform(1.001, 1.5)
        raw = rs.uniform(0.5, 1.5)
        band = "ugrizy"[rs.randint(6)]
        a = atm_psf.AtmosphericPSF(airmass, raw, band, seed=int(rs.randint(2 ** 31)), screen_size=6.4, no2k=True)
        wlen = atm_psf.WLEN_EFF[band]
        target = raw * airmass ** 0.6 * (wlen / 500) ** (-0.3)
        assert a.targetFWHM == target
        np.testing.assert_allclose(atm_psf.vk_seeing(a.r0_500, wlen, a.L0), target, atol=1e-3, rtol=0)
        assert 10.0 <= a.L0 <= 100.0
        assert np.all(a.speeds <= 20.0) and len(a.altitudes) == 6
        np.testing.assert_allclose(a.r0_weights.sum(), 1.0)


def test_screen_structure_function_is_von_karman():
    """The synthesised screen has the von Karman phase structure function (in rad^2 at 500 nm; the screen is in
    nm of path): 6.88 (r/r0)^(5/3) at r << L0, saturating beyond the outer scale."""
    from imsim_amd import fft_draw
    rng = np.random.default_rng(5)
    npix, scale, r0, L0 = 4096, 0.1, 0.2, 25.0
    lags = (2, 5, 10, 20, 50, 100)
    acc = {lag: [] for lag in lags}
    for _ in range(2):
        s = atm_psf.von_karman_screen(npix, scale, r0, L0, rng) * (2 * np.pi / 500.0)    # rad at 500 nm
        for lag in lags:
            acc[lag].append(0.5 * (np.mean((s[:, lag:] - s[:, :-lag]) ** 2) + np.mean((s[lag:, :] - s[:-lag, :]) ** 2)))
    for lag in lags:
        expect = fft_draw.vonkarman_structure_function(np.array([lag * scale]), r0, L0)[0]
        np.testing.assert_allclose(np.mean(acc[lag]), expect, rtol=0.06, err_msg=f"lag {lag}")
    # and the Kolmogorov limit of the theory itself
    np.testing.assert_allclose(fft_draw.vonkarman_structure_function(np.array([0.01]), r0, 1.0e5)[0],
                               6.8839 * (0.01 / r0) ** (5.0 / 3.0), rtol=0.03)


def test_second_kick_table_is_a_proper_cdf():
    r2, cdf = atm_psf.second_kick_table(622.2, 0.17, 8.36, 0.61, 0.2)
    assert cdf[0] == 0.0 and cdf[-1] == 1.0 and np.all(np.diff(cdf) >= 0) and np.all(np.diff(r2) > 0)
    hlr = np.sqrt(np.interp(0.5, cdf, r2))
    assert 0.1 < hlr < 0.6          # arcsec: a sizeable part of 0.8" seeing lives above kcrit = 0.2/r0
    # a larger kcrit leaves less turbulence for the second kick
    r2b, cdfb = atm_psf.second_kick_table(622.2, 0.17, 8.36, 0.61, 1.0)
    assert np.sqrt(np.interp(0.5, cdfb, r2b)) < hlr


def test_r0_500_inverts_the_tokovinin_formula():
    """tests/test_psf.py:228-246 of the reference: for ten random (airmass, rawSeeing, band) the r0_500 the constructor
    solves for reproduces the target FWHM through the von Karman seeing formula to 1e-3 arcsec."""
    rng = np.random.default_rng(57721)
    for _ in range(10):
        airmass = rng.uniform(1.001, 1.5)
        raw_seeing = rng.uniform(0.5, 1.5)
        band = "ugrizy"[rng.integers(6)]
        atm = atm_psf.AtmosphericPSF(airmass, raw_seeing, band, seed=int(rng.integers(2 ** 31)), screen_size=6.4)
        wlen = dict(u=365.49, g=480.03, r=622.20, i=754.06, z=868.21, y=991.66)[band]
        target = raw_seeing * airmass ** 0.6 * (wlen / 500.0) ** (-0.3)
        np.testing.assert_allclose(atm.targetFWHM, target, rtol=1e-12)
        assert 10.0 <= atm.L0 <= 100.0
        np.testing.assert_allclose(atm_psf.vk_seeing(atm.r0_500, wlen, atm.L0), target, atol=1e-3, rtol=0)
    # the truncated log-normal outer scale stays inside [10, 100] m (atmPSF.py:249-252)
    for L0 in (10.0, 25.0, 100.0):
        r = atm_psf.r0_500_for_seeing(622.2, L0, 0.8)
        np.testing.assert_allclose(atm_psf.vk_seeing(r, 622.2, L0), 0.8, atol=1e-6)
