"""get_stamp_size family (imsim/stamp_utils.py:9-354) restated in imsim_amd.catalog."""
import math

import numpy as np

from imsim_amd import catalog


def _jac(q=1.0, beta=0.0):
    return catalog.shear_matrix(np.array([q]), np.array([beta]))


def test_sersic_xvalue_is_normalised_and_transforms():
    # total flux by radial quadrature (circular), and by a grid sum for a sheared n=1 profile
    r = np.geomspace(1e-7, 400.0, 200001)
    for n, hlr in ((1.0, 0.6), (4.0, 0.3)):
        v = catalog.sersic_xvalue(n, np.array([hlr]), np.array([5.0e4]), _jac()[0], r, 0.0 * r)
        np.testing.assert_allclose(np.trapezoid(v * 2 * np.pi * r, r), 5.0e4, rtol=2e-3)
    xs = np.linspace(-12, 12, 1201)
    X, Y = np.meshgrid(xs, xs)
    v = catalog.sersic_xvalue(1.0, np.array([0.6]), np.array([5.0e4]), _jac(0.6, 30.0)[0], X, Y)
    np.testing.assert_allclose(v.sum() * (xs[1] - xs[0]) ** 2, 5.0e4, rtol=2e-2)
    # circular n=1: I(0) = flux b^2 / (2 pi hlr^2), b = 1.6783
    c = catalog.sersic_xvalue(1.0, np.array([0.5]), np.array([1.0]), _jac()[0], 0.0, 0.0)
    np.testing.assert_allclose(c, 1.6783469900166605 ** 2 / (2 * np.pi * 0.25), rtol=1e-9)


def test_bright_galaxies_grow_to_the_surface_brightness_limit():
    kind = np.array([1, 1, 2, 2])
    hlr = np.array([0.4, 1.0, 0.4, 1.0])
    jac = np.tile(_jac()[0], (4, 1))
    ms = np.ones(4)
    base = catalog.gal_stamp_size(kind, hlr, ms)
    faint = catalog.gal_stamp_size(kind, hlr, ms, jac=jac, nominal_flux=np.full(4, 500.0))
    assert np.array_equal(base, faint)                       # below 10 photons per stamp pixel: GoodImageSize only
    prev = faint
    for flux in (1.0e5, 1.0e6, 1.0e7, 1.0e8):
        cur = catalog.gal_stamp_size(kind, hlr, ms, jac=jac, nominal_flux=np.full(4, flux))
        assert np.all(cur >= prev) and np.all(cur <= catalog.NMAX)
        prev = cur
    assert np.any(prev > 2 * base)
    # the edge of the grown object stamp is below the keep level (sqrt(noise_var)/8), one step smaller is not
    keep = math.sqrt(800.0) / 8.0
    size = catalog.gal_stamp_size(kind[:1], hlr[:1], ms[:1], jac=jac[:1], nominal_flux=np.array([1.0e7]))[0]
    gal = math.sqrt(size ** 2 - 16 ** 2)                     # quadrature with the proxy PSF's 16 pixels
    edge = catalog.sersic_xvalue(1.0, hlr[:1], np.array([1.0e7]), jac[0], gal / 2 * 0.2, 0.0)[0]
    assert edge < keep * 1.5
    inner = catalog.sersic_xvalue(1.0, hlr[:1], np.array([1.0e7]), jac[0], gal / 2 / 1.21 * 0.2, 0.0)[0]
    assert inner > keep


def test_huge_stamps_fall_back_to_three_times_the_limit_and_nmax():
    kind, hlr = np.array([2]), np.array([8.0])
    jac = np.tile(_jac()[0], (1, 1))
    s = catalog.gal_stamp_size(kind, hlr, np.ones(1), jac=jac, nominal_flux=np.array([1.0e10]))
    assert s[0] == catalog.NMAX


def test_star_stamp_sizes_reproduce_the_reference_regression_values():
    """tests/test_stamp.py:264-312 of the reference builds stamps for stars of visit 449053 (tests/data/small_opsim_9683.db:
    airmass 1.12386549163403, seeingFwhm500 0.663846738172267, r band) and pins their sizes: 40 x 40 for the stars of
    2 443 and 28 124 photons, 106 x 106 for 292 627 photons, the full 4096 for 5.3e8 photons.  The sky variance of that
    visit comes from the Rubin sky model (not available here); for every plausible value (450..700 e-/pixel at sky
    brightness 21.09 mag/arcsec^2) the restated get_star_stamp_size gives exactly those sizes."""
    airmass, raw_seeing = 1.12386549163403, 0.663846738172267
    flux = np.array([2443.0, 28124.0, 292627.0, 531711520.0])
    for noise_var in (450.0, 500.0, 550.0, 600.0, 650.0, 700.0):
        sizes = catalog.star_stamp_size(flux, noise_var, airmass, raw_seeing, "r")
        assert list(sizes) == [40, 40, 106, 4096], (noise_var, sizes)
    # the folding threshold is rounded down to whole e-folds (stamp_utils.py:137-139): the size is a step function of flux
    steps = catalog.star_stamp_size(np.geomspace(1e5, 1e7, 60), 550.0, airmass, raw_seeing, "r")
    assert len(np.unique(steps)) <= 6 and np.all(np.diff(steps) >= 0)
