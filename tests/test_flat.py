"""LSST_Flat (imsim/flat.py, area branch) on the CPU oracle: the reference's RNG-independent physics
criteria for the Silicon model (tests/test_flats.py:21-60, :62-111, :113-165) pin the oracle's
brighter-fatter and tree-ring implementation; tests/test_parity_gpu.py then ties the GPU to the
oracle bit for bit."""
import numpy as np
import pytest

from imsim_amd import configs, flat, treerings
from oracle import orc_loader


def _stats(img):
    """mean, variance and the neighbour covariances of tests/test_flats.py:50-53 (cov11 averaged over both diagonals)"""
    a = img - img.mean()
    cov10 = np.mean(a[1:, :] * a[:-1, :])
    cov01 = np.mean(a[:, 1:] * a[:, :-1])
    cov11 = 0.5 * (np.mean(a[1:, 1:] * a[:-1, :-1]) + np.mean(a[1:, :-1] * a[:-1, 1:]))
    return img.mean(), img.var(), cov10, cov01, cov11


def test_builder_parameter_surface():
    b = flat.LSST_FlatBuilder()
    with pytest.raises(ValueError):
        b.setup({"xsize": 16, "ysize": 16})                      # counts_per_pixel is required (flat.py:45)
    assert b.setup({"counts_per_pixel": 80000, "max_counts_per_iter": 4000, "xsize": 256, "ysize": 200}) == (256, 200)
    assert b.iterations() == (20, 4000.0)
    b.setup({"counts_per_pixel": 1050, "size": 8})
    assert (b.max_counts_per_iter, b.buffer_size, b.nx, b.ny) == (1000.0, 5, 8, 2)      # defaults, flat.py:52-58
    niter, per = b.iterations()
    assert niter == 2 and per == 525.0


def test_simple_flat_is_poisson():
    """tests/test_flats.py:21-60: no sensor -> mean = var = N within 1 %, negligible covariances"""
    tot = 100_000.0
    o = orc_loader.OracleScene(configs.scene_flat(256, 256, sensor=False))
    img = o.build_flat(tot, 10_000.0, seed=1234)
    mean, var, c10, c01, c11 = _stats(img)
    np.testing.assert_allclose(mean, tot, rtol=1e-2)
    np.testing.assert_allclose(var, tot, rtol=1e-2)
    assert abs(c10) < 1e-2 * tot and abs(c01) < 1e-2 * tot and abs(c11) < 1e-2 * tot


def test_silicon_flat_has_brighter_fatter_covariances():
    """tests/test_flats.py:62-111: 80 000 e-/px in 20 iterations, Silicon sensor, no tree rings.  The reference
    uses 256^2 pixels, where the sampling noise of a covariance (N / 256 = 312) is as large as its cov01 and
    cov11 thresholds; 768^2 brings it down to 104 with the same thresholds."""
    tot = 80_000.0
    o = orc_loader.OracleScene(configs.scene_flat(768, 768, sensor=True))
    img = o.build_flat(tot, 4_000.0, seed=1234)
    mean, var, c10, c01, c11 = _stats(img)
    np.testing.assert_allclose(mean, tot, rtol=1e-2)
    np.testing.assert_allclose(var, tot, rtol=1e-1)
    assert var < tot
    assert c10 > 1e-2 * tot and c01 > 3e-3 * tot and c11 > 2e-3 * tot
    assert c10 > c01 > c11


def test_treerings_flat_variance_follows_the_ring_amplitude():
    """tests/test_flats.py:113-165: f(r) = 0.26 cos(2 pi r / 87) about (-100, -100), 1e5 e-/px in 10 iterations"""
    tot, amp, period = 100_000.0, 0.26, 87.0
    tr = treerings.simple_treerings(amp, period, dr=period / 100.0)
    o = orc_loader.OracleScene(configs.scene_flat(256, 256, sensor=True, treering=tr, treering_center=(-100.0, -100.0)))
    img = o.build_flat(tot, 10_000.0, seed=1234)
    mean, var, c10, c01, c11 = _stats(img)
    pred_var = 0.5 * (tot * amp * 2 * np.pi / period) ** 2 + tot
    np.testing.assert_allclose(mean, tot, rtol=1e-2)
    np.testing.assert_allclose(var, pred_var, rtol=3e-2)
    assert c10 > 0.5 * tot and c01 > 0.5 * tot and c11 > 0.5 * tot


def test_photon_flat_sed_branch():
    """flat.py:237-262: uniform photons with sampled wavelengths through the sensor, one boundary update per
    iteration.  (Small here; tests/test_parity_gpu.py runs the reference-sized flat on the GPU.)"""
    tot = 3_000.0
    scene = configs.scene_flat(64, 64, sensor=True, buffer_size=6)
    scene.track_static_delta = 1
    o = orc_loader.OracleScene(scene)
    b = flat.LSST_FlatBuilder()
    b.setup({"counts_per_pixel": tot, "max_counts_per_iter": 1_000, "xsize": 64, "ysize": 64, "buffer_size": 6})
    assert b.iterations() == (3, 1000.0)
    img = np.array(b.build_image_photons(o, seed=5), dtype=np.float64)
    assert img.shape == (64, 64)
    # photons that diffuse across the edge of the working image are lost there, the buffer keeps the level inside
    np.testing.assert_allclose(img.mean(), tot, rtol=1e-2)
    np.testing.assert_allclose(img.var(), tot, rtol=0.10)
    assert np.all(o.sensor_array("delta") >= 0)
    # no sensor: plain Poisson image
    s2 = configs.scene_flat(64, 64, sensor=False, buffer_size=0)
    o2 = orc_loader.OracleScene(s2)
    b.setup({"counts_per_pixel": 1500.0, "max_counts_per_iter": 1000, "xsize": 64, "ysize": 64, "buffer_size": 0})
    img2 = np.array(b.build_image_photons(o2, seed=6), dtype=np.float64)
    np.testing.assert_allclose(img2.mean(), 1500.0, rtol=1e-2)
    np.testing.assert_allclose(img2.var(), 1500.0, rtol=0.08)
