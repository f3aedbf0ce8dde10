"""Object profiles beyond point / Sersic (imsim/instcat.py:487-546): streaks (galsim.Box) and
galsim.RandomKnots, on the CPU oracle (tests/test_parity_gpu.py ties the GPU to it bit for bit)."""
import numpy as np

from imsim_amd import catalog, configs
from oracle import orc_loader

PIX = 0.2


def _cat(kind, n_obj, **kw):
    rng = np.random.default_rng(3)
    c = dict(x=rng.uniform(60, 196, n_obj), y=rng.uniform(60, 196, n_obj), mag=np.zeros(n_obj),
             nominal_flux=np.full(n_obj, 4000.0), kind=np.full(n_obj, kind, dtype=np.int32), hlr=np.full(n_obj, 0.5),
             q=np.ones(n_obj), pa=np.zeros(n_obj), obj_id=np.arange(n_obj, dtype=np.int64) + 10,
             n_knots=np.zeros(n_obj), box_length=np.zeros(n_obj), box_width=np.zeros(n_obj))
    c.update(kw)
    return c


def _shoot(cat, n_phot=4000):
    scene = configs.scene_c2(nx=256, ny=256)
    scene.psf = []                                     # bare profile
    objects, _ = catalog.build_object_table(cat, np.full(len(cat["x"]), n_phot), stamp_size=128)
    orc = orc_loader.OracleScene(scene)
    pool = orc.shoot_pool(objects).to_host()
    dx = (pool["x"].reshape(len(objects), n_phot) - objects["x0"][:, None]) * PIX     # arcsec
    dy = (pool["y"].reshape(len(objects), n_phot) - objects["y0"][:, None]) * PIX
    return objects, dx, dy


def test_random_knots_are_a_fixed_set_of_gaussian_points():
    n_obj, n_knots, hlr = 300, 7, 0.5
    cat = _cat(catalog.KIND_KNOTS, n_obj, n_knots=np.full(n_obj, float(n_knots)))
    objects, dx, dy = _shoot(cat)
    assert np.all(objects["prof_table"] == -3) and np.all(objects["prof_aux"] == n_knots)
    radii = []
    for i in range(n_obj):
        pts, counts = np.unique(np.stack([dx[i], dy[i]], axis=1), axis=0, return_counts=True)
        assert len(pts) == n_knots                                  # every photon sits on one of the knots
        assert counts.min() > 4000 / n_knots * 0.7                  # equal flux per knot (multinomial)
        radii.append(np.hypot(pts[:, 0], pts[:, 1]))
    radii = np.concatenate(radii)
    # the knots follow the parent Gaussian: median radius = half-light radius
    np.testing.assert_allclose(np.median(radii), hlr, rtol=0.06)
    np.testing.assert_allclose(np.mean(radii ** 2), 2 * (hlr / 1.1774100225154747) ** 2, rtol=0.08)
    # different objects have different knots
    assert not np.allclose(np.sort(dx[0])[:50], np.sort(dx[1])[:50])


def test_knots_are_sheared_like_the_parent_profile():
    n_obj = 400
    cat = _cat(catalog.KIND_KNOTS, n_obj, n_knots=np.full(n_obj, 20.0), q=np.full(n_obj, 0.4), pa=np.full(n_obj, 90.0))
    _, dx, dy = _shoot(cat, n_phot=1000)
    # beta = 90 - pa = 0: major axis along x; area-preserving shear: sigma_x / sigma_y = 1/q
    np.testing.assert_allclose(np.std(dx) / np.std(dy), 1 / 0.4, rtol=0.08)


def test_streak_is_a_uniform_rotated_box():
    n_obj, L, W, pa = 20, 12.0, 0.8, 30.0
    cat = _cat(catalog.KIND_STREAK, n_obj, box_length=np.full(n_obj, L), box_width=np.full(n_obj, W), pa=np.full(n_obj, pa))
    objects, dx, dy = _shoot(cat, n_phot=20000)
    assert np.all(objects["prof_table"] == -2)
    t = np.deg2rad(pa)
    u = np.cos(t) * dx + np.sin(t) * dy                   # along the streak
    v = -np.sin(t) * dx + np.cos(t) * dy
    assert np.abs(u).max() <= L / 2 and np.abs(v).max() <= W / 2
    assert np.abs(u).max() > 0.499 * L and np.abs(v).max() > 0.499 * W
    np.testing.assert_allclose(np.var(u), L * L / 12, rtol=0.01)
    np.testing.assert_allclose(np.var(v), W * W / 12, rtol=0.01)
    assert abs(np.mean(u)) < 0.02 and abs(np.corrcoef(u.ravel(), v.ravel())[0, 1]) < 0.01


def test_stamp_sizes_of_knots_and_streaks():
    cat = _cat(catalog.KIND_KNOTS, 3, n_knots=np.full(3, 5.0), hlr=np.array([0.2, 0.5, 1.5]))
    _, sizes = catalog.build_object_table(cat, np.full(3, 1000))
    assert np.all(np.diff(sizes) > 0) and sizes[0] >= 16
    cat = _cat(catalog.KIND_STREAK, 2, box_length=np.array([10.0, 40.0]), box_width=np.array([1.0, 1.0]))
    _, sizes = catalog.build_object_table(cat, np.full(2, 1000))
    assert sizes[0] >= 2 * 10.0 / PIX and sizes[1] >= 2 * 40.0 / PIX      # GoodImageSize = 2 pi / (stepk scale), stepk = pi / L


def test_double_gaussian_psf_is_a_two_component_mixture():
    """BuildDoubleGaussianPSF (imsim/atmPSF.py:448-486) as a photon operator"""
    from imsim_amd import config
    comp, tab, p0 = config.double_gaussian_psf(0.7)
    alpha = 0.7 / 2.3835
    s1, s2 = np.sqrt(alpha ** 2 - 0.04 / 12), np.sqrt(4 * alpha ** 2 - 0.04 / 12)
    assert comp[2] == s1 and comp[5] == s2 and abs(comp[6] - 1 / 1.1) < 1e-15
    cat = _cat(0, 1)                                        # one point source
    scene = configs.scene_c2(nx=256, ny=256)
    scene.psf = [comp]
    objects, _ = catalog.build_object_table(cat, np.array([400000]), stamp_size=128)
    pool = orc_loader.OracleScene(scene).shoot_pool(objects).to_host()
    dx, dy = (pool["x"] - objects["x0"][0]) * PIX, (pool["y"] - objects["y0"][0]) * PIX
    f = 1 / 1.1
    np.testing.assert_allclose(np.var(dx), f * s1 ** 2 + (1 - f) * s2 ** 2, rtol=0.02)
    np.testing.assert_allclose(np.var(dy), f * s1 ** 2 + (1 - f) * s2 ** 2, rtol=0.02)
    m4 = np.mean(dx ** 4)
    np.testing.assert_allclose(m4, 3 * (f * s1 ** 4 + (1 - f) * s2 ** 4), rtol=0.05)        # heavier tails than one Gaussian
    # its k-table: value 1 at k = 0, the mixture of the two Gaussian transforms elsewhere
    from imsim_amd import tables
    q = np.linspace(0.0, tables.KTABLE_QMAX, tables.KTABLE_NPTS)
    k = q / p0
    np.testing.assert_allclose(tab, f * np.exp(-0.5 * (k * s1) ** 2) + (1 - f) * np.exp(-0.5 * (k * s2) ** 2))
    assert tab[0] == 1.0 and tab[-1] < 1e-12
