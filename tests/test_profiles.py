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


def test_quintic_interpolant_tables():
    """The x-interpolant of galsim.InterpolatedImage defaults to Quintic.  The kernel restated in engine.quintic_kernel must
    interpolate (1 at 0, 0 at the other integers), be a partition of unity and reproduce polynomials up to degree four (what
    defines the Bernstein & Gruen quintic); its positive and negative flux are GalSim's documented 1.1293413... and
    0.1293413...; the sign-change abscissa beyond 2 is GalSim's (25 + sqrt 31) / 11; the sampling table has exact interval
    masses, so the signed weights average to 1 and carry no second moment."""
    from imsim_amd.engine import quintic_kernel as K, interpolant_cdf, QUINTIC_NEGATIVE as Q
    assert K([0.0])[0] == 1.0 and np.all(K(np.array([1.0, 2.0, 3.0, -1.0, -2.0, 3.5])) == 0.0)
    x = np.linspace(0.0, 1.0, 501)
    for p in range(5):
        tot = sum((float(k) ** p) * K(x - k) for k in range(-3, 5))
        np.testing.assert_allclose(tot, x ** p, atol=3e-14)
    assert abs(K([Q[2]])[0]) < 1e-15 and K([Q[2] - 1e-3])[0] > 0 > K([Q[2] + 1e-3])[0] and K([1.5])[0] < 0
    kx, kcdf, norm = interpolant_cdf()
    assert np.all(np.diff(kx) > 0) and np.all(np.diff(kcdf) > 0) and kcdf[0] == 0.0 and kcdf[-1] == 1.0
    for q in Q + (0.0,):
        assert np.any(kx == q) and np.any(kx == -q)          # no interval holds both signs
    pos, neg = 0.5 * (np.sqrt(norm) + 1.0), 0.5 * (np.sqrt(norm) - 1.0)
    assert abs(pos - 1.1293413499280066) < 1e-12 and abs(neg - 0.1293413499280066) < 1e-12
    u = (np.arange(4000000) + 0.5) / 4000000
    i = np.clip(np.searchsorted(kcdf, u, side="right") - 1, 0, len(kx) - 2)
    d = kx[i] + (u - kcdf[i]) / (kcdf[i + 1] - kcdf[i]) * (kx[i + 1] - kx[i])
    ad = np.abs(d)
    w = np.where(((ad > Q[0]) & (ad < Q[1])) | ((ad > Q[2]) & (ad < Q[3])), -1.0, 1.0) * np.sqrt(norm)
    assert abs(w.mean() - 1.0) < 1e-6 and abs((w * d).mean()) < 1e-9 and abs((w * d * d).mean()) < 5e-5


def test_fits_image_profile_with_the_quintic_interpolant():
    """galsim.InterpolatedImage shoots a pixel by its flux and then convolves with the x-interpolant (Quintic by default):
    offsets drawn from |K| per axis, photon fluxes +- (integral |K|)^2 (SBInterpolatedImage::shoot, Interpolant::shoot).  So
    the signed flux per bin must follow the INTERPOLATED image  sum_q v_q K(x - q_x) K(y - q_y), negative lobes included,
    and the signed total is the object's flux."""
    from imsim_amd.engine import quintic_kernel as K, interpolant_cdf
    rng = np.random.default_rng(6)
    img = np.zeros((7, 8))
    img[2:5, 2:6] = rng.uniform(0.5, 3.0, size=(3, 4))
    img[3, 3] = 12.0                                      # a sharp pixel: its side lobes are negative
    scale, n_phot = 0.4, 3000000
    cat = _cat(catalog.KIND_IMAGE, 1, pa=np.zeros(1), image_scale=np.full(1, scale), image_index=np.zeros(1, dtype=np.int64),
               image_extent=np.full(1, 8 * scale))
    scene = configs.scene_c2(nx=256, ny=256)
    scene.psf = []
    scene.image_profiles = [img]
    assert scene.image_interpolant == "quintic"
    objects, _ = catalog.build_object_table(cat, np.full(1, n_phot), stamp_size=128)
    pool = orc_loader.OracleScene(scene).shoot_pool(objects).to_host()
    norm = interpolant_cdf()[2]
    assert set(np.unique(pool["flux"])) == {-norm, norm}
    assert abs(pool["flux"].sum() - n_phot) < 5 * norm * np.sqrt(n_phot)
    u = (pool["x"] - objects["x0"][0]) * PIX / scale + 4.0          # image pixel coordinates, pixel k covers [k, k + 1)
    v = (pool["y"] - objects["y0"][0]) * PIX / scale + 3.5
    assert u.min() >= -3.0 and u.max() <= 11.0 and v.min() >= -3.0 and v.max() <= 10.0
    sub = 2                                                          # bins of half an image pixel
    edges_u, edges_v = np.linspace(-3, 11, 14 * sub + 1), np.linspace(-3, 10, 13 * sub + 1)
    got, _, _ = np.histogram2d(v, u, bins=(edges_v, edges_u), weights=pool["flux"])
    cnt, _, _ = np.histogram2d(v, u, bins=(edges_v, edges_u))
    # expectation: the bin integral of the interpolated image (Gauss-Legendre inside every bin; K is a quintic per half pixel)
    gx, gw = np.polynomial.legendre.leggauss(6)

    def bin_weights(edges, n_pix):
        a, b = edges[:-1], edges[1:]
        xs = 0.5 * (a + b)[:, None] + 0.5 * (b - a)[:, None] * gx[None, :]           # [bin][node]
        centres = np.arange(n_pix) + 0.5
        return (0.5 * (b - a))[:, None] * (K((xs[:, :, None] - centres[None, None, :]).reshape(-1)).reshape(
            xs.shape + (n_pix,)) * gw[None, :, None]).sum(axis=1)                     # [bin][pixel]
    Wu, Wv = bin_weights(edges_u, 8), bin_weights(edges_v, 7)
    want = Wv @ img @ Wu.T / img.sum() * n_phot
    sigma = norm * np.sqrt(np.maximum(cnt, 25.0))
    assert np.all(np.abs(got - want) < 5.5 * sigma)
    assert want.min() < -20 * norm * np.sqrt(25.0) and got[want < -20 * norm * 5].max() < 0      # real negative lobes, seen


def test_fits_image_profile_follows_the_pixels(tmp_path):
    """FITS-stamp objects (galsim.InterpolatedImage, imsim/instcat.py:552-561) with the "nearest" interpolant: photons land in
    the image's pixels in proportion to their values, uniformly inside a pixel, on the image's own pixel scale, rotated by
    -theta."""
    from imsim_amd import fits_io
    rng = np.random.default_rng(5)
    img = np.zeros((9, 12))                               # [ny][nx]
    img[2:7, 3:9] = rng.uniform(0.5, 3.0, size=(5, 6))
    img[0, 0] = -4.0                                      # negative pixels are empty
    img[8, 11] = 2.0
    scale, theta = 0.35, 25.0
    n_obj, n_phot = 6, 60000
    cat = _cat(catalog.KIND_IMAGE, n_obj, pa=np.full(n_obj, theta), image_scale=np.full(n_obj, scale),
               image_index=np.zeros(n_obj, dtype=np.int64), image_extent=np.full(n_obj, 12 * scale))
    scene = configs.scene_c2(nx=256, ny=256)
    scene.psf = []
    scene.image_profiles = [img]
    scene.image_interpolant = "nearest"
    objects, sizes = catalog.build_object_table(cat, np.full(n_obj, n_phot), stamp_size=128)
    assert np.all(objects["prof_table"] == -4) and np.all(objects["prof_scale"] == scale)
    pool = orc_loader.OracleScene(scene).shoot_pool(objects).to_host()
    dx = (pool["x"].reshape(n_obj, n_phot) - objects["x0"][:, None]) * PIX
    dy = (pool["y"].reshape(n_obj, n_phot) - objects["y0"][:, None]) * PIX
    t = np.deg2rad(-theta)                                # undo obj.rotate(-theta)
    u = (np.cos(t) * dx + np.sin(t) * dy) / scale + 6.0   # image pixel coordinates, pixel k covers [k, k+1)
    v = (-np.sin(t) * dx + np.cos(t) * dy) / scale + 4.5
    assert u.min() >= 0 and u.max() <= 12 and v.min() >= 0 and v.max() <= 9
    hist, _, _ = np.histogram2d(v.ravel(), u.ravel(), bins=(9, 12), range=((0, 9), (0, 12)))
    want = np.clip(img, 0, None) / np.clip(img, 0, None).sum() * u.size
    assert hist[0, 0] == 0 and (hist[want == 0] == 0).all()
    assert np.all(np.abs(hist - want)[want > 0] < 5 * np.sqrt(want[want > 0]))
    fx, fy = u - np.floor(u), v - np.floor(v)             # uniform inside the pixel
    assert abs(fx.mean() - 0.5) < 0.005 and abs(fy.mean() - 0.5) < 0.005 and abs(fx.var() - 1 / 12) < 0.002
    # stamp size grows with the extent of the image on the sky
    _, s1 = catalog.build_object_table(cat, np.full(n_obj, n_phot))
    cat2 = dict(cat, image_scale=np.full(n_obj, 4 * scale), image_extent=np.full(n_obj, 48 * scale))
    _, s2 = catalog.build_object_table(cat2, np.full(n_obj, n_phot))
    assert np.all(s2 > s1) and np.all(s1 >= 2 * 12 * scale / PIX)
    # through the instance-catalog reader: the file is found next to the catalog, .gz or not
    import gzip
    fits_io.write_fits(str(tmp_path / "stamp.fits"), [({}, img.astype(np.float32))])
    (tmp_path / "stamp2.fits.gz").write_bytes(gzip.compress((tmp_path / "stamp.fits").read_bytes()))
    from imsim_amd import instcat
    lines = ["object 1 60.0 -38.0 20 sed.txt 0 0 0 0 0 0 stamp.fits 0.35 25.0 none none",
             "object 2 60.0 -38.0 21 sed.txt 0 0 0 0 0 0 stamp2.fits.gz 0.2 0.0 none none",
             "object 3 60.0 -38.0 21 sed.txt 0 0 0 0 0 0 missing.fits 0.2 0.0 none none",
             "object 4 60.0 -38.0 21 sed.txt 0 0 0 0 0 0 point none none"]
    (tmp_path / "cat.txt").write_text("\n".join(lines) + "\n")
    parsed = instcat.parse_objects(str(tmp_path / "cat.txt"))
    assert list(parsed["objtype"]) == [4, 4, 4, 0] and parsed["fits_file"][0].endswith("stamp.fits")
    wcs = configs.scene_c3(nx=256, ny=256, sensor=False).optics.img_wcs
    # put the objects on the CCD by pointing them at the WCS centre
    ra, dec = _wcs_centre(wcs)
    parsed["ra"][:], parsed["dec"][:] = ra, dec
    c = instcat.to_catalog(parsed, wcs, 256, 256, 100.0, 30.0, sort_mag=False)
    assert list(c["kind"]) == [catalog.KIND_IMAGE, catalog.KIND_IMAGE, 0] and c["n_dropped_unsupported"] == 1
    assert len(c["images"]) == 2 and np.allclose(c["images"][0], img) and list(c["image_index"][:2]) == [0, 1]
    assert list(c["image_scale"][:2]) == [0.35, 0.2] and c["image_extent"][0] == 12 * 0.35


def _wcs_centre(w):
    """(ra, dec) of the pixel (128, 128) of a test WCS"""
    from imsim_amd import wcs as wcsmod
    vec = wcsmod.tansip_pix_to_vec(w, np.array([128.0]), np.array([128.0]))
    vec = np.asarray(vec).reshape(3)
    return float(np.arctan2(vec[1], vec[0])), float(np.arcsin(vec[2] / np.linalg.norm(vec)))


def test_general_sersic_index_tables_and_shooting():
    """imsim/instcat.py:511-517: Sersic indices are quantised to 0.05, not collapsed to 1 / 4.  Every index gets its own
    radial table: half of the photons fall inside the half-light radius, and the concentration (r80 / r20) grows with n
    as the Sersic law says."""
    from scipy import special
    scene = configs.scene_c2(nx=256, ny=256)
    scene.psf = []
    ns = np.array([0.5, 1.0, 2.45, 2.5, 4.0, 6.0])
    index = configs.add_sersic_tables(scene, ns)
    assert index[1.0] == 0 and index[4.0] == 1 and scene.sersic_extra_n == (0.5, 2.45, 2.5, 6.0)
    assert configs.add_sersic_tables(scene, [2.5, 3.0])[3.0] == index[6.0] + 1            # idempotent, appends only new ones
    n_obj = len(ns)
    cat = _cat(2, n_obj, sersic_n=ns, hlr=np.full(n_obj, 0.5))
    cat["kind"] = np.where(ns == 1.0, 1, 2).astype(np.int32)
    objects, _ = catalog.build_object_table(cat, np.full(n_obj, 40000), stamp_size=128, sersic_index=scene.sersic_index)
    assert list(objects["prof_table"]) == [index[float(n)] for n in ns]
    orc = orc_loader.OracleScene(scene)
    pool = orc.shoot_pool(objects).to_host()
    r = np.hypot(pool["x"].reshape(n_obj, -1) - objects["x0"][:, None], pool["y"].reshape(n_obj, -1) - objects["y0"][:, None]) * PIX
    conc = []
    for k, n in enumerate(ns):
        assert abs(np.mean(r[k] < 0.5) - 0.5) < 0.012, n                                  # half-light radius
        b = special.gammaincinv(2 * n, 0.5)
        r20, r80 = np.quantile(r[k], [0.2, 0.8])
        want20, want80 = [0.5 * (special.gammaincinv(2 * n, f) / b) ** n for f in (0.2, 0.8)]
        np.testing.assert_allclose([r20, r80], [want20, want80], rtol=0.05)
        conc.append(r80 / r20)
    assert np.all(np.diff(conc) >= -0.02) and conc[-1] > 2 * conc[0]
    # a catalog index without a table is an error, not a silent n = 4
    import pytest
    with pytest.raises(ValueError):
        catalog.build_object_table(cat, np.full(n_obj, 100), stamp_size=64)
    with pytest.raises(ValueError):
        configs.add_sersic_tables(scene, [7.5])


def test_stamp_size_follows_the_sersic_index():
    hlr = np.full(4, 0.5)
    kind = np.array([1, 2, 2, 2])
    sizes = catalog.gal_stamp_size(kind, hlr, np.ones(4), sersic_n=np.array([1.0, 2.0, 4.0, 6.0]))
    assert np.all(np.diff(sizes) > 0)
    assert sizes[2] == catalog.gal_stamp_size(np.array([2]), hlr[:1], np.ones(1))[0]       # n = 4 is the default of kind 2


def test_kolmogorov_table_has_galsims_half_light_radius():
    """galsim.Kolmogorov (the `Kolmogorov` half of imsim/atmPSF.py:534 KolmogorovPSF): GalSim documents
    fwhm = 0.975865 lam / r0 and half_light_radius = 0.554811 lam / r0, i.e. hlr / fwhm = 0.568533.  The radial table the
    photon kernels sample (radius in units of the FWHM, built from the MTF exp(-k^(5/3)) by quadrature) reproduces the
    ratio to 1.1e-4, and its enclosed flux is monotonic up to the shoot accuracy."""
    from imsim_amd import tables
    r2, cdf = tables.kolmogorov_table()
    hlr = float(np.sqrt(np.interp(0.5, cdf, r2)))
    assert abs(hlr / (0.554811 / 0.975865) - 1.0) < 3.0e-4
    assert np.all(np.diff(cdf) >= 0.0) and cdf[0] == 0.0 and abs(cdf[-1] - 1.0) <= tables.SHOOT_ACCURACY
