"""Sky-background inputs (imsim/sky_model.py): fringing against the reference's known answers, the planar gradient."""
import math

import numpy as np
import pytest

from imsim_amd import sky_model


def test_fringing_variation_level_known_answers():
    """tests/test_fringing.py:113-137 of the reference: OH sky-line level for six offsets from the boresight"""
    for ra, dec, level in [(0, 0.1, 1.056503042318907), (0, 0.2, 1.1207294877266138), (0.2, -0.1, 1.0044602251026102),
                           (1.1, 0.2, 1.0166040509448886), (-1.2, 0.5, 1.0389039410245318), (1.2, -0.4, 1.0204232685215646)]:
        f = sky_model.CCD_Fringing(true_center=(math.radians(ra), math.radians(dec)), boresight=(0.0, 0.0), seed=0, spatial_vary=True)
        np.testing.assert_allclose(f.fringe_variation_level(), level, atol=1e-10, rtol=1e-10)


def test_fringing_map_statistics():
    """tests/test_fringing.py:44-59, :91-110: zero amplitude is an error; the map swings by +- amplitude x level around 1, its
    rms along the diagonal is amplitude x level / sqrt 2 (0.0014 to two digits at the reference's offset); the same seed
    gives the same map; without spatial variation the level is 1."""
    cra, cdec, ra, dec = 54.9348753510528, -35.8385705255579, 54.86, -35.76
    f = sky_model.CCD_Fringing(true_center=(math.radians(ra), math.radians(dec)), boresight=(math.radians(cra), math.radians(cdec)),
                               seed=sky_model.sensor_seed("E2V-CCD250-382"), spatial_vary=True)
    xarr, yarr = np.meshgrid(range(1024), range(1000))
    with pytest.raises(ValueError):
        f.calculate_fringe_amplitude(xarr, yarr, amplitude=0, n_side=1024)
    m = f.calculate_fringe_amplitude(xarr, yarr, n_side=1024)
    level = f.fringe_variation_level()
    assert m.shape == (1000, 1024) and 1.0 < level < 1.1
    np.testing.assert_approx_equal(m.max(), 1 + 0.002 * level, significant=4)
    np.testing.assert_approx_equal(m.min(), 1 - 0.002 * level, significant=4)
    np.testing.assert_allclose(np.std(np.diag(m)), 0.002 * level / math.sqrt(2), rtol=0.08)
    m2 = f.calculate_fringe_amplitude(xarr, yarr, n_side=1024)
    assert np.array_equal(m, m2)
    g = sky_model.CCD_Fringing(true_center=(1.0, 0.2), boresight=(0.9, 0.1), seed=5, spatial_vary=False)
    assert g.fringe_variation_level() == 1


def test_sky_gradient_is_the_plane_through_three_levels():
    sky = lambda ra, dec: 1000.0 + 3.0 * ra - 2.0 * dec            # noqa: E731
    pix_to_world = lambda x, y: (0.01 * x + 5.0, 0.02 * y - 3.0)    # noqa: E731
    g = sky_model.SkyGradient(sky, pix_to_world, (2048.0, 2002.0), 4096)
    for (x, y) in ((0.0, 0.0), (4096.0, 0.0), (2048.0, 2002.0), (100.0, 3000.0)):
        want = sky(*pix_to_world(x, y)) / sky(*pix_to_world(2048.0, 2002.0))
        np.testing.assert_allclose(g(x, y), want, rtol=1e-12)
    a, b, c = g.coefficients()
    np.testing.assert_allclose(a + b * 100.0 + c * 3000.0, g(100.0, 3000.0), rtol=1e-13)
