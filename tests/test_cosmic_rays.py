"""Cosmic rays (imsim/cosmic_rays.py), the reference's own unit tests (tests/test_cosmic_rays.py) restated, plus imSim's
catalog file read through the FITS binary-table reader."""
import numpy as np

from imsim_amd.cosmic_rays import CosmicRays, write_cosmic_ray_catalog

TEST_IMAGE = np.array([[0, 10, 0], [20, 30, 20], [0, 40, 0]])


def _catalog(tmp_path):
    f = str(tmp_path / "tmp_cr_catalog.fits")
    write_cosmic_ray_catalog((0, 0, 0, 1, 2), (10, 10, 10, 0, 5), (20, 21, 22, 100, 4000), [x for x in TEST_IMAGE] + [[100], [50]], 1.0, 100,
                             outfile=f)
    return f


def test_read_catalog(tmp_path):
    crs = CosmicRays.read_catalog(_catalog(tmp_path), ccd_rate=None)
    assert len(crs) == 3 and len(crs[0]) == 3
    assert crs[0][0].x0 == 10 and crs[0][0].y0 == 20 and tuple(crs[0][0].pixel_values) == (0, 10, 0)
    assert crs.ccd_rate == 3.0 and crs.num_pix == 100


def test_paint_cr(tmp_path):
    crs = CosmicRays.read_catalog(_catalog(tmp_path), ccd_rate=None)
    imarr = crs.paint_cr(np.zeros((3, 3)), rng=np.random.default_rng(1234), index=0, pixel=(0, 0))
    np.testing.assert_array_equal(TEST_IMAGE, imarr)
    # a hit hanging over the edge is clipped, not an error (the reference swallows the IndexError)
    clipped = crs.paint_cr(np.zeros((3, 3)), rng=np.random.default_rng(1), index=0, pixel=(2, 2))
    assert clipped[2, 2] == 0 and clipped.sum() == 0 + 0     # span row 0 starts with a 0 at its first pixel
    clipped = crs.paint_cr(np.zeros((3, 3)), rng=np.random.default_rng(1), index=0, pixel=(1, 1))
    np.testing.assert_array_equal(clipped, [[0, 0, 0], [0, 0, 10], [0, 20, 30]])


def test_cr_rng_seed(tmp_path):
    f = _catalog(tmp_path)
    im1 = CosmicRays.read_catalog(f).paint(np.zeros((100, 100)), rng=np.random.default_rng(1234), num_crs=10)
    im2 = CosmicRays.read_catalog(f).paint(np.zeros((100, 100)), rng=np.random.default_rng(1234), num_crs=10)
    im3 = CosmicRays.read_catalog(f).paint(np.zeros((100, 100)), rng=np.random.default_rng(1235), num_crs=10)
    np.testing.assert_array_equal(im1, im2)
    assert not np.array_equal(im1, im3) and im1.sum() > 0


def test_imsims_catalog_and_rate():
    """data/cosmic_rays_itl_2017.fits.gz: thousands of hits from ITL darks; a 30 s exposure of a 4k x 4k CCD gets
    exptime x ccd_rate hits on average"""
    crs = CosmicRays()
    assert len(crs) > 1000 and crs.num_pix > 1.5e7 and crs.exptime > 0
    assert all(len(span.pixel_values) > 0 for cr in crs[:50] for span in cr)
    rng = np.random.default_rng(5)
    n = [len(set((crs.draw_hits((4000, 4072), rng, exptime=30.0)[0] // 4072).tolist())) for _ in range(3)]
    expected = 30.0 * crs.ccd_rate * (4000 * 4072) / crs.num_pix
    img = crs.paint(np.zeros((4000, 4072)), rng=np.random.default_rng(6), exptime=30.0)
    assert img.sum() > 0 and 0.2 * expected < np.count_nonzero(img) / 10.0 < 50 * expected and min(n) > 0
