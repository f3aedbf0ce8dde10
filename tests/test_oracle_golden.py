"""Pin the oracle: golden vectors generated from the reference's importable modules
(tests/golden/make_diffraction_golden.py) and the known-answer values the reference's own tests
hold for this path."""
import ctypes as C
import os

import numpy as np
import pytest

from imsim_amd import _abi, diffraction, optics, sensor as sensormod, treerings
from imsim_amd.engine import Scene, SensorSetup, make_slots
from imsim_amd._abi import OBJECT_DTYPE
from oracle import orc_loader

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def vp(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(HERE, "golden", "diffraction_golden.npz"))


def test_directed_dist_and_phi_star_match_reference(golden):
    """imsim/diffraction.py:182-224 (directed_dist, phi_star) on 4000 pupil positions, including
    points exactly on the spider elements."""
    L = orc_loader.load()
    g = golden
    n = len(g["t"])
    o = _abi.Optics()
    diffraction.fill_optics(o, *g["visit_lat_az_alt"][[0, 1, 2]])
    pos = np.ascontiguousarray(g["pos"])
    dist, nrm = np.empty(n), np.empty((n, 2))
    L.orc_test_directed_dist(C.byref(o), vp(pos), vp(dist), vp(nrm), C.c_int64(n))
    np.testing.assert_allclose(dist, g["dist"], rtol=0, atol=5e-16)
    np.testing.assert_allclose(nrm, g["normal"], rtol=0, atol=5e-16)
    ps = np.empty(n)
    wl, d = np.ascontiguousarray(g["wavelength"]), np.ascontiguousarray(g["dist"])
    L.orc_test_phi_star(vp(d), vp(wl), vp(ps), C.c_int64(n))
    np.testing.assert_allclose(ps, g["phi_star"], rtol=2e-15)


@pytest.mark.parametrize("tag,tol", [("visit", 5e-15), ("zenith", 1e-12)])
def test_field_rotation_and_kick_match_reference(golden, tag, tol):
    """field_rotation_matrix + apply_diffraction_delta[_field_rot] + apply_delta_v
    (imsim/diffraction.py:45-179, :318-384).  The near-zenith pointing (alt 89.9 deg, the one the
    reference's tests use) is ill-conditioned, hence its looser tolerance on the rotation."""
    L = orc_loader.load()
    g = golden
    n = len(g["t"])
    la, az, al = g[f"{tag}_lat_az_alt"]
    o = _abi.Optics()
    diffraction.fill_optics(o, la, az, al)
    np.testing.assert_allclose(np.array(o.e_focal), g[f"{tag}_e_focal"], atol=1e-16)
    np.testing.assert_allclose(np.array(o.e_z0), g[f"{tag}_e_z0"], atol=1e-16)
    t = np.ascontiguousarray(g["t"])
    cs = np.empty((n, 2))
    L.orc_test_field_rotation(C.byref(o), vp(t), vp(cs), C.c_int64(n))
    R = g[f"{tag}_R"]
    np.testing.assert_allclose(cs[:, 0], R[:, 0, 0], atol=tol)
    np.testing.assert_allclose(cs[:, 1], R[:, 0, 1], atol=50 * tol)
    np.testing.assert_allclose(-cs[:, 1], R[:, 1, 0], atol=50 * tol)
    pos, wl, gs = (np.ascontiguousarray(g[k]) for k in ("pos", "wavelength", "gauss"))
    for field_rot, key in ((1, "v_rot"), (0, "v_norot")):
        v = np.ascontiguousarray(g["v"]).copy()
        L.orc_test_diffract(C.byref(o), field_rot, vp(pos), vp(t), vp(wl), vp(gs), vp(v), C.c_int64(n))
        ref = g[f"{tag}_{key}"]
        # spec v6 keeps the kicked DIRECTION and leaves the length alone (the reference rescales to the old length,
        # diffraction.py:45-66; nothing downstream depends on it): compare at the reference's length
        v = v * (np.linalg.norm(ref, axis=1) / np.linalg.norm(v, axis=1))[:, None]
        err = np.abs(v - ref).max(axis=1)
        # rotation error times the kick size; kicks are only large exactly on a spider edge
        assert np.median(err) < 1e-15
        assert err.max() < (1e-15 if (tag == "visit" or not field_rot) else 1e-6)


def test_tree_ring_known_answers():
    """tests/test_tree_rings.py:18-38 of the reference: centres and func(5280) for two detectors."""
    tr = treerings.TreeRings(os.path.join(HERE, "golden", "tree_ring_parameters_19mar18_subset.txt"), defer_load=False)
    expect = {"R22_S11": ((-3026.3, -3001.0), 0.0030205), "R34_S22": ((3095.5, -2971.3), -0.0034135)}
    for det, (center, val) in expect.items():
        c = tr.get_center(det)
        assert abs(c[0] - 2048.5 - center[0]) < 0.05 and abs(c[1] - 2048.5 - center[1]) < 0.05
        assert abs(float(tr.get_func(det)(5280.0)) - val) < 5e-7
    with pytest.raises(OSError):
        treerings.TreeRings("invalid.txt")


def test_ray_vector_to_photon_array_known_answer():
    """tests/test_photon_ops.py:668-691 of the reference (R22_S11 focal-plane -> pixel affine)."""
    L = orc_loader.load()
    o = _abi.Optics()
    tel = optics.Telescope([optics.Surface(_abi.IMS_SURF_DETECTOR, 0.0)])
    optics.fill_optics(o, tel, (100.0, 0.0, 2047.5, 0.0, 100.0, 2001.5), 0.0)
    pos = np.array([[1.0, -1.0, 0.0], [2.0, 3.0, 0.0]])
    vel = np.array([[0.0, 0.0, -1.0], [0.25, 0.5, -1.0]])
    out = np.empty((2, 4))
    L.orc_test_ray_to_photon(C.byref(o), vp(pos), vp(vel), vp(out), C.c_int64(2))
    np.testing.assert_array_almost_equal(out[:, 0], [-97952.5, 302047.5])
    np.testing.assert_array_almost_equal(out[:, 1], [102001.5, 202001.5])
    np.testing.assert_array_almost_equal(out[:, 2], [0.0, -0.5])
    np.testing.assert_array_almost_equal(out[:, 3], [0.0, -0.25])


def _moments(img):
    ny, nx = img.shape
    yy, xx = np.mgrid[0:ny, 0:nx].astype(float)
    f = img.sum()
    mx, my = (img * xx).sum() / f, (img * yy).sum() / f
    return (img * (xx - mx) ** 2).sum() / f, (img * (yy - my) ** 2).sum() / f


def _sensor_spot(model_name, n=1000000, seed=1234):
    """The reference's sensor-model test case (tests/test_sensor_models.py:42-59): Gaussian
    sigma 0.3" of 1e6 photons on a 17x17 image at 0.3"/pixel, achromatic."""
    N = 17
    sc = Scene(nx=N, ny=N, seed=seed, psf=[(_abi.IMS_PSF_GAUSSIAN, 0, 0.3, 0.0, 1.0)], ops=[])
    if model_name:
        model = sensormod.load_silicon_model(os.path.join(ROOT, "imsim_amd", "data", "sensor_models", model_name))
        # photons without wavelengths convert 1 micron below the surface in GalSim; a constant
        # 1 micron absorption length reproduces that depth on average
        sc.sensor = SensorSetup(model=model, abs_wl=np.array([300.0, 1100.0]), abs_len=np.array([1.0, 1.0]),
                                slots=make_slots([(1, 1, N, N), (1, 1, N, N)]))
    obj = np.zeros(1, dtype=OBJECT_DTYPE)
    obj["obj_id"], obj["n_phot"], obj["x0"], obj["y0"], obj["flux_per_photon"] = 5, n, 9.0, 9.0, 1.0
    obj["jac"], obj["winv"] = (1, 0, 0, 1), (1 / 0.3, 0, 0, 1 / 0.3)
    obj["prof_table"], obj["sed_table"], obj["sed_wave"] = -1, -1, 600.0
    obj["stamp_xmin"], obj["stamp_xmax"], obj["stamp_ymin"], obj["stamp_ymax"] = 1, N, 1, N
    obj["bf_state"] = 1 if model_name else 0
    o = orc_loader.OracleScene(sc)
    o.render(obj, nrecalc=10000)
    return o.image.astype(float)


@pytest.mark.parametrize("model,mxx,myy", [(None, 1.08142, 1.08299), ("lsst_itl_50_4", 1.29041, 1.29867),
                                           ("lsst_e2v_50_4", 1.30506, 1.32113)])
def test_sensor_model_moments_match_reference_values(model, mxx, myy):
    """Second moments of the spot against the regression values stored in the reference's
    tests/test_sensor_models.py:13-34.  Those values belong to GalSim's own RNG stream, so they are
    statistical targets here: sigma(M) ~ M sqrt(2/N) = 1.8e-3 for each of the two realisations."""
    img = _sensor_spot(model)
    assert img.sum() == pytest.approx(1.0e6, rel=2e-3)
    got = _moments(img)
    tol = 4 * np.hypot(1.8e-3, 1.8e-3)
    assert abs(got[0] - mxx) < tol, got
    assert abs(got[1] - myy) < tol, got


def test_brighter_fatter_makes_spots_larger_and_conserves_flux():
    """The qualitative criteria of tests/test_sensor_models.py:83-118: the peak drops and the
    radius grows with the Silicon models; flux is conserved."""
    none = _sensor_spot(None)
    e2v = _sensor_spot("lsst_e2v_50_4")
    assert e2v.max() < none.max()
    r0 = np.sqrt(sum(_moments(none)))
    r1 = np.sqrt(sum(_moments(e2v)))
    sigma_r = 1.0 / np.sqrt(1e6)
    assert r1 - r0 > 2 * sigma_r
    assert abs(e2v.sum() - none.sum()) / none.sum() < 2e-3


@pytest.mark.parametrize("vendor", ["itl", "e2v"])
def test_sensor_models_of_4_8_and_32_vertices_agree(vendor):
    """tests/test_sensor_models.py:73-125 of the reference: the spot of the sensor-model case drawn with the 4-, 8- and
    32-vertex pixel models of one vendor has the same radius within 2 sigma_r (sigma_r = 1 / sqrt(flux) pixels), a lower
    peak than without a sensor, and is larger than without by more than 2 sigma_r."""
    none = _sensor_spot(None)
    r0 = np.sqrt(sum(_moments(none)))
    sigma_r = 1.0 / np.sqrt(1e6)
    r = {}
    for nv in (4, 8, 32):
        img = _sensor_spot(f"lsst_{vendor}_50_{nv}")
        assert img.max() <= none.max()
        assert img.sum() == pytest.approx(1.0e6, rel=2e-3)
        r[nv] = np.sqrt(sum(_moments(img)))
        assert r[nv] - r0 > 2 * sigma_r
    assert abs(r[8] - r[4]) < 2 * sigma_r, r
    assert abs(r[32] - r[8]) < 2 * sigma_r, r


def test_oracle_reproduces_the_frozen_spec_digests():
    """tests/golden/pipeline_golden.json freezes the numerics spec: C2 image, C3 photon fields
    after the op chain, the LSST_Image brighter-fatter image and the photon-pooling image of small seeded cases."""
    import json
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_pipeline_golden as g
    want = json.load(open(os.path.join(HERE, "golden", "pipeline_golden.json")))
    assert want["spec"] == "v6"
    got = {k: g.digest(v) for k, v in g.cases(g.oracle_backend).items()}
    assert got == want["sha256"]
