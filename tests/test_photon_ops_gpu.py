"""Every registered PhotonOp kind on the GPU against the oracle, plus the reference's own op-level criteria
(tests/test_photon_ops.py): RubinDiffraction then RubinOptics == RubinDiffractionOptics (:281-318), field rotation
on / off (:339-427), BandpassRatio sum(flux) = 0.8 N (:768-790)."""
import math

import numpy as np
import pytest

from imsim_amd import _abi, configs, catalog
from helpers import assert_bits_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def _scene(ops, n=512, **kw):
    sc = configs.scene_c3(nx=n, ny=n, sensor=False, **kw)
    sc.ops = ops
    return sc


def _objects(scene, n_obj=80, n=512, flux_seed=4):
    cat = catalog.synthetic_catalog(n_obj, nx=n, ny=n)
    phot = catalog.realize_fluxes(cat["nominal_flux"], flux_seed)
    objects, _ = configs.c3_objects(cat, phot, scene)
    return objects


def _pools(scene, objects):
    from imsim_amd.engine import Renderer
    from oracle import orc_loader
    r = Renderer(scene)
    pool = r.shoot_photons(objects)
    r.apply_ops(pool)
    r.synchronize()
    orc = orc_loader.OracleScene(scene)
    opool = orc.shoot_pool(objects)
    orc.apply_ops(opool)
    return pool.to_host(), opool.to_host()


SAMPLERS = [(_abi.IMS_OP_TIME_SAMPLER, 0, [0.0, 30.0]), (_abi.IMS_OP_PUPIL_ANNULUS_SAMPLER, 0, [4.18, 2.55])]


@pytest.mark.parametrize("disable_rot", [0.0, 1.0])
def test_standalone_rubin_diffraction_and_rubin_optics_are_bit_exact(torch_cuda, disable_rot):
    """op kinds 5 (RubinDiffraction: xy -> v, spider kick, v -> xy through the inverse WCS chain) and 4 (RubinOptics)
    as separate chain entries, with and without field rotation: every photon field equals the oracle's bits."""
    ops = SAMPLERS + [(_abi.IMS_OP_RUBIN_DIFFRACTION, 0, [1.0, disable_rot]), (_abi.IMS_OP_RUBIN_OPTICS, 0, [1.0, 0.0]),
                      (_abi.IMS_OP_REFRACTION, 0, [3.9])]
    scene = _scene(ops)
    g, o = _pools(scene, _objects(scene))
    assert np.count_nonzero(g["flux"]) > 0.9 * len(g["flux"])
    for f in g:
        assert_bits_equal(g[f], o[f], f"photon field {f} (disable_field_rotation={disable_rot})")


def test_rubin_diffraction_alone_moves_photons_only_slightly(torch_cuda):
    """kind 5 alone: positions change by the diffraction kick only (sub-pixel for almost all photons), slopes untouched."""
    base = _scene(list(SAMPLERS))
    objects = _objects(base)
    g0, _ = _pools(base, objects)
    sc = _scene(SAMPLERS + [(_abi.IMS_OP_RUBIN_DIFFRACTION, 0, [1.0, 0.0])])
    g, o = _pools(sc, objects)
    for f in g:
        assert_bits_equal(g[f], o[f], f"photon field {f}")
    d = np.hypot(g["x"] - g0["x"], g["y"] - g0["y"])
    assert np.median(d) < 0.5 and d.max() > 1e-6
    assert not g["dxdz"].any() and not g["dydz"].any()


def test_diffraction_then_optics_equals_the_fused_op(torch_cuda):
    """tests/test_photon_ops.py:281-318: RubinDiffraction followed by RubinOptics gives the positions of
    RubinDiffractionOptics (assert_array_almost_equal, 6 decimals).  Both chains put the diffraction at op index 2,
    so they draw the same deviate, as the reference's test does with two identically seeded rngs."""
    fused = _scene(SAMPLERS + [(_abi.IMS_OP_RUBIN_DIFFRACTION_OPTICS, 0, [1.0, 0.0])])
    seq = _scene(SAMPLERS + [(_abi.IMS_OP_RUBIN_DIFFRACTION, 0, [1.0, 0.0]), (_abi.IMS_OP_RUBIN_OPTICS, 0, [1.0, 0.0])])
    objects = _objects(fused, n_obj=120)
    gf, _ = _pools(fused, objects)
    gs, _ = _pools(seq, objects)
    ok = (gf["flux"] > 0) & (gs["flux"] > 0)
    assert ok.sum() > 0.9 * len(ok)
    np.testing.assert_array_almost_equal(gf["x"][ok], gs["x"][ok], decimal=6)
    np.testing.assert_array_almost_equal(gf["y"][ok], gs["y"][ok], decimal=6)
    np.testing.assert_allclose(gf["dxdz"][ok], gs["dxdz"][ok], atol=1e-9)
    assert np.array_equal(gf["flux"] > 0, gs["flux"] > 0) or np.mean((gf["flux"] > 0) != (gs["flux"] > 0)) < 1e-4


def _spike_scene(t, disable_rot, altitude=89.9, azimuth=45.0, latitude=-30.24463):
    """One very bright point source without PSF, all photons at time t, pointing close to the zenith (where the
    field rotates fastest), as tests/test_photon_ops.py:339-391 sets it up."""
    from imsim_amd import diffraction
    ops = [(_abi.IMS_OP_TIME_SAMPLER, 0, [t, 0.0]), (_abi.IMS_OP_PUPIL_ANNULUS_SAMPLER, 0, [4.18, 2.55]),
           (_abi.IMS_OP_RUBIN_DIFFRACTION_OPTICS, 0, [1.0, disable_rot])]
    sc = _scene(ops, n=512)
    sc.psf = []
    opt = type(sc.optics).from_buffer_copy(bytes(sc.optics))
    diffraction.fill_optics(opt, math.radians(latitude), math.radians(azimuth), math.radians(altitude))
    sc.optics = opt
    return sc


def _star(scene, n_phot):
    cat = catalog.synthetic_catalog(1, nx=512, ny=512)
    cat["x"][:] = 256.0
    cat["y"][:] = 256.0
    cat["kind"][:] = 0                                        # point source
    cat["nominal_flux"][:] = float(n_phot)
    objects, _ = configs.c3_objects(cat, np.array([n_phot]), scene)
    assert objects["prof_table"][0] == _abi.IMS_PROF_POINT
    return objects


def _spike_angles(x, y, r=20.0):
    cx, cy = np.median(x), np.median(y)
    far = np.hypot(x - cx, y - cy) > r
    return np.arctan2(y[far] - cy, x[far] - cx)


def _cross_angle(theta):
    """Angle of the four-armed cross: the median of the spike angles modulo pi/2, taken around their circular mean
    so that a cross lying near the wrap of the modulo is not biased (the reference's plain `median(angles % (pi/2))`
    assumes it does not)."""
    c = math.atan2(np.mean(np.sin(4 * theta)), np.mean(np.cos(4 * theta))) / 4
    return c + np.median((theta - c + math.pi / 4) % (math.pi / 2) - math.pi / 4)


def _field_rotation_angle(latitude, altitude, azimuth, t):
    """imsim/diffraction.py field_rotation_matrix restated (tests/test_photon_ops.py:417-427)."""
    from imsim_amd import diffraction
    ef = diffraction.e_equatorial(latitude, altitude, azimuth)
    ez0 = diffraction.zenith_direction(latitude)
    w = diffraction.OMEGA_EARTH * t
    ez = np.array([math.cos(latitude) * math.cos(w), math.cos(latitude) * math.sin(w), math.sin(latitude)])
    eh = np.cross(ef, ez)
    g = np.cross(ef, ez0)
    nrm = np.linalg.norm(eh) * np.linalg.norm(g)
    return math.atan2(np.dot(ez, g) / nrm, np.dot(eh, g) / nrm)


def test_spikes_rotate_with_the_field_and_stop_when_disabled(torch_cuda):
    """tests/test_photon_ops.py:339-414: the cross of spider spikes turns by the field-rotation angle between t = 0 and
    t = dt (rtol 0.03); with disable_field_rotation the two photon sets coincide."""
    from imsim_amd.engine import Renderer
    dt = 1.0
    lat, alt, az = math.radians(-30.24463), math.radians(89.9), math.radians(45.0)
    res = {}
    for disable in (0.0, 1.0):
        for t in (0.0, dt):
            sc = _spike_scene(t, disable)
            r = Renderer(sc)
            pool = r.shoot_photons(_star(sc, 1_000_000))
            r.apply_ops(pool)
            r.synchronize()
            g = pool.to_host()
            ok = g["flux"] > 0
            res[(disable, t)] = (g["x"][ok], g["y"][ok], ok)
    a0 = _cross_angle(_spike_angles(*res[(0.0, 0.0)][:2]))
    a1 = _cross_angle(_spike_angles(*res[(0.0, dt)][:2]))
    expected = _field_rotation_angle(lat, alt, az, dt)
    assert abs(expected) > 1e-3                           # near the zenith the field turns by degrees per second
    np.testing.assert_allclose(abs(a1 - a0), abs(expected), rtol=0.03)
    x0, y0, ok0 = res[(1.0, 0.0)]
    x1, y1, ok1 = res[(1.0, dt)]
    assert np.array_equal(ok0, ok1)
    np.testing.assert_array_almost_equal(x0, x1)
    np.testing.assert_array_almost_equal(y0, y1)


def test_bandpass_ratio_reweights_fluxes(torch_cuda):
    """op kind 9 (imsim/photon_ops.py:506-533) with target = 0.8 x initial (tests/test_photon_ops.py:768-790):
    fluxes equal the oracle's bits, sum(flux) = 0.8 N to 1e-3; the rendered image takes the f64 deposit path and agrees
    with the oracle to 1e-12 relative (non-integer sums are order-dependent in the last bits), realized flux included."""
    from imsim_amd import tables
    from imsim_amd.engine import Renderer
    from oracle import orc_loader
    wl, thr = tables.synthetic_r_band()
    bp = tables.Bandpass(wl, thr)
    table, wl_min, wl_step = (bp * 0.8).ratio_table(bp)
    ops = [(_abi.IMS_OP_BANDPASS_RATIO, 0, [])]
    scene = configs.scene_c2(nx=512, ny=512)
    scene.ops = ops
    scene.ratio_tables, scene.ratio_wl_min, scene.ratio_wl_step = table[None, :], wl_min, wl_step
    cat = catalog.synthetic_catalog(300, nx=512, ny=512)
    phot = catalog.realize_fluxes(cat["nominal_flux"], 2)
    objects, _ = catalog.build_object_table(cat, phot)
    r = Renderer(scene)
    pool = r.shoot_photons(objects)
    r.apply_ops(pool)
    r.synchronize()
    orc = orc_loader.OracleScene(scene)
    opool = orc.shoot_pool(objects)
    orc.apply_ops(opool)
    g, o = pool.to_host(), opool.to_host()
    assert_bits_equal(g["flux"], o["flux"], "reweighted flux")
    np.testing.assert_allclose(g["flux"].sum(), 0.8 * len(g["flux"]), rtol=1e-3)
    r2 = Renderer(scene)
    real = torch_cuda.zeros(len(objects), dtype=torch_cuda.float64, device="cuda")
    r2.render(objects, realized=real)
    r2.synchronize()
    orc2 = orc_loader.OracleScene(scene)
    real_o = np.zeros(len(objects))
    orc2.render(objects, realized=real_o)
    img = r2.image.cpu().numpy()
    assert np.array_equal(img != 0, orc2.image64 != 0)                       # pixel indices: exact
    np.testing.assert_allclose(img, orc2.image64, rtol=1e-12, atol=0)
    np.testing.assert_allclose(real.cpu().numpy(), real_o, rtol=1e-12)
    # the same photons with unit flux: every pixel holds 1 / 0.8 of the reweighted charge
    scene.ops, scene.ratio_tables = [], None
    r3 = Renderer(scene)
    r3.render(objects)
    r3.synchronize()
    np.testing.assert_allclose(img, 0.8 * r3.image.cpu().numpy(), rtol=1e-12)
