"""The reference's per-call object contracts over the GPU engine (imsim_amd/photon_ops.py): PhotonOp.applyTo on a caller's
photon array (imsim/photon_ops.py:81, :304, :520; tests/test_photon_ops.py:45-66 style arrays), SiliconSensor.accumulate with
resume / recalc (imsim/photon_pooling.py:195-225), against the oracle's op-by-op passes over the same arrays."""
import math

import numpy as np
import pytest

from imsim_amd import _abi, configs, photon_ops, tables
from helpers import assert_bits_equal

pytestmark = pytest.mark.gpu


def _test_photons(n=20000, seed=42, nx=512, with_pupil=True):
    """create_test_photon_array of the reference's tests (tests/test_photon_ops.py:45-66): seed 42, pupil radii in 2.5 .. 4.2"""
    rng = np.random.default_rng(seed)
    r_uv = rng.uniform(2.5, 4.2, n)
    phi = rng.uniform(0.0, 2.0 * math.pi, n)
    f = dict(x=rng.uniform(100.0, nx - 100.0, n), y=rng.uniform(100.0, nx - 100.0, n), flux=np.ones(n),
             wavelength=np.full(n, 577.6))
    if with_pupil:
        f.update(pupil_u=r_uv * np.cos(phi), pupil_v=r_uv * np.sin(phi), time=rng.uniform(0.0, 30.0, n))
    return photon_ops.PhotonArray(n, **f)


def _oracle_pool(scene, photons, row):
    from oracle import orc_loader
    orc = orc_loader.OracleScene(scene)
    n = len(photons.x)
    pool = orc_loader.HostPool(n)
    for f in photon_ops.FIELDS:
        pool.a[f][:n] = getattr(photons, f)
    objects, prefix = orc._objects(row)
    pool.objects, pool.prefix, pool.offs = objects, prefix, np.array([0, n], dtype=np.int64)
    return orc, pool


@pytest.mark.parametrize("name,disable_rot", [("RubinOptics", False), ("RubinDiffraction", False), ("RubinDiffractionOptics", False),
                                              ("RubinDiffractionOptics", True)])
def test_rubin_ops_apply_to_a_callers_photon_array(name, disable_rot):
    """op.applyTo(photon_array, rng=...) mutates the array in place like the reference's operators; every field equals the
    oracle's pass over the same array, vignetted photons carry flux 0 (imsim/photon_ops.py:502), and the pre-conditions of
    the reference are asserted (pupil and time allocated, :139-140, :327-328)."""
    n = 512
    optics = configs.rubin_optics_struct(n, n)
    op = getattr(photon_ops, name)(optics, shift_photons=False, disable_field_rotation=disable_rot, nx=n, ny=n, obj_id=7)
    photons = _test_photons(nx=n)
    before = {f: getattr(photons, f).copy() for f in photon_ops.FIELDS}
    op.applyTo(photons, rng=1234)
    row = photon_ops._one_row(len(photons), 7, n, n, 1, 1)
    reference = photon_ops.PhotonArray(len(photons), **before)
    orc, pool = _oracle_pool(op.scene, reference, row)
    orc.scene.seed = 1234
    orc.bound.base_params.seed = 1234
    orc.apply_ops(pool)
    got = {f: getattr(photons, f) for f in photon_ops.FIELDS}
    want = pool.to_host()
    for f in ("x", "y", "flux", "dxdz", "dydz"):
        assert_bits_equal(got[f], want[f], f"{name}: photon field {f}")
    assert not np.array_equal(got["x"], before["x"])
    if name != "RubinDiffraction":
        assert np.abs(got["dxdz"]).max() > 0.05                      # the beam arrives at f/1.2: slopes of several tenths
        assert 0.9 < np.count_nonzero(got["flux"]) / len(photons) <= 1.0
    with pytest.raises(AssertionError):
        op.applyTo(_test_photons(100, with_pupil=False), rng=1)


def test_bandpass_ratio_and_the_galsim_ops_on_a_photon_array():
    """BandpassRatio(target = 0.8 x initial): sum(flux) = 0.8 N (tests/test_photon_ops.py:768-790); TimeSampler and
    PupilAnnulusSampler fill time and pupil of an array that has none."""
    wl, thr = tables.synthetic_r_band()
    bp = tables.Bandpass(wl, thr)
    op = photon_ops.BandpassRatio(bp * 0.8, bp, nx=256, ny=256)
    photons = _test_photons(5000, nx=256)
    photons.wavelength[:] = np.random.default_rng(3).uniform(560.0, 680.0, len(photons))
    op.applyTo(photons)
    np.testing.assert_allclose(photons.flux.sum(), 0.8 * len(photons), rtol=1e-12)
    chain = photon_ops.DevicePhotonOp([(_abi.IMS_OP_TIME_SAMPLER, 0, [0.0, 30.0]), (_abi.IMS_OP_PUPIL_ANNULUS_SAMPLER, 0, [4.18, 2.55])],
                                      nx=256, ny=256)
    bare = _test_photons(5000, nx=256, with_pupil=False)
    chain.applyTo(bare, rng=9)
    rr = np.hypot(bare.pupil_u, bare.pupil_v)
    assert rr.min() >= 2.55 and rr.max() <= 4.18 and 0.0 <= bare.time.min() and bare.time.max() <= 30.0
    assert abs(np.mean(rr ** 2) - 0.5 * (4.18 ** 2 + 2.55 ** 2)) < 0.1      # uniform over the annulus


def test_silicon_sensor_accumulate_with_resume_and_recalc():
    """sensor.accumulate(photons, image, resume, recalc) (imsim/photon_pooling.py:195-225): three calls on one exposure --
    fresh, resumed with a recalculation, resumed without -- give the oracle's image, flux and pixel-boundary state, and the
    brighter-fatter feedback of the first 2e5 electrons is in the boundaries the later photons see."""
    n = 64
    setup = configs.silicon_setup(n, n, tree_rings=True)
    sensor = photon_ops.SiliconSensor(setup, n, n, has_angles=True, obj_id=3)
    sensor.updateRNG(77)
    rng = np.random.default_rng(5)

    def spot(k):
        f = dict(x=rng.normal(32.3, 1.5, k), y=rng.normal(31.6, 1.5, k), flux=np.ones(k), wavelength=rng.uniform(550.0, 700.0, k),
                 dxdz=rng.normal(0.0, 0.2, k), dydz=rng.normal(0.0, 0.2, k))
        return photon_ops.PhotonArray(k, **f)
    batches = [spot(200000), spot(50000), spot(30000)]
    image = np.zeros((n, n), dtype=np.float32)
    added = [sensor.accumulate(batches[0], image, resume=False, photon_first=0),
             sensor.accumulate(batches[1], image, resume=True, recalc=True, photon_first=200000),
             sensor.accumulate(batches[2], image, resume=True, recalc=False, photon_first=250000)]
    from oracle import orc_loader
    orc = orc_loader.OracleScene(sensor.scene)
    orc.scene.seed = 77
    orc.bound.base_params.seed = 77
    want_added, first = [], 0
    for k, b in enumerate(batches):
        if k == 1:
            orc.update_distortions(0, 1)
        row = photon_ops._one_row(len(b), 3, n, n, 1, 1)
        row["phot_first"] = first
        first += len(b)
        _, pool = _oracle_pool(sensor.scene, b, row)
        real = np.zeros(1)
        orc.accumulate(pool, realized=real)
        want_added.append(float(real[0]))
    assert added == want_added and sum(added) > 0.95 * 280000
    assert_bits_equal(image, orc.image, "accumulated image")
    r = sensor.renderer()
    for name in ("boundary", "bounds"):
        got = r.bound.sensor_arrays[name].cpu().numpy().view(np.float64)
        assert_bits_equal(got, orc.sensor_array(name)[:len(got)], f"sensor {name}")
    fresh = photon_ops.SiliconSensor(setup, n, n, has_angles=True, obj_id=3).renderer()
    moved = r.bound.sensor_arrays["boundary"].cpu().numpy().view(np.float64) - fresh.bound.sensor_arrays["boundary"].cpu().numpy().view(np.float64)
    assert np.abs(moved).max() > 1e-3                                # 2e5 electrons in a few pixels moved their boundaries
    # an integer image takes the photons through a float64 copy (photon_pooling.py:213-225)
    counts = np.zeros((n, n), dtype=np.int32)
    again = photon_ops.SiliconSensor(setup, n, n, has_angles=True, obj_id=3)
    again.updateRNG(77)
    again.accumulate(batches[0], counts, resume=False)
    first = np.zeros((n, n), dtype=np.float32)
    photon_ops.SiliconSensor(setup, n, n, has_angles=True, obj_id=3)
    s2 = photon_ops.SiliconSensor(setup, n, n, has_angles=True, obj_id=3)
    s2.updateRNG(77)
    s2.accumulate(batches[0], first, resume=False)
    assert np.array_equal(counts, first.astype(np.int32)) and counts.sum() > 190000


def test_stamp_builder_draws_single_objects_like_the_batch_path():
    """LSST_SiliconBuilder.setup / buildPSF / getDrawMethod / draw (imsim/stamp.py:109, :251, :312, :411) one object at a time:
    every stamp equals that object's contribution to a CCD rendered by the batch path, bit for bit, the `base` side channel
    carries nominal / phot / fft / realized flux, a zero-photon object raises SkipThisObject, draw_method is honoured."""
    import torch
    from imsim_amd import catalog, fft_draw, stamp
    from imsim_amd.engine import Renderer
    n = 512
    scene = configs.scene_c3(nx=n, ny=n)
    scene.sensor.scratch_cells = 400_000
    cat = catalog.synthetic_catalog(40, nx=n, ny=n)
    cat["x"][:] = np.clip(cat["x"], 120, n - 120)
    cat["y"][:] = np.clip(cat["y"], 120, n - 120)
    cat["nominal_flux"][:3] = [60000.0, 45.0, 3.0e6]                 # a bright star (brighter-fatter rounds), a faint one, an FFT one
    cat["kind"][:3] = [0, 1, 0]
    cat["sb_flux"] = cat["nominal_flux"] / 80.0
    phot = catalog.realize_fluxes(cat["nominal_flux"], scene.seed)
    phot[5] = 0
    fa, fs = catalog.kolmogorov_gaussian_fwhm(configs.VISIT["airmass"], configs.VISIT["raw_seeing"], "r")
    builder = stamp.LSST_SiliconBuilder(scene, lambda c, p: configs.c3_objects(c, p, scene), kpsf=fft_draw.kolmogorov_gaussian_kpsf(fa, fs),
                                        fwhm_total=float(np.hypot(fa, fs)), fft_sb_thresh=2.0e4, nrecalc=10000)
    rows_all, _ = configs.c3_objects(cat, np.maximum(phot, 1), scene)
    for k in (0, 1, 2, 7, 5):
        base = {"_object": {key: v[k:k + 1] for key, v in cat.items() if isinstance(v, np.ndarray)}, "phot_flux": int(phot[k]), "seed": scene.seed}
        if phot[k] == 0:
            with pytest.raises(stamp.SkipThisObject):
                builder.setup({}, base)
            continue
        xs, ys, image_pos, world_pos = builder.setup({}, base)
        psf = builder.buildPSF({}, base)
        method = builder.getDrawMethod({"draw_method": "auto"}, base)
        assert (method == "fft") == (k == 2) and (psf is not None) == (k == 2)
        row = builder._rows
        img = stamp.StampImage(row["stamp_xmin"][0], row["stamp_xmax"][0], row["stamp_ymin"][0], row["stamp_ymax"][0], dtype=np.float64)
        assert img.array.shape == (ys, xs)
        out = builder.draw(None, img, method, None, {}, base)
        assert out is img and img.added_flux == base["realized_flux"] > 0
        assert base["nominal_flux"] == cat["nominal_flux"][k]
        if method == "fft":
            assert base["phot_flux"] == 0.0 and base["fft_flux"] == cat["nominal_flux"][k]
            np.testing.assert_allclose(img.array.sum(), cat["nominal_flux"][k], rtol=0.05)
            continue
        assert base["phot_flux"] == phot[k] and 0.8 * phot[k] < img.added_flux <= phot[k]
        r = Renderer(scene)
        r.render_lsst_image(row, nrecalc=10000)
        r.synchronize()
        full = r.image64_numpy()
        x0, x1, y0, y1 = img.bounds
        assert_bits_equal(img.array, full[y0 - 1:y1, x0 - 1:x1], f"object {k}: stamp vs batch render")
        assert full.sum() == img.array.sum()
    with pytest.raises(ValueError):
        builder.getDrawMethod({"draw_method": "psychic"}, {})
    assert builder.getDrawMethod({"draw_method": "phot"}, {}) == "phot"
