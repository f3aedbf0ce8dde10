"""The object table built on the device (ims_build_object_table).

CPU: the oracle's restatement (oracle/orc_catalog.c) against the numpy builder, which carries the reference's stamp-size
regression values (tests/test_stamp_size.py): identical integers (photon counts, tables, flags, stamp bounds), floats to
rounding, the local WCS to the 2e-9 by which the analytic jacobian differs from the central difference it replaces.
GPU: the device table equals the oracle's bit for bit, with the fluxes given and with the Poisson realisation done by the
kernel; launch tables gathered from it render the image of the same rows planned on the host."""
import numpy as np
import pytest

from imsim_amd import configs, catalog, _abi
from oracle import orc_loader

VISIT = dict(configs.VISIT)


def _case(n=6000, lens=False, seed=20261001):
    scene = configs.BENCH_CONFIGS["c3"]["scene"]()
    cat = catalog.synthetic_catalog(n, seed=seed, nx=scene.nx, ny=scene.ny)
    if lens:
        rng = np.random.default_rng(5)
        cat["g1"], cat["g2"] = rng.normal(0, 0.03, n), rng.normal(0, 0.03, n)
        cat["mu"] = 1.0 + rng.normal(0, 0.05, n)
    phot = catalog.realize_fluxes(cat["nominal_flux"], scene.seed)
    return scene, cat, phot


@pytest.mark.parametrize("lens", [False, True])
def test_oracle_table_matches_the_numpy_builder(lens):
    scene, cat, phot = _case(lens=lens)
    rows, meta = orc_loader.build_object_table(scene, cat, VISIT, phot)
    ref, sizes = configs.c3_objects(cat, phot, scene)
    keep = phot > 0
    r, m = rows[keep], meta[keep]
    for f in ("obj_id", "n_phot", "phot_first", "prof_table", "sed_table", "flags", "bf_state"):
        assert np.array_equal(r[f], ref[f]), f
    for f in ("x0", "y0", "prof_scale", "flux_per_photon"):
        assert np.array_equal(r[f], ref[f]), f
    for f, tol in (("jac", 1e-14), ("dcr_tanz", 1e-13), ("dcr_sinp", 1e-13), ("dcr_cosp", 1e-13), ("winv", 5e-9)):
        assert np.abs(r[f] - ref[f]).max() <= tol * max(np.abs(ref[f]).max(), 1.0), f
    # every stamp size, the surface-brightness loop of the bright / oversized galaxies included (get_good_phot_stamp_size,
    # imsim/stamp_utils.py:196-220, :300-354, restated in the table builder): the numpy builder's integers
    pend = (m["flags"] & _abi.IMS_META_SIZE_PENDING) != 0
    assert pend.sum() == 0
    assert np.array_equal(m["size"], sizes)
    for f in ("stamp_xmin", "stamp_xmax", "stamp_ymin", "stamp_ymax"):
        assert np.array_equal(r[f], ref[f]), f
    # the field angle of the atmospheric PSF
    thx, thy = configs.field_angles(scene, cat["x"][keep], cat["y"][keep])
    assert np.abs(r["atm_tan_x"] - thx).max() < 1e-12 and np.abs(r["atm_tan_y"] - thy).max() < 1e-12
    # without the tables of that loop the builder flags the bright few and the host's loop closes the gap (the checker's form)
    from imsim_amd import device_table
    rows2, meta2 = orc_loader.build_object_table(scene, cat, VISIT, phot, sizes_on_device=False)
    pend2 = (meta2["flags"] & _abi.IMS_META_SIZE_PENDING) != 0
    assert 0 < pend2[keep].sum() < 0.05 * len(r)
    assert (m["size"][pend2[keep]] != meta2["size"][keep][pend2[keep]]).any()        # the loop did change sizes
    idx, sz, host = device_table.host_fixups(cat, meta2, {})
    assert len(host) == 0
    full = meta2["size"].copy()
    full[idx] = sz
    assert np.array_equal(full[keep], sizes)


def test_oracle_flux_realisation_is_poisson():
    scene, cat, _ = _case(n=20000)
    rows, meta = orc_loader.build_object_table(scene, cat, VISIT, None)
    mean = cat["nominal_flux"]
    z = (meta["n_phot"] - mean) / np.sqrt(mean)
    assert abs(z.mean()) < 0.03 and abs(z.std() - 1.0) < 0.03
    assert np.array_equal(meta["n_phot"], rows["n_phot"])
    again, _ = orc_loader.build_object_table(scene, cat, VISIT, None)
    assert np.array_equal(again["n_phot"], rows["n_phot"])


@pytest.mark.gpu
@pytest.mark.parametrize("given", [True, False])
def test_device_table_is_bit_identical_to_the_oracle(given):
    import torch
    from imsim_amd.engine import Renderer
    from imsim_amd.device_table import DeviceTable
    scene, cat, phot = _case(n=20000, lens=True)
    r = Renderer(scene)
    t = DeviceTable(r, cat, VISIT, phot_flux=phot if given else None)
    torch.cuda.synchronize()
    rows, meta = orc_loader.build_object_table(scene, cat, VISIT, phot if given else None)
    assert not (meta["flags"] & _abi.IMS_META_SIZE_PENDING).any() and not (t.meta["flags"] & _abi.IMS_META_SIZE_PENDING).any()
    got = t.rows_numpy()
    assert got.tobytes() == rows.tobytes()                     # the kernel's surface-brightness loop included: ONE pass
    assert np.array_equal(t.n_phot, rows["n_phot"])
    # the form with the host's loop (the checker): same table
    from imsim_amd import device_table
    t_host = DeviceTable(r, cat, VISIT, phot_flux=phot if given else None, sizes_on_device=False)
    torch.cuda.synchronize()
    assert t_host.rows_numpy().tobytes() == got.tobytes()


def _edge_catalog():
    """A catalog whose first rows sit on the edges of the table builder: fluxes of 0 and far below one photon, the tiny-flux
    and faint / photon-shooting thresholds and their neighbours, a 1e9-photon star and galaxy (stamp capped at the maximum),
    sources far off the CCD on every side, positions exactly on pixel centres and pixel borders, the smallest and largest
    half-light radii and the flattest ellipse of the catalog generator's ranges and beyond."""
    scene, cat, _ = _case(n=400, seed=11)
    flux = [0.0, 1e-6, 0.4, 0.999, 1.0, 9.9, 10.0, 10.1, 31.9, 32.0, 99.9, 100.0, 100.1, 1e4 - 1, 1e4, 1e4 + 1, 1e6, 1e9, 1e9, 3e7]
    k = len(flux)
    cat["nominal_flux"][:k] = flux
    cat["sb_flux"][:k] = np.asarray(flux) / 76.4
    cat["kind"][:k] = [0, 1, 2, 0, 1, 2, 0, 1, 2, 0, 1, 2, 0, 1, 2, 0, 1, 0, 2, 1]
    pos = [(-800.0, 2000.0), (4096.0 + 900.0, 10.0), (2000.0, -650.5), (30.0, 4096.0 + 2000.0), (-3000.0, -3000.0), (0.5, 0.5),
           (1.0, 1.0), (4096.5, 4096.5), (2048.0, 2048.5), (2048.5, 2048.0)]
    for i, (x, y) in enumerate(pos):
        cat["x"][k + i], cat["y"][k + i] = x, y
        cat["nominal_flux"][k + i] = 5000.0 + 700.0 * i
        cat["sb_flux"][k + i] = cat["nominal_flux"][k + i] / 76.4
    j = k + len(pos)
    cat["hlr"][j:j + 6] = [0.01, 0.05, 3.0, 8.0, 0.3, 0.3]
    cat["q"][j:j + 6] = [1.0, 0.05, 0.05, 1.0, 0.051, 0.999]
    cat["kind"][j:j + 6] = [1, 1, 2, 2, 1, 2]
    cat["nominal_flux"][j:j + 6] = [2e5, 2e5, 2e6, 5e3, 3e3, 3e3]
    cat["sb_flux"][j:j + 6] = cat["nominal_flux"][j:j + 6] / 76.4
    return scene, cat


def test_oracle_table_on_the_edges_matches_the_numpy_builder():
    scene, cat = _edge_catalog()
    phot = catalog.realize_fluxes(cat["nominal_flux"], scene.seed)
    assert phot[0] == 0 and phot[1] == 0
    rows, meta = orc_loader.build_object_table(scene, cat, VISIT, phot)
    ref, sizes = configs.c3_objects(cat, phot, scene)
    keep = phot > 0
    r = rows[keep]
    for f in ("obj_id", "n_phot", "prof_table", "sed_table", "flags", "x0", "y0", "prof_scale", "flux_per_photon"):
        assert np.array_equal(r[f], ref[f]), f
    assert not (meta["flags"] & _abi.IMS_META_SIZE_PENDING).any()
    assert np.array_equal(meta["size"][keep], sizes)               # 1e9-photon galaxy, flattest ellipse, largest half-light radius ...
    assert sizes.max() == catalog.NMAX and (sizes == 32).any()
    from imsim_amd import device_table
    rows2, meta2 = orc_loader.build_object_table(scene, cat, VISIT, phot, sizes_on_device=False)
    idx, sz, host = device_table.host_fixups(cat, meta2, {})
    assert len(host) == 0 and len(idx) > 0
    full = meta2["size"].copy()
    full[idx] = sz
    assert np.array_equal(full[keep], sizes)


@pytest.mark.gpu
@pytest.mark.parametrize("given", [True, False])
def test_device_table_on_the_edges_is_bit_identical_to_the_oracle(given):
    import torch
    from imsim_amd.engine import Renderer
    from imsim_amd.device_table import DeviceTable
    from imsim_amd import device_table
    scene, cat = _edge_catalog()
    phot = catalog.realize_fluxes(cat["nominal_flux"], scene.seed) if given else None
    r = Renderer(scene)
    t = DeviceTable(r, cat, VISIT, phot_flux=phot)
    torch.cuda.synchronize()
    rows, meta = orc_loader.build_object_table(scene, cat, VISIT, phot)
    assert t.rows_numpy().tobytes() == rows.tobytes()
    assert np.array_equal(t.n_phot, rows["n_phot"])
    t_host = DeviceTable(r, cat, VISIT, phot_flux=phot, sizes_on_device=False)
    torch.cuda.synchronize()
    assert t_host.rows_numpy().tobytes() == rows.tobytes()
    # and the table renders: LSST_Image from the device table == the oracle on its rows (the three sources above 1e7 photons
    # dimmed to 2e5 for this part: the oracle is one CPU core)
    dim = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in cat.items()}
    dim["nominal_flux"] = np.minimum(cat["nominal_flux"], 2.0e5)
    dim["sb_flux"] = dim["nominal_flux"] / 76.4
    phot2 = catalog.realize_fluxes(dim["nominal_flux"], scene.seed) if given else None
    t2 = DeviceTable(r, dim, VISIT, phot_flux=phot2)
    r.render_lsst_image(t2, nrecalc=10000)
    r.synchronize()
    host_rows = t2.rows_numpy()
    assert host_rows["n_phot"].sum() < 3e6
    orc = orc_loader.OracleScene(scene)
    orc.render_lsst_image(host_rows[host_rows["n_phot"] > 0], nrecalc=10000)
    assert orc.image.sum() > 0
    assert np.array_equal(r.image_numpy().view(np.uint32), orc.image.view(np.uint32))


@pytest.mark.gpu
def test_plan_from_the_device_table_renders_the_host_planned_image():
    import torch
    from imsim_amd.engine import Renderer
    from imsim_amd.device_table import DeviceTable
    scene, cat, phot = _case(n=3000, seed=7)
    r = Renderer(scene)
    t = DeviceTable(r, cat, VISIT, phot_flux=phot)
    real = torch.zeros(t.n, dtype=torch.float64, device="cuda")
    r.render_lsst_image(t, nrecalc=2000, realized=real)
    r.synchronize()
    img = r.image_numpy()
    rows = t.rows_numpy()
    keep = rows["n_phot"] > 0
    r2 = Renderer(scene)
    real2 = torch.zeros(int(keep.sum()), dtype=torch.float64, device="cuda")
    r2.render_lsst_image(rows[keep], nrecalc=2000, realized=real2)
    r2.synchronize()
    assert img.sum() > 0
    assert np.array_equal(img.view(np.uint32), r2.image_numpy().view(np.uint32))
    assert np.array_equal(real.cpu().numpy()[keep], real2.cpu().numpy())
    orc = orc_loader.OracleScene(scene)
    orc.render_lsst_image(rows[keep], nrecalc=2000)
    assert np.array_equal(img.view(np.uint32), orc.image.view(np.uint32))


@pytest.mark.gpu
def test_photon_pooling_from_the_device_table_equals_the_host_table_form():
    """LSST_PhotonPoolingImage semantics (batch shares, whole-CCD recalculation per batch) driven from a DeviceTable: batch
    shares by index arithmetic, shoot and per-batch tables gathered on the device -- the image, the realized fluxes and the
    pixel-boundary state of the host-table form, bit for bit."""
    import torch
    from imsim_amd import photon_pooling, stamp
    from imsim_amd.engine import Renderer
    from imsim_amd.device_table import DeviceTable
    scene, cat, phot = _case(n=4000, seed=11)
    scene.track_static_delta = 1
    scene.sensor.scratch_cells = 0
    out = []
    for device_form in (True, False):
        r = Renderer(scene)
        t = DeviceTable(r, cat, VISIT, phot_flux=phot)
        if device_form:
            n_phot = t.n_phot
            src = t
        else:
            rows = t.rows_numpy()
            n_phot = rows["n_phot"]
            src = rows
        modes = stamp.classify(n_phot.astype(float), 100.0)
        real = torch.zeros(t.n, dtype=torch.float64, device="cuda")
        step = photon_pooling.prepared_image(r, src, modes, nbatch=5, seed=3, realized=real)
        step()
        r.synchronize()
        out.append((r.image_numpy(), real.cpu().numpy(), r.bound.sensor_arrays["boundary"].cpu().numpy()))
    assert out[0][0].sum() > 0
    assert np.array_equal(out[0][0].view(np.uint32), out[1][0].view(np.uint32))
    assert np.array_equal(out[0][1], out[1][1])
    assert np.array_equal(out[0][2], out[1][2])
