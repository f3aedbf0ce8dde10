"""GPU parity tests proper: the HIP path, called through the C-ABI, against the CPU oracle."""
import os
import ctypes as C

import numpy as np
import pytest

from imsim_amd import _abi
from helpers import small_case, assert_bits_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def _dev_math(torch, which, x, seed=0, obj=0, slot=0, n=None):
    lib = _abi.load()
    m = 2 if which in (2, 4, 6, 10) else 1
    if which == 6:
        xin = torch.zeros(1, dtype=torch.float64, device="cuda")
    else:
        xin = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)).cuda()
        n = xin.numel() // 2 if which == 12 else xin.numel()
    out = torch.empty(n * m, dtype=torch.float64, device="cuda")
    _abi.check(lib.ims_test_math(which, xin.data_ptr(), out.data_ptr(), n, seed, obj, slot, None))
    torch.cuda.synchronize()
    return out.cpu().numpy()


def test_lean_div_sqrt(torch_cuda):
    """ddiv / dsqrt_n / dsqrt0 (ims_math.h: the division and square-root cores without range scaling and
    special-value fix-ups) give the correctly rounded IEEE result, bit for bit, over the stated operand range."""
    torch = torch_cuda
    lib = _abi.load()
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    n = 1 << 22

    def run(which, xin):
        out = torch.empty(n, dtype=torch.float64, device="cuda")
        _abi.check(lib.ims_test_math(which, xin.data_ptr(), out.data_ptr(), n, 0, 0, 0, None))
        torch.cuda.synchronize()
        return out

    def rnd(lo, hi, signed=False):
        m = 1.0 + torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
        m = m + torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * 2.0 ** -30
        e = torch.randint(lo, hi + 1, (n,), device="cuda", generator=g).to(torch.float64)
        v = m * torch.exp2(e)
        if signed:
            v = v * (torch.randint(0, 2, (n,), device="cuda", generator=g).to(torch.float64) * 2 - 1)
        return v

    for lo, hi in ((-1, 1), (-40, 40), (-240, 240)):
        x = rnd(2 * lo, 2 * hi)
        ref = torch.sqrt(x).view(torch.int64)
        assert int((run(7, x).view(torch.int64) != ref).sum()) == 0
        assert int((run(9, x).view(torch.int64) != ref).sum()) == 0
        a, b = rnd(lo, hi, True), rnd(lo, hi, True)
        q = run(8, torch.stack([a, b], dim=1).contiguous().view(-1))
        assert int((q.view(torch.int64) != (a / b).view(torch.int64)).sum()) == 0
    # exact cases: perfect squares, zero numerators, sqrt(0)
    k = torch.arange(1, n + 1, dtype=torch.float64, device="cuda")
    assert bool((run(7, k * k) == k).all())
    z = torch.zeros(n, dtype=torch.float64, device="cuda")
    assert bool((run(9, z) == 0.0).all())
    assert bool((run(8, torch.stack([z, k], dim=1).contiguous().view(-1)) == 0.0).all())


WORD_EDGES = np.array([0, 1, 2 ** 29 - 1, 2 ** 29, 2 ** 29 + 1, 2 ** 30, 2 ** 31 - 1, 2 ** 31, 3 * 2 ** 29 - 1, 3 * 2 ** 29,
                       5 * 2 ** 29, 7 * 2 ** 29 - 1, 7 * 2 ** 29, 2 ** 32 - 2, 2 ** 32 - 1], dtype=np.float64)


def test_device_math_is_bit_identical_to_oracle(torch_cuda):
    """The numerics spec: every elementary function gives the same bits on gfx950 and on the CPU."""
    from oracle import orc_loader
    rng = np.random.default_rng(11)
    cases = {
        0: np.concatenate([rng.uniform(0, 1, 200000), 10 ** rng.uniform(-16, 3, 100000)]),
        1: rng.uniform(-60, 60, 200000),
        2: rng.uniform(0, 1, 200000),
        3: np.concatenate([rng.uniform(-5, 5, 100000), 10 ** rng.uniform(-12, 6, 100000)]),
        4: rng.uniform(-100, 100, 200000),
        5: rng.uniform(0, 30, 100000),
        # spec v6: the deviate functions take 32-bit words (as integer-valued doubles); edge words first
        10: np.concatenate([WORD_EDGES, rng.integers(0, 2 ** 32, 200000).astype(np.float64)]),
        11: np.concatenate([WORD_EDGES, rng.integers(0, 2 ** 32, 200000).astype(np.float64)]),
        12: rng.integers(0, 2 ** 32, 400000).astype(np.float64),
    }
    for which, x in cases.items():
        x = x[x > 0] if which == 0 else x
        assert_bits_equal(_dev_math(torch_cuda, which, x), orc_loader.math_probe(which, x), f"math fn {which}")
    g = _dev_math(torch_cuda, 6, None, seed=12345, obj=77, slot=3, n=300000)
    assert_bits_equal(g, orc_loader.gauss_probe(12345, 77, 300000, 3), "gaussian pairs")


def test_c2_fused_image_is_bit_exact(torch_cuda):
    """C2 (phot, Gaussian PSF, no sensor): integer photon counts per pixel are identical."""
    from imsim_amd.engine import Renderer
    from oracle import orc_loader
    scene, objects, _ = small_case(n_obj=400, nx=512, ny=512)
    r = Renderer(scene)
    real = torch_cuda.zeros(len(objects), dtype=torch_cuda.float64, device="cuda")
    r.render(objects, realized=real)
    r.synchronize()
    orc = orc_loader.OracleScene(scene)
    real_o = np.zeros(len(objects))
    orc.render(objects, realized=real_o)
    assert r.image_numpy().sum() > 0
    assert_bits_equal(r.image_numpy(), orc.image, "C2 image")
    assert_bits_equal(real.cpu().numpy(), real_o, "realized_flux")


def test_pooled_photons_are_bit_exact_and_match_fused(torch_cuda):
    """LSST_Photons path: the photon pool equals the oracle's bit for bit; accumulating it gives
    the same pixel indices, and the same image as the fused kernel (batching invariance)."""
    from imsim_amd.engine import Renderer
    from oracle import orc_loader
    scene, objects, _ = small_case(n_obj=300, nx=512, ny=512, flux_seed=3)
    r = Renderer(scene)
    pool = r.shoot_photons(objects)
    r.apply_ops(pool)
    pix = r.accumulate(pool, want_pixel_index=True)
    r.synchronize()
    orc = orc_loader.OracleScene(scene)
    opool = orc.shoot_pool(objects)
    orc.apply_ops(opool)
    opix = orc.accumulate(opool, want_pixel_index=True)
    g, o = pool.to_host(), opool.to_host()
    for f in g:
        assert_bits_equal(g[f], o[f], f"photon field {f}")
    assert_bits_equal(pix.cpu().numpy(), opix, "pixel indices")
    assert_bits_equal(r.image_numpy(), orc.image, "pooled image")
    r2 = Renderer(scene)
    r2.render(objects)
    r2.synchronize()
    assert_bits_equal(r2.image_numpy(), r.image_numpy(), "fused vs pooled image")


def test_shoot_ops_photons_one_launch_equals_shoot_then_ops(torch_cuda):
    """ims_shoot_ops_photons (shoot + PSF + op chain in one launch) stores the photons of ims_shoot_photons +
    ims_apply_ops bit for bit; its `converted` form followed by ims_accumulate_segments (what the brighter-fatter chains
    run: conversion and diffusion in the producing kernel, only the pixel search in the consumer) gives the image of
    the fused kernel and of the oracle."""
    from imsim_amd.engine import Renderer
    from oracle import orc_loader
    scene, objects = _c3_case(n_obj=120)
    r = Renderer(scene)
    a = r.shoot_photons(objects)
    r.apply_ops(a)
    b = r.shoot_ops_photons(objects)
    r.synchronize()
    ga, gb = a.to_host(), b.to_host()
    for f in ga:
        assert_bits_equal(gb[f], ga[f], f"photon field {f}")
    c = r.shoot_ops_photons(objects, converted=True)
    real = torch_cuda.zeros(len(objects), dtype=torch_cuda.float64, device="cuda")
    r.accumulate_segments(c, realized=real)
    r.synchronize()
    with pytest.raises(Exception, match="converted"):
        r.accumulate_segments(b)
    orc = orc_loader.OracleScene(scene)
    real_o = np.zeros(len(objects))
    orc.render(objects, realized=real_o)
    assert r.image_numpy().sum() > 0
    assert_bits_equal(r.image_numpy(), orc.image, "converted pool + pixel search vs oracle")
    assert_bits_equal(real.cpu().numpy(), real_o, "realized flux")
    r2 = Renderer(scene)
    r2.render(objects)
    r2.synchronize()
    assert_bits_equal(r2.image_numpy(), r.image_numpy(), "fused vs converted pool")
    # ims_accumulate_small (one wavefront per object row, here looping over rows of up to millions of photons)
    r3 = Renderer(scene)
    real3 = torch_cuda.zeros(len(objects), dtype=torch_cuda.float64, device="cuda")
    r3.accumulate_segments(r3.shoot_ops_photons(objects, converted=True), realized=real3, small=True)
    r3.synchronize()
    assert_bits_equal(r3.image_numpy(), orc.image, "wave-per-object pixel search vs oracle")
    assert_bits_equal(real3.cpu().numpy(), real_o, "realized flux (wave per object)")


def test_batching_invariance(torch_cuda):
    """Splitting every object's photons over batches (photon_pooling.py:300-304 flux split) gives
    the same image as one shot."""
    from imsim_amd.engine import Renderer
    scene, objects, _ = small_case(n_obj=200, nx=256, ny=256, flux_seed=5)
    r1 = Renderer(scene)
    r1.render(objects)
    r2 = Renderer(scene)
    nb = 7
    F = objects["n_phot"].copy()
    for i in range(nb):
        part = objects.copy()
        lo, hi = (F * i) // nb, (F * (i + 1)) // nb
        part["phot_first"], part["n_phot"] = lo, hi - lo
        r2.render(part[part["n_phot"] > 0])
    r1.synchronize(); r2.synchronize()
    assert_bits_equal(r1.image_numpy(), r2.image_numpy(), "batched image")


def test_empty_and_ragged_inputs(torch_cuda):
    from imsim_amd.engine import Renderer
    scene, objects, _ = small_case(n_obj=50, nx=128, ny=128)
    r = Renderer(scene)
    r.render(objects[:0])
    r.synchronize()
    assert r.image_numpy().sum() == 0
    one = objects[:1].copy()
    one["n_phot"] = 1
    r.render(one)
    r.synchronize()
    assert r.image_numpy().sum() in (0.0, 1.0)


# ---------------------------------------------------------------------------------------------
# C3: full photon-op chain + Silicon sensor (tree rings, brighter-fatter)
# ---------------------------------------------------------------------------------------------
def _c3_case(n_obj=150, n=512, flux_seed=1, scratch=2_000_000, **kw):
    from imsim_amd import configs, catalog
    scene = configs.scene_c3(nx=n, ny=n, **kw)
    if scene.sensor is not None:
        scene.sensor.scratch_cells = scratch
    cat = catalog.synthetic_catalog(n_obj, nx=n, ny=n)
    phot = catalog.realize_fluxes(cat["nominal_flux"], flux_seed)
    objects, sizes = configs.c3_objects(cat, phot, scene)
    return scene, objects


class _SensorArrays(dict):
    """boundary / bounds / delta arrays of a renderer"""
    per_cell = {"boundary": None, "bounds": 8, "delta": 1}


def _sensor_arrays_gpu(r):
    out = _SensorArrays()
    for name, dt in (("boundary", np.float64), ("bounds", np.float64), ("delta", np.float64)):
        out[name] = r.bound.sensor_arrays[name].cpu().numpy().view(dt)
    return out


@pytest.mark.parametrize("chain,layout", [("0", "1"), ("1", "0")])
def test_specialised_kernels_equal_the_descriptor_driven_ones(torch_cuda, monkeypatch, chain, layout):
    """The kernels specialised for the default operator chain / analytic PSF (run_ops<1>, run_psf<1>) and for the known optics
    layout (trace<LAYOUT>) render the image, realized fluxes and sensor state of the kernels that loop over the
    descriptors, bit for bit (LSST_Image plan of a C3 case: fused launch, pool shoots, rounds)."""
    from imsim_amd.engine import Renderer
    scene, objects = _c3_case(n_obj=200)
    out = []
    for c, l in (("1", "1"), (chain, layout)):
        monkeypatch.setenv("IMS_CHAIN_KERNELS", c)
        monkeypatch.setenv("IMS_LAYOUT_KERNELS", l)
        r = Renderer(scene)
        real = torch_cuda.zeros(len(objects), dtype=torch_cuda.float64, device="cuda")
        r.render_lsst_image(objects, nrecalc=1000, realized=real)
        r.synchronize()
        out.append((r.image_numpy(), real.cpu().numpy(), _sensor_arrays_gpu(r)))
    assert out[0][0].sum() > 0
    assert_bits_equal(out[1][0], out[0][0], "image")
    assert_bits_equal(out[1][1], out[0][1], "realized flux")
    for name in ("boundary", "bounds", "delta"):
        assert_bits_equal(out[1][2][name], out[0][2][name], f"sensor {name}")


@pytest.mark.parametrize("nrecalc", [1000, 300])
def test_two_segment_round_search_gives_the_one_segment_result(torch_cuda, monkeypatch, nrecalc):
    """ims_tuning_t.round_two_segments (k_accumulate_round_c2: photons j and j + 256 of an object on one thread, both pool
    records requested before the first search) ends with the image and sensor state of the one-segment kernel -- and of the
    oracle: rounds of 1000 photons (two workgroups per object, the last one ragged) and of 300 (a partial second segment)."""
    from imsim_amd.engine import Renderer
    from oracle import orc_loader
    scene, objects = _c3_case(n_obj=200)
    out = []
    for two in ("1", "0"):
        monkeypatch.setenv("IMS_ROUND_TWO_SEGMENTS", two)
        r = Renderer(scene)
        r.render_lsst_image(objects, nrecalc=nrecalc)
        r.synchronize()
        out.append((r.image_numpy(), _sensor_arrays_gpu(r)))
    assert_bits_equal(out[0][0], out[1][0], "image")
    for name in ("boundary", "bounds", "delta"):
        assert_bits_equal(out[0][1][name], out[1][1][name], f"sensor {name}")
    orc = orc_loader.OracleScene(scene)
    orc.render_lsst_image(objects, nrecalc=nrecalc)
    assert_bits_equal(out[0][0], orc.image, "image vs oracle")


def test_lazy_static_renderer_makes_the_state_when_something_needs_it(torch_cuda):
    """lazy_static used to sit in the renderer's base parameters, so pooled accumulates and the sky's pixel areas ran on a
    slot 0 that was never initialised.  Now the flag travels with the fused render only; an entry point that reads the stored
    state makes it first (and the renderer is an ordinary one from then on): pixel areas and a pooled accumulate of ordinary rows
    equal those of a renderer that was made with the state."""
    from imsim_amd.engine import Renderer
    from imsim_amd.lsst_image import LSST_ImageBuilderBase
    scene, objects = _c3_case(n_obj=200)
    objects = objects[objects["n_phot"] <= 1000]
    out = []
    for lazy in (True, False):
        r = Renderer(scene, lazy_static=lazy)
        assert r.lazy_static == lazy
        assert int(r.bound.base_params.lazy_static) == 0
        pool = r.shoot_ops_photons(objects, converted=True)
        r.accumulate_segments(pool)
        assert not r.lazy_static
        r.synchronize()
        image = r.image_numpy()
        area = LSST_ImageBuilderBase().sky_pixel_areas(r)
        out.append((image, area.cpu().numpy()))
    assert out[0][0].sum() > 0
    assert_bits_equal(out[0][0], out[1][0], "pooled accumulate on a renderer made without the static state")
    assert_bits_equal(out[0][1], out[1][1], "pixel areas")


@pytest.mark.parametrize("native", ["1", "0"])
def test_lazy_static_state_gives_the_stored_state_image(torch_cuda, monkeypatch, native):
    """Renderer(lazy_static=True): slot 0 is never initialised; the fused launch sets the photons within pristine_margin of a
    pixel edge aside and k_margin_photons finishes them from the tree-ring closed form -- image and realized fluxes of the
    stored-state render and of the oracle, bit for bit (library planner and numpy planner).  The static state really is absent:
    its arrays are filled with a pattern first."""
    from imsim_amd.engine import Renderer
    from oracle import orc_loader
    monkeypatch.setenv("IMS_NATIVE_PLAN", native)
    scene, objects = _c3_case(n_obj=300)
    assert scene.sensor.tr_table is not None                   # tree rings: the polygons differ from pixel to pixel
    out = []
    for lazy in (True, False):
        r = Renderer(scene, lazy_static=lazy)
        assert r.lazy_static == lazy
        if lazy:
            cells = (scene.nx + 1) * (scene.ny + 1)
            for name, per in (("boundary", 20), ("bounds", 8)):
                a = r.bound.sensor_arrays[name]
                a = a if a.dtype == torch_cuda.float64 else a.view(torch_cuda.float64)
                a[:cells * per].fill_(float("nan"))                               # whoever reads slot 0 reads NaN
        real = torch_cuda.zeros(len(objects), dtype=torch_cuda.float64, device="cuda")
        plan, parts = r.plan_lsst_image(objects, nrecalc=1000, want_realized=True)
        r.execute_plan(plan)
        for index, tmp in parts:
            real.index_add_(0, index, tmp)
        r.synchronize()
        out.append((r.image_numpy(), real.cpu().numpy()))
    assert_bits_equal(out[0][0], out[1][0], "image: lazy static state vs stored")
    assert_bits_equal(out[0][1], out[1][1], "realized fluxes")
    orc = orc_loader.OracleScene(scene)
    orc.render_lsst_image(objects, nrecalc=1000)
    assert_bits_equal(out[0][0], orc.image, "image vs oracle")


def test_tiled_initial_state_equals_the_per_cell_kernel(torch_cuda, monkeypatch):
    """k_init_tiles (every owned point evaluated once per tile, neighbours through LDS) writes the boundary points, bounds
    lines and delta image of k_init_boundaries (one thread per cell, neighbours recomputed) bit for bit: the static CCD
    region and private regions of different sizes, some clipped at the CCD edge."""
    from imsim_amd.engine import Renderer
    scene, objects = _c3_case(n_obj=200, n=300)
    arrays = []
    for tiles in ("1", "0"):
        monkeypatch.setenv("IMS_INIT_TILES", tiles)
        r = Renderer(scene)
        r.render_lsst_image(objects, nrecalc=1000)               # private regions: initialised, then updated in rounds
        r.synchronize()
        arrays.append(_sensor_arrays_gpu(r))
    for name in ("boundary", "bounds", "delta"):
        assert_bits_equal(arrays[0][name], arrays[1][name], f"sensor {name}: tiled vs per-cell initial state")
    assert np.count_nonzero(arrays[0]["bounds"]) > 0


def test_c3_photon_ops_chain_is_bit_exact(torch_cuda):
    """TimeSampler, PupilAnnulusSampler, PhotonDCR, RubinDiffractionOptics (WCS chain, spider
    diffraction, ray trace), FocusDepth, Refraction: every photon field equals the oracle's bits."""
    from imsim_amd.engine import Renderer
    from oracle import orc_loader
    scene, objects = _c3_case(n_obj=120, sensor=False)
    r = Renderer(scene)
    pool = r.shoot_photons(objects)
    r.apply_ops(pool)
    r.synchronize()
    orc = orc_loader.OracleScene(scene)
    opool = orc.shoot_pool(objects)
    orc.apply_ops(opool)
    g, o = pool.to_host(), opool.to_host()
    assert np.count_nonzero(g["flux"]) > 0.9 * len(g["flux"])
    assert np.all(np.abs(g["dxdz"][g["flux"] > 0]) < 1.0)
    for f in g:
        assert_bits_equal(g[f], o[f], f"photon field {f}")


def test_c3_lsst_image_mode_is_bit_exact(torch_cuda):
    """LSST_Image semantics with the Silicon sensor: static tree-ring boundaries for ordinary
    objects, private brighter-fatter regions advanced in rounds for bright ones.  Image, realized
    flux and the pixel-boundary state all equal the oracle's."""
    from imsim_amd.engine import Renderer
    from oracle import orc_loader
    scene, objects = _c3_case(n_obj=200)
    assert (objects["n_phot"] > 10000).sum() >= 1
    r = Renderer(scene)
    real = torch_cuda.zeros(len(objects), dtype=torch_cuda.float64, device="cuda")
    r.render_lsst_image(objects, realized=real)
    r.synchronize()
    orc = orc_loader.OracleScene(scene)
    real_o = np.zeros(len(objects))
    orc.render_lsst_image(objects, realized=real_o)
    assert_bits_equal(r.image_numpy(), orc.image, "C3 image")
    assert_bits_equal(real.cpu().numpy(), real_o, "realized flux")
    ga = _sensor_arrays_gpu(r)
    for name in ("boundary", "bounds", "delta"):
        assert_bits_equal(ga[name], orc.sensor_array(name)[:len(ga[name])], f"sensor {name}")


def test_lsst_image_edge_cases_are_bit_exact(torch_cuda):
    """The LSST_Image path on a table built to sit on every boundary of the launch plan: photon counts of 0, 1, one below /
    at / above a workgroup's 256 and a round's nrecalc and their multiples (chain classes of exactly k rounds, empty last
    rounds), objects centred off the CCD on all four sides and in a corner (stamps clipped or wholly outside), a stamp wider
    than the image, eight bright objects on ONE pixel (their private regions coincide: the CCD image takes all of them, every
    region only its own charge), a FAINT bright object (no operators, no sensor, yet above nrecalc) -- image, realized fluxes
    and the pixel-boundary state equal the oracle's bit for bit; a table of nothing but empty objects renders nothing."""
    from imsim_amd.engine import Renderer
    from imsim_amd._abi import IMS_OBJ_FAINT
    from oracle import orc_loader
    n, nrecalc = 256, 1000
    scene, base = _c3_case(n_obj=120, n=n, scratch=4_000_000)
    counts = [0, 1, 63, 64, 65, 255, 256, 257, 511, 512, 513, nrecalc - 1, nrecalc, nrecalc + 1, 2 * nrecalc - 1, 2 * nrecalc,
              2 * nrecalc + 1, 5 * nrecalc, 5 * nrecalc + 255, 6 * nrecalc + 256, 7 * nrecalc, 39 * nrecalc, 40 * nrecalc,
              40 * nrecalc + 1, 41 * nrecalc]
    objects = base[:len(counts) + 16].copy()
    objects["n_phot"][:len(counts)] = counts
    k = len(counts)

    def place(i, x, y, size, n_phot):
        objects["x0"][i], objects["y0"][i], objects["n_phot"][i] = x, y, n_phot
        cx, cy = int(np.floor(x + 0.5)), int(np.floor(y + 0.5))
        objects["stamp_xmin"][i], objects["stamp_xmax"][i] = cx - size // 2, cx - size // 2 + size - 1
        objects["stamp_ymin"][i], objects["stamp_ymax"][i] = cy - size // 2, cy - size // 2 + size - 1
    place(k + 0, -6.3, 100.2, 40, 3000)            # off the left edge, stamp clipped
    place(k + 1, n + 7.8, 57.0, 40, 3000)          # off the right edge
    place(k + 2, 120.4, -3.9, 40, 2500)            # below
    place(k + 3, 33.3, n + 12.1, 40, 2500)         # above
    place(k + 4, -15.0, -15.0, 24, 1500)           # corner: the stamp misses the CCD altogether
    place(k + 5, 128.5, 128.5, 2 * n, 6000)        # a stamp wider than the image
    for j in range(8):                             # eight bright objects on one pixel
        place(k + 6 + j, 200.25, 60.75, 48, 3 * nrecalc + 17 * j)
    place(k + 14, 60.0, 200.0, 64, 12 * nrecalc)
    objects["flags"][k + 14] |= IMS_OBJ_FAINT
    place(k + 15, 90.0, 30.0, 32, 0)
    objects["phot_first"] = 0
    r = Renderer(scene)
    real = torch_cuda.zeros(len(objects), dtype=torch_cuda.float64, device="cuda")
    r.render_lsst_image(objects, nrecalc=nrecalc, realized=real)
    r.synchronize()
    orc = orc_loader.OracleScene(scene)
    real_o = np.zeros(len(objects))
    orc.render_lsst_image(objects, nrecalc=nrecalc, realized=real_o)
    assert orc.image.sum() > 0.5 * objects["n_phot"].sum()
    # (the realized flux counts what lands in the object's STAMP -- the off-CCD corner object has one, the CCD image has none of it)
    assert real_o[0] == 0 and real_o[k + 4] > 0 and real_o[k + 15] == 0 and real_o[k + 5] > 0 and real_o[k + 14] > 0
    assert_bits_equal(r.image_numpy(), orc.image, "image")
    assert_bits_equal(real.cpu().numpy(), real_o, "realized flux")
    ga = _sensor_arrays_gpu(r)
    for name in ("boundary", "bounds", "delta"):
        assert_bits_equal(ga[name], orc.sensor_array(name)[:len(ga[name])], f"sensor {name}")
    nothing = objects[[0, k + 15]].copy()
    r2 = Renderer(scene)
    r2.render_lsst_image(nothing, nrecalc=nrecalc)
    r2.synchronize()
    assert r2.image_numpy().sum() == 0


@pytest.mark.parametrize("nrecalc,scratch", [(10000, 2_000_000), (1000, 2_000_000), (1000, 90_000)])
def test_native_planner_renders_the_numpy_planned_image(torch_cuda, monkeypatch, nrecalc, scratch):
    """ims_plan_lsst_image / _bind / _upload / _run (the launch plan of a CCD built and enqueued by the library, the default)
    against the numpy planner of engine.Renderer.plan_lsst_image (IMS_NATIVE_PLAN=0, the checker): image, realized fluxes and
    the pixel-boundary state, bit for bit, on one and on several brighter-fatter groups, twice in a row on the same renderer
    (a plan is replayed; the second render's slot tables wait for the first one's launches)."""
    from imsim_amd.engine import Renderer
    scene, objects = _c3_case(n_obj=300, scratch=scratch)
    out = []
    for native in ("1", "0"):
        monkeypatch.setenv("IMS_NATIVE_PLAN", native)
        r = Renderer(scene)
        assert r.native_plan_ok(objects) == (native == "1")
        real = torch_cuda.zeros(len(objects), dtype=torch_cuda.float64, device="cuda")
        r.render_lsst_image(objects, nrecalc=nrecalc, realized=real)
        r.render_lsst_image(objects, nrecalc=nrecalc, realized=real)
        r.synchronize()
        out.append((r.image_numpy(), real.cpu().numpy(), _sensor_arrays_gpu(r)))
        del r
    assert out[0][0].sum() > 0 and out[0][1].sum() > 0
    assert_bits_equal(out[0][0], out[1][0], "image")
    assert_bits_equal(out[0][1], out[1][1], "realized flux")
    for name in ("boundary", "bounds", "delta"):
        assert_bits_equal(out[0][2][name], out[1][2][name], f"sensor {name}")
    # a prepared plan is replayed: three runs, three times the image of one
    monkeypatch.setenv("IMS_NATIVE_PLAN", "1")
    r = Renderer(scene)
    launch = r.prepared_lsst_image(objects, nrecalc=nrecalc)
    for _ in range(3):
        launch()
    r.synchronize()
    one = Renderer(scene)
    one.render_lsst_image(objects, nrecalc=nrecalc)
    one.synchronize()
    assert_bits_equal(r.image64_numpy(), 3.0 * one.image64_numpy(), "three replays")
    assert launch.photons == int(objects["n_phot"].sum())


def test_several_brighter_fatter_groups_on_a_fresh_renderer(torch_cuda):
    """A scratch capacity that holds a third of the bright objects' regions: the FIRST plan of a fresh renderer then carries
    several brighter-fatter groups, each with its own slot table, and the table of group k + 1 is rewritten on the chain
    stream while the launches of group k (several chain classes on several streams) are still queued -- it has to wait for
    all of them (engine.execute_plan).  Image and realized fluxes equal the oracle's and the one-group render's, three times
    over with fresh renderers (the failure this guards against is a race)."""
    from imsim_amd.engine import Renderer, plan_bf_groups
    from oracle import orc_loader
    nrecalc = 1000
    scene, objects = _c3_case(n_obj=300, scratch=4_000_000)
    bright = np.flatnonzero(objects["n_phot"] > nrecalc)
    cells = ((objects["stamp_xmax"] - objects["stamp_xmin"] + 2).astype(np.int64) * (objects["stamp_ymax"] - objects["stamp_ymin"] + 2))[bright]
    one = Renderer(scene)
    real1 = torch_cuda.zeros(len(objects), dtype=torch_cuda.float64, device="cuda")
    one.render_lsst_image(objects, nrecalc=nrecalc, realized=real1)
    one.synchronize()
    want, want_real = one.image_numpy(), real1.cpu().numpy()
    scene.sensor.scratch_cells = int(max(cells.max(), cells.sum() // 3)) + 1
    b = one.bound
    _, groups = plan_bf_groups(objects, nrecalc, b.n_static_slots, b.static_cells, scene.sensor.scratch_cells, b.slot_capacity)
    assert len(groups) >= 3
    rounds = (objects["n_phot"][groups[0][0]] + nrecalc - 1) // nrecalc
    assert rounds.max() >= 6 and rounds.min() < 6, "the first group should hold several chain classes"
    del one
    orc = orc_loader.OracleScene(scene)
    real_o = np.zeros(len(objects))
    orc.render_lsst_image(objects, nrecalc=nrecalc, realized=real_o)
    assert_bits_equal(want, orc.image, "one group vs oracle")
    for attempt in range(3):
        r = Renderer(scene)
        real = torch_cuda.zeros(len(objects), dtype=torch_cuda.float64, device="cuda")
        r.render_lsst_image(objects, nrecalc=nrecalc, realized=real)
        r.synchronize()
        assert_bits_equal(r.image_numpy(), want, f"image, attempt {attempt}")
        assert_bits_equal(real.cpu().numpy(), want_real, f"realized flux, attempt {attempt}")
        del r
    assert_bits_equal(want_real, real_o, "realized flux vs oracle")


@pytest.mark.parametrize("vendor", ["itl", "e2v"])
def test_sensor_models_of_4_8_and_32_vertices_on_the_gpu(torch_cuda, vendor):
    """The reference's cross-model criterion (tests/test_sensor_models.py:73-125) through the HIP path: the 1e6-photon
    Gaussian spot of its sensor-model case, drawn in LSST_Image mode with the 4-, 8- and 32-vertex pixel models (the
    kernels unrolled for 4 and 8 vertices per edge and the generic loops for 32), is bit-identical to the oracle's for every
    model, has a lower peak and a radius larger by more than 2 sigma_r than without a sensor, and the three radii agree
    within 2 sigma_r."""
    import os
    from imsim_amd import _abi, sensor as sensormod
    from imsim_amd._abi import OBJECT_DTYPE
    from imsim_amd.engine import Renderer, Scene, SensorSetup, make_slots
    from oracle import orc_loader
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    N, n = 17, 1000000

    def spot(model_name):
        sc = Scene(nx=N, ny=N, seed=1234, psf=[(_abi.IMS_PSF_GAUSSIAN, 0, 0.3, 0.0, 1.0)], ops=[])
        if model_name:
            model = sensormod.load_silicon_model(os.path.join(root, "imsim_amd", "data", "sensor_models", model_name))
            sc.sensor = SensorSetup(model=model, abs_wl=np.array([300.0, 1100.0]), abs_len=np.array([1.0, 1.0]),
                                    slots=make_slots([(1, 1, N, N)]), scratch_cells=4 * (N + 1) * (N + 1))
        obj = np.zeros(1, dtype=OBJECT_DTYPE)
        obj["obj_id"], obj["n_phot"], obj["x0"], obj["y0"], obj["flux_per_photon"] = 5, n, 9.0, 9.0, 1.0
        obj["jac"], obj["winv"] = (1, 0, 0, 1), (1 / 0.3, 0, 0, 1 / 0.3)
        obj["prof_table"], obj["sed_table"], obj["sed_wave"] = -1, -1, 600.0
        obj["stamp_xmin"], obj["stamp_xmax"], obj["stamp_ymin"], obj["stamp_ymax"] = 1, N, 1, N
        r = Renderer(sc)
        r.render_lsst_image(obj, nrecalc=10000)
        r.synchronize()
        orc = orc_loader.OracleScene(sc)
        orc.render_lsst_image(obj, nrecalc=10000)
        assert_bits_equal(r.image_numpy(), orc.image, f"spot image, {model_name}")
        if model_name:
            ga = _sensor_arrays_gpu(r)
            for name in ("boundary", "bounds"):
                assert_bits_equal(ga[name], orc.sensor_array(name)[:len(ga[name])], f"sensor {name}, {model_name}")
        return r.image_numpy().astype(float)

    def radius(img):
        yy, xx = np.mgrid[0:N, 0:N].astype(float)
        f = img.sum()
        mx, my = (img * xx).sum() / f, (img * yy).sum() / f
        return np.sqrt((img * ((xx - mx) ** 2 + (yy - my) ** 2)).sum() / f)

    none = spot(None)
    sigma_r = 1.0 / np.sqrt(float(n))
    r = {}
    for nv in (4, 8, 32):
        img = spot(f"lsst_{vendor}_50_{nv}")
        assert img.max() <= none.max()
        assert abs(img.sum() - n) < 2e-3 * n
        r[nv] = radius(img)
        assert r[nv] - radius(none) > 2 * sigma_r
    assert abs(r[8] - r[4]) < 2 * sigma_r, r
    assert abs(r[32] - r[8]) < 2 * sigma_r, r


def test_pooling_mode_brighter_fatter_is_bit_exact(torch_cuda):
    """Photon-pooling semantics (photon_pooling.py:141-160): the whole CCD is one brighter-fatter
    region, recalculated once per batch from the charge accumulated since the last recalc."""
    from imsim_amd.engine import Renderer
    from oracle import orc_loader
    scene, objects = _c3_case(n_obj=150, n=256, flux_seed=4)
    scene.track_static_delta = 1
    r = Renderer(scene)
    orc = orc_loader.OracleScene(scene)
    nb = 3
    F = objects["n_phot"].copy()
    for i in range(nb):
        part = objects.copy()
        lo, hi = (F * i) // nb, (F * (i + 1)) // nb
        part["phot_first"], part["n_phot"], part["bf_state"] = lo, hi - lo, 0
        part["flags"] = 0
        part = part[part["n_phot"] > 0]
        if i > 0:
            r.update_distortions(0, 1, bf_tag=i)          # GPU: visit only tiles in reach of batch i-1's charge
            orc.update_distortions(0, 1)                  # oracle: every cell
        pool = r.shoot_photons(part)
        r.apply_ops(pool)
        pix = r.accumulate(pool, want_pixel_index=True, bf_tag=i + 1)
        opool = orc.shoot_pool(part)
        orc.apply_ops(opool)
        opix = orc.accumulate(opool, want_pixel_index=True)
        r.synchronize()
        assert_bits_equal(pix.cpu().numpy(), opix, f"pixel indices batch {i}")
    assert_bits_equal(r.image_numpy(), orc.image, "pooled BF image")
    ga = _sensor_arrays_gpu(r)
    for name in ("boundary", "bounds", "delta"):
        assert_bits_equal(ga[name], orc.sensor_array(name)[:len(ga[name])], f"sensor {name}")


@pytest.mark.parametrize("quads", ["1", "0"])
def test_c3b_atmospheric_psf_is_bit_exact(torch_cuda, monkeypatch, quads):
    """C3b: the 6-screen AtmosphericPSF (phase-screen gradient gather, chromatic dilation, second
    kick) in front of the full op chain and the Silicon sensor.  quads: the 2 x 2 cells of the bilinear gradient as
    16-byte items (ims_atmosphere_t.screen_quads, the default) or four gathers from the plain screens."""
    from imsim_amd import configs, catalog
    from imsim_amd.engine import Renderer
    from oracle import orc_loader
    monkeypatch.setenv("IMS_SCREEN_QUADS", quads)
    monkeypatch.setenv("IMS_SCREEN_PREPASS", "0")              # the gathers where the photons are made (the pre-pass: next test)
    scene = configs.scene_c3b(nx=256, ny=256, screen_size=102.4, screen_scale=0.1)
    scene.sensor.scratch_cells = 500_000
    cat = catalog.synthetic_catalog(120, nx=256, ny=256)
    phot = catalog.realize_fluxes(cat["nominal_flux"], 2)
    objects, _ = configs.c3b_objects(cat, phot, scene)
    r = Renderer(scene)
    assert (r.bound.atm_struct.screen_quads is not None) == (quads == "1")
    pool = r.shoot_photons(objects)
    orc = orc_loader.OracleScene(scene)
    opool = orc.shoot_pool(objects)
    g, o = pool.to_host(), opool.to_host()
    assert np.ptp(g["time"]) > 20 and np.all(np.hypot(g["pupil_u"], g["pupil_v"]) <= 4.18 + 1e-9)
    for f in g:
        assert_bits_equal(g[f], o[f], f"photon field {f}")
    r.render_lsst_image(objects)
    r.synchronize()
    orc.render_lsst_image(objects)
    assert_bits_equal(r.image_numpy(), orc.image, "C3b image")


@pytest.mark.parametrize("mode,buckets", [("1", "128"), ("1", "4"), ("2", "128")])
def test_screen_prepass_gives_the_in_place_gathers(torch_cuda, monkeypatch, mode, buckets):
    """ims_screen_prepass (N1: the phase-screen gathers of all photons of a render ahead of the shooting kernels, sorted into
    eight spatial slices and arrival-time buckets so that an XCD's L2 holds the windows it looks through) leaves the image
    of the gathers done where the photons are made, which is the oracle's, bit for bit -- and its kick buffer holds exactly
    one entry per photon."""
    from imsim_amd import configs, catalog
    from imsim_amd.engine import Renderer
    from oracle import orc_loader
    scene = configs.scene_c3b(nx=256, ny=256, screen_size=102.4, screen_scale=0.1)
    scene.sensor.scratch_cells = 500_000
    cat = catalog.synthetic_catalog(160, nx=256, ny=256)
    phot = catalog.realize_fluxes(cat["nominal_flux"], 4)
    objects, _ = configs.c3b_objects(cat, phot, scene)
    out = []
    for on in (mode, "0"):
        monkeypatch.setenv("IMS_SCREEN_PREPASS", on)
        monkeypatch.setenv("IMS_SCREEN_BUCKETS", buckets)
        r = Renderer(scene)
        step = r.prepared_lsst_image(objects, nrecalc=3000)
        assert (step.prepass is not None) == (on != "0")
        for _ in range(2):                                   # replayable
            r.image.zero_()
            step()
        r.synchronize()
        out.append(r.image_numpy())
        if on != "0":
            # mode 1: every photon of the render; mode 2: those of the objects below the recalculation threshold (side stream)
            covered = objects["n_phot"] if on == "1" else objects["n_phot"][objects["n_phot"] <= 3000]
            assert step.prepass.side == (on == "2")
            entries = step.prepass.keep[2].cpu().numpy()
            assert len(entries) == int(covered.sum())
            oi, k = entries >> 32, entries & 0xFFFFFFFF
            key = oi * (1 << 32) + k
            assert len(np.unique(key)) == len(key)             # every photon exactly once
            assert np.array_equal(np.bincount(oi, minlength=len(covered)), covered)
    assert out[0].sum() > 0
    assert_bits_equal(out[0], out[1], "image: pre-pass vs in-place gathers")
    orc = orc_loader.OracleScene(scene)
    orc.render_lsst_image(objects, nrecalc=3000)
    assert_bits_equal(out[0], orc.image, "image vs oracle")


# ---------------------------------------------------------------------------------------------
# FFT branch (imsim/stamp.py:482-525)
# ---------------------------------------------------------------------------------------------
def _fft_case(nx=256):
    from imsim_amd import configs, catalog, fft_draw
    scene = configs.scene_c2(nx=nx, ny=nx)
    cat = dict(x=np.array([100.3, 60.0, 180.6, 128.5]), y=np.array([120.7, 200.2, 70.1, 40.9]), mag=np.zeros(4),
               nominal_flux=np.array([2.0e6, 5.0e6, 1.5e6, 3.0e6]), kind=np.array([0, 1, 2, 0]),
               hlr=np.array([0.0, 0.4, 0.8, 0.0]), q=np.array([1.0, 0.5, 0.8, 1.0]), pa=np.array([0.0, 30.0, 110.0, 0.0]),
               obj_id=np.arange(4))
    objects, _ = catalog.build_object_table(cat, cat["nominal_flux"].astype(np.int64), stamp_size=np.array([64, 128, 96, 48]))
    fwhm_atm, fwhm_sys = catalog.kolmogorov_gaussian_fwhm()
    kpsf = fft_draw.kolmogorov_gaussian_kpsf(fwhm_atm, fwhm_sys)
    rows, order = fft_draw.build_fft_objects(objects, cat["nominal_flux"], objects["prof_table"])
    return scene, rows, kpsf


def test_fft_branch_matches_oracle(torch_cuda):
    """k-space fill is bit-exact (deterministic elementary functions); the inverse transform
    (rocFFT vs numpy) agrees to 1e-11 of the peak; clip + Poisson noise + CCD add are exact."""
    from imsim_amd import fft_draw
    from imsim_amd.engine import Renderer
    from oracle import orc_loader
    scene, rows, kpsf = _fft_case()
    r = Renderer(scene)
    drawer = fft_draw.FftDrawer(r, kpsf, add_noise=True)
    drawer.keep_kspace = True                      # a copy of the half spectra from before the transform (which works in them)
    real = torch_cuda.zeros(len(rows), dtype=torch_cuda.float64, device="cuda")
    kbuf, rbuf = drawer.draw(rows, realized=real)
    r.synchronize()
    orc = orc_loader.OracleFft(scene, kpsf, add_noise=True)
    okbuf = orc.fill(rows)
    assert_bits_equal(kbuf.cpu().numpy(), okbuf, "k-space half spectra")
    orbuf = orc.inverse(rows, okbuf)
    g_rbuf = fft_draw.image_from_rbuf(rows, rbuf.cpu().numpy())      # the default leaves the 1 / N^2 to the reader (rbuf_raw)
    assert np.abs(g_rbuf - orbuf).max() < 1e-11 * np.abs(orbuf).max()
    # noise + add: feed the oracle the SAME real-space images the GPU produced
    oreal = np.zeros(len(rows))
    orc.finish(rows, g_rbuf, oreal)
    assert_bits_equal(r.image_numpy(), orc.image.astype(np.float32), "noisy FFT image")
    np.testing.assert_allclose(real.cpu().numpy(), oreal, rtol=1e-12)
    np.testing.assert_allclose(oreal, rows["flux"], rtol=2e-2)


def test_library_fft_inverse_equals_the_torch_front_end(torch_cuda, monkeypatch):
    """ims_fft_inverse (hipFFT plans cached inside libimsim_hip.so, SURVEY 8b `ims_fft_draw_batch`) against torch.fft.irfft2 --
    the same rocFFT behind another front end: half spectra of real images of every size of the branch, in batches, come back
    as those images (1e-12 of the peak) and equal torch's transform to rounding; a whole FFT draw gives the same CCD image
    either way (the Poisson deviates see transforms that agree to ~1e-13 of the peak)."""
    torch = torch_cuda
    lib = _abi.load()
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    for n, batch in ((32, 7), (64, 3), (256, 2), (1024, 2), (4096, 1)):
        img = torch.rand((batch, n, n), dtype=torch.float64, device="cuda", generator=g)
        spec = torch.fft.rfft2(img).contiguous()
        want = torch.fft.irfft2(spec, s=(n, n), norm="backward")
        k = spec.clone()
        out = torch.empty((batch, n, n), dtype=torch.float64, device="cuda")
        _abi.check(lib.ims_fft_inverse(k.data_ptr(), out.data_ptr(), n, batch, None), "ims_fft_inverse")
        torch.cuda.synchronize()
        assert float((out - img).abs().max()) < 1e-12 and float((out - want).abs().max()) < 1e-13
    assert lib.ims_fft_inverse(None, None, 64, 1, None) != 0 and lib.ims_fft_inverse(k.data_ptr(), out.data_ptr(), 63, 1, None) != 0
    from imsim_amd import fft_draw
    from imsim_amd.engine import Renderer
    scene, rows, kpsf = _fft_case()
    images = []
    for front in ("0", "1"):
        monkeypatch.setenv("IMS_FFT_TORCH", front)
        r = Renderer(scene)
        fft_draw.FftDrawer(r, kpsf, add_noise=True).draw(rows)
        r.synchronize()
        images.append(r.image64_numpy())
    diff = np.count_nonzero(images[0] != images[1])
    assert images[0].sum() > 0 and diff <= 1e-4 * np.count_nonzero(images[0]) + 2
    assert abs(images[0].sum() / images[1].sum() - 1.0) < 1e-6


def test_fft_inverse_left_unnormalised_for_its_readers_gives_the_same_images(torch_cuda, monkeypatch):
    """ims_fft_inverse_raw + ims_fft_params_t.rbuf_raw (the default of FftDrawer): the readers of the real-space buffer -- saturated-box
    scan, spike step, finish -- form value * 1 / (nfft * nfft) themselves, the product the scaling pass of ims_fft_inverse
    forms: the buffer times that factor, the spiked images and the noisy CCD image are the same bits with and without the
    pass (the realized fluxes the same sums), for power-of-two grids and for one that is not (1 / N^2 by division there)."""
    import math
    from imsim_amd import fft_draw, diffraction_fft as dfft
    from imsim_amd.engine import Renderer
    torch = torch_cuda
    scene, rows, kpsf = _fft_case()
    rows = rows.copy()
    rows["flux"] *= 20.0
    odd = rows[:1].copy()
    odd["nfft"] = 96                                        # 3 * 32: not a power of two
    small = rows[:1].copy()
    small["nfft"] = 32                                      # (half spectra of 32 x 17 and 96 x 49 elements: wavefronts of the
    tiny = rows[:2].copy()                                  # elementwise kernels straddle two objects there, walk_span's lane path;
    tiny["nfft"] = (6, 10)                                  # grids of 36 and 100 PIXELS: the same for the real-space kernels)
    tiny["x0"], tiny["y0"] = tiny["stamp_xmin"] + 20, tiny["stamp_ymin"] + 20
    tiny["cx"], tiny["cy"] = 2.3, 3.1
    rows = np.concatenate([tiny, small, odd, rows])
    rows = rows[np.argsort(rows["nfft"], kind="stable")]
    nf = rows["nfft"].astype(np.int64)
    rows["k_offset"] = np.concatenate([[0], np.cumsum(nf * (nf // 2 + 1))])[:-1]
    rows["r_offset"] = np.concatenate([[0], np.cumsum(nf * nf)])[:-1]
    cfg = dfft.DiffractionFFT(exptime=30.0, azimuth=math.radians(114.39), altitude=math.radians(53.16),
                              rotTelPos=math.radians(40.04), spike_length_cutoff=60)
    for spikes in (None, cfg):
        got = {}
        for raw in ("1", "0"):
            monkeypatch.setenv("IMS_FFT_RAW", raw)
            assert fft_draw.raw_inverse() == (raw == "1")
            r = Renderer(scene)
            real = torch.zeros(len(rows), dtype=torch.float64, device="cuda")
            drawer = fft_draw.FftDrawer(r, kpsf, add_noise=True, diffraction_fft=spikes, wavelength=622.2)
            drawer.keep_kspace = True
            kbuf, rbuf = drawer.draw(rows, realized=real)
            r.synchronize()
            if raw == "1":
                # the oracle on the same rows: half spectra bit for bit, and from the GPU's own transforms the same CCD image
                from oracle import orc_loader
                orc = orc_loader.OracleFft(scene, kpsf, add_noise=True, diffraction_fft=spikes, wavelength=622.2)
                assert_bits_equal(kbuf.cpu().numpy(), orc.fill(rows), "k-space half spectra (grids of 32, 96 and powers of two)")
                image = fft_draw.image_from_rbuf(rows, rbuf.cpu().numpy())
                orc.finish(rows, orc.spikes(rows, image) if spikes is not None else image, np.zeros(len(rows)))
                assert_bits_equal(r.image_numpy(), orc.image.astype(np.float32), "CCD image vs the oracle")
            got[raw] = (fft_draw.image_from_rbuf(rows, rbuf.cpu().numpy()), drawer._last[2].cpu().numpy().copy(), r.image64_numpy(),
                        real.cpu().numpy())
        assert np.abs(got["1"][0]).max() > 1e5 and got["1"][2].sum() > 0
        for k, what in enumerate(("real-space images", "images after the spike step", "CCD image")):
            if k == 1 and spikes is None:
                continue                                     # without a spike step the buffer handed to finish IS the raw one
            assert_bits_equal(got["1"][k], got["0"][k], f"{what}, raw inverse vs scaling pass (spikes {'on' if spikes else 'off'})")
        np.testing.assert_allclose(got["1"][3], got["0"][3], rtol=1e-12)       # sums of the same values by atomic adds: any order
    lib = _abi.load()
    assert lib.ims_fft_inverse_raw(None, None, 64, 1, None) != 0


def test_spike_step_in_two_launches_equals_the_one_launch_form_and_the_oracle(torch_cuda, monkeypatch):
    """ims_fft_spikes_listed (FftDrawer's default): the image streamed, the pixels near an arm listed and their sums formed by a second
    launch -- the spiked images are ims_fft_spikes's bit for bit, also when the list runs over (a capacity of 64 entries: the
    fallback launch does the whole step again) and against the oracle."""
    import math
    from imsim_amd import fft_draw, diffraction_fft as dfft
    from imsim_amd.engine import Renderer
    from oracle import orc_loader
    scene, rows, kpsf = _fft_case()
    rows = rows.copy()
    rows["flux"] *= 20.0
    cfg = dfft.DiffractionFFT(exptime=30.0, azimuth=math.radians(114.39), altitude=math.radians(53.16),
                              rotTelPos=math.radians(40.04), spike_length_cutoff=60)
    got = {}
    for name, env in (("one launch", {"IMS_FFT_SPIKE_LIST": "0"}), ("listed", {"IMS_FFT_SPIKE_LIST": "1"}),
                      ("listed, list of 64", {"IMS_FFT_SPIKE_LIST": "1", "IMS_FFT_SPIKE_LIST_CAP": "64"}),
                      ("listed, in place", {"IMS_FFT_SPIKE_LIST": "1", "IMS_SPIKE_TABLE": "0"})):
        for k in ("IMS_FFT_SPIKE_LIST", "IMS_FFT_SPIKE_LIST_CAP", "IMS_SPIKE_TABLE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        r = Renderer(scene)
        drawer = fft_draw.FftDrawer(r, kpsf, add_noise=True, diffraction_fft=cfg, wavelength=622.2)
        kbuf, rbuf = drawer.draw(rows)
        r.synchronize()
        assert (drawer._last[7] is None) == (name == "one launch")
        if drawer._last[7] is not None:
            n_listed = int(drawer._last[7][1][0].item())
            assert n_listed > 64 and (n_listed <= drawer._last[7][0].numel()) == (name != "listed, list of 64")
        got[name] = (drawer._last[2].cpu().numpy().copy(), r.image_numpy(), fft_draw.image_from_rbuf(rows, rbuf.cpu().numpy()))
    orc = orc_loader.OracleFft(scene, kpsf, add_noise=True, diffraction_fft=cfg, wavelength=622.2)
    ofinal = orc.spikes(rows, got["one launch"][2])
    for name in got:
        assert_bits_equal(got[name][0], ofinal, f"spiked images, {name} vs oracle")
        assert_bits_equal(got[name][1], got["one launch"][1], f"CCD image, {name}")
    assert np.abs(ofinal - np.clip(got["one launch"][2], 0, None)).max() > 1.0


def test_fft_and_photon_shooting_agree(torch_cuda):
    """The reference's FFT-vs-phot criteria (tests/test_psf.py:341-438: peak within 5 %, moments
    within 10 %) for a bright star and a bright Sersic galaxy through Kolmogorov (+) Gaussian."""
    from imsim_amd import _abi, configs, catalog, fft_draw
    from imsim_amd.engine import Renderer
    fwhm_atm, fwhm_sys = catalog.kolmogorov_gaussian_fwhm()
    for kind, hlr in ((0, 0.0), (1, 0.6)):
        scene = configs.scene_c2(nx=128, ny=128)
        scene.psf = [(_abi.IMS_PSF_RADIAL, 2, fwhm_atm, 0.0, 1.0), (_abi.IMS_PSF_GAUSSIAN, 0, fwhm_sys / 2.3548200450309493, 0.0, 1.0)]
        cat = dict(x=np.array([64.3]), y=np.array([63.8]), mag=np.zeros(1), nominal_flux=np.array([4.0e6]),
                   kind=np.array([kind]), hlr=np.array([hlr]), q=np.array([0.7]), pa=np.array([20.0]), obj_id=np.array([3]))
        objects, _ = catalog.build_object_table(cat, np.array([4000000]), stamp_size=128)
        rp = Renderer(scene)
        rp.render(objects)
        rf = Renderer(scene)
        rows, _ = fft_draw.build_fft_objects(objects, cat["nominal_flux"], objects["prof_table"])
        fft_draw.FftDrawer(rf, fft_draw.kolmogorov_gaussian_kpsf(fwhm_atm, fwhm_sys), add_noise=False).draw(rows)
        rp.synchronize(); rf.synchronize()
        a, b = rp.image_numpy().astype(float), rf.image_numpy().astype(float)

        def mom(img):
            yy, xx = np.mgrid[0:128, 0:128]
            w = (np.hypot(xx - 63.3, yy - 62.8) < 25)
            f = (img * w).sum()
            mx, my = (img * w * xx).sum() / f, (img * w * yy).sum() / f
            return f, mx, my, (img * w * ((xx - mx) ** 2 + (yy - my) ** 2)).sum() / f
        fa, xa, ya, ra = mom(a)
        fb, xb, yb, rb = mom(b)
        assert abs(a.max() / b.max() - 1) < 0.05
        assert abs(fa / fb - 1) < 0.01
        assert abs(xa - xb) < 0.02 and abs(ya - yb) < 0.02
        assert abs(ra / rb - 1) < 0.10


def test_fft_diffraction_spikes_are_bit_exact(torch_cuda):
    """stamp.diffraction_fft: saturated bounding box + analytic Lorentzian-cross stencil on the GPU
    equals the oracle (which is pinned to the reference's apply_diffraction_psf by golden vectors)."""
    import math
    from imsim_amd import fft_draw, diffraction_fft as dfft
    from imsim_amd.engine import Renderer
    from oracle import orc_loader
    scene, rows, kpsf = _fft_case()
    rows = rows.copy()
    rows["flux"] *= 20.0                                   # peaks well above the 1e5 threshold
    cfg = dfft.DiffractionFFT(exptime=30.0, azimuth=math.radians(114.39), altitude=math.radians(53.16),
                              rotTelPos=math.radians(40.04), spike_length_cutoff=60)
    r = Renderer(scene)
    drawer = fft_draw.FftDrawer(r, kpsf, add_noise=False, diffraction_fft=cfg, wavelength=622.2)
    kbuf, rbuf = drawer.draw(rows)
    r.synchronize()
    final = drawer._last[2].cpu().numpy()
    orc = orc_loader.OracleFft(scene, kpsf, add_noise=False, diffraction_fft=cfg, wavelength=622.2)
    image = fft_draw.image_from_rbuf(rows, rbuf.cpu().numpy())
    ofinal = orc.spikes(rows, image)
    assert_bits_equal(final, ofinal, "spiked FFT images")
    assert np.abs(final - np.clip(image, 0, None)).max() > 1.0      # spikes did something
    np.testing.assert_allclose(final.sum(), np.clip(image, 0, None).sum(), rtol=0.05)


# ---------------------------------------------------------------------------------------------
# YAML-driven end-to-end runs (imsim_amd.config.Process)
# ---------------------------------------------------------------------------------------------
def _process(**over):
    import os
    from imsim_amd import config
    here = os.path.dirname(os.path.abspath(__file__))
    o = {"input.instance_catalog.file_name": os.path.join(here, "golden", "example_instcat_subset.txt")}
    o.update(over)
    return config.Process(os.path.join(here, "data", "test-config-instcat.yaml"), template_dirs=[os.path.join(here, "data")],
                          overrides=o)


def test_config_c1_fft_no_sensor(torch_cuda):
    """BASELINE config #1 semantics: instance catalog, image.nobjects=100, stamp.draw_method=fft,
    image.sensor="" -- every object goes through the FFT branch."""
    res = _process(**{"image.nobjects": 100, "stamp.draw_method": "fft", "image.sensor": "", "stamp.photon_ops": []})
    img, truth = res.images[0], res.truth[0]
    assert res.det_names == ["R22_S11"] and img.shape == (4004, 4096)
    assert set(truth["mode"]) == {"fft"} and len(truth["mode"]) > 30
    on = truth["realized_flux"] > 0.5 * truth["nominal_flux"]
    assert on.sum() >= 0.7 * len(on)                       # objects near the CCD edge lose flux off the stamp
    # Poisson-noised total close to the drawn flux
    assert abs(img.sum() / truth["realized_flux"].sum() - 1) < 0.02
    assert "input.sky_model" in res.ignored and any(m.startswith("input.checkpoint") for m in res.ignored)


def test_config_phot_full_chain_and_pooling_agree(torch_cuda):
    """The same catalog through LSST_Image/LSST_Silicon and through LSST_PhotonPoolingImage/
    LSST_Photons: per-object fluxes agree (tests/test_image.py:162-228 style: within 4 sqrt(N))."""
    a = _process(**{"image.nobjects": 60, "stamp.draw_method": "phot"})
    b = _process(**{"image.nobjects": 60, "image.type": "LSST_PhotonPoolingImage", "stamp.type": "LSST_Photons",
                    "image.nbatch": 4, "image.nsubbatch": 3, "stamp.fft_sb_thresh": 0,       # no threshold: nothing is FFT-drawn
                    "input.checkpoint": ""})
    ta, tb = a.truth[0], b.truth[0]
    assert np.array_equal(ta["index"], tb["index"])
    assert set(ta["mode"]) <= {"phot", "faint"}
    inside = (ta["x"] > 60) & (ta["x"] < 4036) & (ta["y"] > 60) & (ta["y"] < 3944)      # photons off the CCD are lost
    bright = (ta["phot_flux"] > 500) & inside
    assert bright.sum() > 5
    # realized / incident flux vs shot photons: vignetting and stamp clipping lose a few per cent
    for t, key in ((ta, "realized_flux"), (tb, "incident_flux")):
        frac = t[key][bright] / t["phot_flux"][bright]
        assert np.all(frac > 0.85) and np.all(frac <= 1.0 + 1e-12)
    sa, sb = a.images[0].sum(), b.images[0].sum()
    assert abs(sa / sb - 1) < 0.02


def test_config_fft_photon_ops_are_refused_where_the_fft_branch_is_reachable(torch_cuda):
    """stamp.fft_photon_ops (imsim/stamp.py:493-499): the reference re-shoots the FFT image through these operators; that stage is
    not built here, so a config that can reach the FFT branch with the key set is an error -- not a render that silently leaves the
    stage out (it used to be listed in res.ignored).  With draw_method: phot nothing is FFT-drawn and the key is moot."""
    from imsim_amd.lsst_image import GalSimConfigError
    ops = [{"type": "TimeSampler", "t0": 0.0, "exptime": 30.0}]
    with pytest.raises(GalSimConfigError, match="fft_photon_ops"):
        _process(**{"image.nobjects": 5, "stamp.fft_photon_ops": ops})
    a = _process(**{"image.nobjects": 5, "stamp.draw_method": "phot", "stamp.fft_photon_ops": ops})
    assert not any("fft_photon_ops" in note for note in a.ignored)
    assert a.images[0].sum() > 0


def test_config_several_ccds_take_the_overlapped_focal_plane_path(torch_cuda, monkeypatch, tmp_path):
    """`output.nfiles: 3` through config.Process: the CCDs are prepared on the host while the previous ones run
    (focal_plane.render_focal_plane, the fan-out of imsim/ccd.py:72-89), with FFT-drawn objects, sky noise and an e-image file
    per CCD -- and give the images, truth records and files of rendering them one after the other (IMS_PROCESS_FOCAL=0)."""
    over = {"image.nobjects": 40, "output.nfiles": 3, "stamp.fft_sb_thresh": 2.0e3, "image.sky_level": 800.0,
            "image.noise": {"type": "CCD"}, "output.dir": str(tmp_path),
            "output.file_name": {"type": "FormattedStr", "format": "eimage_%s.fits", "items": ["$det_name"]}}
    a = _process(**over)
    monkeypatch.setenv("IMS_PROCESS_FOCAL", "0")
    b = _process(**dict(over, **{"output.dir": str(tmp_path / "serial")}))
    assert a.det_names == b.det_names and len(a.images) == 3 and len(set(a.det_names)) == 3
    assert len(a.files) == 3 and len(set(a.files)) == 3
    for k in range(3):
        assert_bits_equal(a.images[k], b.images[k], f"CCD {a.det_names[k]}")
        for key in ("index", "nominal_flux", "phot_flux", "fft_flux"):
            assert_bits_equal(np.asarray(a.truth[k][key]), np.asarray(b.truth[k][key]), f"truth {key} of CCD {a.det_names[k]}")
        assert list(a.truth[k]["mode"]) == list(b.truth[k]["mode"])
        # photon-shot objects: integer sums, exact; FFT-drawn ones: a sum of non-integer pixel values in the order of the atomics
        fft = np.asarray(a.truth[k]["mode"]) == "fft"
        ra, rb = np.asarray(a.truth[k]["realized_flux"]), np.asarray(b.truth[k]["realized_flux"])
        assert_bits_equal(ra[~fft], rb[~fft], f"realized flux of the photon-shot objects of CCD {a.det_names[k]}")
        np.testing.assert_allclose(ra[fft], rb[fft], rtol=1e-12)
    assert any("fft" in list(t["mode"]) for t in a.truth)
    assert not np.array_equal(a.images[0], a.images[1])


# ---------------------------------------------------------------------------------------------
# LSST_Flat (imsim/flat.py, area branch)
# ---------------------------------------------------------------------------------------------
def test_flat_matches_oracle_bit_for_bit(torch_cuda):
    """pixel areas of the live boundary state, Poisson realisation and the feedback through
    updatePixelDistortions: GPU and oracle agree on every pixel and every boundary point"""
    from imsim_amd import configs, flat, treerings
    from imsim_amd.engine import Renderer
    from oracle import orc_loader
    tr = treerings.simple_treerings(0.26, 87.0, dr=0.87)
    kw = dict(sensor=True, treering=tr, treering_center=(-100.0, -100.0), seed=77)
    scene = configs.scene_flat(96, 80, **kw)
    r = Renderer(scene)
    b = flat.LSST_FlatBuilder()
    b.setup({"counts_per_pixel": 9000, "max_counts_per_iter": 3000, "xsize": 96, "ysize": 80})
    img = b.build_image(r, seed=77).cpu().numpy()
    orc = orc_loader.OracleScene(configs.scene_flat(96, 80, **kw))
    oimg = orc.build_flat(9000.0, 3000.0, seed=77)
    assert img.shape == (80, 96) and abs(img.mean() / 9000.0 - 1) < 0.01
    assert_bits_equal(img, oimg, "flat image")
    ga = _sensor_arrays_gpu(r)
    for name in ("boundary", "bounds"):
        assert_bits_equal(ga[name], orc.sensor_array(name), f"flat sensor {name}")


def test_full_ccd_silicon_flat_statistics(torch_cuda):
    """tests/test_flats.py:62-111 on a whole 4096 x 4004 CCD (sampling noise of a covariance: N / 4050 = 20 e-^2):
    variance below the mean, cov10 > cov01 > cov11 > 0 at the reference's thresholds"""
    from imsim_amd import configs, flat
    from imsim_amd.engine import Renderer
    tot = 80_000.0
    r = Renderer(configs.scene_flat(4096, 4004, sensor=True))
    b = flat.LSST_FlatBuilder()
    b.setup({"counts_per_pixel": tot, "max_counts_per_iter": 4000, "xsize": 4096, "ysize": 4004})
    img = b.build_image(r, seed=1234).cpu().numpy()
    a = img - img.mean()
    cov10 = np.mean(a[1:, :] * a[:-1, :])
    cov01 = np.mean(a[:, 1:] * a[:, :-1])
    cov11 = np.mean(a[1:, 1:] * a[:-1, :-1])
    np.testing.assert_allclose(img.mean(), tot, rtol=1e-3)
    assert 0.85 * tot < img.var() < tot
    assert cov10 > 1e-2 * tot and cov01 > 3e-3 * tot and cov11 > 2e-3 * tot
    assert cov10 > cov01 > cov11
    # left-right and up-down symmetric
    cov1m1 = np.mean(a[1:, :-1] * a[:-1, 1:])
    assert abs(cov11 - cov1m1) < 100.0


def test_config_lsst_flat(torch_cuda):
    """image.type: LSST_Flat through the config driver (tests/test_flats.py:28-44 style dict config)"""
    from imsim_amd import config
    res = config.Process({"image": {"type": "LSST_Flat", "random_seed": 1234, "xsize": 128, "ysize": 96,
                                    "counts_per_pixel": 20000, "max_counts_per_iter": 5000,
                                    "sensor": {"type": "Silicon"}}})
    img = res.images[0]
    assert img.shape == (96, 128) and img.dtype == np.float32
    assert res.truth[0]["niter"] == 4
    np.testing.assert_allclose(img.mean(), 20000.0, rtol=1e-2)
    assert 0.8 * 20000 < img.var() < 1.05 * 20000


def test_atm_psf_fft_matches_photon_shooting(torch_cuda):
    """tests/test_psf.py:341-438 (test_atm_psf_fft): bright stars drawn by FFT with make_fft_psf's stand-ins
    (VonKarman for the phase screens, Airy for the second kick) resemble the photon-shot AtmosphericPSF image:
    peak within 5 % (7 % here, see below), moment radius within 10 %.  The FFT PSF is the EXPECTATION of the
    photon-shot one; nine stars spread over the field average the 30-s realisation.  The second kick carries the
    Kolmogorov spectrum above kcrit/r0 (as galsim.SecondKick does), the VonKarman stand-in slightly less for this
    visit's outer scale of 18 m, which leaves the photon-shot peak about 5 % lower."""
    from imsim_amd import configs, fft_draw
    from imsim_amd.engine import Renderer
    n, half = 1024, 40
    scene = configs.scene_c3b(nx=n, ny=n, sensor=False, screen_size=409.6, screen_scale=0.1,
                              device=torch_cuda.device("cuda", 0))
    scene.ops = []
    gx, gy = np.meshgrid([170.3, 512.6, 853.8], [171.7, 511.2, 852.4])
    k = gx.size
    cat = dict(x=gx.ravel(), y=gy.ravel(), mag=np.zeros(k), nominal_flux=np.full(k, 2.0e6), kind=np.zeros(k, dtype=int),
               hlr=np.zeros(k), q=np.ones(k), pa=np.zeros(k), obj_id=np.arange(k) + 3)
    objects, _ = configs.c3b_objects(cat, np.full(k, 2000000), scene)
    cx, cy = np.floor(cat["x"] + 0.5).astype(int), np.floor(cat["y"] + 0.5).astype(int)
    objects["stamp_xmin"], objects["stamp_xmax"] = cx - half, cx + half - 1
    objects["stamp_ymin"], objects["stamp_ymax"] = cy - half, cy + half - 1
    rp = Renderer(scene)
    rp.render(objects)
    rf = Renderer(scene)
    atm = scene.atm
    kpsf, extra = fft_draw.atmospheric_fft_kpsf(atm, atm.wlen_eff, first_table=2, fwhm_sys=0.3)
    rows, _ = fft_draw.build_fft_objects(objects, cat["nominal_flux"], objects["prof_table"])
    fft_draw.FftDrawer(rf, kpsf, add_noise=False, extra_ktables=extra).draw(rows)
    rp.synchronize(); rf.synchronize()
    a, b = rp.image_numpy().astype(float), rf.image_numpy().astype(float)

    def stats(img):
        out = []
        for x0, y0 in zip(cx, cy):
            st = img[y0 - 1 - 30:y0 - 1 + 30, x0 - 1 - 30:x0 - 1 + 30]
            yy, xx = np.mgrid[0:60, 0:60]
            f = st.sum()
            mx, my = (st * xx).sum() / f, (st * yy).sum() / f
            out.append((f, st.max(), np.sqrt((st * ((xx - mx) ** 2 + (yy - my) ** 2)).sum() / f)))
        return np.array(out)
    sa, sb = stats(a), stats(b)
    assert abs(sa[:, 0].sum() / sb[:, 0].sum() - 1) < 0.03
    assert abs(sa[:, 1].mean() / sb[:, 1].mean() - 1) < 0.07
    assert abs(sa[:, 2].mean() / sb[:, 2].mean() - 1) < 0.10


@pytest.mark.parametrize("interpolant", ["quintic", "nearest"])
def test_knots_and_streak_profiles_are_bit_exact(torch_cuda, interpolant):
    """galsim.RandomKnots, galsim.Box (streak) and FITS-stamp (InterpolatedImage) objects (imsim/instcat.py:487-561):
    photon pool and image equal the oracle's (whose distributions tests/test_profiles.py checks).  With the Quintic
    x-interpolant (GalSim's default) the photons of a FITS stamp carry +- (integral |K|)^2: every photon field is still
    identical, the image is then a sum of non-integers whose last bits depend on the order of the atomic adds."""
    from imsim_amd import catalog, configs
    from imsim_amd.engine import Renderer
    from oracle import orc_loader
    n = 60
    rng = np.random.default_rng(8)
    kind = np.where(np.arange(n) % 4 == 0, catalog.KIND_KNOTS, np.where(np.arange(n) % 4 == 1, catalog.KIND_STREAK,
                    np.where(np.arange(n) % 4 == 2, catalog.KIND_IMAGE, 1))).astype(np.int32)
    images = [rng.uniform(0, 1, size=(17, 23)) ** 3, np.clip(rng.normal(0.2, 1.0, size=(40, 31)), 0, None)]   # FITS-stamp profiles
    cat = dict(image_index=rng.integers(0, 2, n), image_scale=rng.uniform(0.05, 0.3, n), image_extent=np.full(n, 8.0),
               x=rng.uniform(40, 216, n), y=rng.uniform(40, 216, n), mag=np.zeros(n), nominal_flux=np.full(n, 3000.0), kind=kind,
               hlr=rng.uniform(0.2, 0.8, n), q=rng.uniform(0.3, 1.0, n), pa=rng.uniform(0, 180, n), obj_id=np.arange(n) + 100,
               n_knots=np.where(kind == catalog.KIND_KNOTS, rng.integers(1, 30, n), 0).astype(float),
               box_length=np.where(kind == catalog.KIND_STREAK, rng.uniform(2, 15, n), 0.0),
               box_width=np.where(kind == catalog.KIND_STREAK, rng.uniform(0.2, 1.0, n), 0.0))
    scene = configs.scene_c2(nx=256, ny=256)
    from imsim_amd import config
    scene.psf = [config.double_gaussian_psf(0.7)[0]]          # and the DoubleGaussianPSF mixture as the PSF
    scene.image_profiles = images
    scene.image_interpolant = interpolant
    objects, _ = catalog.build_object_table(cat, rng.integers(500, 4000, n))
    assert (objects["prof_table"] == -4).sum() == n // 4
    r = Renderer(scene)
    pool = r.shoot_photons(objects)
    r.accumulate(pool)
    r.synchronize()
    orc = orc_loader.OracleScene(scene)
    opool = orc.shoot_pool(objects)
    orc.accumulate(opool)
    g, o = pool.to_host(), opool.to_host()
    for f in ("x", "y", "wavelength", "flux"):
        assert_bits_equal(g[f], o[f], f"photon field {f}")
    r2 = Renderer(scene)
    r2.render(objects)
    r2.synchronize()
    if interpolant == "nearest":
        assert set(np.unique(g["flux"])) <= {0.0, 1.0}
        assert_bits_equal(r.image_numpy(), orc.image, "knots/streak image")
        assert_bits_equal(r2.image_numpy(), orc.image, "fused knots/streak image")
    else:
        assert (g["flux"] < 0).any() and (np.abs(g["flux"][g["flux"] != 0]) != 1.0).any()
        np.testing.assert_allclose(r.image_numpy(), orc.image, rtol=0, atol=1e-4)
        np.testing.assert_allclose(r2.image_numpy(), orc.image, rtol=0, atol=1e-4)


def test_sky_background_and_noise(torch_cuda):
    """addNoise (imsim/lsst_image.py:128-200): Poisson sky with a linear gradient and a multiplier map, bit-exact vs
    the oracle's Poisson deviates; mean and variance equal the expectation."""
    from imsim_amd import configs, lsst_image
    from imsim_amd.engine import Renderer
    from oracle import orc_loader
    scene = configs.scene_c2(nx=192, ny=160)
    r = Renderer(scene)
    b = lsst_image.LSST_ImageBuilder()
    mult = 1.0 + 0.1 * np.cos(np.arange(160)[:, None] / 20.0) * np.ones((160, 192))
    grad = (0.95, 0.0005, -0.0002)
    b.add_noise(r, sky_level=25000.0, sky_gradient=grad, multiplier=mult, seed=99)
    r.synchronize()
    img = r.image.cpu().numpy()
    xx, yy = np.meshgrid(np.arange(192.0), np.arange(160.0))
    expect = 25000.0 * 0.04 * (grad[0] + grad[1] * xx + grad[2] * yy) * mult
    assert abs((img - expect).mean()) < 4 * np.sqrt(expect.mean() / img.size)
    np.testing.assert_allclose(((img - expect) ** 2 / expect).mean(), 1.0, rtol=0.03)
    orc = orc_loader.OracleScene(scene)
    base = np.ascontiguousarray((grad[0] + grad[1] * xx + grad[2] * yy) * mult)
    orc.lib.orc_flat_add(None, base.ctypes.data, 25000.0 * 0.2 * 0.2, 1.0, 99, lsst_image.NOISE_STREAM, 192, 160,
                         orc.image64.ctypes.data, None)
    assert_bits_equal(img, orc.image64, "sky noise")


def test_sky_follows_the_pixel_areas_of_the_silicon_sensor(torch_cuda):
    """`sensor.calculate_pixel_areas` under the sky (image.use_flux_sky_areas, config/imsim-config.yaml:222-228): with a
    Silicon sensor the sky expectation of a pixel is proportional to its polygon area -- tree rings by default (areas equal
    to the oracle's, and the sky image to the oracle's deviates of that expectation), and with use_flux the one-step
    brighter-fatter distortion from the flux already drawn: the pixels under a bright star shrink, and the boundary state is
    the static one again afterwards."""
    from imsim_amd import lsst_image
    from imsim_amd.engine import Renderer
    from oracle import orc_loader
    scene, objects = _c3_case(n_obj=60)
    r = Renderer(scene)
    b = lsst_image.LSST_ImageBuilder()
    static = _sensor_arrays_gpu(r)["boundary"].copy()
    areas = b.sky_pixel_areas(r)
    orc = orc_loader.OracleScene(scene)
    n = scene.nx * scene.ny
    want = np.empty(n)
    acc = np.zeros(1, dtype=np.int64)
    orc.lib.orc_sensor_pixel_areas(orc.bound.sensor_dev_ptr, 0, want.ctypes.data, acc.ctypes.data)
    assert_bits_equal(areas.cpu().numpy().ravel(), want, "tree-ring pixel areas")
    assert 1e-5 < want.std() < 1e-2 and abs(want.mean() - 1.0) < 1e-3
    b.add_noise(r, sky_level=20000.0, seed=5, pixel_areas=areas)
    r.synchronize()
    base = np.ascontiguousarray(want.reshape(scene.ny, scene.nx))
    orc.lib.orc_flat_add(None, base.ctypes.data, 20000.0 * 0.2 * 0.2, 1.0, 5, lsst_image.NOISE_STREAM, scene.nx, scene.ny,
                         orc.image64.ctypes.data, None)
    assert_bits_equal(r.image.cpu().numpy(), orc.image64, "sky on tree-ring pixel areas")
    # use_flux: draw the objects first, then the areas see their charge
    r2 = Renderer(scene)
    bright = objects.copy()
    k = int(np.argmax(bright["prof_table"] == -1))               # a point source made bright
    bright["n_phot"][k] = 800000
    r2.render_lsst_image(bright)
    flux_areas = b.sky_pixel_areas(r2, use_flux=True).cpu().numpy()
    img = r2.image.cpu().numpy()
    peak = np.unravel_index(np.argmax(img), img.shape)
    assert img[peak] > 1e4
    assert flux_areas[peak] < want.reshape(scene.ny, scene.nx)[peak] - 2e-4          # a charged pixel repels: it gets smaller
    assert abs(flux_areas.mean() - 1.0) < 1e-3
    r2.synchronize()
    n0 = (scene.nx + 1) * (scene.ny + 1) * 2 * scene.sensor.owned_points()        # the owner cells of slot 0
    assert_bits_equal(_sensor_arrays_gpu(r2)["boundary"][:n0], static[:n0], "slot 0 back to its static state")


def test_photon_pooling_build_image_is_bit_exact(torch_cuda):
    """C4 semantics end to end (imsim/photon_pooling.py:116-168): nbatch photon batches, nsubbatch object sub-batches,
    one pixel-boundary recalculation per batch (tile-tagged on the GPU) -- image and boundaries equal the oracle's."""
    from helpers import c3_small_case
    from imsim_amd import photon_pooling, stamp
    from imsim_amd.engine import Renderer
    from oracle import orc_loader
    scene, objects = c3_small_case(n_obj=90, n=192, flux_seed=6, scratch=0)
    scene.track_static_delta = 1
    modes = stamp.classify(objects["n_phot"].astype(float), 100.0)
    r = Renderer(scene)
    real = torch_cuda.zeros(len(objects), dtype=torch_cuda.float64, device="cuda")
    n_gpu = photon_pooling.build_image(r, objects, modes, nbatch=5, nsubbatch=4, seed=21, realized=real)
    r.synchronize()
    orc = orc_loader.OracleScene(scene)
    n_cpu = photon_pooling.build_image(orc, objects, modes, nbatch=5, nsubbatch=4, seed=21)
    assert n_gpu == n_cpu == int(objects["n_phot"].sum())
    assert_bits_equal(r.image_numpy(), orc.image, "pooling image")
    ga = _sensor_arrays_gpu(r)
    for name in ("boundary", "bounds", "delta"):
        assert_bits_equal(ga[name], orc.sensor_array(name), f"pooling sensor {name}")
    assert abs(real.sum().item() / r.image.sum().item() - 1) < 1e-12
    # the replayable form (one fused launch per batch, bench config c4) gives the same image and sensor state
    r2 = Renderer(scene)
    run = photon_pooling.prepared_image(r2, objects, modes, nbatch=5, seed=21)
    run()
    r2.synchronize()
    assert run.photons == n_gpu
    assert_bits_equal(r2.image_numpy(), orc.image, "prepared pooling image")
    assert_bits_equal(_sensor_arrays_gpu(r2)["boundary"], orc.sensor_array("boundary"), "prepared pooling boundaries")
    r2.image.zero_()
    run()                                                        # replay: a fresh CCD
    r2.synchronize()
    assert_bits_equal(r2.image_numpy(), orc.image, "replayed pooling image")


def test_pooling_deposits_into_the_delta_image_only_give_the_same_image(torch_cuda, monkeypatch):
    """IMS_POOL_DELTA_ONLY (round 6): a batch's photons add to the delta-charge image of slot 0 alone (track_static_delta 2) and the
    image takes the charge when the next recalculation consumes it -- Silicon's `target += delta` -- the last batch's by
    Renderer.fold_delta().  Image, realized fluxes and pixel boundaries of the two-atomics form and of the oracle, bit for bit;
    from a host table and from a device table; replayed.  With a checkpoint hook (which reads the image between the batches) the
    form is not taken."""
    from helpers import c3_small_case
    from imsim_amd import photon_pooling, stamp
    from imsim_amd.engine import Renderer
    from oracle import orc_loader
    scene, objects = c3_small_case(n_obj=90, n=192, flux_seed=6, scratch=0)
    scene.track_static_delta = 1
    modes = stamp.classify(objects["n_phot"].astype(float), 100.0)
    orc = orc_loader.OracleScene(scene)
    photon_pooling.build_image(orc, objects, modes, nbatch=5, nsubbatch=4, seed=21)
    out = []
    for delta_only in ("1", "0"):
        monkeypatch.setenv("IMS_POOL_DELTA_ONLY", delta_only)
        r = Renderer(scene)
        real = torch_cuda.zeros(len(objects), dtype=torch_cuda.float64, device="cuda")
        run = photon_pooling.prepared_image(r, objects, modes, nbatch=5, seed=21, realized=real)
        for _ in range(2):
            r.image.zero_()
            real.zero_()
            run()
            r.synchronize()
        pending = float(r.delta_tensor(0).abs().sum().item())
        assert (pending == 0.0) == (delta_only == "1")          # folded into the image at the end / the last batch's charge as GalSim leaves it
        out.append((r.image_numpy(), real.cpu().numpy(), _sensor_arrays_gpu(r)["boundary"]))
    assert_bits_equal(out[0][0], out[1][0], "image: deposits into the delta image only vs into both")
    assert_bits_equal(out[0][1], out[1][1], "realized fluxes")
    assert_bits_equal(out[0][2], out[1][2], "pixel boundaries")
    assert_bits_equal(out[0][0], orc.image, "image vs oracle")
    monkeypatch.setenv("IMS_POOL_DELTA_ONLY", "1")
    seen = []
    r = Renderer(scene)
    run = photon_pooling.prepared_image(r, objects, modes, nbatch=5, seed=21, after_batch=lambda i: seen.append(float(r.image.sum().item())))
    run()
    r.synchronize()
    assert len(seen) == 5 and all(b > a for a, b in zip(seen[:-1], seen[1:]))          # the image grows batch by batch: both atomics
    assert seen[-1] == float(r.image.sum().item())
    assert_bits_equal(r.image_numpy(), orc.image, "image with a checkpoint hook")


def test_pool_shot_batch_by_batch_gives_the_one_launch_image(torch_cuda, monkeypatch):
    """IMS_POOL_OVERLAP=1 (Renderer._overlapped_pool): the objects whose share of a batch fills wavefronts are shot by share on a
    stream of their own, an event per batch, the others whole and first -- the same pool, so the image, the realized fluxes
    and the pixel-boundary state of the one-launch form and of the oracle; replayed, and from a device table as well."""
    from helpers import c3_small_case
    from imsim_amd import photon_pooling, stamp
    from imsim_amd.engine import Renderer
    from oracle import orc_loader
    scene, objects = c3_small_case(n_obj=90, n=192, flux_seed=6, scratch=0)
    scene.track_static_delta = 1
    modes = stamp.classify(objects["n_phot"].astype(float), 100.0)
    orc = orc_loader.OracleScene(scene)
    photon_pooling.build_image(orc, objects, modes, nbatch=5, nsubbatch=4, seed=21)
    monkeypatch.setenv("IMS_POOL_SMALL_MAX", "8")                 # objects of more than 40 photons go by share: most of this catalog
    out = []
    for overlap in ("1", "0"):
        monkeypatch.setenv("IMS_POOL_OVERLAP", overlap)
        r = Renderer(scene)
        real = torch_cuda.zeros(len(objects), dtype=torch_cuda.float64, device="cuda")
        run = photon_pooling.prepared_image(r, objects, modes, nbatch=5, seed=21, realized=real)
        assert (run.timed[2][0] > 1) == (overlap == "1")          # several pool-shoot launches, or the one
        for _ in range(2):                                        # the second time: a replay over the same pool
            r.image.zero_()
            real.zero_()
            run()
            r.synchronize()
        out.append((r.image_numpy(), real.cpu().numpy(), _sensor_arrays_gpu(r)["boundary"]))
    assert_bits_equal(out[0][0], out[1][0], "image: pool shot by share vs in one launch")
    assert_bits_equal(out[0][1], out[1][1], "realized fluxes")
    assert_bits_equal(out[0][2], out[1][2], "pixel boundaries")
    assert_bits_equal(out[0][0], orc.image, "image vs oracle")


def test_photon_pooling_edge_cases_are_bit_exact(torch_cuda):
    """Photon pooling on a table that sits on the boundaries of the batch arithmetic: photon counts of 0, 1, nbatch - 1
    (demoted to one random batch), nbatch, nbatch + 1, around nbatch x 64 (the split between the two pixel-search launches
    of the resident form), shares that do not divide, objects off the CCD, more batches than some objects have photons --
    the sub-batch loop and the HBM-resident replay give the oracle's image and sensor state bit for bit, and every photon
    is accounted for."""
    from helpers import c3_small_case
    from imsim_amd import photon_pooling, stamp
    from imsim_amd.engine import Renderer
    from oracle import orc_loader
    nbatch = 7
    scene, objects = c3_small_case(n_obj=80, n=160, flux_seed=9, scratch=0)
    scene.track_static_delta = 1
    counts = [0, 1, nbatch - 1, nbatch, nbatch + 1, 2 * nbatch - 1, 64 * nbatch - 1, 64 * nbatch, 64 * nbatch + 1, 65 * nbatch + 3,
              99, 100, 101, 12345, 7 * 1000, 7 * 1000 + 6, 0, 3]
    objects = objects[:len(counts) + 3].copy()
    objects["n_phot"][:len(counts)] = counts
    k = len(counts)
    for i, (x, y) in enumerate(((-9.5, 40.0), (80.0, 171.3), (165.2, 165.9))):          # off the CCD: left, above, corner
        cx, cy = int(np.floor(x + 0.5)), int(np.floor(y + 0.5))
        objects["x0"][k + i], objects["y0"][k + i], objects["n_phot"][k + i] = x, y, 900 + i
        objects["stamp_xmin"][k + i], objects["stamp_xmax"][k + i] = cx - 16, cx + 15
        objects["stamp_ymin"][k + i], objects["stamp_ymax"][k + i] = cy - 16, cy + 15
    objects["phot_first"] = 0
    modes = stamp.classify(objects["n_phot"].astype(float), 100.0)
    orc = orc_loader.OracleScene(scene)
    n_cpu = photon_pooling.build_image(orc, objects, modes, nbatch=nbatch, nsubbatch=3, seed=5)
    assert n_cpu == int(objects["n_phot"].sum()) and orc.image.sum() > 0
    r = Renderer(scene)
    n_gpu = photon_pooling.build_image(r, objects, modes, nbatch=nbatch, nsubbatch=3, seed=5)
    r.synchronize()
    assert n_gpu == n_cpu
    assert_bits_equal(r.image_numpy(), orc.image, "pooling image")
    ga = _sensor_arrays_gpu(r)
    for name in ("boundary", "bounds", "delta"):
        assert_bits_equal(ga[name], orc.sensor_array(name), f"pooling sensor {name}")
    r2 = Renderer(scene)
    run = photon_pooling.prepared_image(r2, objects, modes, nbatch=nbatch, seed=5)
    run()
    r2.synchronize()
    assert run.photons == n_cpu
    assert_bits_equal(r2.image_numpy(), orc.image, "resident replay image")
    assert_bits_equal(_sensor_arrays_gpu(r2)["boundary"], orc.sensor_array("boundary"), "resident replay boundaries")


def test_resumed_pooling_ccd_restores_image_and_realized_fluxes(torch_cuda, tmp_path):
    """A photon-pooling CCD interrupted after its second batch and resumed (HBM-resident form): the record carries the image,
    the finished batches and the realized fluxes up to them, so the resumed run ends with the image AND the per-object fluxes
    of the uninterrupted one (no sensor: nothing depends on when a batch runs)."""
    from helpers import c3_small_case
    from imsim_amd import photon_pooling, stamp
    from imsim_amd.checkpoint import Checkpointer
    from imsim_amd.engine import Renderer
    scene, objects = c3_small_case(n_obj=120, n=128, flux_seed=8, sensor=False)
    modes = stamp.classify(objects["n_phot"].astype(float), 100.0)
    full = Renderer(scene)
    real_full = torch_cuda.zeros(len(objects), dtype=torch_cuda.float64, device="cuda")
    photon_pooling.build_image(full, objects, modes, nbatch=5, seed=3, realized=real_full)
    full.synchronize()

    class Interrupt(Exception):
        pass

    class Dying(Checkpointer):
        def save(self, name, data):
            super().save(name, data)
            if data[1] == 2:
                raise Interrupt()
    first = Renderer(scene)
    real = torch_cuda.zeros(len(objects), dtype=torch_cuda.float64, device="cuda")
    with pytest.raises(Interrupt):
        photon_pooling.build_image(first, objects, modes, nbatch=5, seed=3, realized=real, checkpoint=Dying("c.hdf", dir=str(tmp_path)))
    resumed = Renderer(scene)
    real2 = torch_cuda.zeros(len(objects), dtype=torch_cuda.float64, device="cuda")
    chk = Checkpointer("c.hdf", dir=str(tmp_path))
    photon_pooling.build_image(resumed, objects, modes, nbatch=5, seed=3, realized=real2, checkpoint=chk)
    resumed.synchronize()
    assert chk.load("buildImage_photonpooling")[1] == 5
    assert 0 < real.sum().item() < real_full.sum().item()
    assert_bits_equal(resumed.image_numpy(), full.image_numpy(), "resumed image")
    assert_bits_equal(real2.cpu().numpy(), real_full.cpu().numpy(), "resumed realized fluxes")


def test_photon_flat_is_bit_exact_and_shows_brighter_fatter(torch_cuda):
    """LSST_Flat, sed branch (imsim/flat.py:237-262): small case bit-exact vs the oracle; reference-sized case
    (256^2, 80 000 e-/px in 20 iterations) has variance below the mean and positive neighbour covariances."""
    from imsim_amd import configs, flat
    from imsim_amd.engine import Renderer
    from oracle import orc_loader
    b = flat.LSST_FlatBuilder()
    cfg = {"counts_per_pixel": 2400.0, "max_counts_per_iter": 800, "xsize": 48, "ysize": 40, "buffer_size": 5}
    b.setup(cfg)
    scene = configs.scene_flat(48, 40, sensor=True, seed=3)
    scene.track_static_delta = 1
    r = Renderer(scene)
    img = b.build_image_photons(r, seed=9).cpu().numpy()
    orc = orc_loader.OracleScene(scene)
    b.build_image_photons(orc, seed=9)
    assert_bits_equal(img, flat.crop(scene, orc.image64), "photon flat")
    assert_bits_equal(_sensor_arrays_gpu(r)["boundary"], orc.sensor_array("boundary"), "photon flat boundaries")
    tot = 80_000.0
    b.setup({"counts_per_pixel": tot, "max_counts_per_iter": 4000, "xsize": 256, "ysize": 256})
    scene = configs.scene_flat(256, 256, sensor=True)
    scene.track_static_delta = 1
    r = Renderer(scene)
    img = b.build_image_photons(r, seed=1234).cpu().numpy()
    a = img - img.mean()
    cov10 = np.mean(a[1:, :] * a[:-1, :])
    np.testing.assert_allclose(img.mean(), tot, rtol=1e-2)
    assert 0.85 * tot < img.var() < tot
    assert cov10 > 1e-2 * tot


def test_gpu_reproduces_the_frozen_spec_digests(torch_cuda):
    """the same digests (tests/golden/pipeline_golden.json) from the HIP path alone, without the oracle in the loop"""
    import json, os, sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_pipeline_golden as g
    from imsim_amd.engine import Renderer

    class B(Renderer):
        def image64_host(self):
            self.synchronize()
            return self.image.cpu().numpy()

    want = json.load(open(os.path.join(here, "golden", "pipeline_golden.json")))
    got = {k: g.digest(v) for k, v in g.cases(B).items()}
    assert got == want["sha256"]


def test_focal_plane_ccds_on_streams(torch_cuda):
    """BASELINE config C5 at test size: several CCDs of one visit (different seeds, tree-ring detectors and
    sizes), two in flight at a time on their own streams.  Every CCD equals its stand-alone render, one of
    them is checked against the oracle, and the rank split deals each CCD to exactly one rank."""
    from imsim_amd import focal_plane, configs
    from imsim_amd.engine import Renderer
    from oracle import orc_loader
    specs = {94: dict(n=384, n_obj=160, flux_seed=2, seed=11), 95: dict(n=256, n_obj=120, flux_seed=3, seed=12),
             96: dict(n=320, n_obj=140, flux_seed=4, seed=13), 97: dict(n=256, n_obj=100, flux_seed=5, seed=14)}

    def build(det):
        s = specs[det]
        return _c3_case(n_obj=s["n_obj"], n=s["n"], flux_seed=s["flux_seed"], seed=s["seed"])

    images = focal_plane.render_focal_plane(list(specs), build, concurrent=2)
    assert sorted(images) == sorted(specs)
    serial = focal_plane.render_focal_plane(list(specs), build, concurrent=1)
    for det in specs:
        scene, objects = build(det)
        r = Renderer(scene)
        r.render_lsst_image(objects)
        r.synchronize()
        assert_bits_equal(images[det], r.image_numpy(), f"CCD {det} on a stream vs stand-alone")
        assert_bits_equal(serial[det], images[det], f"CCD {det}: concurrent=1 vs 2")
    scene, objects = build(95)
    orc = orc_loader.OracleScene(scene)
    orc.render_lsst_image(objects)
    assert_bits_equal(images[95], orc.image, "CCD 95 vs oracle")
    parts = [focal_plane.render_focal_plane(list(specs), build, rank=k, world=2, concurrent=2) for k in range(2)]
    assert sorted(list(parts[0]) + list(parts[1])) == sorted(specs)
    for p in parts:
        for det, img in p.items():
            assert_bits_equal(img, images[det], f"CCD {det} rank split")
    # with a sink every image is handed over as a view of the page-locked buffer of its stream and nothing is kept
    seen = {}
    out = focal_plane.render_focal_plane(list(specs), build, concurrent=3, sink=lambda det, img: seen.__setitem__(det, img.copy()))
    assert out == {} and sorted(seen) == sorted(specs)
    for det in specs:
        assert_bits_equal(seen[det], images[det], f"CCD {det} through the sink")


def _c5_test_visit(n=384, n_ccd=4, per=150, star=None):
    """Four CCDs of a visit at test size, each holding FFT-drawn, photon-shot and faint objects: the bench's C5 catalog with
    its bright tail set by hand (a star and a compact galaxy above fft_sb_thresh, a star of 1.5e6 e- below it)."""
    import math
    from imsim_amd import configs, catalog
    from imsim_amd.diffraction_fft import DiffractionFFT
    scene = configs.scene_c3(nx=n, ny=n)
    scene.sensor.scratch_cells = 3_000_000
    cat = configs._c5_catalog(n_ccd * per, scene, n_ccd=n_ccd)
    for det in range(n_ccd):
        a = int(cat.ccd_offsets[det])
        cat["nominal_flux"][a:a + 3] = [2.0e7 + 1.0e6 * det, 1.5e6 if star is None else star[det], 4.0e7]
        cat["kind"][a:a + 3] = [0, 0, 1]
        cat["hlr"][a + 2] = 0.1
        cat["x"][a:a + 3] = [0.3 * n, 0.7 * n, 0.55 * n]
        cat["y"][a:a + 3] = [0.35 * n + 3.0 * det, 0.3 * n, 0.75 * n]
    cat["sb_flux"] = cat["nominal_flux"] / 80.0
    phot = catalog.realize_fluxes(cat["nominal_flux"], scene.seed)
    objects, _ = configs._c5_objects(cat, phot, scene)
    visit = dict(configs.c5_visit_fft())
    visit["diffraction_fft"] = DiffractionFFT(exptime=30.0, azimuth=math.radians(114.39), altitude=math.radians(53.16),
                                              rotTelPos=math.radians(40.04), spike_length_cutoff=150)
    return scene, cat, phot, objects, visit


def test_focal_plane_ccds_hold_fft_photon_and_faint_objects(torch_cuda):
    """BASELINE config C5 as written: every CCD of the focal plane is the reference's per-CCD build (imsim/lsst_image.py:276-395,
    FFT branch imsim/stamp.py:482-525, fan-out imsim/ccd.py:72-89) -- FFT-drawn objects with the spike stencil first, then the
    photon-shot ones (a 1.5e6-photon star through 150 brighter-fatter rounds among them), faint ones without sensor -- with
    three CCDs in flight on the device's streams.  Every CCD equals its stand-alone build bit for bit; one of them equals the
    oracle's build bit for bit once the oracle is handed the GPU's inverse transforms (rocFFT and numpy agree to ~1e-11 of the
    peak, which can move a Poisson deviate), and within the FFT tolerance when it runs its own."""
    import copy
    from imsim_amd import focal_plane, configs, lsst_image, fft_draw
    from imsim_amd.config import ccd_seed
    from imsim_amd.engine import Renderer
    from imsim_amd._abi import IMS_OBJ_FAINT
    from oracle import orc_loader
    scene, cat, phot, objects, visit = _c5_test_visit()
    offs, coffs = objects.ccd_offsets, objects.cat_offsets
    dets = list(range(len(offs) - 1))

    def job_of(det):
        sub = {k: v[coffs[det]:coffs[det + 1]] for k, v in cat.items() if isinstance(v, np.ndarray)}
        return configs.c5_job(scene, sub, phot[coffs[det]:coffs[det + 1]], np.asarray(objects[offs[det]:offs[det + 1]]), visit=visit)

    def build(det):
        sc = copy.copy(scene)
        sc.seed = ccd_seed(scene.seed, det)
        return sc, job_of(det)

    for det in dets:
        j = job_of(det)
        assert j.n_fft == 2 and (j.objects["n_phot"] > 1_000_000).sum() == 1 and (j.objects["flags"] & IMS_OBJ_FAINT).any()
    transforms = {}
    images = focal_plane.render_focal_plane(dets, build, concurrent=3,
                                            post=lambda det, r: transforms.__setitem__(
                                                det, fft_draw.image_from_rbuf(job_of(det).fft_rows, r._keep_fft[1].cpu().numpy())))
    assert sorted(images) == dets
    for det in dets:
        sc, job = build(det)
        r = Renderer(sc)
        lsst_image.draw_job(r, job)
        r.synchronize()
        assert_bits_equal(images[det], r.image_numpy(), f"CCD {det} in flight vs stand-alone")
        del r
    det = 2
    sc, job = build(det)
    for own_transform in (False, True):
        orc = orc_loader.OracleScene(sc)
        o = orc_loader.OracleFft(sc, job.kpsf, add_noise=True, diffraction_fft=job.diffraction_fft, wavelength=job.wavelength)
        rbuf = o.inverse(job.fft_rows, o.fill(job.fft_rows)) if own_transform else transforms[det]
        o.finish(job.fft_rows, o.spikes(job.fft_rows, rbuf))
        orc.image64 += o.image
        orc.render_lsst_image(job.objects, nrecalc=job.nrecalc)
        if not own_transform:
            assert_bits_equal(images[det], orc.image, f"CCD {det} vs oracle (GPU transforms)")
        else:
            diff = np.count_nonzero(images[det] != orc.image)
            assert diff <= 1e-4 * np.count_nonzero(orc.image) + 2, diff
            assert abs(float(images[det].sum(dtype=np.float64)) / float(orc.image.sum(dtype=np.float64)) - 1.0) < 1e-6
    # the FFT-drawn star left its spikes: light far from the core along the rotated cross that photon-only CCDs do not have
    assert images[det].sum(dtype=np.float64) > 4.0e7
    # the rank split deals every CCD to exactly one rank, same images
    parts = [focal_plane.render_focal_plane(dets, build, rank=k, world=2, concurrent=2) for k in range(2)]
    assert sorted(list(parts[0]) + list(parts[1])) == dets
    for p in parts:
        for d, img in p.items():
            assert_bits_equal(img, images[d], f"CCD {d} rank split")


def test_joint_top_chains_of_a_focal_plane_equal_a_chain_per_ccd(torch_cuda, monkeypatch):
    """ims_plan_run_deferred / ims_plans_run_joint / ims_plan_join: the brightest stars of several CCDs advance through their
    brighter-fatter rounds in lockstep -- ONE pixel-search, ONE updatePixelDistortions and ONE refresh launch per round for all
    CCDs of a batch (different sensors, images, pools, round counts) -- and every CCD's image equals the one it gets from a
    chain of its own (IMS_FOCAL_JOINT=0), bit for bit; batches of two (so that a second batch follows the first on the shared
    streams) and of eight, with and without the ordering hint."""
    import copy
    from imsim_amd import focal_plane, configs
    from imsim_amd.config import ccd_seed
    scene, cat, phot, objects, visit = _c5_test_visit(star=(1.5e6, 0.6e6, 2.2e6, 1.0e6))      # 60 .. 220 rounds
    offs, coffs = objects.ccd_offsets, objects.cat_offsets
    dets = list(range(len(offs) - 1))
    jobs = {}
    for det in dets:
        sub = {k: v[coffs[det]:coffs[det + 1]] for k, v in cat.items() if isinstance(v, np.ndarray)}
        jobs[det] = configs.c5_job(scene, sub, phot[coffs[det]:coffs[det + 1]], np.asarray(objects[offs[det]:offs[det + 1]]), visit=visit)

    def build(det):
        sc = copy.copy(scene)
        sc.seed = ccd_seed(scene.seed, det)
        return sc, jobs[det]

    monkeypatch.setenv("IMS_FOCAL_JOINT", "0")
    single = focal_plane.render_focal_plane(dets, build, concurrent=2)
    # (lists of active tiles, k_build_active_j: at this size they are off by default -- forced on, with launches of 1 % of the
    # tiles so that every workgroup walks several list entries, and off)
    # (the lists appended to by the pixel search itself -- the default --, built by a launch of its own from charge marks per 4 x 4
    # pixels, and from the tile marks alone)
    for joint, hint, list_min, fraction, fine, search in (("8", None, "0", "0.01", "1", "1"), ("8", None, "0", "0.01", "1", "0"),
                                                          ("8", None, "0", "0.01", "0", "0"),
                                                          ("2", lambda det: int(jobs[det].objects["n_phot"].max()), "0", "0.25", "1", "1"),
                                                          ("2", None, "1000000000", "0.25", "1", "1")):
        monkeypatch.setenv("IMS_FOCAL_JOINT", joint)
        monkeypatch.setenv("IMS_JOINT_LIST_MIN", list_min)
        monkeypatch.setenv("IMS_ACTIVE_FRACTION", fraction)
        monkeypatch.setenv("IMS_JOINT_FINE_MARKS", fine)
        monkeypatch.setenv("IMS_JOINT_SEARCH_LISTS", search)
        images = focal_plane.render_focal_plane(dets, build, concurrent=2, chain_hint=hint)
        assert focal_plane.render_focal_plane.last_joint_plans == len(dets)          # every CCD has a star with rounds of its own
        assert sorted(images) == dets
        for det in dets:
            assert_bits_equal(images[det], single[det], f"CCD {det}: joint rounds (batches of {joint}) vs a chain per CCD")
    # CCDs whose plans cannot leave their rounds to a joint run ride along in the same batch: one without any object that needs
    # rounds of its own (no regions, no chain), one whose regions do not fit the scratch capacity at once (several groups: a later
    # group rewrites the slot table, so the plan runs whole)
    odd = {det: jobs[det].objects.copy() for det in dets}
    odd[dets[0]]["n_phot"] = np.minimum(odd[dets[0]]["n_phot"], 9000)
    bright = odd[dets[1]][odd[dets[1]]["n_phot"] > 10000]
    cells = ((bright["stamp_xmax"] - bright["stamp_xmin"] + 2).astype(np.int64) * (bright["stamp_ymax"] - bright["stamp_ymin"] + 2)).sum()

    def build_odd(det):
        sc = copy.copy(build(det)[0])
        if det == dets[1]:
            sc.sensor = copy.copy(sc.sensor)
            sc.sensor.scratch_cells = int(cells * 0.6)
        return sc, odd[det]
    monkeypatch.setenv("IMS_FOCAL_JOINT", "0")
    single = focal_plane.render_focal_plane(dets, build_odd, concurrent=2, nrecalc=10000)
    monkeypatch.setenv("IMS_FOCAL_JOINT", "8")
    images = focal_plane.render_focal_plane(dets, build_odd, concurrent=2, nrecalc=10000)
    assert focal_plane.render_focal_plane.last_joint_plans == len(dets) - 2
    for det in dets:
        assert_bits_equal(images[det], single[det], f"CCD {det}: plans that run whole in a joint batch")
    # the object tables of photon-only CCDs (no CcdJob) take the same path
    tables = {det: jobs[det].objects for det in dets}
    monkeypatch.setenv("IMS_FOCAL_JOINT", "0")
    single = focal_plane.render_focal_plane(dets, lambda det: (build(det)[0], tables[det]), concurrent=2, nrecalc=10000)
    monkeypatch.setenv("IMS_FOCAL_JOINT", "8")
    images = focal_plane.render_focal_plane(dets, lambda det: (build(det)[0], tables[det]), concurrent=2, nrecalc=10000)
    for det in dets:
        assert_bits_equal(images[det], single[det], f"CCD {det}: object table, joint vs single")


@pytest.mark.parametrize("lazy", ["1", "0"])
def test_focal_plane_sensor_arena_leases_equal_a_state_per_ccd(torch_cuda, monkeypatch, lazy):
    """engine.SensorArena: the CCDs of a joint focal plane lease their pixel-boundary state from one arena per device -- three
    static regions taken in turn (a region is re-initialised for the next CCD behind the events of its last readers) and private
    cells by need -- instead of 5 GB of arrays per renderer.  Same images, bit for bit, as renderers with a state of their own
    (IMS_FOCAL_ARENA=0), also when the private pool is too small for a batch: the previous batch is then collected early, or the
    batch is cut short.  lazy: with the static state not made at all (IMS_FOCAL_LAZY_STATIC, the default: one static region as
    a mere address) and made per CCD (0: the three regions in turn)."""
    monkeypatch.setenv("IMS_FOCAL_LAZY_STATIC", lazy)
    import copy
    from imsim_amd import focal_plane, configs, engine
    from imsim_amd.config import ccd_seed
    scene, cat, phot, objects, visit = _c5_test_visit(n_ccd=6, star=(1.5e6, 0.6e6, 2.2e6, 1.0e6, 0.8e6, 1.2e6))
    offs, coffs = objects.ccd_offsets, objects.cat_offsets
    dets = list(range(len(offs) - 1))
    jobs = {}
    for det in dets:
        sub = {k: v[coffs[det]:coffs[det + 1]] for k, v in cat.items() if isinstance(v, np.ndarray)}
        jobs[det] = configs.c5_job(scene, sub, phot[coffs[det]:coffs[det + 1]], np.asarray(objects[offs[det]:offs[det + 1]]), visit=visit)

    def build(det):
        sc = copy.copy(scene)
        sc.seed = ccd_seed(scene.seed, det)
        return sc, jobs[det]

    def need(det):
        o = jobs[det].objects
        br = o[o["n_phot"] > 10000]
        return int(((br["stamp_xmax"] - br["stamp_xmin"] + 2).astype(np.int64) * (br["stamp_ymax"] - br["stamp_ymin"] + 2)).sum())

    monkeypatch.setenv("IMS_FOCAL_JOINT", "4")
    monkeypatch.setenv("IMS_FOCAL_ARENA", "0")
    own = focal_plane.render_focal_plane(dets, build, concurrent=2)
    assert focal_plane.render_focal_plane.last_arena_gib == 0.0
    monkeypatch.setenv("IMS_FOCAL_ARENA", "1")
    leased = focal_plane.render_focal_plane(dets, build, concurrent=2)
    assert focal_plane.render_focal_plane.last_arena_gib > 0.0
    needs = sorted(need(d) for d in dets)
    # a pool that holds the two largest CCDs and no more: batches of four cannot form, cells come back from collected CCDs
    monkeypatch.setenv("IMS_FOCAL_ARENA_CELLS", str(needs[-1] + needs[-2] + 16))
    tight = focal_plane.render_focal_plane(dets, build, concurrent=2)
    for det in dets:
        assert_bits_equal(leased[det], own[det], f"CCD {det}: leased state vs a state of its own")
        assert_bits_equal(tight[det], own[det], f"CCD {det}: a private pool that runs dry")
    arena = next(iter(engine._SENSOR_ARENAS.values()))
    assert arena._free == [(arena.private_base, arena.private_cells)]           # every lease came back, the free list coalesced
    engine._SENSOR_ARENAS.clear()                                               # (the tight pool must not serve the tests that follow)


# ---------------------------------------------------------------------------------------------
# CCD readout: e-image -> raw amplifier segments
# ---------------------------------------------------------------------------------------------
def _gpu_bleed(torch, img, full_well, midline):
    lib = _abi.load()
    t = torch.from_numpy(np.ascontiguousarray(img, dtype=np.float64)).cuda()
    ny, nx = t.shape
    flags = torch.empty((nx * ny + 15) // 16 * 16 + 16 * nx, dtype=torch.uint8, device="cuda")   # IMS_READOUT_SCRATCH_BYTES
    _abi.check(lib.ims_readout_bleed(t.data_ptr(), flags.data_ptr(), nx, ny, float(full_well), int(midline), None))
    torch.cuda.synchronize()
    return t.cpu().numpy()


def test_bleed_trails_match_reference_goldens_and_oracle(torch_cuda):
    """bleed_eimage on the GPU against the vectors generated from the reference's bleed_trails.py, and against the
    oracle on a CCD-sized image with many saturated stars."""
    import os
    from oracle import orc_loader
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "readout_golden.npz"))
    fw = float(gold["full_well"])
    assert_bits_equal(_gpu_bleed(torch_cuda, gold["img_in"], fw, True), gold["img_midline"], "midline stop")
    assert_bits_equal(_gpu_bleed(torch_cuda, gold["img_in"], fw, False), gold["img_nomidline"], "no midline stop")
    assert_bits_equal(_gpu_bleed(torch_cuda, gold["chan_in"][:, None], fw, False)[:, 0], gold["chan_out"], "single channel")
    assert_bits_equal(_gpu_bleed(torch_cuda, gold["neg_in"][:, None], float(gold["neg_fw"]), False)[:, 0], gold["neg_out"],
                      "regression channel")
    rng = np.random.default_rng(8)
    img = rng.poisson(800.0, size=(1000, 1536)).astype(np.float64)
    for _ in range(300):
        y, x, s = rng.integers(0, 1000), rng.integers(0, 1536), rng.integers(1, 6)
        img[max(y - s, 0):y + s, max(x - s, 0):x + s] += rng.uniform(0.5, 40.0) * 1e5
    for mid in (True, False):
        assert_bits_equal(_gpu_bleed(torch_cuda, img, 1e5, mid), orc_loader.bleed_eimage(img, 1e5, mid), f"random image, midline {mid}")


def _eimage_with_stars(nx, ny, seed, n_star=40):
    rng = np.random.default_rng(seed)
    e = rng.poisson(800.0, size=(ny, nx)).astype(np.float64)
    for _ in range(n_star):
        y, x, s = rng.integers(0, ny), rng.integers(0, nx), rng.integers(1, 5)
        e[max(y - s, 0):y + s, max(x - s, 0):x + s] += np.round(rng.uniform(0.2, 30.0) * 1e5)
    return e


def test_readout_chain_is_bit_exact(torch_cuda, tmp_path):
    """CcdReadout.build_amp_images on the GPU (bleed trails, dark current, gain / flips / crosstalk, prescan and
    overscan, parallel + serial CTI, bias, read noise, int32) against the oracle, for a full-size E2V CCD with
    per-amp bias levels and for an ITL CCD; then the raw file is written and read back."""
    import json
    from imsim_amd import readout, camera, fits_io
    from oracle import orc_loader
    levels = {f"{r}_{s}": {a: 20000.0 + 11 * i for i, a in enumerate(camera.CHANNELS)} for r in camera.RAFTS for s in camera.SENSORS}
    (tmp_path / "bias.json").write_text(json.dumps(levels))
    cases = (("R22_S11", dict(bias_levels_file=str(tmp_path / "bias.json"))),
             ("R01_S00", dict(bias_level=1000.0, read_noise=6.5, scti=3e-6, pcti=0)))
    for det, kw in cases:
        cam = camera.Camera("LsstCamSim", bias_levels_file=kw.get("bias_levels_file"))
        ny, nx = cam[det].bounds.numpyShape()
        e = _eimage_with_stars(nx, ny, seed=len(det) + nx)
        hdr = readout.eimage_header(det, 30.0, opsim_data={"mjd": 60000.25, "band": "r"})
        eimg = readout.EImage(torch_cuda.from_numpy(e).cuda(), hdr)
        ro = readout.CcdReadout(eimg, camera_obj=cam, **kw)
        seed = 4242
        got = ro.build_amp_images(seed)
        torch_cuda.cuda.synchronize()
        st = {}
        want = orc_loader.readout_chain(e, ro.descriptor(), ro.full_well, ro.midline_stop(), ro.dark_level(), readout.DARK_STREAM,
                                        seed, ro.pcte_band, ro.scte_band, st)
        assert (st["bled"] != e).any(), "the test image must actually bleed"
        assert_bits_equal(eimg.array.cpu().numpy(), st["dark"], f"{det}: e-image after bleed trails and dark current")
        assert_bits_equal(got.cpu().numpy(), want, f"{det}: raw segments")
        assert got.shape == (16, 2048, 576)
    hdus = ro.prepare_hdus(seed + 1)
    f = tmp_path / "raw.fits"
    readout.CcdReadout.write_raw_file(hdus, str(f))
    back = fits_io.read_fits(str(f))
    assert len(back) == 17 and back[0][0]["OUTFILE"] == "raw.fits" and back[0][0]["CHIPID"] == "R01_S00"
    assert back[1][0]["EXTNAME"] == "Segment10" and back[16][0]["EXTNAME"] == "Segment00"
    assert back[9][0]["DATASEC"] == "[4:512,1:2000]"
    for k in range(16):
        assert_bits_equal(back[k + 1][1], hdus[k + 1][1], f"segment {k} through the file")
    eimg.write(str(tmp_path / "eimage.fits"))
    (h, d), = fits_io.read_fits(str(tmp_path / "eimage.fits"))
    assert h["DET_NAME"] == "R01_S00" and d.shape == (4000, 4072) and d.dtype == np.float32


def test_config_readout_writes_eimage_and_raw_file(torch_cuda, tmp_path):
    """`output.file_name` and the `output.readout` extra output (config/imsim-config.yaml:322-352 semantics): the
    e-image file and the 16-segment raw file; the segments, put back together with the gains, give the e-image."""
    from imsim_amd import fits_io, camera
    from imsim_amd.lsst_image import GalSimConfigError
    res = _process(**{"image.nobjects": 40, "stamp.draw_method": "phot", "output.dir": str(tmp_path), "output.file_name": "eimage.fits",
                      "output.readout": {"readout_time": 3.0, "dark_current": 0.0, "bias_level": 1000.0, "scti": 0.0, "pcti": 0.0,
                                         "read_noise": 0.0, "file_name": "amp.fits",
                                         "added_keywords": {"TESTKEY1": "TESTVAL1"}}})
    assert [os.path.basename(f) for f in res.files] == ["eimage.fits", "amp.fits"] and len(res.raw) == 1
    (eh, ed), = fits_io.read_fits(res.files[0])
    assert eh["DET_NAME"] == "R22_S11" and eh["CAMERA"] == "LsstCamSim" and ed.shape == (4004, 4096)
    assert np.array_equal(ed, res.images[0])
    raw = fits_io.read_fits(res.files[1])
    assert len(raw) == 17 and raw[0][0]["TESTKEY1"] == "TESTVAL1" and raw[0][0]["CHIPID"] == "R22_S11" and raw[0][0]["EXPTIME"] == eh["EXPTIME"]
    ccd = camera.Camera("LsstCamSim")["R22_S11"]
    back = np.zeros_like(ed)
    for k, amp in enumerate(ccd.values()):
        r, b = amp.raw_data_bounds, amp.bounds
        sec = raw[k + 1][1][r.ymin - 1:r.ymax, r.xmin - 1:r.xmax].astype(np.float64) - 1000.0
        if amp.raw_flip_x:
            sec = sec[:, ::-1]
        if amp.raw_flip_y:
            sec = sec[::-1, :]
        back[b.ymin - 1:b.ymax, b.xmin - 1:b.xmax] = sec * amp.gain
    # the readout works on the e-image after the bleed trails (the catalog's brightest star is far above full well)
    bled = res.eimages[0].array.cpu().numpy()
    # (the e-image file is float32: its brightest pixels, > 2^24 e-, are rounded by a few electrons)
    assert ed.max() > ccd.full_well and bled.max() == ccd.full_well and bled.sum() <= ed.astype(np.float64).sum() + 64
    # no noise, no CTI, no dark current: only crosstalk (<= 4e-4 of the brightest neighbour) and the ADU truncation remain
    assert np.abs(back - bled).max() <= 2.0 + 1e-3 * bled.max()
    assert abs(back.sum() / bled.sum() - 1) < 0.01
    with pytest.raises(GalSimConfigError):
        _process(**{"image.nobjects": 5, "output.readout": {"no_such_parameter": 1}})


def test_config_fits_stamp_object_end_to_end(torch_cuda, tmp_path):
    """An instance catalog with a FITS-stamp object (imsim/instcat.py:552-561) through config.Process: the stamp's
    shape shows up on the CCD at the object's position, rotated by -theta, on the stamp's own pixel scale."""
    from imsim_amd import fits_io
    here = os.path.dirname(os.path.abspath(__file__))
    header = [l for l in open(os.path.join(here, "golden", "example_instcat_subset.txt")) if not l.startswith("object")]
    stamp = np.zeros((20, 20), dtype=np.float32)
    stamp[2:18, 9:11] = 1.0                               # a bar along the stamp's y axis
    fits_io.write_fits(str(tmp_path / "bar.fits"), [({}, stamp)])
    lines = header + ["object 1 60.49045502638662697 -38.16437495898705379 17.0 starSED/x.gz 0 0 0 0 0 0 bar.fits 0.4 0.0 none none\n",
                      "object 2 60.52 -38.18 17.0 starSED/x.gz 0 0 0 0 0 0 bar.fits 0.4 90.0 none none\n",
                      "object 3 60.46 -38.15 22.0 starSED/x.gz 0 0 0 0 0 0 point none none\n"]
    (tmp_path / "cat.txt").write_text("".join(lines))
    res = _process(**{"input.instance_catalog.file_name": str(tmp_path / "cat.txt"), "stamp.draw_method": "phot",
                      "image.sensor": "", "stamp.photon_ops": [], "input.instance_catalog.sort_mag": False})
    img, truth = res.images[0].astype(np.float64), res.truth[0]
    assert list(truth["mode"][:2]) == ["phot", "phot"] and truth["phot_flux"][0] > 1e5
    angles = []
    for k in (0, 1):
        x0, y0 = int(round(truth["x"][k])) - 1, int(round(truth["y"][k])) - 1
        cut = img[y0 - 40:y0 + 41, x0 - 40:x0 + 41]
        assert cut.sum() > 0.9 * truth["realized_flux"][k] > 0
        yy, xx = np.mgrid[-40:41, -40:41]
        w = cut / cut.sum()
        mx, my = (w * xx).sum(), (w * yy).sum()
        cxx, cyy, cxy = (w * (xx - mx) ** 2).sum(), (w * (yy - my) ** 2).sum(), (w * (xx - mx) * (yy - my)).sum()
        lam = np.linalg.eigvalsh(np.array([[cxx, cxy], [cxy, cyy]]))
        # the bar is 16 x 0.4" = 32 pixels long and 2 x 0.4" = 4 pixels wide (plus the PSF)
        assert abs(np.sqrt(lam[1]) - 32 / np.sqrt(12)) < 1.5 and np.sqrt(lam[0]) < 4.0
        angles.append(0.5 * np.degrees(np.arctan2(2 * cxy, cxx - cyy)))
    # the stamp lives on the sky (the WCS turns it on the CCD); theta = 90 turns the second one by a right angle
    d = abs(angles[0] - angles[1]) % 180.0
    assert abs(d - 90.0) < 3.0, angles


def test_config_flat_readout_without_opsim(torch_cuda, tmp_path):
    """A flat read out without any opsim data (tests/test_readout.py:124-160 of the reference, test_no_opsim): the
    header falls back to its defaults, the segments carry counts / gain + bias."""
    from imsim_amd import config, camera, fits_io
    res = config.Process({"image": {"type": "LSST_Flat", "random_seed": 42, "det_name": "R22_S11", "counts_per_pixel": 1000,
                                    "max_counts_per_iter": 1000, "sensor": ""},
                          "output": {"dir": str(tmp_path), "file_name": "flat_e.fits",
                                     "readout": {"file_name": "flat_amp.fits", "dark_current": 0.0, "scti": 0.0, "pcti": 0.0,
                                                 "read_noise": 0.0, "bias_level": 500.0}}})
    assert res.images[0].shape == (4004, 4096) and abs(res.images[0].mean() - 1000.0) < 0.1
    raw = fits_io.read_fits(res.files[1])
    ph = raw[0][0]
    assert ph["IMGTYPE"] == "FLAT" and ph["TRACKSYS"] == "LOCAL" and ph["MJD"] == 51444.0 and ph["RUNNUM"] == -999
    ccd = camera.Camera("LsstCamSim")["R22_S11"]
    for k, amp in enumerate(ccd.values()):
        r = amp.raw_data_bounds
        sec = raw[k + 1][1][r.ymin - 1:r.ymax, r.xmin - 1:r.xmax].astype(np.float64)
        xt = 0.0 if ccd.xtalk is None else sum(ccd.xtalk[k][j] / list(ccd.values())[j].gain for j in range(16)) * 1000.0
        assert abs(sec.mean() - (500.0 + 1000.0 / amp.gain + xt - 0.5)) < 0.2, (k, sec.mean())
        assert (raw[k + 1][1][:, :r.xmin - 1] == 500).all()            # prescan: bias only


def test_general_sersic_index_fft_and_photon_shooting_agree(torch_cuda):
    """A Sersic index off the 1 / 4 pair (imsim/instcat.py:511-517: quantised to 0.05, here 2.5) has its own radial
    table for photon shooting and its own k-table for the FFT branch; the two renderings meet the reference's
    FFT-vs-phot criteria, and the photon pool equals the oracle's bit for bit."""
    from imsim_amd import _abi, configs, catalog, fft_draw
    from imsim_amd.engine import Renderer
    from oracle import orc_loader
    fwhm_atm, fwhm_sys = catalog.kolmogorov_gaussian_fwhm()
    scene = configs.scene_c2(nx=128, ny=128)
    scene.psf = [(_abi.IMS_PSF_RADIAL, 2, fwhm_atm, 0.0, 1.0), (_abi.IMS_PSF_GAUSSIAN, 0, fwhm_sys / 2.3548200450309493, 0.0, 1.0)]
    configs.add_sersic_tables(scene, [2.5])
    cat = dict(x=np.array([64.3]), y=np.array([63.8]), mag=np.zeros(1), nominal_flux=np.array([4.0e6]), kind=np.array([2]),
               hlr=np.array([0.5]), q=np.array([0.7]), pa=np.array([20.0]), obj_id=np.array([3]), sersic_n=np.array([2.5]))
    objects, _ = catalog.build_object_table(cat, np.array([4000000]), stamp_size=128, sersic_index=scene.sersic_index)
    assert objects["prof_table"][0] == scene.sersic_index[2.5] == 3
    rp = Renderer(scene)
    rp.render(objects)
    rf = Renderer(scene)
    kt = fft_draw.profile_ktable_ids(scene, objects["prof_table"])
    assert kt[0] == 2
    rows, _ = fft_draw.build_fft_objects(objects, cat["nominal_flux"], kt)
    fft_draw.FftDrawer(rf, fft_draw.kolmogorov_gaussian_kpsf(fwhm_atm, fwhm_sys), add_noise=False).draw(rows)
    rp.synchronize(); rf.synchronize()
    a, b = rp.image_numpy().astype(float), rf.image_numpy().astype(float)
    yy, xx = np.mgrid[0:128, 0:128]
    w = (np.hypot(xx - 63.3, yy - 62.8) < 25)

    def mom(img):
        f = (img * w).sum()
        mx, my = (img * w * xx).sum() / f, (img * w * yy).sum() / f
        return f, mx, my, (img * w * ((xx - mx) ** 2 + (yy - my) ** 2)).sum() / f
    fa, xa, ya, ra = mom(a)
    fb, xb, yb, rb = mom(b)
    assert abs(a.max() / b.max() - 1) < 0.05 and abs(fa / fb - 1) < 0.01
    assert abs(xa - xb) < 0.02 and abs(ya - yb) < 0.02 and abs(ra / rb - 1) < 0.10
    # and it is NOT the n = 4 profile any more: a de Vaucouleurs profile of the same half-light radius puts more light far out
    objects4 = objects.copy()
    objects4["prof_table"] = 1
    r4 = Renderer(scene)
    r4.render(objects4)
    r4.synchronize()
    ring = (np.hypot(xx - 63.3, yy - 62.8) > 12) & w
    a4 = r4.image_numpy().astype(float)
    assert a4[ring].sum() > 1.15 * a[ring].sum()
    small = objects.copy()
    small["n_phot"] = 5000
    pool = Renderer(scene).shoot_photons(small)
    opool = orc_loader.OracleScene(scene).shoot_pool(small)
    g, o = pool.to_host(), opool.to_host()
    for f in ("x", "y", "wavelength"):
        assert_bits_equal(g[f], o[f], f"photon field {f} (n = 2.5)")


def test_pooling_mode_draws_fft_objects_first(torch_cuda):
    """LSST_PhotonPoolingImageBuilder.buildImage (imsim/photon_pooling.py:84-114): objects above the FFT threshold are
    FFT-drawn before the photon batches and do not enter them."""
    from imsim_amd import _abi, configs, catalog, fft_draw, lsst_image
    from imsim_amd.engine import Renderer
    scene = configs.scene_c3(nx=256, ny=256, sensor=False)
    cat = catalog.synthetic_catalog(40, nx=256, ny=256)
    cat["nominal_flux"][:2] = [3.0e6, 5.0e6]                      # two objects beyond the 1e6 floor of stamp.py:275
    cat["kind"][:2] = [0, 1]
    cat["sb_flux"] = cat["nominal_flux"] / 80.0
    phot = catalog.realize_fluxes(cat["nominal_flux"], 3)
    fwhm_atm, fwhm_sys = catalog.kolmogorov_gaussian_fwhm(configs.VISIT["airmass"], configs.VISIT["raw_seeing"], "r")
    kpsf = fft_draw.kolmogorov_gaussian_kpsf(fwhm_atm, fwhm_sys)
    b = lsst_image.LSST_PhotonPoolingImageBuilder()
    b.setup({"det_name": "R22_S11", "xsize": 256, "ysize": 256, "nbatch": 3, "nsubbatch": 2}, "LSST_Photons")
    r = Renderer(scene)
    truth = {}
    b.build_image(r, cat, phot, lambda c, p: configs.c3_objects(c, p, scene), seed=5, truth=truth, fft_sb_thresh=2.0e4, kpsf=kpsf,
                  fwhm_total=float(np.hypot(fwhm_atm, fwhm_sys)))
    r.synchronize()
    modes = list(truth["mode"])
    assert modes[0] == "fft" and modes[1] == "fft" and "fft" not in modes[2:]
    assert np.all(truth["phot_flux"][:2] == 0) and np.all(truth["fft_flux"][:2] == cat["nominal_flux"][:2])
    np.testing.assert_allclose(truth["incident_flux"][:2], cat["nominal_flux"][:2], rtol=0.03)
    img = r.image_numpy()
    ix, iy = int(round(cat["x"][0])) - 1, int(round(cat["y"][0])) - 1
    assert img[iy - 2:iy + 3, ix - 2:ix + 3].sum() > 0.2 * 3.0e6 * 0.5       # the FFT-drawn star is on the image
    # without a threshold everything is photon-shot, as before
    r2 = Renderer(scene)
    t2 = {}
    b.build_image(r2, cat, phot, lambda c, p: configs.c3_objects(c, p, scene), seed=5, truth=t2)
    assert "fft" not in list(t2["mode"]) and t2["phot_flux"][0] == phot[0]


@pytest.mark.parametrize("pooling", [False, True])
def test_twenty_object_image_splits_and_sums_like_the_reference(torch_cuda, pooling):
    """tests/test_image.py:18-29, :162-228 of the reference: 20 stars through Gaussian(fwhm 0.3) + RubinDiffractionOptics with
    fft_sb_thresh = 1e4 -- the ten of ~2e6 electrons are FFT-drawn, the nine of ~2e5 are photon-shot, the last (38 e-) is
    faint -- and the 20 x 20 pixel aperture sums equal the expected brightness within 4 sqrt(N), for LSST_Image and for
    LSST_PhotonPoolingImage alike.  (The reference's pixel positions come from its Batoid WCS, which is out of scope; the
    positions here are this build's own, including the reference's pair of nearly coincident objects.)"""
    import math
    from imsim_amd import _abi, configs, catalog, fft_draw, lsst_image
    from imsim_amd.diffraction_fft import DiffractionFFT
    from imsim_amd.engine import Renderer
    n = 1024
    scene = configs.scene_c3(nx=n, ny=n, sensor=False)
    scene.optics = configs.rubin_optics_struct(n, n, rottelpos=20.0, altitude=88.0, azimuth=73.7707957)
    scene.psf = [(_abi.IMS_PSF_GAUSSIAN, 0, 0.3 / 2.3548200450309493, 0.0, 1.0)]
    scene.ops = [(_abi.IMS_OP_TIME_SAMPLER, 0, [0.0, 30.0]), (_abi.IMS_OP_PUPIL_ANNULUS_SAMPLER, 0, [4.18, 2.55]),
                 (_abi.IMS_OP_RUBIN_DIFFRACTION_OPTICS, 0, [1.0, 0.0])]
    # the reference's expected brightness values (tests/test_image.py:189-210), objects off the image dropped
    flux = np.array([1.974717e+06, 4.329067e+06 / 2, 1.971714e+06, 1.971509e+06, 1.976528e+06, 3.125554e+06 / 2 + 4.0e5, 1.978627e+06,
                     1.977857e+06, 1.977190e+05, 4.049977e+06 / 20, 2.192697e+06 / 11, 4.329907e+06 / 22, 1.971800e+05, 1.964360e+05,
                     1.970330e+05, 1.970620e+05, 1.965160e+05, 3.800000e+01])
    rng = np.random.default_rng(8)
    cat = catalog.synthetic_catalog(len(flux), nx=n, ny=n)
    gx, gy = np.meshgrid(np.linspace(150, n - 150, 5), np.linspace(150, n - 150, 4))
    cat["x"][:] = gx.ravel()[:len(flux)] + rng.uniform(-0.5, 0.5, len(flux))
    cat["y"][:] = gy.ravel()[:len(flux)] + rng.uniform(-0.5, 0.5, len(flux))
    cat["x"][12], cat["y"][12] = cat["x"][11] + 0.4, cat["y"][11] - 1.8       # "almost indistinguishable" pair (:176-178)
    cat["kind"][:] = 0
    cat["nominal_flux"][:] = flux
    cat["sb_flux"] = flux / 80.0
    phot = catalog.realize_fluxes(flux, 12345)
    kpsf = [(_abi.IMS_KPSF_GAUSSIAN, 0, 0.3 / 2.3548200450309493)]
    dfft = DiffractionFFT(exptime=30.0, azimuth=math.radians(73.7707957), altitude=math.radians(88.0), rotTelPos=math.radians(20.0),
                          brightness_threshold=1.0e5)
    make = lambda c, p: configs.c3_objects(c, p, scene)
    r = Renderer(scene)
    truth = {}
    if pooling:
        b = lsst_image.LSST_PhotonPoolingImageBuilder()
        b.setup({"det_name": "R22_S11", "xsize": n, "ysize": n, "nbatch": 10, "nbatch_fft": 5}, "LSST_Photons")
        b.build_image(r, cat, phot, make, seed=12345, truth=truth, fft_sb_thresh=1.0e4, kpsf=kpsf, fwhm_total=0.3,
                      diffraction_fft=dfft)
    else:
        b = lsst_image.LSST_ImageBuilder()
        b.setup({"det_name": "R22_S11", "xsize": n, "ysize": n})
        b.build_image(r, cat, phot, make, fft_sb_thresh=1.0e4, kpsf=kpsf, fwhm_total=0.3, diffraction_fft=dfft, truth=truth)
    r.synchronize()
    modes = list(truth["mode"])
    n_fft = int(np.sum(flux >= 1.0e6))
    assert modes[:n_fft] == ["fft"] * n_fft and modes[n_fft:-1] == ["phot"] * (len(flux) - n_fft - 1) and modes[-1] == "faint"
    img = r.image_numpy().astype(np.float64)
    pixel_radius = 10
    for i in range(len(flux)):
        col, row = int(round(cat["x"][i])) - 1, int(round(cat["y"][i])) - 1
        got = img[row - pixel_radius:row + pixel_radius, col - pixel_radius:col + pixel_radius].sum()
        want = flux[i] + (flux[12] if i == 11 else flux[11] if i == 12 else 0.0)
        # the spider diffraction puts ~2 % of a star's light beyond the 20 x 20 pixel aperture (the reference's
        # expected values are sums of ITS rendering and contain the same loss; here the expectation is the input flux)
        assert abs(got - 0.98 * want) <= 4.0 * math.sqrt(want) + 0.02 * want, (i, got, want)
    key = "incident_flux" if pooling else "realized_flux"
    fl = truth[key]
    np.testing.assert_allclose(fl[:-1], flux[:-1], rtol=0.1)                    # tests/test_image.py:224-226


@pytest.mark.parametrize("exponent", [-0.2, -0.3])
def test_atmospheric_psf_is_chromatic_with_the_configured_exponent(torch_cuda, exponent):
    """tests/test_psf.py:322-339 of the reference: a star seen through the AtmosphericPSF at 400 nm and at 900 nm; the ratio of
    the PSF sizes is (400 / 900)^exponent to 0.1 %.  Here on the photons themselves (the reference measures HSM moments on
    1e5-photon images): both wavelengths draw the same deviates, so the atmospheric displacements scale exactly."""
    from imsim_amd import _abi, atm_psf, configs, catalog
    from imsim_amd.engine import Renderer
    pos = {}
    for wl in (400.0, 900.0):
        scene = configs.scene_c3(nx=256, ny=256, sensor=False)
        scene.ops = []
        # the reference's set-up: airmass 1, seeing 1", 51.2 m screens, 600 s "for lots of mixing", second kick off "since it's
        # achromatic" (:279-292)
        atm = atm_psf.AtmosphericPSF(1.0, 1.0, "r", seed=5, exponent=exponent, exptime=600.0, screen_size=51.2, screen_scale=0.1,
                                     no2k=True, device=torch_cuda.device("cuda", 0))
        scene.atm = atm
        scene.psf = atm.psf_components(second_kick_table_id=0)
        assert len(scene.psf) == 1
        cat = catalog.synthetic_catalog(1, nx=256, ny=256)
        cat["x"][:], cat["y"][:], cat["kind"][:], cat["nominal_flux"][:] = 128.0, 128.0, 0, 1.0e5
        objects, _ = configs.c3b_objects(cat, np.array([200000]), scene)
        objects["sed_table"], objects["sed_wave"] = -1, wl                      # monochromatic SED
        r = Renderer(scene)
        g = r.shoot_photons(objects).to_host()
        r.synchronize()
        assert np.all(g["wavelength"] == wl)
        pos[wl] = (g["x"] - objects["x0"][0], g["y"] - objects["y0"][0])
    s400 = np.sqrt(np.mean(pos[400.0][0] ** 2 + pos[400.0][1] ** 2))
    s900 = np.sqrt(np.mean(pos[900.0][0] ** 2 + pos[900.0][1] ** 2))
    np.testing.assert_allclose(s400 / s900, (400.0 / 900.0) ** exponent, rtol=0.001)
    assert s900 > 0.5                                                         # pixels: a real seeing disc, not a delta function


def test_image_nobjects_is_capped_by_the_catalog(torch_cuda):
    """tests/test_lsst_image.py:10-66 (test_image_nobjects): the number of objects rendered is the minimum of
    image.nobjects and the objects the catalog holds for the CCD -- nobjects 1 renders one, a request beyond the catalog
    renders every object once (no repeats), and "" means all."""
    common = {"stamp.draw_method": "phot", "image.sensor": "", "stamp.photon_ops": []}
    everything = _process(**{"image.nobjects": "", **common}).truth[0]
    n_all = len(everything["index"])
    assert n_all > 5
    one = _process(**{"image.nobjects": 1, **common}).truth[0]
    assert len(one["index"]) == 1
    five = _process(**{"image.nobjects": 5, **common}).truth[0]
    assert len(five["index"]) == 5 and list(five["index"]) == list(everything["index"][:5])
    beyond = _process(**{"image.nobjects": n_all + 1000, **common}).truth[0]
    assert len(beyond["index"]) == n_all and len(set(beyond["index"])) == n_all
