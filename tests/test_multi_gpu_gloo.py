"""The N>1 path on CPU: two and EIGHT gloo ranks shard the catalog (or deal the CCDs of a focal plane), each renders its
share (with the oracle, since the product has no CPU path), the images are sum-reduced onto rank 0 and must equal the
single-process render bit for bit (shard invariance of the counter-addressed random streams).  Eight is the node size
BASELINE.json names (1/2/4/8 GPUs); the reference's own fan-out is CCDs over `output.nproc` workers (imsim/ccd.py:72-89)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _c3_case():
    from helpers import c3_small_case
    return c3_small_case(n_obj=160, n=256, flux_seed=3, scratch=400_000)


def _worker(rank, world, port, out_path, mode="c2"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import small_case
    from imsim_amd import parallel
    from oracle import orc_loader
    if mode == "c3":
        scene, objects = _c3_case()
    else:
        scene, objects, _ = small_case(n_obj=120, nx=256, ny=256, flux_seed=9)
    mine = parallel.shard_objects(objects, rank, world)
    counts = torch.tensor([len(mine), int(mine["n_phot"].sum())])
    gathered = [torch.zeros_like(counts) for _ in range(world)]
    dist.all_gather(gathered, counts)
    orc = orc_loader.OracleScene(scene)
    if mode == "c3":
        orc.render_lsst_image(mine, nrecalc=2000)         # full op chain, Silicon sensor, brighter-fatter chains per object
    else:
        orc.render(mine)
    img = torch.from_numpy(orc.image64.copy())
    # what bench.py checks on every rank's own image before the int32 exchange may be used
    ok = torch.tensor([1.0 if parallel.integer_counts_ok(img, world) else 0.0])
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    assert float(ok.item()) == 1.0
    img_i = img.clone()
    parallel.reduce_image(img, dst=0)
    parallel.reduce_image(img_i, dst=0, integer_counts=True)        # the int32 exchange bench.py uses
    if rank == 0:
        np.savez(out_path, image=img.numpy(), image_int=img_i.numpy(), counts=torch.stack(gathered).numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_render_equals_single_process(tmp_path, world):
    from helpers import small_case
    from oracle import orc_loader
    out = str(tmp_path / "reduced.npz")
    from helpers import free_port
    port = free_port()
    mp.start_processes(_worker, args=(world, port, out), nprocs=world, join=True, start_method="spawn")
    res = np.load(out)
    scene, objects, _ = small_case(n_obj=120, nx=256, ny=256, flux_seed=9)
    orc = orc_loader.OracleScene(scene)
    orc.render(objects)
    assert np.array_equal(res["image"], orc.image64)
    assert np.array_equal(res["image_int"], orc.image64) and res["image_int"].dtype == np.float64
    counts = res["counts"]
    assert counts.shape[0] == world
    assert counts[:, 0].sum() == len(objects)
    assert counts[:, 1].sum() == objects["n_phot"].sum()
    # the load is balanced up to the weight of the single heaviest (indivisible) object
    assert counts[:, 1].max() - counts[:, 1].min() <= objects["n_phot"].max()


@pytest.mark.parametrize("world", [2, 8])
def test_c3_sharded_lsst_image_reduce_equals_single_process(tmp_path, world):
    """BASELINE config 3 in small: full photon-op chain, Silicon sensor with tree rings, objects above nrecalc running their
    own brighter-fatter rounds; objects dealt by `parallel.assign_ranks`' cost model to 2 and to 8 ranks, the int32 and the f64
    form of the image reduce both equal to one process bit for bit."""
    from oracle import orc_loader
    from helpers import free_port
    from imsim_amd import parallel
    out = str(tmp_path / "reduced_c3.npz")
    mp.start_processes(_worker, args=(world, free_port(), out, "c3"), nprocs=world, join=True, start_method="spawn")
    res = np.load(out)
    scene, objects = _c3_case()
    assert (objects["n_phot"] > 2000).sum() >= 3            # brighter-fatter chains are exercised
    orc = orc_loader.OracleScene(scene)
    orc.render_lsst_image(objects, nrecalc=2000)
    assert orc.image64.sum() > 0
    assert np.array_equal(res["image"], orc.image64)
    assert np.array_equal(res["image_int"], orc.image64)
    assert parallel.integer_counts_ok(torch.from_numpy(orc.image64), world)
    counts = res["counts"]
    assert counts[:, 0].sum() == len(objects) and counts[:, 1].sum() == objects["n_phot"].sum()
    # the cost model keeps objects whole and gives the owner of the longest chain little else
    owner = parallel.assign_ranks(objects["n_phot"], world, nrecalc=2000)
    assert len(np.unique(owner)) == world


def test_shard_objects_is_a_partition():
    from helpers import small_case
    from imsim_amd import parallel
    _, objects, _ = small_case(n_obj=101, nx=256, ny=256)
    for world in (1, 2, 3, 8):
        ids = np.concatenate([parallel.shard_objects(objects, r, world)["obj_id"] for r in range(world)])
        assert sorted(ids) == sorted(objects["obj_id"])


def test_unit_flux_path_knows_what_makes_a_photon_carry_other_than_one_electron():
    """parallel.unit_flux_path decides whether an exchange may run on int32 copies: not with BandpassRatio in the chain, not
    with a flux_per_photon other than 1, not with FITS-stamp objects shot through an interpolant (+- (integral |K|)^2)."""
    from helpers import small_case
    from imsim_amd import parallel, _abi
    scene, objects, _ = small_case(n_obj=20, nx=128, ny=128)
    assert parallel.unit_flux_path(scene, objects)
    scaled = objects.copy()
    scaled["flux_per_photon"][3] = 0.5
    assert not parallel.unit_flux_path(scene, scaled)
    scene.image_profiles = [np.ones((4, 4))]
    assert parallel.unit_flux_path(scene, objects)                 # no object samples the image
    stamp_obj = objects.copy()
    stamp_obj["prof_table"][5] = _abi.IMS_PROF_IMAGE
    assert not parallel.unit_flux_path(scene, stamp_obj) and not parallel.unit_flux_path(scene)
    scene.image_interpolant = "nearest"
    assert parallel.unit_flux_path(scene, stamp_obj)
    scene.ops = list(scene.ops) + [(_abi.IMS_OP_BANDPASS_RATIO, 0, [])]
    assert not parallel.unit_flux_path(scene, objects)


def _pooling_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import c3_small_case
    from imsim_amd import parallel, photon_pooling, stamp
    from oracle import orc_loader
    scene, objects = c3_small_case(n_obj=60, n=128, flux_seed=5, scratch=0)
    scene.track_static_delta = 1
    orc = orc_loader.OracleScene(scene)
    modes = stamp.classify(objects["n_phot"].astype(float), 100.0)
    n = photon_pooling.build_image(orc, objects, modes, nbatch=4, nsubbatch=3, seed=11, rank=rank, world=world)
    img = torch.from_numpy(orc.image64.copy())
    parallel.reduce_image(img, dst=0)
    tot = torch.tensor([n])
    dist.all_reduce(tot)
    if rank == 0:
        np.savez(out_path, image=img.numpy(), boundary=orc.sensor_array("boundary"), photons=tot.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_photon_pooling_with_brighter_fatter_equals_single_process(tmp_path, world):
    """in pooling mode all objects share the sensor state; the ranks (2 and 8) all-reduce the delta
    charge before every recalculation (imsim/photon_pooling.py:159) and must end with the single-process image and boundaries."""
    from helpers import c3_small_case
    from imsim_amd import photon_pooling, stamp
    from oracle import orc_loader
    out = str(tmp_path / "pooled.npz")
    from helpers import free_port
    port = free_port()
    mp.start_processes(_pooling_worker, args=(world, port, out), nprocs=world, join=True, start_method="spawn")
    res = np.load(out)
    scene, objects = c3_small_case(n_obj=60, n=128, flux_seed=5, scratch=0)
    scene.track_static_delta = 1
    orc = orc_loader.OracleScene(scene)
    modes = stamp.classify(objects["n_phot"].astype(float), 100.0)
    n = photon_pooling.build_image(orc, objects, modes, nbatch=4, nsubbatch=3, seed=11)
    assert int(res["photons"][0]) == n == int(objects["n_phot"].sum())
    assert np.array_equal(res["image"], orc.image64)
    assert np.array_equal(res["boundary"], orc.sensor_array("boundary"))
    # brighter-fatter did act: the boundaries differ from a run without recalculation
    ref = orc_loader.OracleScene(scene)
    photon_pooling.build_image(ref, objects, modes, nbatch=1, nsubbatch=3, seed=11)
    assert not np.array_equal(ref.sensor_array("boundary"), orc.sensor_array("boundary"))


def test_ccds_are_dealt_round_robin():
    from imsim_amd import parallel
    dets = list(range(189))
    parts = [parallel.shard_ccds(dets, r, 8) for r in range(8)]
    assert sorted(sum(parts, [])) == dets
    assert max(len(p) for p in parts) == 24 and min(len(p) for p in parts) == 23       # ceil(189 / 8)
    assert parts[3][:3] == [3, 11, 19]
    assert parallel.shard_ccds(dets, 0, 1) == dets


def _focal_ccd(det, n=128, n_obj=40):
    """one small CCD of a focal plane: its own catalog, seed and photon streams (what configs._c5_step deals per CCD)"""
    import copy
    from imsim_amd import configs, catalog
    from imsim_amd.config import ccd_seed
    base = configs.scene_c3(nx=n, ny=n)
    base.sensor.scratch_cells = 200_000
    sc = copy.copy(base)
    sc.seed = base.seed if det == 0 else ccd_seed(base.seed, det)
    cat = catalog.synthetic_catalog(n_obj, seed=20261001 + det, nx=n, ny=n)
    phot = catalog.realize_fluxes(cat["nominal_flux"], sc.seed)
    objects, _ = configs.c3_objects(cat, phot, sc)
    return sc, objects


def _focal_worker(rank, world, port, out_path, n_ccd):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from imsim_amd import parallel
    from oracle import orc_loader
    mine = parallel.shard_ccds(list(range(n_ccd)), rank, world)
    images = {}
    for det in mine:
        sc, objects = _focal_ccd(det)
        orc = orc_loader.OracleScene(sc)
        orc.render_lsst_image(objects, nrecalc=2000)
        images[det] = orc.image64.copy()
    box = [None] * world
    dist.gather_object(images, box if rank == 0 else None, dst=0)       # stands for the per-CCD files: no exchange in the data path
    if rank == 0:
        merged = {}
        for part in box:
            assert not (set(part) & set(merged))
            merged.update(part)
        np.savez(out_path, **{f"ccd{d}": v for d, v in merged.items()}, owners=np.array([len(p) for p in box]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_focal_plane_ccd_deal_equals_single_process(tmp_path, world):
    """BASELINE config 5's partition in small (imsim/ccd.py:72-89: CCDs are independent jobs): CCD i -> rank i mod world, every
    CCD with its own catalog and seed; whichever rank renders a CCD, its image is the one a single process renders."""
    from helpers import free_port
    from oracle import orc_loader
    n_ccd = 11                                             # not a multiple of 8: ranks own 2 or 1 CCDs
    out = str(tmp_path / "focal.npz")
    mp.start_processes(_focal_worker, args=(world, free_port(), out, n_ccd), nprocs=world, join=True, start_method="spawn")
    res = np.load(out)
    assert res["owners"].sum() == n_ccd and res["owners"].max() - res["owners"].min() <= 1
    sums = []
    for det in range(n_ccd):
        sc, objects = _focal_ccd(det)
        orc = orc_loader.OracleScene(sc)
        orc.render_lsst_image(objects, nrecalc=2000)
        assert np.array_equal(res[f"ccd{det}"], orc.image64), det
        sums.append(orc.image64.sum())
    assert len(set(sums)) == n_ccd                        # the CCDs do differ (own catalogs and seeds)


def test_bench_launcher_stops_the_other_ranks_when_one_dies():
    """`bench.py --gpus 2` starts its own ranks; a rank that exits (here: before the rendezvous) must end the run at once
    with a non-zero code instead of leaving rank 0 in init_process_group until the collective timeout."""
    import subprocess
    import time
    env = dict(os.environ, IMS_BENCH_FAIL_RANK="1")
    env.pop("WORLD_SIZE", None)
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline", "--no-cold"], env=env, capture_output=True, timeout=120)
    assert p.returncode != 0
    assert time.time() - t0 < 60.0
    assert b"rank 1 exited with code 3" in p.stderr
