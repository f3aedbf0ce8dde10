"""The N>1 path on CPU: two gloo ranks shard the catalog, each renders its share (with the oracle,
since the product has no CPU path), the images are sum-reduced onto rank 0 and must equal the
single-process render bit for bit (shard invariance of the counter-addressed random streams)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import small_case
    from imsim_amd import parallel
    from oracle import orc_loader
    scene, objects, _ = small_case(n_obj=120, nx=256, ny=256, flux_seed=9)
    mine = parallel.shard_objects(objects, rank, world)
    counts = torch.tensor([len(mine), int(mine["n_phot"].sum())])
    gathered = [torch.zeros_like(counts) for _ in range(world)]
    dist.all_gather(gathered, counts)
    orc = orc_loader.OracleScene(scene)
    orc.render(mine)
    img = torch.from_numpy(orc.image64.copy())
    img_i = img.clone()
    parallel.reduce_image(img, dst=0)
    parallel.reduce_image(img_i, dst=0, integer_counts=True)        # the int32 exchange bench.py uses
    if rank == 0:
        np.savez(out_path, image=img.numpy(), image_int=img_i.numpy(), counts=torch.stack(gathered).numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_render_equals_single_process(tmp_path):
    from helpers import small_case
    from oracle import orc_loader
    out = str(tmp_path / "reduced.npz")
    from helpers import free_port
    port = free_port()
    mp.start_processes(_worker, args=(2, port, out), nprocs=2, join=True, start_method="spawn")
    res = np.load(out)
    scene, objects, _ = small_case(n_obj=120, nx=256, ny=256, flux_seed=9)
    orc = orc_loader.OracleScene(scene)
    orc.render(objects)
    assert np.array_equal(res["image"], orc.image64)
    assert np.array_equal(res["image_int"], orc.image64) and res["image_int"].dtype == np.float64
    counts = res["counts"]
    assert counts[:, 0].sum() == len(objects)
    assert counts[:, 1].sum() == objects["n_phot"].sum()
    # the load is balanced up to the weight of the single heaviest (indivisible) object
    assert abs(counts[0, 1] - counts[1, 1]) <= objects["n_phot"].max()


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


def test_two_rank_photon_pooling_with_brighter_fatter_equals_single_process(tmp_path):
    """in pooling mode all objects share the sensor state; the ranks all-reduce the delta
    charge before every recalculation and must end with the single-process image and boundaries."""
    from helpers import c3_small_case
    from imsim_amd import photon_pooling, stamp
    from oracle import orc_loader
    out = str(tmp_path / "pooled.npz")
    from helpers import free_port
    port = free_port()
    mp.start_processes(_pooling_worker, args=(2, port, out), nprocs=2, join=True, start_method="spawn")
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
