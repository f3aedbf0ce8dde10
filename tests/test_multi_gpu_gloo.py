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
    img = torch.from_numpy(orc.image.copy())
    parallel.reduce_image(img, dst=0)
    if rank == 0:
        np.savez(out_path, image=img.numpy(), counts=torch.stack(gathered).numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_render_equals_single_process(tmp_path):
    from helpers import small_case
    from oracle import orc_loader
    out = str(tmp_path / "reduced.npz")
    port = 29500 + (os.getpid() % 2000)
    mp.start_processes(_worker, args=(2, port, out), nprocs=2, join=True, start_method="spawn")
    res = np.load(out)
    scene, objects, _ = small_case(n_obj=120, nx=256, ny=256, flux_seed=9)
    orc = orc_loader.OracleScene(scene)
    orc.render(objects)
    assert np.array_equal(res["image"], orc.image)
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
