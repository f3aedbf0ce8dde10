"""The N > 1 path through the HIP Renderer: two ranks on ONE GPU (gloo for the exchanges, which RCCL refuses to
do between two ranks of one device) must reproduce the single-rank image bit for bit, both for the LSST_Image
object sharding + image reduce (C3) and for photon pooling with the delta-charge all-reduce before every
recalculation (C4), and `bench.py --gpus 2` must start its own ranks and report n_gpus 2."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _worker(rank, world, port, out_path, mode):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import c3_small_case
    from imsim_amd import parallel, photon_pooling, stamp
    from imsim_amd.engine import Renderer
    if mode == "lsst_image":
        scene, objects = c3_small_case(n_obj=300, n=512, flux_seed=3)
        r = Renderer(scene, "cuda:0")
        step = r.prepared_lsst_image(parallel.shard_objects(objects, rank, world), nrecalc=2000)
    else:
        scene, objects = c3_small_case(n_obj=400, n=256, flux_seed=5, scratch=0)
        scene.track_static_delta = 1
        r = Renderer(scene, "cuda:0")
        modes = stamp.classify(objects["n_phot"].astype(float), 100.0)
        step = photon_pooling.prepared_image(r, objects, modes, nbatch=5, seed=11, rank=rank, world=world)
    for _ in range(2):                       # replayable: the second pass must give the same image
        r.image.zero_()
        step()
        parallel.reduce_image(r.image, dst=0, integer_counts=True)
    torch.cuda.synchronize()
    if rank == 0:
        np.savez(out_path, image=r.image.cpu().numpy(), boundary=r.bound.sensor_arrays["boundary"].cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def _run(tmp_path, mode):
    import torch.multiprocessing as mp
    out = str(tmp_path / f"{mode}.npz")
    from helpers import free_port
    port = free_port()
    mp.start_processes(_worker, args=(2, port, out, mode), nprocs=2, join=True, start_method="spawn")
    return np.load(out)


def test_two_ranks_lsst_image_reduce_equals_single_rank(tmp_path):
    from helpers import c3_small_case
    from imsim_amd.engine import Renderer
    from oracle import orc_loader
    res = _run(tmp_path, "lsst_image")
    scene, objects = c3_small_case(n_obj=300, n=512, flux_seed=3)
    r = Renderer(scene, "cuda:0")
    r.prepared_lsst_image(objects, nrecalc=2000)()
    r.synchronize()
    single = r.image.cpu().numpy()
    assert single.sum() > 0 and (objects["n_phot"] > 2000).sum() >= 2       # brighter-fatter chains are exercised
    assert np.array_equal(res["image"], single)
    orc = orc_loader.OracleScene(scene)
    orc.render_lsst_image(objects, nrecalc=2000)
    assert np.array_equal(res["image"], orc.image64)


def test_two_ranks_photon_pooling_allreduce_equals_single_rank(tmp_path):
    from helpers import c3_small_case
    from imsim_amd import photon_pooling, stamp
    from imsim_amd.engine import Renderer
    res = _run(tmp_path, "pooling")
    scene, objects = c3_small_case(n_obj=400, n=256, flux_seed=5, scratch=0)
    scene.track_static_delta = 1
    modes = stamp.classify(objects["n_phot"].astype(float), 100.0)
    r = Renderer(scene, "cuda:0")
    photon_pooling.prepared_image(r, objects, modes, nbatch=5, seed=11)()
    r.synchronize()
    assert np.array_equal(res["image"], r.image.cpu().numpy())
    # every rank ran the identical updatePixelDistortions on the summed charge: same boundaries as one rank
    assert np.array_equal(res["boundary"], r.bound.sensor_arrays["boundary"].cpu().numpy())
    # and the recalculations did act
    r0 = Renderer(scene, "cuda:0")
    photon_pooling.prepared_image(r0, objects, modes, nbatch=1, seed=11)()
    r0.synchronize()
    assert not np.array_equal(r0.bound.sensor_arrays["boundary"].cpu().numpy(), res["boundary"])


@pytest.mark.parametrize("config", ["c3", "c4"])
def test_bench_starts_its_own_ranks(config):
    """`python bench.py --gpus 2` with no launcher: the parent starts two ranks (on this one-GPU box a dry run over
    gloo) and relays rank 0's JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--config", config, "--n-objects", "3000", "--no-cpu-baseline"], env=env, capture_output=True,
                       text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["shared_gpu"] is True


def _bench_dump(tmp_path, config, gpus, extra_env=None, n_objects=3000):
    out = tmp_path / f"dump_{config}_{gpus}.{'json' if config == 'c5' else 'npz'}"
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(IMS_BENCH_DUMP=str(out), IMS_BENCH_SHARE_GPU="1", **(extra_env or {}))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "1", "--warmup", "0",
                        "--config", config, "--n-objects", str(n_objects), "--no-cpu-baseline", "--no-cold"], env=env,
                       capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == gpus and line["value"] > 0
    if gpus > 1:
        assert line["shared_gpu"] is True            # a dry run of the multi-rank logic on ONE GPU: says nothing about scaling
    return out, line


@pytest.mark.parametrize("config", ["c3", "c4"])
def test_eight_ranks_reduce_to_the_one_rank_image(tmp_path, config):
    """The node size BASELINE.json names, as a dry run on one GPU (eight rank processes on cuda:0, exchanges over gloo): C3's
    objects dealt by `parallel.assign_ranks` to eight ranks and summed by the int32 image reduce, C4's delta-charge all-reduce
    before every recalculation (imsim/photon_pooling.py:159) between eight ranks -- the image on rank 0 equals the one-rank image
    bit for bit, and every rank's own image passed integer_counts_ok(image, 8) (bench.py raises otherwise)."""
    import torch
    from imsim_amd import parallel
    one, _ = _bench_dump(tmp_path, config, 1)
    eight, line = _bench_dump(tmp_path, config, 8)
    a, b = np.load(one), np.load(eight)
    assert a["image"].sum() > 0
    assert np.array_equal(a["image"], b["image"])
    assert bool(b["integer_counts_ok"][0])
    assert parallel.integer_counts_ok(torch.from_numpy(a["image"]), 8)
    assert "over 8 rank(s)" in line["config"]["sharding"]


def test_eight_ranks_deal_the_ccds_of_a_focal_plane(tmp_path):
    """BASELINE config 5's partition with eight ranks (imsim/ccd.py:72-89: CCD i -> rank i mod 8, no exchange), dry run on one GPU:
    sixteen CCDs of the bench visit (FFT-drawn, photon-shot and faint objects, the bright tail) -- every CCD's float32 image has
    the CRC it has when one rank renders them all, and the ranks own two CCDs each."""
    env = {"IMS_C5_CCDS": "16"}
    one, _ = _bench_dump(tmp_path, "c5", 1, env, n_objects=189 * 400)
    eight, line = _bench_dump(tmp_path, "c5", 8, env, n_objects=189 * 400)
    a = json.load(open(one))["ccd_crc32_and_rank"]
    b = json.load(open(eight))["ccd_crc32_and_rank"]
    assert sorted(a) == sorted(b) == sorted(str(d) for d in range(16))
    assert {d: v[0] for d, v in a.items()} == {d: v[0] for d, v in b.items()}
    assert len(set(v[0] for v in a.values())) == 16                 # the CCDs differ
    assert all(v[1] == int(d) % 8 for d, v in b.items())            # CCD i -> rank i mod 8


_RCCL_SCRIPT = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import torch
import torch.distributed as dist
from imsim_amd import parallel
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
torch.cuda.set_device(0)
assert dist.get_backend() == "nccl"
g = torch.Generator(device="cuda"); g.manual_seed(3)
counts = torch.randint(0, 70000, (4096, 4096), device="cuda", generator=g).to(torch.float64)
want = counts.clone()
parallel.reduce_image(counts, dst=0, integer_counts=True)            # int32 copy -> RCCL reduce -> back
assert torch.equal(counts, want)
real = torch.rand((512, 512), dtype=torch.float64, device="cuda", generator=g) * 3.7
want = real.clone()
parallel.reduce_image(real, dst=0)                                   # f64 RCCL reduce
assert torch.equal(real, want)
delta = torch.randint(0, 500, (1025 * 1025,), device="cuda", generator=g).to(torch.float64)
want = delta.clone()
parallel.allreduce_delta(delta, integer_counts=True)                 # int32 RCCL all-reduce
parallel.allreduce_delta(delta)                                      # f64 RCCL all-reduce
assert torch.equal(delta, want)
dist.barrier()
t = torch.tensor([1.25], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)                             # bench.py's timing reduction
assert float(t.item()) == 1.25
torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL_OK")
"""


def test_rccl_executes_every_exchange_of_the_path_with_one_rank(tmp_path):
    """The one GPU of the test box cannot hold two RCCL ranks, so the two-rank tests above exchange over gloo.  This one
    runs the nccl (= RCCL) backend itself: a one-rank process group bound to cuda:0, and every collective the path issues
    -- the int32 and the f64 form of the CCD image reduce, both forms of the delta-charge all-reduce, the barrier and the
    MAX all-reduce of bench.py's timing -- as self-exchanges (IMS_EXCHANGE_SINGLE_RANK=1).  What it cannot show is the
    transport between GPUs."""
    from helpers import free_port
    script = tmp_path / "rccl_one_rank.py"
    script.write_text(_RCCL_SCRIPT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), IMS_EXCHANGE_SINGLE_RANK="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "RCCL_OK" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]


_LIB_RCCL_SCRIPT = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import torch
from imsim_amd import parallel
torch.cuda.set_device(0)
comm = parallel.LibraryComm(0, 1, "cuda:0")                            # ims_comm_unique_id + ims_comm_init inside libimsim_hip.so
parallel.install(comm)
assert parallel.exchanging(1)
g = torch.Generator(device="cuda"); g.manual_seed(3)
counts = torch.randint(0, 70000, (4096, 4096), device="cuda", generator=g).to(torch.float64)
want = counts.clone()
assert comm.count_inexact(counts) == 0
parallel.reduce_image(counts, dst=0, integer_counts=True)              # f64 -> int32, ncclReduce, -> f64, all in the library
assert torch.equal(counts, want)
bad = counts.clone(); bad[5, 7] = 0.5; bad[9, 9] = 3.0e9; bad[1, 1] = -1.0
assert comm.count_inexact(bad) == 3
real = torch.rand((512, 512), dtype=torch.float64, device="cuda", generator=g) * 3.7
want = real.clone()
parallel.reduce_image(real, dst=0)                                     # f64 ncclReduce
assert torch.equal(real, want)
delta = torch.randint(0, 500, (1025 * 1025,), device="cuda", generator=g).to(torch.float64)
want = delta.clone()
parallel.allreduce_delta(delta, integer_counts=True)                   # int32 ncclAllReduce
parallel.allreduce_delta(delta)                                        # f64 ncclAllReduce
assert torch.equal(delta, want)
torch.cuda.synchronize()
parallel.install(None)
comm.destroy()
print("LIB_RCCL_OK")
"""


def test_the_librarys_own_rccl_communicator_runs_every_exchange_with_one_rank(tmp_path):
    """SURVEY 8(b) lists the exchanges in the C-ABI: ims_comm_unique_id / ims_comm_init / ims_reduce_image /
    ims_allreduce_delta / ims_count_inexact call RCCL from inside libimsim_hip.so (looked up at run time, the copy torch
    loaded), no torch.distributed in the data path.  One rank, self-exchanges: the int32 and f64 forms of both collectives
    leave the data as it was, the exactness check counts what an int32 exchange would not carry."""
    script = tmp_path / "lib_rccl_one_rank.py"
    script.write_text(_LIB_RCCL_SCRIPT)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "LIB_RCCL_OK" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]
