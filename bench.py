#!/usr/bin/env python
"""bench.py -- objects/sec into one 4k x 4k LSST CCD (photon-shooting path), BASELINE.json's metric.

  python bench.py --gpus N --steps K --warmup W [--config c2|c3] [--no-cpu-baseline]

One step = one pass of the hot path over the whole synthetic instance catalog with
the object table already resident in HBM.  For N > 1 (launched by torch.distributed.run, one rank
per GPU) the objects are dealt round-robin by flux to the ranks, each rank renders its share into
its own CCD image and the images are summed onto rank 0 with an RCCL reduce inside the timed
region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s peak
OBJECT_ROW_BYTES = 256    # ims_object_t
IMAGE_RMW_BYTES = 8       # one fp32 read + write per photon


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default=os.environ.get("IMSIM_BENCH_CONFIG", "c3"))
    ap.add_argument("--n-objects", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="objects in the CPU-baseline sample")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from imsim_amd import configs, catalog, _abi, parallel
    from imsim_amd.engine import Renderer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # IMS_BENCH_SHARE_GPU=1 is a dry run of the multi-rank path on a box with ONE GPU (all ranks on cuda:0, gloo instead
    # of RCCL, which refuses two ranks on one device): it exercises sharding, barriers and the timing reduction only.
    share_gpu = os.environ.get("IMS_BENCH_SHARE_GPU", "0") == "1"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    device = "cuda:0" if share_gpu else f"cuda:{local_rank}"
    torch.cuda.set_device(device)

    cfg = configs.BENCH_CONFIGS[args.config]
    n_obj = args.n_objects or cfg["n_objects"]
    scene = cfg["scene"]()
    cat = catalog.synthetic_catalog(n_obj, nx=scene.nx, ny=scene.ny)
    phot = catalog.realize_fluxes(cat["nominal_flux"], scene.seed)
    objects, sizes = cfg["objects"](cat, phot, scene)
    mine = parallel.shard_objects(objects, rank, world)

    renderer = Renderer(scene, device)
    step = cfg["make_step"](renderer, mine)
    lib = _abi.load()

    def full_step():
        renderer.image.zero_()
        step()
        # unit photon fluxes: every pixel is an integer count; the brightest pixel of this catalog holds ~1e8 < 2^31
        parallel.reduce_image(renderer.image, dst=0, integer_counts=True)

    for _ in range(args.warmup):
        full_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    lib.ims_enable_timing(cfg["timed_kernel"])
    t0 = time.perf_counter()
    for _ in range(args.steps):
        full_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ms, nl = _abi.C.c_float(), _abi.C.c_int()
    have_ms = lib.ims_last_kernel_ms(_abi.C.byref(ms), _abi.C.byref(nl)) == 0
    lib.ims_enable_timing(0)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    n_total_obj = len(objects)
    n_total_phot = int(objects["n_phot"].sum())
    ms_per_step = 1e3 * elapsed / args.steps
    value = n_total_obj * args.steps / elapsed

    # roofline of the dominant kernel of this workload (configs.py names it): its launches are bracketed
    # by hipEvent pairs inside the library, on the stream they run on; achieved = algorithmic bytes per
    # launch / mean launch duration.  `traffic` = HBM bytes per launch from the committed rocprofv3 PMC
    # passes of the same command (profiles/hbm_traffic.json), corrected as MI355X_MICROARCH.md prescribes.
    launches_per_step, algo_bytes_step = step.timed[cfg["timed_kernel"]]
    n_launch = max(int(nl.value), 1) if have_ms else 1
    bytes_per_launch = algo_bytes_step / max(launches_per_step, 1)
    k_ms = float(ms.value) / n_launch if have_ms else float("nan")
    achieved = bytes_per_launch / (k_ms * 1e-3) / 1e9 if have_ms else float("nan")
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": hbm_traffic(args.config, cfg["kernel"], world),
                "kernel": cfg["kernel"], "mean_launch_ms": k_ms, "timed_launches_per_step": launches_per_step,
                "kernel_ms_per_step": float(ms.value) / args.steps if have_ms else None,
                "algorithmic_bytes_per_launch": bytes_per_launch, "photons_per_step": step.photons,
                "limiter": "f64 VALU issue (rocprofv3 PMC, profiles/round1_c3_final_sq_pmc.txt: VALU busy 89 % of the "
                           "SIMD cycles of this kernel); HBM is the stated bound of SURVEY 8(d), not the measured one"}

    out = {
        "metric": "objects/sec into one 4k x 4k LSST CCD (photon-shooting path)",
        "value": value, "unit": "objects/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": cfg["workload"], "n_objects": n_total_obj, "n_photons": n_total_phot,
                   "image": [scene.nx, scene.ny], "sharding": f"objects round-robin by flux over {world} rank(s)"},
        "photons_per_s": n_total_phot * args.steps / elapsed,
        "roofline": roofline,
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg, scene, objects, args.cpu_sample, device)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def hbm_traffic(config, kernel, world):
    """HBM bytes per launch of `kernel` measured with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate
    passes) on this workload at one GPU; None when no committed measurement matches."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "hbm_traffic.json")
    if world != 1 or not os.path.exists(path):
        return None
    with open(path) as fh:
        table = json.load(fh)
    entry = table.get(config, {}).get(kernel)
    return float(entry["hbm_bytes_per_launch"]) if entry else None


def cpu_baseline(cfg, scene, objects, n_sample, device):
    """The oracle ("port") timed on one host core over a bounded sample of the same workload; the same sample is then
    rendered by a fresh GPU renderer and the two CCD images are compared pixel by pixel (parity check
    inside the measurement run)."""
    import torch
    from imsim_amd.engine import Renderer
    from oracle import orc_loader
    n_sample = n_sample or cfg["cpu_sample"]
    rng = np.random.default_rng(99)
    idx = np.sort(rng.choice(len(objects), size=min(n_sample, len(objects)), replace=False))
    sample = objects[idx]
    orc = orc_loader.OracleScene(cfg["cpu_scene"](scene))
    t0 = time.perf_counter()
    cfg["cpu_step"](orc, sample)
    dt = time.perf_counter() - t0
    gpu = Renderer(cfg["cpu_scene"](scene), device)
    cfg["make_step"](gpu, sample)()
    torch.cuda.synchronize()
    got, want = gpu.image_numpy(), orc.image
    parity = {"checked": "float32 CCD image of the CPU sample, GPU vs oracle, every pixel",
              "pixels": int(want.size), "nonzero_pixels": int(np.count_nonzero(want)),
              "bit_identical": bool(np.array_equal(got.view(np.uint32), np.asarray(want, dtype=np.float32).view(np.uint32)))}
    del gpu
    return {"parity": parity, "value": len(sample) / dt, "unit": "objects/s", "cores": 1, "kind": "port",
            "sample": f"{len(sample)} objects drawn at random from the same catalog "
                      f"({int(sample['n_phot'].sum())} photons, {dt:.1f} s)",
            "photons_per_s": float(sample["n_phot"].sum()) / dt,
            "host_cpus": os.cpu_count()}


if __name__ == "__main__":
    main()
