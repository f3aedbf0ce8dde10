#!/usr/bin/env python
"""bench.py -- objects/sec into one 4k x 4k LSST CCD (photon-shooting path), BASELINE.json's metric.

  python bench.py --gpus N --steps K --warmup W [--config c2|c3|c3b|c4|c5|fft|fftx|fftxs] [--no-cpu-baseline] [--no-extra-configs]

One step = one pass of the hot path over the whole synthetic instance catalog with
the object table already resident in HBM.  For N > 1 there is one rank per GPU: either started by
`torch.distributed.run` (RANK / LOCAL_RANK / WORLD_SIZE in the environment), or -- when bench.py is
started plainly with --gpus N -- by this script itself, which starts the N rank processes BEFORE
anything touches the GPU and relays rank 0's JSON line.  Objects are dealt to the ranks by photon count,
each rank renders its share into its own CCD image and the images are summed onto rank 0 with an RCCL
reduce inside the timed region (photon-pooling mode, --config c4: plus an all-reduce of the delta-charge
image before every pixel-boundary recalculation).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s peak
# f64 vector issue peak: 256 CUs x 4 SIMDs x 16 lanes per cycle x 2.4 GHz (an f64 wave64 instruction issues over 4 cycles)
F64_LANE_OPS_PEAK = 256 * 4 * 16 * 2.4e9
# SIMD time of one wavefront instruction by class [ns], tools/dbg/instr_rate.hip on MI355X (EXPERIMENTS.md, round 2): the weights of
# roofline.step.floor_ms_class_weighted.  Keys = the SQ_INSTS_VALU_* counters of the class pass.
CLASS_NS = {"fma_f64": 2.43, "mul_f64": 2.19, "add_f64": 1.98, "trans_f64": 6.8, "int64": 1.8, "cvt": 1.8, "int32": 1.45, "other": 1.4}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default=os.environ.get("IMSIM_BENCH_CONFIG", "c3"))
    ap.add_argument("--n-objects", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-allcore", action="store_true", help="skip the all-core CPU leg")
    ap.add_argument("--no-cold", action="store_true", help="skip the cold (plan + upload + run) render timing")
    ap.add_argument("--cpu-sample", type=int, default=0, help="objects in the one-core CPU-baseline sample")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="skip the short runs of the other BASELINE configs (extra.configs of the default c3 line)")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes.  Nothing in this
    (parent) process initialises the GPU: torch.cuda.device_count() only counts devices.  With fewer visible
    GPUs than ranks the run is a DRY RUN of the multi-rank path (all ranks on cuda:0, gloo instead of RCCL): it is
    flagged in the JSON line (`shared_gpu`) and says nothing about scaling."""
    import socket
    import torch
    n = args.gpus
    ndev = torch.cuda.device_count()
    share = ndev < n or os.environ.get("IMS_BENCH_SHARE_GPU", "0") == "1"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        if share:
            env["IMS_BENCH_SHARE_GPU"] = "1"
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    # watchdog: a rank that dies (say in init_process_group) must not leave the others waiting for the collective timeout.
    # Rank 0's output is drained by a thread; the children are polled, and the first non-zero exit ends the rest.
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            rc = p.poll()
            if rc is not None and rc != 0:
                failed = (r, rc)
                break
        time.sleep(0.2)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.time() + 5.0
        for p in procs:
            try:
                p.wait(timeout=max(t_end - time.time(), 0.1))
            except subprocess.TimeoutExpired:
                p.kill()
        print(f"bench.py: rank {failed[0]} exited with code {failed[1]}; the other ranks were stopped", file=sys.stderr)
    rcs = [p.wait() for p in procs]
    reader.join(timeout=5.0)
    if failed is None:
        sys.stdout.write(b"".join(c for c in chunks if c).decode())
        sys.stdout.flush()
    return max(abs(rc) for rc in rcs) or (1 if failed is not None else 0)


# The other BASELINE.json configs on the same record as the headline: each as a CHILD process of its own
# -- `bench.py --config X --steps K --warmup W` with a small one-core oracle sample, whose image the child's GPU renderer must
# reproduce (the child's cpu_baseline.parity) -- started BEFORE this process touches the GPU, one after the other.
# (name, objects of the one-core oracle sample, environment, timed steps, warm-up steps: the sub-millisecond configs take more steps,
# and the FFT branch three warm-ups -- a (size, batch) pair that comes again gets its batched hipFFT plan, milliseconds the second time)
# (C5 is the WHOLE visit of 189 CCDs -- rounds 4 and 5 ran its first 32 here; `fftx` is the FFT branch at throughput: 5 000 stars on
# 1024^2 .. 4096^2 grids with the spike stencil)
EXTRA_CONFIGS = (("c2", 2000, {}, 10, 2), ("c3b", 1500, {}, 2, 1), ("c4", 1500, {}, 2, 1), ("c5", 1200, {}, 2, 1),
                 ("fft", 0, {}, 10, 3), ("fftx", 2, {}, 2, 1), ("fftxs", 2, {}, 1, 1))


def extra_configs(budget_s=420.0):
    out = {}
    t_start = time.perf_counter()
    for name, sample, env_add, n_steps, n_warm in EXTRA_CONFIGS:
        left = budget_s - (time.perf_counter() - t_start)
        if left < 10.0:
            out[name] = {"skipped": "time budget of the extra configs used up"}
            continue
        cmd = [sys.executable, os.path.abspath(__file__), "--config", name, "--steps", str(n_steps), "--warmup", str(n_warm), "--no-cold",
               "--no-cpu-allcore", "--no-extra-configs"]
        if sample:
            cmd += ["--cpu-sample", str(sample)]
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "IMSIM_BENCH_CONFIG")}
        env.update(env_add)
        t0 = time.perf_counter()
        try:
            p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=left)
            line = json.loads(p.stdout.strip().splitlines()[-1]) if p.returncode == 0 and p.stdout.strip() else None
        except subprocess.TimeoutExpired:
            p, line = None, None
        if line is None:
            out[name] = {"failed": True, "returncode": p.returncode if p is not None else "timeout",
                         "stderr_tail": (p.stderr[-400:] if p is not None else "")}
            continue
        par = (line.get("cpu_baseline") or {}).get("parity") or {}
        entry = {"ms_per_step": line["ms_per_step"], "objects_per_s": line["value"], "photons_per_s": line.get("photons_per_s"),
                 "steps": line["steps"], "warmup": line["warmup"],
                 "bit_identical": par.get("bit_identical"), "parity_sample": (line.get("cpu_baseline") or {}).get("sample"),
                 "cpu_oracle_objects_per_s": (line.get("cpu_baseline") or {}).get("value"),
                 "workload": line["config"]["workload"], "n_objects": line["config"]["n_objects"],
                 "wall_s": time.perf_counter() - t0}
        # the child's own roofline figures: the whole step against its VALU issue floors (flat and class-weighted), the dominant
        # kernel against HBM, the FFT branch's bytes against HBM
        rf = line.get("roofline") or {}
        if rf.get("step"):
            entry["roofline_step"] = {k: rf["step"].get(k) for k in ("valu_wave_insts", "floor_ms", "frac", "floor_ms_class_weighted",
                                                                     "frac_class_weighted", "sustained_clock_ghz", "source")}
        entry["dominant_kernel"] = {"kernel": rf.get("kernel"), "hbm_frac": rf.get("frac"), "achieved_GBps": rf.get("achieved"),
                                    "traffic_bytes_per_launch": rf.get("traffic")}
        if rf.get("branch"):
            entry["branch"] = rf["branch"]
        if rf.get("phases"):
            entry["phases"] = rf["phases"]
        if "within_tolerance" in par:          # FFT-drawn stamps: library transforms agree to ~1e-11 of the peak (parity_mode close)
            entry["within_tolerance"] = par["within_tolerance"]
            entry["differing_pixels"] = par.get("differing_pixels")
        if env_add:
            entry["env"] = env_add
        out[name] = entry
    return out


JSON_FD = [1]


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    # read by the HSA runtime when the GPU is first touched (dmabuf IPC: RCCL between processes needs it on this pool)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # keep stdout for the JSON line: everything else this process (and the libraries it loads) prints goes to stderr
    sys.stdout.flush()
    JSON_FD[0] = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("IMS_BENCH_FAIL_RANK") == str(rank) and world > 1:
        sys.exit(3)                     # test hook of the launcher's watchdog (tests/test_multi_gpu_gloo.py)
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; the launcher's world size is used", file=sys.stderr)

    import torch
    import torch.distributed as dist
    from imsim_amd import configs, catalog, _abi, parallel

    cfg = configs.BENCH_CONFIGS[args.config]
    # The CPU legs run FIRST, before this process initialises the GPU (the all-core leg forks workers).  Building the
    # scene / catalog is host-only, except C3b whose phase screens are generated on the GPU.
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    cpu_first = want_cpu and not cfg.get("scene_needs_gpu", False)

    # IMS_BENCH_SHARE_GPU=1 is a dry run of the multi-rank path on a box with ONE GPU (all ranks on cuda:0, gloo instead
    # of RCCL, which refuses two ranks on one device): it exercises sharding, exchanges and the timing reduction only.
    share_gpu = os.environ.get("IMS_BENCH_SHARE_GPU", "0") == "1"
    device = "cuda:0" if share_gpu else f"cuda:{local_rank}"

    timing = {}

    def build_inputs():
        n_obj = args.n_objects or cfg["n_objects"]
        scene = cfg["scene"]()
        cat = cfg["catalog"](n_obj, scene) if "catalog" in cfg else catalog.synthetic_catalog(n_obj, nx=scene.nx, ny=scene.ny)
        t0 = time.perf_counter()
        phot = catalog.realize_fluxes(cat["nominal_flux"], scene.seed)
        objects, _ = cfg["objects"](cat, phot, scene)
        timing["object_table_ms"] = 1e3 * (time.perf_counter() - t0)
        timing["cat"] = cat
        return scene, objects

    # the other BASELINE configs, each a short child run with its own oracle parity flag -- before this process touches the GPU
    extra_cfg = None
    if (rank == 0 and world == 1 and args.config == "c3" and not args.no_extra_configs and not args.no_cpu_baseline
            and not args.n_objects and os.environ.get("IMS_BENCH_EXTRA", "1") != "0"):
        extra_cfg = extra_configs()

    cpu = None
    if cpu_first:
        scene, objects = build_inputs()
        cpu = cpu_legs(cfg, scene, objects, args)

    torch.cuda.set_device(device)
    from imsim_amd.engine import Renderer
    if cfg.get("focal", False):
        # hipFFT's start-up (a fresh process's first plan of every transform size compiles its kernels: seconds) beside the
        # host's own start-up work, as a visit would do it beside reading its catalogs
        from imsim_amd import focal_plane
        if os.environ.get("IMS_BENCH_EARLY_WARM", "1") != "0":
            focal_plane.warm_fft(device)
    if not cpu_first:
        scene, objects = build_inputs()
    # The renderer -- and with it the plan streams of the device -- comes BEFORE the RCCL communicator: measured with one rank
    # (IMS_BENCH_RCCL_ONE_RANK=1, tools/dbg/rccl_step_time.py), a C3 step takes 35 - 39 ms when the process group is
    # initialised first and 24.0 - 24.3 ms (23.8 without a communicator) when the streams have run something first: HIP
    # maps streams onto four hardware queues in the order they come into use, and a plan stream that shares a queue with
    # another one serialises the step.
    renderer = Renderer(scene, device)
    if not cfg.get("focal", False):
        # (a focal plane's CCDs share four streams that are first used by the first CCD: running something on THOSE ahead
        # of time was measured to hurt -- C5 1 326 -> 2 434 ms -- and C5 is the same 1 326 ms with and without a communicator)
        renderer.touch_streams()

    # IMS_BENCH_RCCL_ONE_RANK=1 (with one rank): the RCCL process group is created and the image reduce of every step runs as a
    # self-exchange -- the nccl code path of the N > 1 run on a box with one GPU (says nothing about the transport)
    one_rank_rccl = world == 1 and os.environ.get("IMS_BENCH_RCCL_ONE_RANK", "0") == "1"
    if one_rank_rccl:
        os.environ["IMS_EXCHANGE_SINGLE_RANK"] = "1"
        if "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as sk:                  # a free port: a fixed one may still be held by the previous run
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or one_rank_rccl:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # IMS_COMM=lib (default): the exchanges of the path run on the library's own RCCL communicator (ims_comm_init /
        # ims_reduce_image / ims_allreduce_delta inside libimsim_hip.so); torch.distributed over gloo only carries the 128-byte
        # id, the barriers and the MAX of the timing.  IMS_COMM=torch: torch.distributed's nccl backend does the exchanges (the
        # checker; the form of rounds 1 - 3).
        use_lib_comm = os.environ.get("IMS_COMM", "lib") == "lib" and not share_gpu
        if share_gpu or use_lib_comm:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        if use_lib_comm:
            lib_comm = parallel.LibraryComm(rank, world, device)
            parallel.install(lib_comm)

    step = cfg["make_step"](renderer, objects, rank, world)
    lib = _abi.load()
    ctl_device = "cpu" if (dist.is_initialized() and dist.get_backend() == "gloo") else device

    unit_flux = parallel.unit_flux_path(scene, objects)

    def full_step():
        renderer.image.zero_()
        step()
        # unit photon fluxes: every pixel is an integer count (checked once, after the timed region: integer_counts_ok)
        if cfg.get("reduce", True):
            parallel.reduce_image(renderer.image, dst=0, integer_counts=unit_flux)

    for _ in range(args.warmup):
        full_step()
    torch.cuda.synchronize()
    if (world > 1 or one_rank_rccl) and unit_flux and cfg.get("reduce", True):
        # the int32 exchange is exact only if EVERY rank's own image is an integer count below 2^31 / world: checked once on the
        # per-rank images (before any reduce touches them), outside the timed region, the verdict agreed by all ranks
        renderer.image.zero_()
        step()
        torch.cuda.synchronize()
        ok = parallel.integer_counts_ok(renderer.image, world)
        flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=ctl_device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if float(flag.item()) != 1.0:
            raise RuntimeError("bench.py: a rank's CCD image is not an integer count below 2^31 / world -- int32 exchange invalid")
    if world > 1 or one_rank_rccl:
        dist.barrier()
    torch.cuda.synchronize()
    lib.ims_enable_timing(cfg["timed_kernel"])
    t0 = time.perf_counter()
    step_ends = []
    for _ in range(args.steps):
        full_step()
        step_ends.append(time.perf_counter())         # (a focal-plane step ends with every image on the host: these are step times)
    torch.cuda.synchronize()
    if world > 1 or one_rank_rccl:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ms, nl = _abi.C.c_float(), _abi.C.c_int()
    have_ms = lib.ims_last_kernel_ms(_abi.C.byref(ms), _abi.C.byref(nl)) == 0
    lib.ims_enable_timing(0)
    if world > 1 and unit_flux and cfg.get("reduce", True) and rank == 0:
        # the int32 exchange is only exact for integer counts whose sum over the ranks fits: the reduced image must qualify
        if not parallel.integer_counts_ok(renderer.image, 1):
            raise RuntimeError("bench.py: the reduced CCD image is not an integer count below 2^31 -- int32 exchange invalid")
    if world > 1 or one_rank_rccl:
        t = torch.tensor([elapsed], dtype=torch.float64, device=ctl_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    dump = os.environ.get("IMS_BENCH_DUMP")
    if dump:
        # test hook (tests/test_multi_gpu_hip.py): what the last step left -- the reduced f64 CCD image on rank 0, or for a focal
        # plane the CRC of every CCD's float32 image from whichever rank rendered it -- so that an N-rank run can be compared with
        # a one-rank run bit for bit
        hashes = getattr(step, "hashes", None)
        if hashes is not None:
            box = [None] * world
            if world > 1:
                dist.gather_object(dict(hashes), box if rank == 0 else None, dst=0)
            else:
                box = [dict(hashes)]
            if rank == 0:
                merged = {}
                for k, part in enumerate(box):
                    for det, h in part.items():
                        merged[str(det)] = [int(h), k]
                with open(dump, "w") as fh:
                    json.dump({"ccd_crc32_and_rank": merged}, fh)
        elif rank == 0:
            np.savez(dump, image=renderer.image.cpu().numpy(),
                     integer_counts_ok=np.array([parallel.integer_counts_ok(renderer.image, 1)]))

    n_total_obj = len(objects)
    fft_mask = getattr(objects, "fft_mask", None)           # a focal plane's FFT-drawn objects shoot no photons
    n_total_phot = int(objects["n_phot"].sum() if fft_mask is None else np.asarray(objects["n_phot"])[~fft_mask].sum())
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
    prof = profile_entry(args.config, cfg["kernel"], world)
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": float(prof["hbm_bytes_per_launch"]) if prof and "hbm_bytes_per_launch" in prof else None,
                "kernel": cfg["kernel"], "mean_launch_ms": k_ms, "timed_launches_per_step": launches_per_step,
                "kernel_ms_per_step": float(ms.value) / args.steps if have_ms else None,
                "algorithmic_bytes_per_launch": bytes_per_launch, "photons_per_step": step.photons}
    # The measured limiter of the photon kernels is f64 VALU issue, not HBM: lane-operations per second of this kernel
    # = (VALU instructions per 64-photon wave, from the committed SQ PMC pass) x 64 lanes x waves per launch / the
    # live launch duration, against the 3.9e13 lane-ops/s at which the chip issues f64 instructions.
    if prof and "valu_insts_per_wave" in prof and have_ms and getattr(step, "timed_waves", None):
        waves_per_launch = step.timed_waves[cfg["timed_kernel"]] / max(launches_per_step, 1)
        lane_ops = prof["valu_insts_per_wave"] * 64.0 * waves_per_launch / (k_ms * 1e-3)
        roofline["f64_valu_issue"] = {"achieved": lane_ops, "peak": F64_LANE_OPS_PEAK, "unit": "lane-ops/s",
                                      "frac": lane_ops / F64_LANE_OPS_PEAK,
                                      "valu_insts_per_wave": prof["valu_insts_per_wave"],
                                      "waves_per_launch": waves_per_launch,
                                      "note": "primary limiter: f64 VALU issue (SQ PMC in profiles/); the launch shares the GPU "
                                              "with the concurrent brighter-fatter chains, so frac is of the whole chip"}
    roofline["limiter"] = ("f64 VALU issue (rocprofv3 SQ PMC in profiles/); HBM is the stated bound of SURVEY 8(d), "
                           "not the measured one")
    # The whole step against its instruction floor: every VALU instruction of every kernel of a step (SQ_INSTS_VALU of the
    # committed SQ pass, profiles/hbm_traffic.json `_step`), issued at one wave-instruction per SIMD every 4 cycles on
    # 1024 SIMDs at 2.4 GHz, against the measured step -- the figure that says how far the step is from its own arithmetic
    pstep = profile_entry(args.config, "_step", world)
    if pstep:
        insts = pstep.get("valu_wave_insts_per_step")
        if insts is None:                       # a focal plane: counted per CCD render on a part of the visit
            insts = pstep["valu_wave_insts_per_ccd"] * getattr(step, "n_ccds", 1)
        floor_ms = insts / (256 * 4 * 2.4e9 / 4.0) * 1e3
        roofline["step"] = {"valu_wave_insts": insts, "floor_ms": floor_ms,
                            "frac": floor_ms / ms_per_step, "ms_per_step": ms_per_step, "source": pstep.get("sq_source"),
                            "note": "VALU issue floor of ALL kernels of one step / the measured step; floor_ms: every wave-instruction "
                                    "4 cycles at 2.4 GHz on 1024 SIMDs; floor_ms_class_weighted: the step's instructions by class "
                                    "(SQ_INSTS_VALU_* pass) x the SIMD time of one wavefront instruction of that class as measured by "
                                    "tools/dbg/instr_rate.hip (f64 fma 2.43 / mul 2.19 / add 1.98 ns, f64 rcp-rsq-sqrt 6.8, 64-bit integer "
                                    "and conversions 1.8, 32-bit integer 1.45, everything else 1.4); sustained_clock_ghz: "
                                    "GRBM_GUI_ACTIVE / wall of the photon kernels in a counter pass of the same command"}
        mix = pstep.get("class_mix_per_step") or (
            {k: v * getattr(step, "n_ccds", 1) for k, v in pstep["class_mix_per_ccd"].items()} if pstep.get("class_mix_per_ccd") else None)
        if mix:
            known = sum(mix.get(k, 0.0) for k in CLASS_NS if k != "other")
            other = max(insts - known, 0.0)
            ns = sum(mix.get(k, 0.0) * CLASS_NS[k] for k in CLASS_NS if k != "other") + other * CLASS_NS["other"]
            fw = ns / 1024.0 * 1e-6
            roofline["step"].update(floor_ms_class_weighted=fw, frac_class_weighted=fw / ms_per_step,
                                    class_mix={**{k: mix.get(k, 0.0) for k in CLASS_NS if k != "other"}, "other": other},
                                    mix_source=pstep.get("mix_source"))
        if pstep.get("sustained_clock_ghz"):
            roofline["step"]["sustained_clock_ghz"] = pstep["sustained_clock_ghz"]
            roofline["step"]["clock_source"] = pstep.get("clock_source")
    phases = profile_entry(args.config, "_phases", world)
    if phases:
        roofline["phases"] = phases
    # The other large kernel of a step with brighter-fatter chains is the pixel search of the rounds (k_accumulate_round): its
    # launches are timed the same way in a few extra steps, and the line names as `roofline.kernel` whichever of the two has
    # the larger summed launch time per step -- the kernel that is dominant in the shipped kernel trace.
    if have_ms and 4 in getattr(step, "timed", {}) and step.timed[4][0] > 0 and cfg["timed_kernel"] == 2:
        extra_steps = 3
        lib.ims_enable_timing(4)
        for _ in range(extra_steps):
            full_step()
        torch.cuda.synchronize()
        ms4, nl4 = _abi.C.c_float(), _abi.C.c_int()
        ok4 = lib.ims_last_kernel_ms(_abi.C.byref(ms4), _abi.C.byref(nl4)) == 0 and nl4.value > 0
        lib.ims_enable_timing(0)
        if ok4:
            n4, bytes4 = step.timed[4]
            mean4 = float(ms4.value) / nl4.value
            per_launch4 = bytes4 / n4
            prof4 = profile_entry(args.config, "k_accumulate_round<4>", world)
            rounds = {"bound": "hbm", "kernel": "k_accumulate_round<4>", "achieved": per_launch4 / (mean4 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                      "unit": "GB/s", "frac": per_launch4 / (mean4 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      "traffic": float(prof4["hbm_bytes_per_launch"]) if prof4 and "hbm_bytes_per_launch" in prof4 else None,
                      "mean_launch_ms": mean4, "timed_launches_per_step": n4, "kernel_ms_per_step": float(ms4.value) / extra_steps,
                      "algorithmic_bytes_per_launch": per_launch4,
                      "limiter": "memory latency: dependent gathers of the pool record, the 64-byte bounds line and the polygon points "
                                 "(SQ_WAIT_ANY / SQ_WAVE_CYCLES in profiles/)"}
            if rounds["kernel_ms_per_step"] > (roofline["kernel_ms_per_step"] or 0.0):
                first = {k: roofline.pop(k) for k in list(roofline) if k in rounds or k in ("f64_valu_issue", "photons_per_step")}
                roofline.update(rounds)
                roofline["other"] = first
            else:
                roofline["other"] = rounds
            roofline["dominant_by"] = "summed launch time per step (hipEvent pairs on the launch streams; launches of different streams overlap)"
    if getattr(step, "branch_bytes", None):
        # FFT branch: SURVEY 8(d)'s 24 N^2 B per object over the whole step (fill + transform + finish)
        roofline["branch"] = {"algorithmic_bytes_per_step": step.branch_bytes,
                              "achieved": step.branch_bytes / (ms_per_step * 1e-3) / 1e9, "unit": "GB/s",
                              "frac": step.branch_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS}

    out = {
        "metric": cfg.get("metric", "objects/sec into one 4k x 4k LSST CCD (photon-shooting path)"),
        "value": value, "unit": "objects/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": cfg["workload"] + ("; " + step.workload_note if getattr(step, "workload_note", None) else ""),
                   "n_objects": n_total_obj, "n_photons": n_total_phot,
                   "image": [scene.nx, scene.ny],
                   "sharding": cfg.get("sharding", "objects dealt by photon count") + f" over {world} rank(s)"},
        "photons_per_s": n_total_phot * args.steps / elapsed,
        "roofline": roofline,
    }
    if share_gpu and world > 1:
        out["shared_gpu"] = True        # dry run: all ranks on ONE GPU over gloo -- not a scaling measurement
    if one_rank_rccl:
        out["rccl_one_rank"] = True     # the step includes the int32 copy and an RCCL self-reduce of the CCD image
    if world > 1 or one_rank_rccl:
        out["exchange"] = ("libimsim_hip.so over RCCL (ims_reduce_image / ims_allreduce_delta)" if parallel._LIBRARY_COMM[0] is not None
                           else ("torch.distributed " + dist.get_backend()))

    if rank == 0 and world == 1 and not args.no_cold and ("cold" in cfg or "end_to_end" in cfg):
        out["extra"] = cfg["cold"](scene, objects, device) if "cold" in cfg else {}
        # host side of LSST_SiliconBuilder.setup for the whole catalog (Poisson fluxes, stamp sizes, local WCS, DCR angles)
        out["extra"]["object_table_ms"] = timing.get("object_table_ms")
        if "end_to_end" in cfg:
            renderer = step = None                   # its pool and private regions make room for the fresh renderer
            torch.cuda.empty_cache()
            out["extra"].update(cfg["end_to_end"](scene, timing["cat"], device))
    if want_cpu:
        if cpu is None:
            cpu = cpu_legs(cfg, scene, objects, args, fork_ok=False)
        out["cpu_baseline"] = cpu_parity(cfg, scene, cpu, device)
    if cfg.get("focal") and rank == 0:
        out.setdefault("extra", {})["step_ms"] = [round(1e3 * (b - a), 1) for a, b in zip([t0] + step_ends[:-1], step_ends)]
    if extra_cfg is not None:
        out.setdefault("extra", {})["configs"] = extra_cfg
    if rank == 0:
        # ONE JSON line and nothing else on stdout: libraries write there too (RCCL prints its version banner through C
        # stdio, flushed at exit -- i.e. AFTER a line printed here), so file descriptor 1 was pointed at stderr for the
        # whole run (main) and the line goes to the descriptor that was stdout
        os.write(JSON_FD[0], (json.dumps(out) + "\n").encode())
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def profile_entry(config, kernel, world):
    """Per-launch figures of `kernel` measured with rocprofv3 --pmc on this workload at one GPU (separate passes:
    FETCH_SIZE / WRITE_SIZE -> hbm_bytes_per_launch, SQ_INSTS_VALU / SQ_WAVES -> valu_insts_per_wave); None when no
    committed measurement matches."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if world != 1 or not os.path.exists(path):
        return None
    with open(path) as fh:
        table = json.load(fh)
    return table.get(config, {}).get(kernel)


# ---------------------------------------------------------------------------------------------
# CPU baseline: the oracle ("port") on the host cores -- pure CPU, runs before the GPU is initialised
# ---------------------------------------------------------------------------------------------
_FORK_STATE = {}


def _allcore_worker(k):
    """One forked worker: renders its share of the catalog into a private image with the oracle scene it inherited
    (the static pixel-boundary state of the CCD is shared copy-on-write; private regions and images are its own)."""
    st = _FORK_STATE
    part = st["parts"][k]
    orc = st["orc"]
    orc.image64 = np.zeros_like(orc.image64)
    t0 = time.perf_counter()
    _cpu_step(st["cfg"])(orc, part)
    return (len(part), int(part["n_phot"].sum()), float(orc.image64.sum()), time.perf_counter() - t0)


def _cpu_step(cfg):
    """the oracle-side step of a config; the FFT branch's lives here because the package never imports the oracle"""
    if cfg.get("cpu_step") is not None:
        return cfg["cpu_step"]
    if cfg.get("cpu_sample_of") is not None:
        return _cpu_job_step

    def fft_step(orc_scene, sample):
        from oracle import orc_loader
        kw = {}
        if cfg.get("fft_spikes"):
            from imsim_amd import configs
            v = configs.c5_visit_fft()
            kw = dict(diffraction_fft=v["diffraction_fft"], wavelength=v["wavelength"])
        o = orc_loader.OracleFft(orc_scene.scene, cfg["fft_kpsf"](), add_noise=True, **kw)
        rows = cfg["fft_rows"](sample)
        rbuf = o.inverse(rows, o.fill(rows))
        if cfg.get("fft_spikes"):
            rbuf = o.spikes(rows, rbuf)
        o.finish(rows, rbuf)
        orc_scene.image64 += o.image
    return fft_step


def _cpu_job_step(orc_scene, sample):
    """One CCD's build on the checker, in the reference's order (imsim/lsst_image.py:342-368): the FFT-drawn objects (k-space
    fill, numpy inverse transform, spike stencil, clip + Poisson noise, stamp -> CCD), then the photon-shot ones."""
    from oracle import orc_loader
    job = sample.job
    if job.n_fft:
        o = orc_loader.OracleFft(orc_scene.scene, job.kpsf, add_noise=True, diffraction_fft=job.diffraction_fft,
                                 wavelength=job.wavelength, extra_ktables=job.extra_ktables)
        rows = job.fft_rows
        rbuf = o.inverse(rows, o.fill(rows))
        if job.diffraction_fft is not None and job.diffraction_fft.enabled:
            rbuf = o.spikes(rows, rbuf)
        o.finish(rows, rbuf)
        orc_scene.image64 += o.image
    if len(job.objects):
        orc_scene.render_lsst_image(job.objects, nrecalc=job.nrecalc)


def cpu_legs(cfg, scene, objects, args, fork_ok=True):
    """(i) ONE core over a bounded random sample (imSim itself never threads, imsim/__init__.py:2-10); (ii) all host
    cores over the whole catalog, forked workers with private images (mirrors output.nproc / image.nproc process
    parallelism), objects dealt longest-first by photon count."""
    import multiprocessing as mp
    from oracle import orc_loader
    from imsim_amd import parallel
    n_sample = args.cpu_sample or cfg["cpu_sample"]
    rng = np.random.default_rng(99)
    if cfg.get("cpu_sample_of") is not None:
        sample = cfg["cpu_sample_of"](objects, scene, args.cpu_sample)   # a whole CCD of a focal plane, as the job the GPU runs
        n_phot_sample = int(sample.job.objects["n_phot"].sum())
        what = ((f"CCD 0 of the focal plane whole" if not args.cpu_sample else
                 f"CCD 0 of the focal plane, its first {args.cpu_sample} objects of at most 3e5 photons and its FFT-drawn objects on grids "
                 f"up to 2048^2, as one job") +
                f": {len(sample)} objects, {sample.job.n_fft} of them FFT-drawn, {n_phot_sample} photons shot")
    else:
        pool = np.arange(len(objects))
        if cfg.get("fft_spikes"):
            # (the checker's spike stencil on a 4096^2 grid takes minutes on one core: the sample is drawn from the stars whose stamps fit 2048^2)
            pool = np.flatnonzero((objects["stamp_xmax"] - objects["stamp_xmin"] + 1) <= 2048)
        idx = np.sort(rng.choice(pool, size=min(n_sample, len(pool)), replace=False))
        sample = objects[idx]
        n_phot_sample = int(sample["n_phot"].sum())
        what = f"{len(sample)} objects drawn at random from the same catalog ({n_phot_sample} photons"
    cpu_scene = cfg["cpu_scene"](scene)
    orc = orc_loader.OracleScene(cpu_scene)
    t0 = time.perf_counter()
    _cpu_step(cfg)(orc, sample)
    dt = time.perf_counter() - t0
    res = {"value": len(sample) / dt, "unit": "objects/s", "cores": 1, "kind": "port",
           "sample": f"{what}, {dt:.1f} s)" if what.count("(") > what.count(")") else f"{what} ({dt:.1f} s)",
           "photons_per_s": float(n_phot_sample) / dt, "host_cpus": os.cpu_count(),
           "_sample": sample, "_image": orc.image}
    if fork_ok and not args.no_cpu_allcore and cfg.get("cpu_allcore", True):
        cores = len(os.sched_getaffinity(0))
        nproc = max(1, cores)
        # plain longest-processing-time dealing by photon count: a CPU worker's time follows its photons (the GPU cost model
        # of parallel.assign_ranks, which prices brighter-fatter rounds, left the workers 2.6 x out of balance at 64 workers)
        owner = parallel.assign_ranks(objects["n_phot"], nproc, round_photons=0.0)
        parts = [objects[owner == k] for k in range(nproc)]
        per_worker = np.array([int(p["n_phot"].sum()) for p in parts], dtype=np.float64)
        fresh = orc_loader.OracleScene(cpu_scene)
        _FORK_STATE.update(parts=parts, orc=fresh, cfg=cfg)
        t0 = time.perf_counter()
        with mp.get_context("fork").Pool(nproc) as pool:
            done = pool.map(_allcore_worker, range(nproc), chunksize=1)
        dt_all = time.perf_counter() - t0
        _FORK_STATE.clear()
        n_done = sum(d[0] for d in done)
        res["all_cores"] = {"value": n_done / dt_all, "unit": "objects/s", "cores": nproc, "kind": "port",
                            "sample": f"the whole catalog ({n_done} objects, {sum(d[1] for d in done)} photons) over "
                                      f"{nproc} forked workers with private images, {dt_all:.1f} s wall "
                                      f"(slowest worker {max(d[3] for d in done):.1f} s)",
                            "electrons": sum(d[2] for d in done),
                            "photon_imbalance_max_over_mean": float(per_worker.max() / max(per_worker.mean(), 1.0)),
                            # (dealt longest-first by photon count: what is left of the imbalance is ONE object -- an object's
                            # brighter-fatter rounds are a serial chain that no worker can share)
                            "largest_object_photons": int(objects["n_phot"].max()),
                            "mean_photons_per_worker": float(per_worker.mean()),
                            "imbalance_without_the_largest_objects": float(
                                np.sort(per_worker)[-min(4, nproc):][0] / max(per_worker.mean(), 1.0)) if nproc > 4 else None,
                            "slowest_worker_s": max(d[3] for d in done), "wall_s": dt_all,
                            # what the same cores sustain once the work balances (a visit deals 189 CCDs' worth of objects):
                            # objects / mean worker time -- the figure to hold against the GPU's, not the wall of ONE CCD
                            "value_if_balanced": n_done / (sum(d[3] for d in done) / nproc)}
    return res


def cpu_parity(cfg, scene, cpu, device):
    """parity check inside the measurement run -- the sample the oracle was timed on is rendered by a
    fresh GPU renderer and the two float32 CCD images are compared pixel by pixel."""
    import torch
    from imsim_amd.engine import Renderer
    sample, want = cpu.pop("_sample"), cpu.pop("_image")
    gpu = Renderer(cfg["cpu_scene"](scene), device)
    cfg["make_step"](gpu, sample, 0, 1)()
    torch.cuda.synchronize()
    got = gpu.image_numpy()
    cpu["parity"] = {"checked": "float32 CCD image of the CPU sample, GPU vs oracle, every pixel",
                     "pixels": int(want.size), "nonzero_pixels": int(np.count_nonzero(want)),
                     "bit_identical": bool(np.array_equal(got.view(np.uint32), np.asarray(want, dtype=np.float32).view(np.uint32)))}
    if cfg.get("parity_mode") == "close":
        # FFT branch: the k-space values are bit-identical, the library transforms (rocFFT vs numpy) agree to ~1e-11 of the
        # peak, so a Poisson deviate can differ where its mean sits within that of a decision boundary
        w = np.asarray(want, dtype=np.float64)
        diff = np.abs(got.astype(np.float64) - w)
        flux_ratio = float(got.sum(dtype=np.float64) / max(w.sum(), 1e-300))
        cpu["parity"].update(differing_pixels=int(np.count_nonzero(diff)), max_abs_diff=float(diff.max()), flux_ratio=flux_ratio,
                             tolerance="pixels that differ < 1e-4 of the non-zero ones; total flux within 1e-6",
                             within_tolerance=bool(np.count_nonzero(diff) <= 1e-4 * max(np.count_nonzero(w), 1) + 2
                                                   and abs(flux_ratio - 1.0) < 1e-6))
    del gpu
    return cpu


if __name__ == "__main__":
    main()
