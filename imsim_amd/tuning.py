"""The ONE place where this package reads the process environment.

Every switch of the host code and every field of the library's `ims_tuning_t` (include/imsim_hip.h) is named here with its
default and what it selects; the rest of the package asks `env()` / `flag()` / `number()`, and `sync_library()` hands the
library-side choices to libimsim_hip.so through `ims_set_tuning` -- the library itself reads no environment variable (apart
from the file names of the libraries it looks up at run time, csrc/ims_libs.h).  All alternatives compute the same bits by
another route; the defaults are the measured-fastest forms (EXPERIMENTS.md).  A name that is not listed here is an error, so
a misspelt switch cannot silently select the default.
"""
import ctypes as C
import os

# name -> (default, what it selects).  Defaults are strings as they would stand in the environment (None = unset).
KNOWN = {
    # -- the library's ims_tuning_t (sync_library) --
    "IMS_CHAIN_KERNELS": ("1", "kernels specialised for the default operator chain / analytic PSF; 0 = loops over the descriptors"),
    "IMS_LAYOUT_KERNELS": ("1", "ray trace unrolled for a known optics layout; 0 = loop over the surfaces"),
    "IMS_PSF_SCREENS_KERNEL": ("1", "straight-line PSF code for imSim's default AtmosphericPSF; 0 = the component loop"),
    "IMS_PHOTON_LDS": (None, "dynamic LDS bytes a photon-kernel launch asks for without using it (default: 41 984 for launches "
                             "that gather phase screens, else 0)"),
    "IMS_ROUND_COMPACT": ("1", "pixel search of a round launched with a 120-byte argument block; 0 = the 1.4-KB launch parameters"),
    "IMS_INIT_TILES": ("1", "tiled initial pixel-boundary state (k_init_tiles); 0 = one thread per cell"),
    "IMS_UPD_DPP": ("1", "updatePixelDistortions with its table delivered by DPP broadcasts for small launches"),
    "IMS_UPD_DPP_MAX": ("128", "... for launches of at most this many tiles"),
    "IMS_JOINT_LISTS": ("1", "joint rounds: update / refresh over lists of the tiles with charge in reach"),
    "IMS_JOINT_LIST_MIN": ("1024", "... for rounds of more than this many tiles"),
    "IMS_ACTIVE_FRACTION": ("0.25", "... workgroups launched per tile of the round"),
    "IMS_JOINT_FINE_MARKS": ("1", "... from charge marks per 4 x 4 pixels: a tile is listed when charge lies within the update's reach of it"),
    "IMS_JOINT_SEARCH_LISTS": ("1", "... appended to by the pixel search where the charge lands; 0 = a launch of its own scans the marks (k_build_active_j)"),
    "IMS_ROUND_TWO_SEGMENTS": ("0", "pixel search of a round with two 256-photon segments per workgroup (both pool records requested up front)"),
    # -- engine --
    "IMS_FOCAL_GC": ("0", "1 = Python's cyclic garbage collector stays on while a focal plane renders (a generation-2 pass inside the call: "
                          "80 ms of a 1.2-s visit)"),
    "IMS_POOL_DELTA_ONLY": ("1", "photon pooling: a batch's deposits go to the delta-charge image only and the image takes them at the recalculation "
                                 "(one atomic add per photon; Silicon's target += delta); 0 = image and delta image both"),
    "IMS_SPIKE_TABLE": ("1", "FFT branch: the non-zero entries of the spike stencil from a table made once per visit (ims_fft_spike_table); "
                             "0 = every stencil value evaluated in place"),
    "IMS_FOCAL_PINNED": ("64", "focal plane: page-locked float32 image buffers (67 MB each for a 4096 x 4096 CCD) in flight at most; kept for "
                               "the life of the process"),
    "IMS_FFT_SPIKE_LIST": ("1", "FFT branch: the spike step in two launches (ims_fft_spikes_listed: the image streamed, the arms' pixels listed and "
                                "summed with every lane at work); 0 = one launch (ims_fft_spikes), the same bits"),
    "IMS_FFT_SPIKE_LIST_CAP": ("0", "entries of that list (0 = a eighth of the draw's pixels); a list that runs over costs one more launch"),
    "IMS_FFT_RAW": ("1", "FFT branch: the inverse transform's 1 / N^2 applied by the kernel that reads the real-space buffer next "
                         "(ims_fft_inverse_raw + ims_fft_params_t.rbuf_raw); 0 = a scaling pass of its own (the same bits)"),
    "IMS_SCREEN_PREPASS": ("0", "phase-screen gathers ahead of the shooting kernels (1: every photon, 2: ordinary objects on a side stream)"),
    "IMS_SCREEN_BUCKETS": ("128", "arrival-time buckets of the pre-pass"),
    "IMS_SCREEN_QUADS": (None, "phase screens also as 2 x 2 cells of 16 bytes (default 1 unless the pre-pass covers every photon)"),
    "IMS_STREAM_PRIORITIES": ("-1,0,0,0,0", "HIP priorities of the chain / bulk / chain1 / chain2 / chain3 plan streams"),
    "IMS_STREAM_TOUCH": ("0,1,2,3,4", "Renderer.touch_streams: order in which the plan streams (chain, bulk, chain1, chain2, chain3) are first used"),
    "IMS_UPLOAD_SYNC": ("0", "engine.upload_async as the synchronous copy it replaced"),
    "IMS_BF_TAGS": ("0", "tile marks in LSST_Image chains (the update skips tiles without charge in reach)"),
    "IMS_CHAIN_CLASSES": ("40,6", "round counts that cut the bright objects into concurrent brighter-fatter chains"),
    "IMS_LAZY_STATIC": ("0", "Renderer without an explicit lazy_static: 1 = slot 0 without stored state where it applies (LSST_Image renders only).  "
                             "A single CCD makes its state once per renderer, not per step, and its second launch is a serial tail of the "
                             "wide-launch stream: C3 24.2 ms either way, same image"),
    "IMS_NATIVE_PLAN": ("1", "launch plan of a CCD built and enqueued by the library (ims_plan_*); 0 = the numpy planner (the checker)"),
    "IMS_POOL_RESIDENT": ("1", "photon pooling with the pool of all batches in HBM; 0 = one fused launch per batch"),
    "IMS_POOL_OVERLAP": ("0", "photon pooling: 1 = the pool shot batch by batch on a stream of its own, ahead of the batches' pixel searches "
                              "and recalculations (measured: C4 244 against 236 ms -- two wide kernels share the wave slots, they do not add up)"),
    "IMS_POOL_SMALL_MAX": ("64", "per-batch shares up to this many photons go through ims_accumulate_small"),
    "IMS_POOL_SPATIAL": ("1", "shoot table of pooling mode in spatial order"),
    "IMS_FFT_TORCH": ("0", "inverse transforms of the FFT branch through torch.fft instead of ims_fft_inverse (the checker)"),
    "IMS_EXCHANGE_SINGLE_RANK": ("0", "run the exchanges of a one-rank process group as self-exchanges"),
    # -- focal plane --
    "IMS_FOCAL_STREAMS": ("1", "four plan streams by role for all CCDs of a device; 0 = a set per renderer"),
    "IMS_FOCAL_TOUCH": ("mid,top0,pre,bulk", "order in which the focal-plane role streams (top0 = joint rounds, pre, bulk, mid) are first used, right "
                                            "where they are made (HIP binds a stream to a hardware queue at its first use, queue = index mod 4: "
                                            "with all four used there C5 takes 1.74 .. 1.79 s in any order, 2.4 s when the FFT warm-up threads "
                                            "come first); empty = no touch"),
    "IMS_FOCAL_TOPS": ("2", "streams for the long top chains of a device"),
    "IMS_FOCAL_CONCURRENT": ("4", "bench C5: CCDs in flight on the rolling-window path"),
    "IMS_FOCAL_THREADS": ("1", "host threads that enqueue CCDs (rolling window)"),
    "IMS_FOCAL_INIT": ("bulk", "rolling window: stream of a CCD's static-state initialisation"),
    "IMS_FOCAL_COPY": ("mid", "rolling window: stream of a CCD's image copy"),
    "IMS_FOCAL_ANCHOR": ("top", "rolling window: stream a CCD's plan is anchored to"),
    "IMS_FOCAL_FFT": (None, "stream of a CCD's FFT-drawn objects (joint path: mid; rolling window: top)"),
    "IMS_FOCAL_JOINT": ("20", "CCDs per batch whose chains advance in joint launches (0 / 1: a chain per CCD).  Measured with four batches alive "
                              "(189 CCDs): 16 -> 1.57 s, 18 / 19 -> 1.59, 20 -> 1.49 - 1.51, 21 -> 1.56, 22 / 24 / 27 -> 1.58"),
    "IMS_FOCAL_JOINT_MAX_BRIGHT": ("600", "CCDs with more objects of their own rounds than this take the rolling window"),
    "IMS_FOCAL_AHEAD": ("pre:1", "the host enqueues a CCD's front only when the front n CCDs before has run on that stream"),
    "IMS_NO_HINT": ("0", "ignore the chain-length hint that batches CCDs of similar chains"),
    "IMS_FOCAL_PRE_PRIORITY": ("0", "joint path: priority of the stream of the FFT draws / initial states / first pool slices"),
    "IMS_FOCAL_JOINT_INIT": ("pre", "joint path: the stream a CCD's renderer is initialised on"),
    "IMS_FOCAL_ARENA": ("1", "joint path: pixel-boundary state leased from one arena per device (engine.SensorArena); 0 = per renderer"),
    "IMS_FOCAL_ARENA_CELLS": (None, "the arena's private pool in owner cells (tests: a pool that runs dry)"),
    "IMS_FOCAL_ARENA_FACTOR": ("0.8", "private cells per CCD the arena's pool is sized for, as a multiple of the first CCD's need"),
    "IMS_FOCAL_STATIC_REGIONS": ("3", "static regions of the arena, taken in turn"),
    "IMS_FOCAL_LAZY_STATIC": ("1", "joint path: a CCD's static pixel-boundary state is not made; the photons near a pixel edge are finished by a "
                                   "second launch from the tree-ring closed form (ims_render_params_t.lazy_static); 0 = k_init_tiles per CCD"),
    "IMS_FOCAL_ALIVE": ("4", "joint path: batches alive at a time (in their rounds and tails, being enqueued).  With every role stream on a "
                             "hardware queue of its own: 2 -> 1.77 s, 3 -> 1.67, 4 -> 1.68 (1.54 with IMS_FOCAL_PHOTON_LDS), 5 -> 1.61 with it"),
    "IMS_FOCAL_PHOTON_LDS": ("41984", "joint path: IMS_PHOTON_LDS while a focal plane renders -- the photon kernels capped at three workgroups "
                                      "per CU leave wave slots to the rounds (C5 1.68 -> 1.54 s); empty = as elsewhere"),
    "IMS_FFT_WARM": ("1", "focal_plane.warm_fft makes the hipFFT plans on background threads; 0 = plans are made where they are first needed "
                          "(rocprofv3's counter passes crash with launches from several host threads)"),
    "IMS_FOCAL_JOINT_THREAD": ("1", "joint path: the rounds of a batch are enqueued by a second host thread while the first goes on with the next fronts"),
    "IMS_FOCAL_COARSE_SLICES": ("1", "joint path: a chain class's photons shot in two launches (round 0, the rest) instead of up to six slices"),
    "IMS_FOCAL_TRACE": ("0", "print, per CCD, when its work ended on every stream and when the host enqueued it"),
    "IMS_PROCESS_FOCAL": ("1", "config.Process with several CCDs on the overlapped focal-plane path"),
    "IMS_PROCESS_CONCURRENT": ("3", "... CCDs in flight"),
    # -- bench / tests --
    "IMS_C5_CCDS": (None, "bench.py --config c5 on the first N CCDs at the full per-CCD workload"),
    "IMS_BENCH_DUMP": (None, "bench.py: file that receives the last step's image (or per-CCD CRCs): the N-rank tests"),
    # -- files --
    "IMSIM_HIP_LIB": (None, "another build of the library (A/B measurements)"),
    "ROCFFT_RTC_CACHE_PATH": (None, "rocFFT's file of run-time compiled kernels; unset: <user cache dir>/imsim_amd/rocfft_kernels.db (fft_kernel_cache)"),
    "XDG_CACHE_HOME": (None, "the user's cache directory (default ~/.cache)"),
    "IMSIM_DATA_DIR": (None, "imSim's data directory"),
    "IMSIM_CONFIG_DIR": (None, "imSim's config directory (template lookup)"),
    "SIMS_SED_LIBRARY_DIR": (None, "the SED library of instance catalogs"),
}


_SCOPED = {}                  # switches a caller set for the duration of a call (scoped): below the environment, above the defaults


def env(name, default=None):
    """The value of a known switch: the environment's, else what a caller set for the time being (scoped), else `default` when
    given, else the listed default."""
    if name not in KNOWN:
        raise KeyError(f"tuning.env: {name!r} is not a known switch (imsim_amd/tuning.py lists them)")
    v = os.environ.get(name)
    if v is not None:
        return v
    if name in _SCOPED:
        return _SCOPED[name]
    return default if default is not None else KNOWN[name][0]


class scoped:
    """`with tuning.scoped(IMS_PHOTON_LDS="41984"): ...` -- defaults of this process while the block runs (the environment still
    wins; None or "" leaves a switch alone).  Process-wide, like the library's tuning block it ends up in: for callers that own
    the device for the duration (a focal plane), not for concurrent renders with different wishes."""
    def __init__(self, **values):
        for k in values:
            if k not in KNOWN:
                raise KeyError(k)
        self.values = {k: str(v) for k, v in values.items() if v is not None and str(v) != ""}
        self.before = {}

    def __enter__(self):
        self.before = {k: _SCOPED.get(k) for k in self.values}
        _SCOPED.update(self.values)
        return self

    def __exit__(self, *exc):
        for k, v in self.before.items():
            if v is None:
                _SCOPED.pop(k, None)
            else:
                _SCOPED[k] = v
        # the library's process-wide block follows at once (not at the next renderer construction or plan run: a renderer built
        # before the block and used after it would otherwise launch with the scope's settings)
        if _LIB[0] is not None:
            sync_library(_LIB[0])
        return False


def flag(name, default=None):
    """True unless the switch is "0" / unset-with-default-"0" """
    v = env(name, default)
    return v is not None and v != "0" and v != ""


def number(name, cast=int, default=None):
    v = env(name, default)
    return None if v is None or v == "" else cast(v)


def setdefault(name, value):
    """a default the CALLER chooses for this process (bench C5: four top-chain streams)"""
    if name not in KNOWN:
        raise KeyError(name)
    os.environ.setdefault(name, str(value))


def fft_kernel_cache(seed=None):
    """rocFFT compiles the kernels of a transform size at run time (0.5 - 1.4 s per size, profiles/round5_hipfft_plan_cost.log)
    and keeps them in a file, by default under the home directory -- which a batch node's user may not have, and then every
    process pays again.  Unless ROCFFT_RTC_CACHE_PATH is set, point it at <user cache dir>/imsim_amd/rocfft_kernels.db (the
    temporary directory when the home is not writable), copied from `seed` (the package's lib/rocfft_kernels.db, made on a
    GPU box by tools/make_fft_cache.py) when it does not exist yet.  With the kernels on file a plan costs 3 - 14 ms
    (profiles/round5_c5_cold.log).  Called by _abi.load(), before the first plan.  Returns the path in use."""
    import shutil
    import tempfile
    have = os.environ.get("ROCFFT_RTC_CACHE_PATH")
    if have:
        return have
    roots = [os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache"),
             os.path.join(tempfile.gettempdir(), f"imsim_amd_{os.getuid()}")]
    try:
        os.makedirs(roots[1], mode=0o700, exist_ok=True)
    except OSError:
        pass
    for k, root in enumerate(roots):
        d = os.path.join(root, "imsim_amd")
        try:
            os.makedirs(d, mode=0o700, exist_ok=True)
            if k > 0:
                # the fall-back lives under the shared temporary directory at a predictable name: rocFFT LOADS code objects from
                # this file, so the directory (and its parent) must be ours alone -- not a symlink, owned by this user, writable by
                # nobody else; otherwise the variable stays unset
                ok = True
                for q in (root, d):
                    st = os.lstat(q)
                    import stat as _stat
                    ok = ok and _stat.S_ISDIR(st.st_mode) and st.st_uid == os.getuid() and not (st.st_mode & 0o022)
                if not ok:
                    continue
            path = os.path.join(d, "rocfft_kernels.db")
            if os.path.islink(path):
                continue
            if not os.path.exists(path) and seed and os.path.exists(seed):
                tmp = f"{path}.{os.getpid()}"
                shutil.copyfile(seed, tmp)
                os.replace(tmp, path)                    # (several ranks start at once: whole file or none)
            elif not os.access(d, os.W_OK):
                continue
        except OSError:
            continue
        os.environ["ROCFFT_RTC_CACHE_PATH"] = path
        return path
    return None


class Tuning(C.Structure):
    """ims_tuning_t (include/imsim_hip.h)"""
    _fields_ = [("chain_kernels", C.c_int32), ("layout_kernels", C.c_int32), ("psf_screens_kernel", C.c_int32), ("photon_lds", C.c_int32),
                ("round_compact", C.c_int32), ("init_tiles", C.c_int32), ("upd_dpp", C.c_int32), ("joint_lists", C.c_int32),
                ("upd_dpp_max", C.c_int64), ("joint_list_min", C.c_int64), ("active_fraction", C.c_double),
                ("round_two_segments", C.c_int32), ("joint_fine_marks", C.c_int32),
                ("joint_search_lists", C.c_int32), ("pad", C.c_int32)]


def library_tuning():
    """ims_tuning_t as the environment asks for it"""
    t = Tuning()
    t.chain_kernels = 1 if flag("IMS_CHAIN_KERNELS") else 0
    t.layout_kernels = 1 if flag("IMS_LAYOUT_KERNELS") else 0
    t.psf_screens_kernel = 1 if flag("IMS_PSF_SCREENS_KERNEL") else 0
    lds = number("IMS_PHOTON_LDS")
    t.photon_lds = -1 if lds is None else int(lds)
    t.round_compact = 1 if flag("IMS_ROUND_COMPACT") else 0
    t.init_tiles = 1 if flag("IMS_INIT_TILES") else 0
    t.upd_dpp = 1 if flag("IMS_UPD_DPP") else 0
    t.joint_lists = 1 if flag("IMS_JOINT_LISTS") else 0
    t.upd_dpp_max = number("IMS_UPD_DPP_MAX")
    t.joint_list_min = number("IMS_JOINT_LIST_MIN")
    t.active_fraction = number("IMS_ACTIVE_FRACTION", float)
    t.round_two_segments = 1 if flag("IMS_ROUND_TWO_SEGMENTS") else 0
    t.joint_fine_marks = 1 if flag("IMS_JOINT_FINE_MARKS") else 0
    t.joint_search_lists = 1 if flag("IMS_JOINT_SEARCH_LISTS") else 0
    return t


_LAST = [None]
_LIB = [None]                    # the library sync_library was last called with (scoped.__exit__ re-synchronises it)


def sync_library(lib):
    """Hand the library-side choices to libimsim_hip.so (ims_set_tuning) when they differ from what it was last given; called
    wherever the host is about to enqueue launches (renderer construction, plan runs, joint runs)."""
    _LIB[0] = lib
    t = library_tuning()
    raw = bytes(t)
    if raw != _LAST[0]:
        rc = lib.ims_set_tuning(C.byref(t))
        if rc != 0:
            msg = lib.ims_last_error()
            raise RuntimeError(f"ims_set_tuning failed ({rc}): {msg.decode() if msg else ''}")
        _LAST[0] = raw
