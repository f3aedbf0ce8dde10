"""Make the file of rocFFT's run-time compiled kernels for the transform sizes of the FFT branch (32 .. 8192, one stamp per
plan and batched), on a GPU box:

    python tools/make_fft_cache.py gpurun_out/rocfft_kernels.db

and copy it to imsim_amd/lib/rocfft_kernels.db (a built artefact like the library: not in the history).  A process that finds
no kernel file of its own starts from a copy of it (imsim_amd/tuning.py fft_kernel_cache) and its first plan of a size costs
milliseconds instead of 0.5 - 1.4 s.  rocFFT checks the file against its own version and the GPU: a file from another
installation is ignored, not wrong."""
import ctypes as C
import os
import sys
import time

out = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else "rocfft_kernels.db")
if os.path.exists(out):
    os.remove(out)
os.environ["ROCFFT_RTC_CACHE_PATH"] = out
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from imsim_amd import _abi  # noqa: E402

lib = _abi.load()
lib.ims_fft_warm.argtypes = [C.c_int32, C.c_void_p]
stream = torch.cuda.Stream()
for n in (32, 64, 128, 256, 512, 1024, 2048, 4096, 8192):
    t0 = time.perf_counter()
    _abi.check(lib.ims_fft_warm(n, C.c_void_p(stream.cuda_stream)), "ims_fft_warm")
    t1 = time.perf_counter()
    batch = 3 if n <= 2048 else 1
    k = torch.zeros(batch * n * (n // 2 + 1) * 2, dtype=torch.float64, device="cuda")
    r = torch.empty(batch * n * n, dtype=torch.float64, device="cuda")
    with torch.cuda.stream(stream):
        for _ in range(3):                 # (a batch that comes again gets the batched plan)
            _abi.check(lib.ims_fft_inverse(C.c_void_p(k.data_ptr()), C.c_void_p(r.data_ptr()), n, batch, C.c_void_p(stream.cuda_stream)), "ims_fft_inverse")
    stream.synchronize()
    print(f"{n:5d}^2: first plan {1e3 * (t1 - t0):7.1f} ms, batched plans + transforms {1e3 * (time.perf_counter() - t1):7.1f} ms", flush=True)
print(out, os.path.getsize(out) if os.path.exists(out) else "NOT WRITTEN")
