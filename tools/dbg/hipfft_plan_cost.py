"""What the first hipFFT plan of a process costs and where the time goes: loading the library, the first plan, further sizes,
and what a second process finds cached.  Run under gpurun: python tools/dbg/hipfft_plan_cost.py [label]"""
import ctypes as C
import glob
import os
import sys
import time

t0 = time.perf_counter()
import torch
torch.zeros(1, device="cuda")
torch.cuda.synchronize()
print(f"torch + first HIP call: {time.perf_counter() - t0:.2f} s")
t0 = time.perf_counter()
lib = C.CDLL("libhipfft.so")
lib.hipfftPlanMany.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int]
lib.hipfftExecZ2D.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
print(f"dlopen libhipfft.so: {1e3 * (time.perf_counter() - t0):.1f} ms")
HIPFFT_Z2D = 0x6c
for nfft in (1024, 4096, 2048, 512, 1024):
    plan = C.c_void_p()
    n = (C.c_int * 2)(nfft, nfft)
    inembed = (C.c_int * 2)(nfft, nfft // 2 + 1)
    onembed = (C.c_int * 2)(nfft, nfft)
    t0 = time.perf_counter()
    rc = lib.hipfftPlanMany(C.byref(plan), 2, n, inembed, 1, nfft * (nfft // 2 + 1), onembed, 1, nfft * nfft, HIPFFT_Z2D, 1)
    t1 = time.perf_counter()
    k = torch.zeros(nfft * (nfft // 2 + 1), dtype=torch.complex128, device="cuda")
    r = torch.empty(nfft * nfft, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    rc2 = lib.hipfftExecZ2D(plan, C.c_void_p(k.data_ptr()), C.c_void_p(r.data_ptr()))
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    rc3 = lib.hipfftExecZ2D(plan, C.c_void_p(k.data_ptr()), C.c_void_p(r.data_ptr()))
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print(f"  {nfft:5d}^2: plan {1e3 * (t1 - t0):8.1f} ms (rc {rc}), first exec {1e3 * (t3 - t2):7.1f} ms, second exec {1e3 * (t4 - t3):6.2f} ms")
home = os.path.expanduser("~")
found = []
for pat in (home + "/.cache/**/*rocfft*", home + "/.cache/**/*.db", "/tmp/**/*rocfft*"):
    found += glob.glob(pat, recursive=True)
print("cache files:", [(f, os.path.getsize(f)) for f in sorted(set(found))][:10], " ROCFFT_RTC_CACHE_PATH =", os.environ.get("ROCFFT_RTC_CACHE_PATH"))
