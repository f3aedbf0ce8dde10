// hipFFT and RCCL behind the C-ABI without a link-time dependency: the entry points are looked up at first use.
//
// Why not -lhipfft -lrccl: a Python host loads torch first, and torch ships its own copies of libhipfft / librccl /
// libamdhip64 under names without a version suffix; a DT_NEEDED on /opt/rocm's "libhipfft.so.0" would bring a SECOND copy
// into the process, bound to a second HIP runtime that knows nothing of the caller's streams and device pointers.  So: first
// the copy that is already loaded (RTLD_NOLOAD), then the ordinary search path (a C / C++ host links nothing else and gets
// /opt/rocm's).  IMS_HIPFFT_LIB / IMS_RCCL_LIB name a file explicitly.
#pragma once

#include <dlfcn.h>
#include <hipfft/hipfft.h>
#include <rccl/rccl.h>

namespace ims_libs {

static void* open_lib(const char* env, const char* const* names)
{
    const char* forced = getenv(env);
    if (forced && *forced) return dlopen(forced, RTLD_NOW | RTLD_GLOBAL);
    for (int k = 0; names[k]; ++k) {
        void* h = dlopen(names[k], RTLD_NOW | RTLD_NOLOAD);
        if (h) return h;
    }
    for (int k = 0; names[k]; ++k) {
        void* h = dlopen(names[k], RTLD_NOW | RTLD_GLOBAL);
        if (h) return h;
    }
    return nullptr;
}

struct Fft {
    decltype(&hipfftCreate) create = nullptr;
    decltype(&hipfftPlanMany) plan_many = nullptr;
    decltype(&hipfftSetStream) set_stream = nullptr;
    decltype(&hipfftExecZ2D) exec_z2d = nullptr;
    decltype(&hipfftDestroy) destroy = nullptr;
    bool ok = false;
};

static const Fft* fft()
{
    static Fft f;
    static bool tried = false;
    static std::mutex m;
    std::lock_guard<std::mutex> lock(m);
    if (!tried) {
        tried = true;
        static const char* const names[] = { "libhipfft.so", "libhipfft.so.0", nullptr };
        void* h = open_lib("IMS_HIPFFT_LIB", names);
        if (h) {
            f.create = (decltype(f.create))dlsym(h, "hipfftCreate");
            f.plan_many = (decltype(f.plan_many))dlsym(h, "hipfftPlanMany");
            f.set_stream = (decltype(f.set_stream))dlsym(h, "hipfftSetStream");
            f.exec_z2d = (decltype(f.exec_z2d))dlsym(h, "hipfftExecZ2D");
            f.destroy = (decltype(f.destroy))dlsym(h, "hipfftDestroy");
            f.ok = f.create && f.plan_many && f.set_stream && f.exec_z2d && f.destroy;
        }
    }
    return f.ok ? &f : nullptr;
}

struct Rccl {
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclReduce) reduce = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    bool ok = false;
};

static const Rccl* rccl()
{
    static Rccl r;
    static bool tried = false;
    static std::mutex m;
    std::lock_guard<std::mutex> lock(m);
    if (!tried) {
        tried = true;
        static const char* const names[] = { "librccl.so", "librccl.so.1", nullptr };
        void* h = open_lib("IMS_RCCL_LIB", names);
        if (h) {
            r.get_unique_id = (decltype(r.get_unique_id))dlsym(h, "ncclGetUniqueId");
            r.comm_init_rank = (decltype(r.comm_init_rank))dlsym(h, "ncclCommInitRank");
            r.comm_destroy = (decltype(r.comm_destroy))dlsym(h, "ncclCommDestroy");
            r.reduce = (decltype(r.reduce))dlsym(h, "ncclReduce");
            r.all_reduce = (decltype(r.all_reduce))dlsym(h, "ncclAllReduce");
            r.error_string = (decltype(r.error_string))dlsym(h, "ncclGetErrorString");
            r.ok = r.get_unique_id && r.comm_init_rank && r.comm_destroy && r.reduce && r.all_reduce && r.error_string;
        }
    }
    return r.ok ? &r : nullptr;
}

}  // namespace ims_libs
