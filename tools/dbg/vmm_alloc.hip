// What a fresh process pays for tens of GiB of device memory, by route (round 5: the sensor arena of a focal plane is 60 GiB and
// one hipMalloc of it costs 22.6 ms per GiB):
//   1. hipMalloc of one block of G GiB
//   2. hipMalloc of G blocks of 1 GiB, of 4 G blocks of 256 MiB
//   3. ONE virtual range of G GiB (hipMemAddressReserve) backed by chunks of C MiB (hipMemCreate + hipMemMap), access set once
// each followed by a fill of everything (the memory is there: bytes / s of the fill).
// hipcc --offload-arch=gfx950 -O2 -o /tmp/vmm_alloc tools/dbg/vmm_alloc.hip && /tmp/vmm_alloc 48
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void k_fill(unsigned long long* p, size_t n, unsigned long long v)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v + i;
}
__global__ void k_check(const unsigned long long* p, size_t n, unsigned long long v, unsigned long long* bad)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned long long c = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) c += (p[i] != v + i);
    if (c) atomicAdd(bad, c);
}

static int fill(void* p, size_t bytes, const char* what)
{
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 2; ++rep) {
        const double t0 = now();
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (unsigned long long*)p, bytes / 8, 7ull);
        CK(hipDeviceSynchronize());
        const double dt = now() - t0;
        printf("    %s fill %d: %.2f ms (%.2f TB/s)\n", what, rep, 1e3 * dt, bytes / dt / 1e12);
    }
    return 0;
}

int main(int argc, char** argv)
{
    const size_t G = argc > 1 ? (size_t)atoi(argv[1]) : 48;
    const int route = argc > 2 ? atoi(argv[2]) : 0;           // 0: all, in the order 3 (VMM), 2 (blocks), 1 (one block)
    CK(hipSetDevice(0));
    CK(hipFree(nullptr));
    unsigned long long* bad;
    CK(hipMalloc((void**)&bad, 8));
    CK(hipMemset(bad, 0, 8));
    const size_t GiB = (size_t)1 << 30;
    if (route == 0 || route == 3) {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        size_t gran = 0;
        CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
        printf("VMM: recommended granularity %zu KiB\n", gran >> 10);
        for (size_t chunk_mib : { (size_t)1024, (size_t)256, (size_t)64 }) {
            const size_t chunk = chunk_mib << 20, n = G * GiB / chunk;
            double t0 = now();
            void* base = nullptr;
            CK(hipMemAddressReserve(&base, G * GiB, 0, nullptr, 0));
            const double t_res = now() - t0;
            std::vector<hipMemGenericAllocationHandle_t> h(n);
            double t_create = 0, t_map = 0;
            for (size_t k = 0; k < n; ++k) {
                double a = now();
                CK(hipMemCreate(&h[k], chunk, &prop, 0));
                double b = now();
                CK(hipMemMap((char*)base + k * chunk, chunk, 0, h[k], 0));
                double c = now();
                t_create += b - a; t_map += c - b;
            }
            hipMemAccessDesc acc = {};
            acc.location = prop.location;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            double a = now();
            CK(hipMemSetAccess(base, G * GiB, &acc, 1));
            const double t_acc = now() - a;
            printf("VMM %zu GiB in chunks of %zu MiB: reserve %.1f ms, create %.1f ms, map %.1f ms, set access %.1f ms: total %.1f ms\n", G, chunk_mib,
                   1e3 * t_res, 1e3 * t_create, 1e3 * t_map, 1e3 * t_acc, 1e3 * (now() - t0));
            if (fill(base, G * GiB, "VMM")) return 1;
            hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, (const unsigned long long*)base, G * GiB / 8, 7ull, bad);
            unsigned long long nb = 1;
            CK(hipMemcpy(&nb, bad, 8, hipMemcpyDeviceToHost));
            printf("    check across the chunk seams: %llu words differ\n", nb);
            t0 = now();
            CK(hipMemUnmap(base, G * GiB));
            for (size_t k = 0; k < n; ++k) CK(hipMemRelease(h[k]));
            CK(hipMemAddressFree(base, G * GiB));
            printf("    unmap + release: %.1f ms\n", 1e3 * (now() - t0));
        }
    }
    if (route == 0 || route == 2) {
        for (size_t chunk_mib : { (size_t)4096, (size_t)1024, (size_t)256 }) {
            const size_t chunk = chunk_mib << 20, n = G * GiB / chunk;
            std::vector<void*> p(n);
            double t0 = now();
            for (size_t k = 0; k < n; ++k) CK(hipMalloc(&p[k], chunk));
            printf("hipMalloc %zu blocks of %zu MiB: %.1f ms\n", n, chunk_mib, 1e3 * (now() - t0));
            t0 = now();
            for (size_t k = 0; k < n; ++k)
                hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, 0, (unsigned long long*)p[k], chunk / 8, 7ull);
            CK(hipDeviceSynchronize());
            printf("    fill of all blocks: %.2f ms\n", 1e3 * (now() - t0));
            t0 = now();
            for (size_t k = 0; k < n; ++k) CK(hipFree(p[k]));
            printf("    free: %.1f ms\n", 1e3 * (now() - t0));
        }
    }
    if (route == 0 || route == 1) {
        for (size_t g : { (size_t)8, G }) {
            void* p = nullptr;
            double t0 = now();
            CK(hipMalloc(&p, g * GiB));
            printf("hipMalloc one block of %zu GiB: %.1f ms\n", g, 1e3 * (now() - t0));
            if (fill(p, g * GiB, "block")) return 1;
            t0 = now();
            CK(hipFree(p));
            printf("    free: %.1f ms\n", 1e3 * (now() - t0));
        }
    }
    return 0;
}
