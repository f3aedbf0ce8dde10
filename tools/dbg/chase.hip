// Dependent-load latency on gfx950 as the brighter-fatter rounds see it: one lane chases a random cycle of 64-byte lines over
// footprints of 256 KB .. 8 GB, (a) warm (second pass of the same kernel), (b) cold after a kernel boundary, the lines last
// written by ANOTHER kernel that ran on all XCDs.  Reports ns per load.
//   hipcc --offload-arch=gfx950 -O3 tools/dbg/chase.hip -o /tmp/chase && /tmp/chase
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>
#include <algorithm>

__global__ void k_write(uint64_t* buf, const uint32_t* next, size_t n_lines)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_lines; i += (size_t)gridDim.x * blockDim.x)
        buf[i * 8] = (uint64_t)next[i] * 8;
}

__global__ void k_chase(const uint64_t* buf, int hops, uint64_t* out, unsigned long long* ticks, int passes)
{
    uint64_t p = 0;
    for (int pass = 0; pass < passes; ++pass) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (int h = 0; h < hops; ++h) p = buf[p];
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        ticks[pass] = t1 - t0 + (p & 0);          // 100 MHz counter
    }
    out[0] = p;
}

// 64 lanes, each on its own chain (lane l starts at line l): a wavefront's load returns when its slowest lane has its line
__global__ void k_chase64(const uint64_t* buf, int hops, uint64_t* out, unsigned long long* ticks, int passes)
{
    uint64_t p = (uint64_t)threadIdx.x * 8;
    for (int pass = 0; pass < passes; ++pass) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (int h = 0; h < hops; ++h) p = buf[p];
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0) ticks[pass] = t1 - t0;
    }
    out[threadIdx.x] = p;
}

int main()
{
    const size_t sizes[] = { 256ull << 10, 2ull << 20, 16ull << 20, 128ull << 20, 1ull << 30, 8ull << 30 };
    uint64_t* out; unsigned long long* ticks;
    hipMalloc(&out, 8 * 64); hipMalloc(&ticks, 64);
    for (size_t bytes : sizes) {
        const size_t n = bytes / 64;
        const int hops = 2000;
        // a random cycle over `hops + 1` lines spread uniformly over the footprint (not over all lines: building an 8 GB cycle is slow)
        std::mt19937_64 rng(1234);
        std::vector<uint32_t> pick(hops + 1);
        pick[0] = 0;
        for (int i = 1; i <= hops; ++i) pick[i] = (uint32_t)(rng() % n);
        std::vector<uint32_t> next_sparse(n ? 0 : 0);
        uint64_t* buf; uint32_t* next;
        if (hipMalloc(&buf, bytes) != hipSuccess) { printf("alloc %zu failed\n", bytes); continue; }
        hipMalloc(&next, n * 4);
        hipMemset(next, 0, n * 4);
        for (int i = 0; i < hops; ++i) hipMemcpy(next + pick[i], &pick[i + 1], 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_write, dim3(2048), dim3(256), 0, 0, buf, next, n);
        hipDeviceSynchronize();
        unsigned long long h[4];
        // cold: first kernel after the writer
        hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, 0, buf, hops, out, ticks, 2);
        hipMemcpy(h, ticks, 16, hipMemcpyDeviceToHost);
        printf("%8.1f MB: cold after writer kernel %7.1f ns/load, second pass in the same kernel %7.1f ns/load", bytes / 1048576.0,
               h[0] * 10.0 / hops, h[1] * 10.0 / hops);
        // again after a kernel boundary without a writer in between
        hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, 0, buf, hops, out, ticks, 1);
        hipMemcpy(h, ticks, 8, hipMemcpyDeviceToHost);
        printf(", next kernel (nobody wrote) %7.1f ns/load\n", h[0] * 10.0 / hops);
        // vector path: 64 chains of 200 hops each, interleaved through the footprint
        {
            const int vh = 200;
            std::vector<uint32_t> cur(64), nx(64);
            for (int l = 0; l < 64; ++l) cur[l] = l;
            hipMemset(next, 0, n * 4);
            for (int i = 0; i < vh; ++i) {
                for (int l = 0; l < 64; ++l) { nx[l] = (uint32_t)(64 + rng() % (n - 64)); hipMemcpy(next + cur[l], &nx[l], 4, hipMemcpyHostToDevice); }
                cur = nx;
            }
            hipLaunchKernelGGL(k_write, dim3(2048), dim3(256), 0, 0, buf, next, n);
            hipDeviceSynchronize();
            hipLaunchKernelGGL(k_chase64, dim3(1), dim3(64), 0, 0, buf, vh, out, ticks, 2);
            hipMemcpy(h, ticks, 16, hipMemcpyDeviceToHost);
            printf("            64 lanes, vector loads: cold after writer %7.1f ns/load, second pass %7.1f ns/load\n", h[0] * 10.0 / vh, h[1] * 10.0 / vh);
        }
        hipFree(buf); hipFree(next);
    }
    return 0;
}
