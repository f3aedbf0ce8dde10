// gather_rate: the rate at which gfx950 serves RANDOM 64-byte lines out of a buffer far larger than its caches, as a function of
// the memory-level parallelism the kernel offers -- K independent 16-byte loads in flight per lane (each from a line of its
// own), W wavefronts per SIMD.  This is the access pattern of the phase-screen gathers of the 6-layer AtmosphericPSF
// (imsim/atmPSF.py:298-336 -> ims::screen_gradient: per photon and layer one 16-byte 2 x 2 cell out of a 1.6 .. 6.4 GB table),
// where rocprofv3's TCC counters showed 15.6 G fabric read requests per second (profiles/round4_c3b_tcc_pmc.txt) -- is that the
// fabric's ceiling for this pattern or the kernel's?
//
//   hipcc --offload-arch=gfx950 -O3 -o gather_rate tools/dbg/gather_rate.hip && ./gather_rate [buffer GiB, default 1.6]
//
// Method: the buffer holds zeros; a lane's next K addresses are hash(counter) + (the sum of what it just loaded), so a batch of
// K loads cannot be issued before the previous batch has arrived (the compiler cannot know the sum is zero), while the K loads
// of a batch are independent.  Grid = 256 CUs x W workgroups of 256 lanes (one wavefront per SIMD each), so W wavefronts per SIMD
// are resident.  Reported: G lines/s = lanes x K x iterations / time, and the GB/s that is in 64-byte lines.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int K>
__global__ __launch_bounds__(256) void k_gather(const uint4* __restrict__ buf, uint32_t line_mask, int iters, uint32_t* __restrict__ sink)
{
    const uint32_t lane = blockIdx.x * 256u + threadIdx.x;
    uint32_t carry = 0u, acc = 0u;
    uint32_t ctr = lane * 2654435761u;
    for (int it = 0; it < iters; ++it) {
        uint4 v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t line = (mix(ctr + (uint32_t)k * 0x9e3779b9u) + carry) & line_mask;     // a random 64-byte line
            v[k] = buf[(size_t)line * 4u + ((ctr >> 3) & 3u)];                                   // one 16-byte item of it
        }
        uint32_t s = 0u;
#pragma unroll
        for (int k = 0; k < K; ++k) s += v[k].x + v[k].y + v[k].z + v[k].w;
        carry = s;                 // zero at run time; the next batch's addresses wait for this batch
        acc += s;
        ctr += 0x632be5abu;
    }
    if (acc == 0xdeadbeefu) sink[lane] = acc;
}

template <int K>
static double run(const uint4* buf, uint32_t line_mask, int waves_per_simd, int iters, uint32_t* sink, int n_cu)
{
    const int grid = n_cu * waves_per_simd;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k_gather<K>, dim3(grid), dim3(256), 0, 0, buf, line_mask, iters / 4 + 1, sink);       // warm-up
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(k_gather<K>, dim3(grid), dim3(256), 0, 0, buf, line_mask, iters, sink);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    const double loads = (double)grid * 256.0 * K * iters;
    return loads / (ms * 1e-3);
}

int main(int argc, char** argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 1.6;
    // lines: the largest power of two that fits (a mask, as the power-of-two screens use)
    uint64_t lines = 1;
    while ((lines << 1) * 64ull <= (uint64_t)(gib * 1073741824.0)) lines <<= 1;
    const size_t bytes = lines * 64ull;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    uint4* buf = nullptr;
    uint32_t* sink = nullptr;
    CHECK(hipMalloc((void**)&buf, bytes));
    CHECK(hipMemset(buf, 0, bytes));
    CHECK(hipMalloc((void**)&sink, (size_t)n_cu * 16 * 256 * sizeof(uint32_t)));
    CHECK(hipDeviceSynchronize());
    printf("# gather_rate on %s: %d CUs, buffer %.2f GiB = %llu lines of 64 B; 16 useful bytes per line\n", prop.name, n_cu,
           bytes / 1073741824.0, (unsigned long long)lines);
    printf("# rows: wavefronts per SIMD; columns: independent loads in flight per lane; cells: G lines/s (GB/s of 64-B lines)\n");
    const int Ws[] = { 1, 2, 3, 4, 5, 8 };
    printf("%-6s %16s %16s %16s %16s %16s\n", "W\\K", "1", "2", "4", "8", "12");
    for (int w : Ws) {
        printf("%-6d", w);
        const int iters = 2000;
        const double r1 = run<1>(buf, (uint32_t)(lines - 1), w, iters, sink, n_cu);
        const double r2 = run<2>(buf, (uint32_t)(lines - 1), w, iters, sink, n_cu);
        const double r4 = run<4>(buf, (uint32_t)(lines - 1), w, iters / 2, sink, n_cu);
        const double r8 = run<8>(buf, (uint32_t)(lines - 1), w, iters / 4, sink, n_cu);
        const double r12 = run<12>(buf, (uint32_t)(lines - 1), w, iters / 4, sink, n_cu);
        const double rs[] = { r1, r2, r4, r8, r12 };
        for (double r : rs) printf(" %7.2f (%6.0f)", r / 1e9, r * 64.0 / 1e9);
        printf("\n");
    }
    CHECK(hipFree(buf));
    CHECK(hipFree(sink));
    return 0;
}
