// atomic_rate: the rate at which gfx950 executes f64 `unsafeAtomicAdd` (global_atomic_add_f64, no return) on a 128-MB image --
// the deposit of the pixel search: one add to the CCD image and, in photon-pooling mode with brighter-fatter on, a second add to
// the 128-MB delta-charge array (ims_accumulate / k_accumulate_small / tile_flush; SiliconSensor.accumulate as called at
// imsim/photon_pooling.py:195-225).  The sibling of gather_rate.hip: is the C4 pixel search (150 M photons in 5.6 ms = 27 G photons/s,
// 54 G atomics/s) at the memory system's ceiling for this access pattern, or below it?
//
//   hipcc --offload-arch=gfx950 -O3 -o atomic_rate tools/dbg/atomic_rate.hip && ./atomic_rate
//
// Patterns (per lane and iteration one pixel of a 4096 x 4096 f64 image):
//   random      every lane a pixel of its own anywhere on the image (worst case: one 64-B line per lane)
//   wave-local  the 64 lanes of a wavefront fall into ONE 32 x 32-pixel patch at a random place (a faint object's share of a batch:
//               k_accumulate_small, one wavefront per object), the patch changes every iteration
//   wave-core   the same with an 8 x 8 patch (a star's core: many lanes on the same line and the same address)
//   stream      lane l of workgroup b adds to pixel (b * 256 + l) + iteration * grid (perfectly coalesced: the whole-CCD sweeps)
// Columns: one array (image) / two arrays (image + delta, the same pixel in both).  Rows: W wavefronts per SIMD resident.
// Reported: G atomic adds / s (and GB/s if every add moved its 8 bytes once).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

constexpr int NXI = 4096, NYI = 4096;

template <int PATTERN, int ARRAYS>
__global__ __launch_bounds__(256) void k_atomic(double* __restrict__ image, double* __restrict__ delta, int iters)
{
    const uint32_t lane = blockIdx.x * 256u + threadIdx.x;
    const uint32_t wave = lane >> 6;
    uint32_t ctr = lane * 2654435761u;
    for (int it = 0; it < iters; ++it) {
        uint32_t px, py;
        if (PATTERN == 0) {
            const uint32_t h = mix(ctr);
            px = h & (NXI - 1); py = (h >> 12) & (NYI - 1);
        } else if (PATTERN == 1 || PATTERN == 2) {
            const uint32_t span = PATTERN == 1 ? 32u : 8u;
            const uint32_t hw = mix(wave * 0x9e3779b9u + (uint32_t)it * 0x85ebca6bu);      // the wavefront's patch
            const uint32_t hl = mix(ctr);
            px = ((hw & (NXI - 1)) & ~(span - 1u)) + (hl & (span - 1u));
            py = (((hw >> 12) & (NYI - 1)) & ~(span - 1u)) + ((hl >> 8) & (span - 1u));
        } else {
            const uint64_t p = ((uint64_t)lane + (uint64_t)it * gridDim.x * 256ull) & ((uint64_t)NXI * NYI - 1ull);
            px = (uint32_t)(p & (NXI - 1)); py = (uint32_t)(p >> 12);
        }
        const size_t idx = (size_t)py * NXI + px;
        unsafeAtomicAdd(image + idx, 1.0);
        if (ARRAYS == 2) unsafeAtomicAdd(delta + idx, 1.0);
        ctr += 0x632be5abu;
    }
}

template <int PATTERN, int ARRAYS>
static double run(double* image, double* delta, int waves_per_simd, int iters, int n_cu)
{
    const int grid = n_cu * waves_per_simd;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k_atomic<PATTERN, ARRAYS>), dim3(grid), dim3(256), 0, 0, image, delta, iters / 4 + 1);       // warm-up
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((k_atomic<PATTERN, ARRAYS>), dim3(grid), dim3(256), 0, 0, image, delta, iters);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    return (double)grid * 256.0 * iters * ARRAYS / (ms * 1e-3);
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    const size_t bytes = (size_t)NXI * NYI * sizeof(double);
    double *image = nullptr, *delta = nullptr;
    CHECK(hipMalloc((void**)&image, bytes));
    CHECK(hipMalloc((void**)&delta, bytes));
    CHECK(hipMemset(image, 0, bytes));
    CHECK(hipMemset(delta, 0, bytes));
    CHECK(hipDeviceSynchronize());
    printf("# atomic_rate on %s: %d CUs, two f64 arrays of %d x %d (128 MiB each)\n", prop.name, n_cu, NXI, NYI);
    printf("# cells: G atomic adds / s (GB/s at 8 B each); columns: pattern x arrays (1 = image, 2 = image + delta)\n");
    printf("%-4s %15s %15s %15s %15s %15s %15s %15s %15s\n", "W", "random/1", "random/2", "wave32x32/1", "wave32x32/2", "wave8x8/1", "wave8x8/2",
           "stream/1", "stream/2");
    const int Ws[] = { 1, 2, 4, 8 };
    for (int w : Ws) {
        const int iters = 400;
        const double r[] = { run<0, 1>(image, delta, w, iters, n_cu), run<0, 2>(image, delta, w, iters, n_cu),
                             run<1, 1>(image, delta, w, iters, n_cu), run<1, 2>(image, delta, w, iters, n_cu),
                             run<2, 1>(image, delta, w, iters, n_cu), run<2, 2>(image, delta, w, iters, n_cu),
                             run<3, 1>(image, delta, w, iters, n_cu), run<3, 2>(image, delta, w, iters, n_cu) };
        printf("%-4d", w);
        for (double v : r) printf(" %7.2f (%5.0f)", v / 1e9, v * 8.0 / 1e9);
        printf("\n");
    }
    // sanity: the adds happened (stream pattern alone would make every pixel equal)
    double probe[4];
    CHECK(hipMemcpy(probe, image, sizeof(probe), hipMemcpyDeviceToHost));
    printf("# image[0..3] = %.0f %.0f %.0f %.0f\n", probe[0], probe[1], probe[2], probe[3]);
    CHECK(hipFree(image));
    CHECK(hipFree(delta));
    return 0;
}
