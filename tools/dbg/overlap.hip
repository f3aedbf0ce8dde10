// Do chains of short dependent kernels on different streams overlap?  Each stream runs N dependent launches of a kernel with
// G workgroups of 256 threads that sleep ~T us (s_sleep loop on the wall clock).  (round 4: the brighter-fatter chains of
// different CCDs took 139 ms side by side against 158 ms one after the other.)
// hipcc --offload-arch=gfx950 -O2 tools/dbg/overlap.hip -o /tmp/overlap && /tmp/overlap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void busy(long long ticks, int active_wgs, double* sink)
{
    if ((int)blockIdx.x >= active_wgs) return;                 // most workgroups leave at once (tiles without charge)
    const long long t0 = wall_clock64();
    double acc = 0.0;
    while (wall_clock64() - t0 < ticks) { acc += 1.0; __builtin_amdgcn_s_sleep(8); }
    if (sink && acc < 0.0) *sink = acc;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const int n_streams = 4, n = 2000;
    std::vector<hipStream_t> st(n_streams);
    for (auto& s : st) (void)hipStreamCreateWithPriority(&s, hipStreamNonBlocking, -1);
    for (int grid : { 1, 40, 2048 })
        for (int active : { 1, 40, 2048 }) {
            if (active > grid) continue;
            for (int k = 0; k < 50; ++k) hipLaunchKernelGGL(busy, dim3(grid), dim3(256), 0, st[0], 1000, active, (double*)nullptr);
            (void)hipDeviceSynchronize();
            double t0 = now();
            for (int k = 0; k < n; ++k) hipLaunchKernelGGL(busy, dim3(grid), dim3(256), 0, st[0], 1000, active, (double*)nullptr);
            (void)hipDeviceSynchronize();
            const double one = (now() - t0) * 1e3;
            t0 = now();
            for (int k = 0; k < n; ++k)
                for (int s = 0; s < n_streams; ++s) hipLaunchKernelGGL(busy, dim3(grid), dim3(256), 0, st[s], 1000, active, (double*)nullptr);
            (void)hipDeviceSynchronize();
            const double four = (now() - t0) * 1e3;
            printf("grid %4d workgroups, %4d of them busy 10 us: one stream %.1f ms (%.1f us per launch), four streams side by side %.1f ms (x %.2f)\n",
                   grid, active, one, one / n * 1e3, four, four / one);
        }
    return 0;
}
