// How deep is a HIP stream's queue before hipLaunchKernel blocks the host, and does a host thread that is blocked on one
// stream's full queue hold up another thread's launches on another stream?  (round 4: C5 is bound by the enqueue of
// thousands of dependent ~17 us launches per CCD)
// hipcc --offload-arch=gfx950 -O2 tools/dbg/queue_depth.hip -o /tmp/queue_depth -lpthread && /tmp/queue_depth
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

__global__ void spin(long long cycles, int* sink)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) { }
    if (sink && threadIdx.x == 9999) *sink = 1;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void run(hipStream_t st, int n, long long cycles, const char* tag, std::vector<double>* out)
{
    double t0 = now(), last = t0;
    for (int k = 0; k < n; ++k) {
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, cycles, (int*)nullptr);
        if ((k + 1) % 512 == 0) {
            const double t = now();
            if (out) out->push_back((t - last) / 512 * 1e6);
            else printf("%s launches %5d .. %5d: %.2f us per launch on the host\n", tag, k - 511, k, (t - last) / 512 * 1e6);
            last = t;
        }
    }
    const double t1 = now();
    hipStreamSynchronize(st);
    printf("%s: host done after %.1f ms, GPU done after %.1f ms\n", tag, (t1 - t0) * 1e3, (now() - t0) * 1e3);
}

int main()
{
    hipStream_t a, b;
    hipStreamCreate(&a); hipStreamCreate(&b);
    const long long cycles = 2000;                      // wall_clock64 ticks at 100 MHz: 20 us
    run(a, 512, 100, "warm", nullptr);
    printf("-- one thread, one stream, 8192 launches of 20 us\n");
    run(a, 8192, cycles, "A", nullptr);
    printf("-- two threads, two streams, 8192 launches of 20 us each\n");
    std::vector<double> ra, rb;
    std::thread ta([&] { run(a, 8192, cycles, "A", &ra); });
    std::thread tb([&] { run(b, 8192, cycles, "B", &rb); });
    ta.join(); tb.join();
    for (size_t k = 0; k < ra.size(); ++k) printf("block %2zu: A %.2f  B %.2f us per launch\n", k, ra[k], k < rb.size() ? rb[k] : 0.0);
    printf("-- one thread alternating between the two streams (interleaved enqueue)\n");
    double t0 = now();
    for (int k = 0; k < 8192; ++k) {
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, cycles, (int*)nullptr);
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, cycles, (int*)nullptr);
    }
    double t1 = now();
    hipDeviceSynchronize();
    printf("interleaved: host %.1f ms, GPU %.1f ms for 2 x 8192 launches of 20 us (one stream alone: %.1f ms)\n", (t1 - t0) * 1e3, (now() - t0) * 1e3, 8192 * 0.02);
    return 0;
}
