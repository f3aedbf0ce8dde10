// launch-latency probe: dependent small kernels in one stream
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
struct Big { double v[200]; };
__global__ void k_empty(int* p) { if (p && threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void k_big(Big b, int* p) { if (p && threadIdx.x == 0 && blockIdx.x == 0) p[0] += (int)b.v[3]; }
__global__ void k_touch(double* buf, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) buf[i] += 1.0; }
template <class F> double run(const char* name, int n, F f, hipStream_t s)
{
    for (int i = 0; i < 50; ++i) f(i);
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) f(i);
    auto t1 = std::chrono::steady_clock::now();
    hipStreamSynchronize(s);
    auto t2 = std::chrono::steady_clock::now();
    double enq = std::chrono::duration<double, std::micro>(t1 - t0).count() / n;
    double tot = std::chrono::duration<double, std::micro>(t2 - t0).count() / n;
    printf("%-44s enqueue %7.2f us/launch   total %7.2f us/launch\n", name, enq, tot);
    return tot;
}
int main()
{
    hipStream_t s, s2; hipStreamCreate(&s);
    int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, hi);
    int* p; hipMalloc(&p, 4); hipMemset(p, 0, 4);
    double* buf; size_t n = 64u << 20; hipMalloc(&buf, n * 8); hipMemset(buf, 0, n * 8);
    Big b{}; b.v[3] = 1.0;
    const int N = 2000;
    run("empty kernel, 1 block", N, [&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, p); }, s);
    run("empty kernel, 1024 blocks", N, [&](int) { hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, s, p); }, s);
    run("1.6 KB by-value args", N, [&](int) { hipLaunchKernelGGL(k_big, dim3(64), dim3(256), 0, s, b, p); }, s);
    run("high-priority stream, empty", N, [&](int) { hipLaunchKernelGGL(k_empty, dim3(64), dim3(256), 0, s2, p); }, s2);
    run("touch 2 MB (dirty lines) ", N, [&](int) { hipLaunchKernelGGL(k_touch, dim3(1024), dim3(256), 0, s, buf, (size_t)262144); }, s);
    run("touch 64 MB", 300, [&](int) { hipLaunchKernelGGL(k_touch, dim3(32768), dim3(256), 0, s, buf, (size_t)8388608); }, s);
    // a big dirty footprint made by one kernel, then small dependent kernels
    hipLaunchKernelGGL(k_touch, dim3(262144), dim3(256), 0, s, buf, n);
    run("small kernels after 512 MB touched", N, [&](int) { hipLaunchKernelGGL(k_empty, dim3(64), dim3(256), 0, s, p); }, s);
    // graph of 2000 dependent kernels
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_big, dim3(64), dim3(256), 0, s, b, p);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    auto t1 = std::chrono::steady_clock::now();
    printf("%-44s total %7.2f us/kernel\n", "hipGraph of 2000 dependent kernels", std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
    return 0;
}
