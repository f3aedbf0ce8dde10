// cost of counter-based generators on gfx950: Philox4x32-10 vs Threefry4x32-{20,12}
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ void philox(uint32_t (&c)[4], uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ uint32_t rotl(uint32_t x, int n) { return __builtin_rotateleft32(x, n); }
template <int ROUNDS>
__device__ __forceinline__ void threefry(uint32_t (&x)[4], const uint32_t (&key)[4])
{
    constexpr int R[8][2] = {{10, 26}, {11, 21}, {13, 27}, {23, 5}, {6, 20}, {17, 11}, {25, 10}, {18, 20}};
    uint32_t ks[5] = {key[0], key[1], key[2], key[3], 0x1BD11BDAu ^ key[0] ^ key[1] ^ key[2] ^ key[3]};
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] += ks[i];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        if (r % 2 == 0) {
            x[0] += x[1]; x[1] = rotl(x[1], R[r % 8][0]) ^ x[0];
            x[2] += x[3]; x[3] = rotl(x[3], R[r % 8][1]) ^ x[2];
        } else {
            x[0] += x[3]; x[3] = rotl(x[3], R[r % 8][0]) ^ x[0];
            x[2] += x[1]; x[1] = rotl(x[1], R[r % 8][1]) ^ x[2];
        }
        if (r % 4 == 3) {
            const int s = r / 4 + 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) x[i] += ks[(s + i) % 5];
            x[3] += (uint32_t)s;
        }
    }
}
template <int WHICH>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters)
{
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint32_t c[4] = {t, (uint32_t)it, 7u, 99u};
        if (WHICH == 0) philox(c, 12345u, 678u);
        else { const uint32_t key[4] = {12345u, 678u, 0u, 0u}; if (WHICH == 1) threefry<20>(c, key); else threefry<12>(c, key); }
        acc ^= c[0] ^ c[1] ^ c[2] ^ c[3];
    }
    out[t] = acc;
}
template <int W> float run(uint32_t* out, int iters)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<W>, dim3(8192), dim3(256), 0, 0, out, iters);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<W>, dim3(8192), dim3(256), 0, 0, out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main()
{
    uint32_t* out; hipMalloc(&out, 8192 * 256 * 4);
    const int iters = 200;
    const double n = 8192.0 * 256 * iters;
    float p = run<0>(out, iters), t20 = run<1>(out, iters), t12 = run<2>(out, iters);
    printf("philox4x32-10   %.3f ms  %.1f Gblocks/s\nthreefry4x32-20 %.3f ms  %.1f Gblocks/s\nthreefry4x32-12 %.3f ms  %.1f Gblocks/s\n",
           p, n / p / 1e6, t20, n / t20 / 1e6, t12, n / t12 / 1e6);
    uint32_t c[4]; (void)c;
    return 0;
}
