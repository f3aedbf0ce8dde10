// Does v_mfma_f64_16x16x4_f64 round like a chain of four IEEE FMAs, and in which k order?
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/dbg/mfma_f64_order.hip -o /tmp/mfma && /tmp/mfma
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <random>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, const double* B, const double* C, double* D, int n)
{
    // one wave per problem: A[16][4], B[4][16], C[16][16] row-major
    const int p = blockIdx.x, l = threadIdx.x;
    const double a = A[p * 64 + (l & 15) * 4 + (l >> 4)];
    const double b = B[p * 64 + (l >> 4) * 16 + (l & 15)];
    d4 c;
    for (int r = 0; r < 4; ++r) c[r] = C[p * 256 + ((l >> 4) + 4 * r) * 16 + (l & 15)];
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[p * 256 + ((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}
int main()
{
    const int n = 4096;
    std::mt19937_64 g(12345);
    std::uniform_real_distribution<double> u(-1.0, 1.0);
    double *A = new double[n * 64], *B = new double[n * 64], *C = new double[n * 256], *D = new double[n * 256];
    for (int i = 0; i < n * 64; ++i) { A[i] = u(g) * std::ldexp(1.0, (int)(g() % 20) - 10); B[i] = u(g) * std::ldexp(1.0, (int)(g() % 20) - 10); }
    for (int i = 0; i < n * 256; ++i) C[i] = u(g);
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, n * 64 * 8); hipMalloc(&dB, n * 64 * 8); hipMalloc(&dC, n * 256 * 8); hipMalloc(&dD, n * 256 * 8);
    hipMemcpy(dA, A, n * 64 * 8, hipMemcpyHostToDevice); hipMemcpy(dB, B, n * 64 * 8, hipMemcpyHostToDevice);
    hipMemcpy(dC, C, n * 256 * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n), dim3(64), 0, 0, dA, dB, dC, dD, n);
    hipMemcpy(D, dD, n * 256 * 8, hipMemcpyDeviceToHost);
    long fwd = 0, rev = 0, tot = 0, sum_first = 0;
    for (int p = 0; p < n; ++p)
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                const double c0 = C[p * 256 + i * 16 + j];
                double f = c0, r = c0, s = 0.0;
                for (int kk = 0; kk < 4; ++kk) f = std::fma(A[p * 64 + i * 4 + kk], B[p * 64 + kk * 16 + j], f);
                for (int kk = 3; kk >= 0; --kk) r = std::fma(A[p * 64 + i * 4 + kk], B[p * 64 + kk * 16 + j], r);
                for (int kk = 0; kk < 4; ++kk) s = std::fma(A[p * 64 + i * 4 + kk], B[p * 64 + kk * 16 + j], s);
                s = s + c0;
                const double d = D[p * 256 + i * 16 + j];
                ++tot;
                fwd += std::memcmp(&d, &f, 8) == 0;
                rev += std::memcmp(&d, &r, 8) == 0;
                sum_first += std::memcmp(&d, &s, 8) == 0;
            }
    printf("elements %ld: equal to the forward FMA chain c=fma(a_k,b_k,c), k=0..3: %ld; reversed chain: %ld; products summed first, then + c: %ld\n",
           tot, fwd, rev, sum_first);
    return 0;
}
