// v_fmac_f64_dpp row_newbcast on gfx950: (1) semantics -- every lane of a row of 16 must see d of lane SEL of ITS row, result
// bit-equal to fma(d_sel, w, acc); (2) issue cost against the plain v_fma_f64, 1 and 4 wavefronts per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/dbg/dpp_f64.hip -o /tmp/dpp_f64 && /tmp/dpp_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <cstring>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

__global__ void k_sem(const double* d, const double* w, const double* acc, double* out)
{
    const int t = threadIdx.x;
    double a3 = acc[t], a9 = acc[t], a15 = acc[t], a0 = acc[t];
    const double dd = d[t], ww = w[t];
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a3) : "v"(dd), "v"(ww));
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:9 row_mask:0xf bank_mask:0xf" : "+v"(a9) : "v"(dd), "v"(ww));
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:15 row_mask:0xf bank_mask:0xf" : "+v"(a15) : "v"(dd), "v"(ww));
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "+v"(a0) : "v"(dd), "v"(ww));
    out[t] = a3; out[64 + t] = a9; out[128 + t] = a15; out[192 + t] = a0;
}

template <int OP>
__global__ __launch_bounds__(256) void k_rate(double* out, int trips)
{
    double a0 = 2.0 + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    const double b = 1.0000001 + threadIdx.x * 1e-9, c = 1e-9;
    for (int t = 0; t < trips; ++t) {
        if (OP == 0) { REP64(asm volatile("v_fma_f64 %0, %4, %5, %0\nv_fma_f64 %1, %4, %5, %1\nv_fma_f64 %2, %4, %5, %2\nv_fma_f64 %3, %4, %5, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 1) { REP64(asm volatile("v_fmac_f64_dpp %0, %4, %5 row_newbcast:1 row_mask:0xf bank_mask:0xf\nv_fmac_f64_dpp %1, %4, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf\nv_fmac_f64_dpp %2, %4, %5 row_newbcast:9 row_mask:0xf bank_mask:0xf\nv_fmac_f64_dpp %3, %4, %5 row_newbcast:13 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 2) { REP64(asm volatile("v_fmac_f64 %0, %4, %5\nv_fmac_f64 %1, %4, %5\nv_fmac_f64 %2, %4, %5\nv_fmac_f64 %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3;
}

int main()
{
    double hd[64], hw[64], ha[64], ho[256];
    for (int i = 0; i < 64; ++i) { hd[i] = 1.0 / (3.0 + i); hw[i] = 0.1 + 0.37 * i; ha[i] = -2.0 + 0.01 * i * i; }
    double *d, *w, *a, *o;
    hipMalloc(&d, 512); hipMalloc(&w, 512); hipMalloc(&a, 512); hipMalloc(&o, 2048);
    hipMemcpy(d, hd, 512, hipMemcpyHostToDevice); hipMemcpy(w, hw, 512, hipMemcpyHostToDevice); hipMemcpy(a, ha, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_sem, dim3(1), dim3(64), 0, 0, d, w, a, o);
    hipMemcpy(ho, o, 2048, hipMemcpyDeviceToHost);
    const int sels[4] = { 3, 9, 15, 0 };
    int bad = 0;
    for (int q = 0; q < 4; ++q)
        for (int t = 0; t < 64; ++t) {
            const double ref = fma(hd[(t & ~15) + sels[q]], hw[t], ha[t]);
            if (memcmp(&ref, &ho[64 * q + t], 8) != 0) { if (bad < 5) printf("MISMATCH sel %d lane %d: %.17g vs %.17g\n", sels[q], t, ho[64 * q + t], ref); ++bad; }
        }
    printf("semantics: %d mismatches of 256 (row_newbcast:n reads lane n of the lane's own row of 16)\n", bad);

    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    double* out;
    hipMalloc(&out, sizeof(double) * 256 * n_cu * 16);
    typedef void (*kern_t)(double*, int);
    kern_t tab[3] = { k_rate<0>, k_rate<1>, k_rate<2> };
    const char* names[3] = { "v_fma_f64", "v_fmac_f64_dpp row_newbcast", "v_fmac_f64" };
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int trips = 2000;
    for (int occ = 0; occ < 2; ++occ) {
        const int wg = occ == 0 ? 4 : 1;
        for (int op = 0; op < 3; ++op) {
            hipLaunchKernelGGL(tab[op], dim3(n_cu * wg), dim3(256), 0, 0, out, 10);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(tab[op], dim3(n_cu * wg), dim3(256), 0, 0, out, trips);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("%-30s waves/SIMD %d  %8.3f ms  %6.2f ns per wavefront instruction\n", names[op], wg, ms, ms * 1e6 / ((double)trips * 256 * wg));
        }
    }
    return bad ? 1 : 0;
}
