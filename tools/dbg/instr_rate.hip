// Issue cost of single VALU instructions on gfx950: every wavefront runs trips x 256 instructions of one kind on four
// independent registers; 4 or 1 wavefronts per SIMD on every CU.  Reports time per wavefront instruction per SIMD and the
// ratio to v_fma_f64 (4 cycles at full rate).
//   hipcc --offload-arch=gfx950 -O3 tools/dbg/instr_rate.hip -o /tmp/instr_rate && /tmp/instr_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
#define FOUR(ins, tail) ins " %0, %0" tail "\n" ins " %1, %1" tail "\n" ins " %2, %2" tail "\n" ins " %3, %3" tail

enum { FMA64, MUL64, ADD64, FMA32, PKFMA32, RCP64, RSQ64, MULHI, MULLO, MAD64, MUL24, XOR, MOV64, READLANE, CVT, N_OPS };
static const char* NAMES[N_OPS] = { "v_fma_f64", "v_mul_f64", "v_add_f64", "v_fma_f32", "v_pk_fma_f32", "v_rcp_f64", "v_rsq_f64",
                                    "v_mul_hi_u32", "v_mul_lo_u32", "v_mad_u64_u32", "v_mul_u32_u24", "v_xor_b32", "v_mov_b64",
                                    "v_readlane_b32", "v_cvt_f64_u32" };

template <int OP>
__global__ __launch_bounds__(256) void k(double* out, int trips, uint32_t seed)
{
    double a0 = seed + 2.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    const double b = 1.0000001 + threadIdx.x * 1e-9, c = 1e-9;
    float f0 = seed, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;
    const float fb = 1.0001f, fc = 1e-6f;
    uint32_t u0 = seed + threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, s0 = 0;
    const uint32_t ub = 0xD2511F53u;
    uint64_t w0 = u0, w1 = u1, w2 = u2, w3 = u3;
    for (int t = 0; t < trips; ++t) {
        if (OP == FMA64) { REP64(asm volatile(FOUR("v_fma_f64", ", %4, %5") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == MUL64) { REP64(asm volatile(FOUR("v_mul_f64", ", %4") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (OP == ADD64) { REP64(asm volatile(FOUR("v_add_f64", ", %4") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (OP == FMA32) { REP64(asm volatile(FOUR("v_fma_f32", ", %4, %5") : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(fb), "v"(fc));) }
        if (OP == PKFMA32) { REP64(asm volatile(FOUR("v_pk_fma_f32", ", %4, %5") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == RCP64) { REP64(asm volatile(FOUR("v_rcp_f64", "") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == RSQ64) { REP64(asm volatile(FOUR("v_rsq_f64", "") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == MULHI) { REP64(asm volatile(FOUR("v_mul_hi_u32", ", %4") : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(ub));) }
        if (OP == MULLO) { REP64(asm volatile(FOUR("v_mul_lo_u32", ", %4") : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(ub));) }
        if (OP == MUL24) { REP64(asm volatile(FOUR("v_mul_u32_u24", ", %4") : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(ub));) }
        if (OP == XOR) { REP64(asm volatile(FOUR("v_xor_b32", ", %4") : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(ub));) }
        if (OP == MOV64) { REP64(asm volatile("v_mov_b64 %0, %4\nv_mov_b64 %1, %4\nv_mov_b64 %2, %4\nv_mov_b64 %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (OP == CVT) { REP64(asm volatile("v_cvt_f64_u32 %0, %4\nv_cvt_f64_u32 %1, %4\nv_cvt_f64_u32 %2, %4\nv_cvt_f64_u32 %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(ub));) }
        if (OP == MAD64) { REP64(asm volatile("v_mad_u64_u32 %0, vcc, %4, %4, %0\nv_mad_u64_u32 %1, vcc, %4, %4, %1\nv_mad_u64_u32 %2, vcc, %4, %4, %2\nv_mad_u64_u32 %3, vcc, %4, %4, %3" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(ub) : "vcc");) }
        if (OP == READLANE) { REP64(asm volatile("v_readlane_b32 %1, %0, 3\nv_readlane_b32 %1, %0, 5\nv_readlane_b32 %1, %0, 7\nv_readlane_b32 %1, %0, 9" : "+v"(u0), "+s"(s0));) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + f0 + f1 + f2 + f3 + u0 + u1 + u2 + u3 + s0 + (double)(w0 + w1 + w2 + w3);
}

typedef void (*kern_t)(double*, int, uint32_t);
template <int OP> void fill(kern_t* t) { t[OP] = k<OP>; fill<OP + 1>(t); }
template <> void fill<N_OPS>(kern_t*) {}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    double* out;
    hipMalloc(&out, sizeof(double) * 256 * n_cu * 16);
    kern_t tab[N_OPS];
    fill<0>(tab);
    const int trips = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int occ = 0; occ < 2; ++occ) {
        const int wg_per_cu = occ == 0 ? 4 : 1;   // 256-thread workgroups: 4 -> 4 wavefronts per SIMD, 1 -> 1 per SIMD
        double ref_ms = 0;
        for (int op = 0; op < N_OPS; ++op) {
            hipLaunchKernelGGL(tab[op], dim3(n_cu * wg_per_cu), dim3(256), 0, 0, out, 10, 1u);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(tab[op], dim3(n_cu * wg_per_cu), dim3(256), 0, 0, out, trips, 1u);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ref_ms == 0) ref_ms = ms;
            const double n_inst = (double)trips * 256 * wg_per_cu;   // wavefront instructions per SIMD
            printf("%-16s waves/SIMD %d  %8.3f ms  %6.2f ns per wavefront instruction   x%.2f of v_fma_f64\n", NAMES[op], wg_per_cu, ms,
                   ms * 1e6 / n_inst, ms / ref_ms);
        }
    }
    return 0;
}
