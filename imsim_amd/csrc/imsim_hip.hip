// imsim_hip.hip -- kernels and C-ABI entry points of libimsim_hip.so (gfx950 / MI355X only).
//
// Launch geometry: one 256-thread workgroup (4 wavefronts of 64) per *segment* = up to seg_size
// photons of ONE object, so the object row, its stamp and its tables are wave-uniform (scalar
// loads) and the per-object flux reduction is a wavefront shuffle + one atomic per wave.
// Segments are dealt to the 8 XCDs in contiguous ranges (block b runs on XCD b%8 on this part),
// so that objects that are neighbours in the (spatially sorted) table share an XCD's L2 for the
// pixel-boundary state and the image lines they touch.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include "ims_photon.h"
#include "ims_fft.h"

using namespace ims;

static thread_local char g_err[512] = "";
static int set_err(int code, const char* msg)
{
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}
static int hip_err(hipError_t e, const char* what)
{
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return IMS_ERR_HIP;
}
// Which of its equivalent kernel forms the library launches is decided by ONE explicit block of settings (ims_tuning_t,
// include/imsim_hip.h) that the caller may replace with ims_set_tuning -- never by the caller's environment: the library itself
// reads no environment variable except the file names of the libraries it looks up at run time (csrc/ims_libs.h).
static ims_tuning_t tuning_defaults()
{
    ims_tuning_t t;
    t.chain_kernels = 1; t.layout_kernels = 1; t.psf_screens_kernel = 1; t.photon_lds = -1;
    t.round_compact = 1; t.init_tiles = 1; t.upd_dpp = 1; t.joint_lists = 1;
    t.upd_dpp_max = 128; t.joint_list_min = 1024; t.active_fraction = 0.25;
    t.round_two_segments = 0; t.joint_fine_marks = 1; t.joint_search_lists = 1; t.pad = 0;
    return t;
}
static ims_tuning_t g_tune = tuning_defaults();
#define HIP_TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hip_err(e_, #call); } while (0)

// Timing of the dominant kernel: every launch of the selected entry point between ims_enable_timing(which)
// and the query is bracketed by a hipEvent pair on the launch stream (which: 1 = ims_shoot_accumulate,
// 2 = ims_shoot_ops_photons, 3 = ims_fft_kspace_fill).
#include <vector>
#include <mutex>
#include <unordered_map>
#include <map>
#include <tuple>
static int g_timing = 0;
static std::vector<hipEvent_t> g_events;   // pairs
static size_t g_events_used = 0;
static std::mutex g_state_mutex;           // the event tables below may be reached from several host threads (focal plane: one per CCD in flight)

struct LaunchTimer {
    hipStream_t st;
    size_t slot;
    bool on;
    LaunchTimer(hipStream_t s, int which) : st(s), slot(0), on(g_timing == which)
    {
        if (on) {
            hipEvent_t first;
            {
                std::lock_guard<std::mutex> lock(g_state_mutex);
                if (g_events_used + 2 > g_events.size()) {
                    hipEvent_t a, b;
                    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
                    g_events.push_back(a); g_events.push_back(b);
                }
                slot = g_events_used;
                g_events_used += 2;
                first = g_events[slot];
                second = g_events[slot + 1];
            }
            (void)hipEventRecord(first, st);
        }
    }
    ~LaunchTimer() { if (on) (void)hipEventRecord(second, st); }
    hipEvent_t second;
};

// -DIMS_PROBE (measurement builds only, tools/dbg/round_probe.py): the stamps of ims_photon.h's PROBE / PROBE_WG, copied out
#ifdef IMS_PROBE
extern "C" int ims_probe_read(unsigned long long* out32, unsigned long long* wg512)
{
    if (wg512 && hipMemcpyFromSymbol(wg512, HIP_SYMBOL(ims::g_probe_wg), 512 * sizeof(unsigned long long)) != hipSuccess) return -1;
    return hipMemcpyFromSymbol(out32, HIP_SYMBOL(ims::g_probe), 32 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
// -DIMS_HIST (measurement builds only, tools/dbg/c5_tile_hist.py): what a listed tile of a joint round holds and costs -- per
// number of charged cells in the update's 23 x 23 halo (bins 0 .. 62, 63 = more): tiles, 10-ns ticks from the tile's entry to
// its last store (thread 0), charged cells inside the tile itself
#ifdef IMS_HIST
__device__ unsigned long long g_hist[3][64];
extern "C" int ims_hist_read(unsigned long long* out192, int reset)
{
    if (hipMemcpyFromSymbol(out192, HIP_SYMBOL(g_hist), 192 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        static unsigned long long zero[192];
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_hist), zero, sizeof(zero)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

#ifdef IMS_EXTRA_LAUNCHES
__global__ void k_nop(int* p) { if (p) *p = 0; }      // (measurement builds: empty launches on the joint stream -- what a kernel boundary costs a round)
#endif
// ---------------- segment -> (object, first photon) ----------------
constexpr int N_XCD = 8;

__device__ __forceinline__ int64_t xcd_segment(int64_t b, int64_t n_segments)
{
    // block b runs on XCD (b % 8): give each XCD one contiguous range of segments
    const int64_t per = (n_segments + N_XCD - 1) / N_XCD;
    return (b % N_XCD) * per + (b / N_XCD);
}

__device__ __forceinline__ int64_t find_object(const int64_t* __restrict__ prefix, int64_t n_objects, int64_t seg)
{
    int64_t lo = 0, hi = n_objects;
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (prefix[mid] <= seg) lo = mid; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// run one photon through shoot -> psf -> shift -> ops
template <int PSF = 0>
__device__ __forceinline__ void make_photon(const ims_render_params_t& P, const ims_object_t& o, int64_t k, Rng& rng, Photon& ph)
{
    rng_reset(rng);
    shoot(P, o, k, rng, ph);
    run_psf<PSF>(P, o, k, rng, ph);
    ph.x = o.x0 + ph.x;
    ph.y = o.y0 + ph.y;
}

// ---------------- per-workgroup charge tile ----------------
// All photons of a workgroup belong to ONE object, and a bright star puts thousands of photons per
// round into a few dozen pixels: global atomics on those hot addresses serialise in L2.  Each
// workgroup therefore first sums its photons into a 32x32-pixel LDS tile centred on the object
// (ds_add_f32), then flushes each non-empty tile cell with ONE global atomic to the CCD image (and
// to the delta-charge image when the object's region is brighter-fatter tracked).  Photons outside
// the tile go straight to global memory.  Unit fluxes keep every partial sum an exact integer, so
// the result stays independent of the order of the atomics.  A photon whose flux is not exactly 1
// (BandpassRatio reweighting, flux_per_photon != 1) never enters the float tile: it is added to the
// f64 image directly, so no rounding to binary32 happens anywhere on its way.
constexpr int CT = 32;

// Joint rounds whose update / refresh walk LISTS of tiles: the pixel search appends a tile the first time charge lands within the
// update's reach of it (k_accumulate_round_j; ims_tuning_t.joint_search_lists), instead of a launch of its own that scans the
// charge marks of every tile of every region afterwards (k_build_active_j: 13 - 32 us of every round of a joint run).
struct JointAcc;
struct JointRound;
struct TileLister {
    unsigned long long* list = nullptr; // entries: chain << 58 | slot of the class << 32 | tile of the slot; NULL: no listing (the default)
    int* count = nullptr;               // entries so far (this round's parity)
    const JointAcc* J = nullptr;        // per chain: tile words, tiles before each slot of the class, the class's first slot
    const JointRound* R = nullptr;      // per chain: tiles of the regions that go on after this round (only those are updated)
    int chain = 0;
    unsigned int round_word = 0u;       // this round + 1
};
// (resolved where the lists are written, at the end of the workgroup: kept live through the pixel search, the chain's table
// entries cost the search kernel its register allocation -- 96 registers and scratch instead of 93)
struct TileListerChain { unsigned int* words; int64_t tile_base; int lo; bool on; };
__device__ __forceinline__ TileListerChain lister_chain(const TileLister& L, int slot_idx);

struct ChargeTile {
    int x0, y0;                // pixel coordinates of tile cell (0,0)
    bool track;                // also add to the slot's delta image
    ims_bf_slot_t slot;
    TileLister lister{};       // joint rounds with search-side lists (by value: through a pointer the block lived in scratch); list == NULL: off
    int slot_idx = 0;          // the region's slot
};

__device__ __forceinline__ void tile_begin(float* tile, ChargeTile& ct, const ims_render_params_t& P, const ims_object_t& o,
                                           bool silicon, int n_thr = 256, const TileLister* lister = nullptr)
{
    for (int e = threadIdx.x; e < CT * CT; e += n_thr) tile[e] = 0.0f;
    ct.x0 = (int)floor(o.x0 + 0.5) - CT / 2;
    ct.y0 = (int)floor(o.y0 + 0.5) - CT / 2;
    ct.track = silicon && !(o.flags & IMS_OBJ_FAINT) && (o.bf_state > 0 || P.track_static_delta);
    ct.lister.list = nullptr; ct.slot_idx = 0;
    if (ct.track) { ct.slot_idx = slot_index(P, o); ct.slot = P.sensor->bf_slots[ct.slot_idx]; }
    if (ct.track && lister != nullptr && lister->list != nullptr) ct.lister = *lister;
    __syncthreads();
}

// bf_tag != 0: remember which 16x16 tile of the region received charge (byte at the tile's first cell)
__device__ __forceinline__ void mark_tile_charge(const ims_render_params_t& P, int64_t offset, int nx, int di, int dj)
{
    if (P.bf_tag == 0u) return;
    unsigned char* tc = P.sensor->bf_tile_charge;
    if (tc == nullptr) return;
    tc[offset + (int64_t)(dj & ~15) * (nx + 1) + (di & ~15)] = (unsigned char)P.bf_tag;
    // and which 4 x 4 block of it: the byte of the block's cell (1, 1) -- never a tile's first cell, and inside the region's
    // (nx + 1) x (ny + 1) owner cells because (di & ~3) + 1 <= di + 1 <= nx.  Read by k_build_active_j.
    tc[offset + (int64_t)((dj & ~3) + 1) * (nx + 1) + ((di & ~3) + 1)] = (unsigned char)P.bf_tag;
}

// Charge landed on pixel (di, dj) of the region: list every tile whose update or bounds refresh can depend on it and that no
// deposit of this round has listed yet.  The update of tile [tx0, tx0 + 15]^2 reads the delta charge of [tx0 - 4, tx0 + 18]^2, the
// bounds of its pixels also hang on the first column / row of the right / upper tile, which move with charge up to tx0 + 19: the
// tiles tx with tx0 - 4 <= di <= tx0 + 19, the same in y -- one, two or four tiles.  A tile's word holds the last round that
// listed it: the exchange decides who appends (a stale plain read only costs the exchange).
__device__ __forceinline__ void list_tiles_in_reach(const ChargeTile& ct, int di, int dj)
{
    const TileLister& L = ct.lister;
    const TileListerChain C = lister_chain(L, ct.slot_idx);
    if (!C.on) return;
    const int tiles_x = (ct.slot.nx + 1 + 15) >> 4, tiles_y = (ct.slot.ny + 1 + 15) >> 4;
    int ax = (di - 4) >> 4, bx = (di + 4) >> 4, ay = (dj - 4) >> 4, by = (dj + 4) >> 4;
    if (ax < 0) ax = 0;
    if (ay < 0) ay = 0;
    if (bx > tiles_x - 1) bx = tiles_x - 1;
    if (by > tiles_y - 1) by = tiles_y - 1;
    for (int ty = ay; ty <= by; ++ty)
        for (int tx = ax; tx <= bx; ++tx) {
            const int t = ty * tiles_x + tx;
            unsigned int* w = C.words + (C.tile_base + t);
            if (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == L.round_word) continue;
            if (atomicExch(w, L.round_word) == L.round_word) continue;
            const int k = atomicAdd(L.count, 1);
            L.list[k] = ((unsigned long long)L.chain << 58) | ((unsigned long long)(unsigned int)C.lo << 32) | (unsigned int)t;
        }
}

// list_now: the deposit lists its tiles itself (the stragglers outside the workgroup's LDS tile); the flush of the LDS tile
// collects the tiles of all its deposits first (tile_flush)
__device__ __forceinline__ void deposit_global(const ims_render_params_t& P, const ChargeTile& ct, int ix, int iy, double flux,
                                               bool list_now = true)
{
    const int px = ix - P.xmin, py = iy - P.ymin;
    bool to_image = px >= 0 && px < P.nx && py >= 0 && py < P.ny;
    if (ct.track) {
        const int di = ix - ct.slot.xmin, dj = iy - ct.slot.ymin;
        if (di >= 0 && di < ct.slot.nx && dj >= 0 && dj < ct.slot.ny) {
            unsafeAtomicAdd(P.sensor->bf_delta + (ct.slot.offset + (int64_t)dj * (ct.slot.nx + 1) + di), flux);
            mark_tile_charge(P, ct.slot.offset, ct.slot.nx, di, dj);
            if (list_now && ct.lister.list != nullptr) list_tiles_in_reach(ct, di, dj);
            // track_static_delta 2 (photon pooling, slot 0 = the image): the image takes the charge FROM the delta image when
            // the recalculation consumes it -- GalSim's `target += delta` -- so the deposit is one atomic add, not two
            if (P.track_static_delta == 2) to_image = false;
        }
    }
    if (to_image) unsafeAtomicAdd(P.image + ((int64_t)py * P.nx + px), flux);
}

__device__ __forceinline__ void tile_deposit(float* tile, const ChargeTile& ct, const ims_render_params_t& P, int ix, int iy,
                                             double flux)
{
    const int tx = ix - ct.x0, ty = iy - ct.y0;
    if (flux == 1.0 && tx >= 0 && tx < CT && ty >= 0 && ty < CT) atomicAdd(&tile[ty * CT + tx], 1.0f);
    else deposit_global(P, ct, ix, iy, flux);
}

__device__ __forceinline__ void tile_flush(float* tile, const ChargeTile& ct, const ims_render_params_t& P, int n_thr = 256)
{
    // search-side lists: the deposits of the 32 x 32 LDS tile reach at most 4 x 4 tiles of the region (pixels d0 - 4 .. d0 + 35):
    // their bits are collected in LDS, then ONE lane per tile in reach asks the tile's word, and ONE atomic per workgroup makes
    // room in the list (a deposit-by-deposit form spent the round in exchanges and in additions to the one counter)
    __shared__ unsigned int reach_bits;
    __shared__ int n_new, base_new;
    __shared__ int new_tile[16];
    const bool listing = ct.lister.list != nullptr;      // uniform over the workgroup (one object, one slot)
    if (listing && threadIdx.x == 0) { reach_bits = 0u; n_new = 0; }
    __syncthreads();
    PROBE(5);
    PROBE_WG(6, 0);
    const int tiles_x = listing ? ((ct.slot.nx + 1 + 15) >> 4) : 0, tiles_y = listing ? ((ct.slot.ny + 1 + 15) >> 4) : 0;
    int ox = listing ? ((ct.x0 - ct.slot.xmin - 4) >> 4) : 0, oy = listing ? ((ct.y0 - ct.slot.ymin - 4) >> 4) : 0;
    if (ox < 0) ox = 0;
    if (oy < 0) oy = 0;
    for (int e = threadIdx.x; e < CT * CT; e += n_thr) {
        const float v = tile[e];
        if (v != 0.0f) {
            const int ix = ct.x0 + e % CT, iy = ct.y0 + e / CT;
            deposit_global(P, ct, ix, iy, (double)v, false);
            if (listing) {
                const int di = ix - ct.slot.xmin, dj = iy - ct.slot.ymin;
                if (di >= 0 && di < ct.slot.nx && dj >= 0 && dj < ct.slot.ny) {
                    // tiles ax .. bx by ay .. by, relative to the window's first tile: a 4-bit run in x times a 4-bit run in y
                    int ax = ((di - 4) >> 4) - ox, bx = ((di + 4) >> 4) - ox, ay = ((dj - 4) >> 4) - oy, by = ((dj + 4) >> 4) - oy;
                    if (ax < 0) ax = 0;
                    if (ay < 0) ay = 0;
                    const unsigned int xm = ((2u << bx) - 1u) & ~((1u << ax) - 1u) & 0xFu;
                    const unsigned int ym = ((2u << by) - 1u) & ~((1u << ay) - 1u) & 0xFu;
                    const unsigned int rows = (ym & 1u) | ((ym & 2u) << 3) | ((ym & 4u) << 6) | ((ym & 8u) << 9);
                    atomicOr(&reach_bits, xm * rows);
                }
            }
        }
    }
    if (!listing) return;
    __syncthreads();
    const TileLister& L = ct.lister;
    if (reach_bits == 0u) return;
    const TileListerChain C = lister_chain(L, ct.slot_idx);
    if (!C.on) return;
    if (threadIdx.x < 16 && ((reach_bits >> threadIdx.x) & 1u)) {
        const int tx = ox + (int)(threadIdx.x & 3), ty = oy + (int)(threadIdx.x >> 2);
        if (tx < tiles_x && ty < tiles_y) {
            const int t = ty * tiles_x + tx;
            unsigned int* w = C.words + (C.tile_base + t);
            if (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != L.round_word && atomicExch(w, L.round_word) != L.round_word)
                new_tile[atomicAdd(&n_new, 1)] = t;
        }
    }
    __syncthreads();
    if (n_new > 0) {
        if (threadIdx.x == 0) base_new = atomicAdd(L.count, n_new);
        __syncthreads();
        if ((int)threadIdx.x < n_new)
            L.list[base_new + threadIdx.x] = ((unsigned long long)L.chain << 58) | ((unsigned long long)(unsigned int)C.lo << 32) | (unsigned int)new_tile[threadIdx.x];
    }
}

// ---------------- fused kernel: LSST_Silicon draw (phot) + stamp->CCD add ----------------
#ifndef IMS_FUSED_WAVES
#define IMS_FUSED_WAVES 4
#endif
#ifndef IMS_CHAIN_WAVES
#define IMS_CHAIN_WAVES 4          // wavefronts per SIMD the kernels specialised for the default chain are compiled for
#endif
template <int CHAIN, int PSF = 0, unsigned long long LAYOUT = 0ull>
__global__ __launch_bounds__(256, (CHAIN == 1) ? IMS_CHAIN_WAVES : IMS_FUSED_WAVES) void k_shoot_accumulate(const ims_render_params_t P)
{
    const int64_t per = (P.n_segments + N_XCD - 1) / N_XCD;
    const int64_t b = blockIdx.x;
    if ((b / N_XCD) >= per) return;
    const int64_t seg = xcd_segment(b, P.n_segments);
    if (seg >= P.n_segments) return;
    const int64_t oi = P.seg_object ? (int64_t)P.seg_object[seg] : find_object(P.seg_prefix, P.n_objects, seg);
    const ims_object_t& o = P.objects[oi];
    const int64_t seg_in_obj = seg - P.seg_prefix[oi];
    const int64_t j0 = seg_in_obj * P.seg_size;
    int64_t j1 = j0 + P.seg_size;
    if (j1 > o.n_phot) j1 = o.n_phot;
    // Wavefronts of the workgroup that hold no photon of this segment (a 19-photon object uses one of four) leave at
    // once: they would only sit in the barriers and keep the wave slots other segments could run in.  The hardware
    // barrier counts the waves that are still alive.
    const int n_thr = (((int)(j1 - j0) + 63) >> 6) << 6;
    if ((int)threadIdx.x >= n_thr) return;
    const bool silicon = (P.sensor != nullptr) && (P.sensor->kind == IMS_SENSOR_SILICON);
    const bool has_angles = (CHAIN == 1) ? true : chain_has_angles(P);
    __shared__ float tile[CT * CT];
    ChargeTile ct;
    tile_begin(tile, ct, P, o, silicon, n_thr);
    double added = 0.0;
    // exactly one photon per thread (seg_size == workgroup size): no photon loop, so the compiler
    // cannot hoist the chain's uniform operands across iterations into registers
    const int64_t j = j0 + threadIdx.x;
    if (j < j1) {
        const int64_t k = o.phot_first + j;
        Photon ph;
        Rng rng;
        make_photon<PSF>(P, o, k, rng, ph);
        run_ops<CHAIN, LAYOUT>(P, o, k, rng, ph);
        int ix, iy;
        if (ph.flux != 0.0 && land(P, o, k, rng, ph, silicon, has_angles, ix, iy, seg * 4 + (int64_t)(threadIdx.x >> 6))) {
            added += ph.flux;
            tile_deposit(tile, ct, P, ix, iy, ph.flux);
        }
    }
    tile_flush(tile, ct, P, n_thr);
    if (P.realized_flux != nullptr) {
        const double tot = wave_sum(added);
        if ((threadIdx.x & 63) == 0 && tot != 0.0) unsafeAtomicAdd(P.realized_flux + oi, tot);
    }
}

// ---------------- pooled path ----------------
// LSST_PhotonsBuilder.draw for all objects of a sub-batch, written straight into the merged pool.
// MODE 1 additionally runs the configured photon-op chain before storing (used to pre-compute the
// sensor-independent part of bright objects' photons, see Renderer.plan_lsst_image); MODE 2 also runs the half of
// SiliconSensor.accumulate that does not depend on the pixel boundaries and stores the `converted` pool format
// (ims_photons_t.converted), so that the latency-bound rounds of a brighter-fatter chain only do the pixel search.
// pool.pupil_u / pupil_v / time / obj_index may be NULL (not stored).
template <int MODE, int CHAIN = 0, int PSF = 0, unsigned long long LAYOUT = 0ull>
__global__ __launch_bounds__(256, (CHAIN == 1) ? IMS_CHAIN_WAVES : IMS_FUSED_WAVES) void k_shoot_photons(const ims_render_params_t P,
                                                                        const int64_t* __restrict__ photon_offset,
                                                                        const ims_photons_t pool)
{
    const int64_t per = (P.n_segments + N_XCD - 1) / N_XCD;
    const int64_t b = blockIdx.x;
    if ((b / N_XCD) >= per) return;
    const int64_t seg = xcd_segment(b, P.n_segments);
    if (seg >= P.n_segments) return;
    const int64_t oi = P.seg_object ? (int64_t)P.seg_object[seg] : find_object(P.seg_prefix, P.n_objects, seg);
    const ims_object_t& o = P.objects[oi];
    const int64_t j = (seg - P.seg_prefix[oi]) * P.seg_size + threadIdx.x;
    if (j >= o.n_phot) return;
    const int64_t k = o.phot_first + j;
    Photon ph;
    Rng rng;
    make_photon<PSF>(P, o, k, rng, ph);
    if (MODE >= 1) run_ops<CHAIN, LAYOUT>(P, o, k, rng, ph);
    const int64_t i = photon_offset[oi] + j;
    if (MODE == 2) {
        const bool silicon = (P.sensor != nullptr) && (P.sensor->kind == IMS_SENSOR_SILICON);
        double x0 = ph.x, y0 = ph.y, flux = ph.flux, zs = 1.0;
        if (silicon && !(o.flags & IMS_OBJ_FAINT) && flux != 0.0) {
            double zconv;
            bool coin;
            if (land_convert(P, o, k, rng, ph, (CHAIN == 1) ? true : chain_has_angles(P), x0, y0, zconv, coin)) {
                const double zf = dtanh_pos(ddiv(zconv, 12.0));
                zs = coin ? -zf : zf;
            } else flux = 0.0;
        }
        pool.x[i] = x0; pool.y[i] = y0; pool.flux[i] = flux; pool.dxdz[i] = zs;
        return;
    }
    pool.x[i] = ph.x; pool.y[i] = ph.y; pool.flux[i] = ph.flux;
    pool.dxdz[i] = ph.dxdz; pool.dydz[i] = ph.dydz; pool.wavelength[i] = ph.wl;
    if (pool.pupil_u) { pool.pupil_u[i] = ph.pu; pool.pupil_v[i] = ph.pv; pool.time[i] = ph.t; }
    if (pool.obj_index) pool.obj_index[i] = (int32_t)oi;
}

__device__ __forceinline__ void load_photon(const ims_photons_t& pool, int64_t i, Photon& ph)
{
    ph.x = pool.x[i]; ph.y = pool.y[i]; ph.flux = pool.flux[i]; ph.dxdz = pool.dxdz[i]; ph.dydz = pool.dydz[i];
    ph.wl = pool.wavelength[i]; ph.pu = pool.pupil_u[i]; ph.pv = pool.pupil_v[i]; ph.t = pool.time[i];
}

// `for op in photon_ops: op.applyTo(photons, ...)` (photon_pooling.py:154-155): one pass, all ops
__global__ __launch_bounds__(256) void k_apply_ops(const ims_render_params_t P, const int64_t* __restrict__ photon_offset,
                                                   const ims_photons_t pool)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < pool.n; i += stride) {
        const int32_t oi = pool.obj_index[i];
        const ims_object_t& o = P.objects[oi];
        const int64_t k = o.phot_first + (i - photon_offset[oi]);
        Photon ph;
        load_photon(pool, i, ph);
        Rng rng;
        rng_reset(rng);
        for (int q = 0; q < P.n_ops; ++q) apply_op(P, q, o, k, rng, ph);
        pool.x[i] = ph.x; pool.y[i] = ph.y; pool.flux[i] = ph.flux;
        pool.dxdz[i] = ph.dxdz; pool.dydz[i] = ph.dydz;
        pool.pupil_u[i] = ph.pu; pool.pupil_v[i] = ph.pv; pool.time[i] = ph.t;
    }
}

// sensor.accumulate(photons, full_image, ...) (photon_pooling.py:195-225)
__global__ __launch_bounds__(256) void k_accumulate(const ims_render_params_t P, const int64_t* __restrict__ photon_offset,
                                                    const ims_photons_t pool, int32_t* __restrict__ pixel_index_out)
{
    const bool silicon = (P.sensor != nullptr) && (P.sensor->kind == IMS_SENSOR_SILICON);
    const bool has_angles = chain_has_angles(P);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < pool.n; i += stride) {
        const int32_t oi = pool.obj_index[i];
        const ims_object_t& o = P.objects[oi];
        const int64_t k = o.phot_first + (i - photon_offset[oi]);
        if (pixel_index_out) pixel_index_out[i] = -1;
        Photon ph;
        load_photon(pool, i, ph);
        if (ph.flux == 0.0) continue;
        int ix, iy;
        Rng rng;
        rng_reset(rng);
        if (!land(P, o, k, rng, ph, silicon, has_angles, ix, iy)) continue;
        if (P.realized_flux != nullptr) unsafeAtomicAdd(P.realized_flux + oi, ph.flux);
        bool in_delta = false;
        if (silicon && !(o.flags & IMS_OBJ_FAINT) && (o.bf_state > 0 || P.track_static_delta)) {
            const ims_bf_slot_t bs = P.sensor->bf_slots[slot_index(P, o)];
            const int di = ix - bs.xmin, dj = iy - bs.ymin;
            if (di >= 0 && di < bs.nx && dj >= 0 && dj < bs.ny) {
                unsafeAtomicAdd(P.sensor->bf_delta + (bs.offset + (int64_t)dj * (bs.nx + 1) + di), ph.flux);
                mark_tile_charge(P, bs.offset, bs.nx, di, dj);
                in_delta = P.track_static_delta == 2;
            }
        }
        const int px = ix - P.xmin, py = iy - P.ymin;
        if (px < 0 || px >= P.nx || py < 0 || py >= P.ny) continue;
        const int64_t pidx = (int64_t)py * P.nx + px;
        if (!in_delta) unsafeAtomicAdd(P.image + pidx, ph.flux);
        if (pixel_index_out) pixel_index_out[i] = (int32_t)pidx;
    }
}

// sensor.accumulate for one ROUND of the bright objects: segment-mapped like the fused kernel, but
// the photon (already through the op chain AND the boundary-independent half of the sensor step: the
// `converted` pool format) is loaded from the pool at pool_start[object] + j.
// One workgroup = the photons [j0, j0 + 256) of object `oi` (clipped to j_end).
template <int NV = 0, int WG = 256>
__device__ __forceinline__ void accumulate_segment(const ims_render_params_t& Pg, const ims_photons_t& pool,
                                                   const int64_t* __restrict__ pool_start, int64_t oi, int64_t j0, int64_t j_end,
                                                   const TileLister* lister = nullptr)
{
    const ims_object_t& og = Pg.objects[oi];
    const int64_t left = j_end - j0;                                            // photons of this segment (>= 1)
    const int n_thr = left >= WG ? WG : ((((int)left + 63) >> 6) << 6);
    if ((int)threadIdx.x >= n_thr) return;                                      // photon-less wavefronts leave at once
    const int64_t j = j0 + threadIdx.x;
    // (The prologue of a round's pixel search is a chain of ~17 dependent scalar loads -- the fields of the chain's argument block, of
    // the object row and of the sensor block, each fetched where it is first used, each a cold round trip behind the kernel boundary.
    // Holding them all in one batch with empty asm statements cut the chain to 7 but took the kernel from 94 registers to 128 + 64 ..
    // 156 B of scratch: not kept.  What is kept: the chain of a joint launch found with ONE vector load (joint_chain), and the
    // photon's record requested as soon as its address is known, all four fields at once -- the three others used to wait for the
    // flux -- so that its trip runs beside the rest of the chain: C3 24.08 -> 23.94 ms, C5 6.92 -> 6.83 ms per CCD, round 6.)
    const ims_object_t& o = og;
    const ims_render_params_t& P = Pg;
    const int64_t pstart = pool_start[oi];
    const bool have_sensor = Pg.sensor != nullptr;
    const bool silicon = have_sensor && (P.sensor->kind == IMS_SENSOR_SILICON);
    // the photon's record: all four fields requested before anything of it is looked at (the three others used to wait for the flux)
    const int64_t ip = pstart + (j < j_end ? j : j0);
    const double x0 = pool.x[ip], y0 = pool.y[ip], flux_in = pool.flux[ip], zs = pool.dxdz[ip];
    __shared__ float tile[CT * CT];
    ChargeTile ct;
    PROBE(1);
    tile_begin(tile, ct, P, o, silicon, n_thr, lister);
    PROBE(2);
    PROBE_WG(1, 0);
    double added = 0.0;
    if (j < j_end) {
        int ix, iy;
        // the photon arrives at its conversion depth, diffused (converted pool): only the pixel search is left
        const double flux = flux_in;
        bool ok = false;
        if (flux != 0.0) {
            PROBE(3);
            PROBE_WG(2, 0);
            if (!silicon || (o.flags & IMS_OBJ_FAINT)) {
                ix = (int)floor(x0 + 0.5); iy = (int)floor(y0 + 0.5);
                ok = !(ix < o.stamp_xmin || ix > o.stamp_xmax || iy < o.stamp_ymin || iy > o.stamp_ymax);
            } else {
                const bool coin = __double_as_longlong(zs) < 0;
                ok = land_search<NV, true>(P, o, x0, y0, fabs(zs), coin, ix, iy);
            }
        }
        PROBE(4);
        PROBE_WG(5, 0);
        if (ok) {
            added = flux;
            tile_deposit(tile, ct, P, ix, iy, flux);
        }
    }
    tile_flush(tile, ct, P, n_thr);
    PROBE(6);
    PROBE_WG(7, 0);
    if (P.realized_flux != nullptr) {
        const double tot = wave_sum(added);
        if ((threadIdx.x & 63) == 0 && tot != 0.0) unsafeAtomicAdd(P.realized_flux + oi, tot);
    }
}

// TWO segments of one object per workgroup (photons j0 + t and j0 + 256 + t on thread t): both pool records are requested before
// the first search, so the second photon's record arrives behind the first photon's dependent round trips, and a round has half
// the workgroups to find wave slots for.  The two bodies are written out (a rolled loop lost the register allocation in round 3).
// Same photons, same LDS tile arithmetic (integer counts): the same image bits.  ims_tuning_t.round_two_segments.
template <int NV = 0>
__device__ __forceinline__ void accumulate_segment2(const ims_render_params_t& P, const ims_photons_t& pool,
                                                    const int64_t* __restrict__ pool_start, int64_t oi, int64_t j0, int64_t j_end)
{
    const ims_object_t& o = P.objects[oi];
    const int64_t left = j_end - j0;                                            // photons of this pair of segments (>= 1)
    const int n_thr = left >= 256 ? 256 : ((((int)left + 63) >> 6) << 6);
    if ((int)threadIdx.x >= n_thr) return;                                      // photon-less wavefronts leave at once
    const int64_t ja = j0 + threadIdx.x, jb = ja + 256;
    const bool silicon = (P.sensor != nullptr) && (P.sensor->kind == IMS_SENSOR_SILICON);
    __shared__ float tile[CT * CT];
    ChargeTile ct;
    tile_begin(tile, ct, P, o, silicon, n_thr);
    double added = 0.0;
    const bool has_a = ja < j_end, has_b = jb < j_end;
    const int64_t ia = pool_start[oi] + (has_a ? ja : j0), ib = pool_start[oi] + (has_b ? jb : j0);
    // both records in flight before anything is used
    const double xa = pool.x[ia], ya = pool.y[ia], fa = pool.flux[ia], za = pool.dxdz[ia];
    const double xb = pool.x[ib], yb = pool.y[ib], fb = pool.flux[ib], zb = pool.dxdz[ib];
    const bool plain = !silicon || (o.flags & IMS_OBJ_FAINT);
    if (has_a && fa != 0.0) {
        int ix, iy;
        bool ok;
        if (plain) {
            ix = (int)floor(xa + 0.5); iy = (int)floor(ya + 0.5);
            ok = !(ix < o.stamp_xmin || ix > o.stamp_xmax || iy < o.stamp_ymin || iy > o.stamp_ymax);
        } else ok = land_search<NV, true>(P, o, xa, ya, fabs(za), __double_as_longlong(za) < 0, ix, iy);
        if (ok) { added += fa; tile_deposit(tile, ct, P, ix, iy, fa); }
    }
    if (has_b && fb != 0.0) {
        int ix, iy;
        bool ok;
        if (plain) {
            ix = (int)floor(xb + 0.5); iy = (int)floor(yb + 0.5);
            ok = !(ix < o.stamp_xmin || ix > o.stamp_xmax || iy < o.stamp_ymin || iy > o.stamp_ymax);
        } else ok = land_search<NV, true>(P, o, xb, yb, fabs(zb), __double_as_longlong(zb) < 0, ix, iy);
        if (ok) { added += fb; tile_deposit(tile, ct, P, ix, iy, fb); }
    }
    tile_flush(tile, ct, P, n_thr);
    if (P.realized_flux != nullptr) {
        const double tot = wave_sum(added);
        if ((threadIdx.x & 63) == 0 && tot != 0.0) unsafeAtomicAdd(P.realized_flux + oi, tot);
    }
}

template <int NV>
__global__ __launch_bounds__(256, (NV == 8) ? 2 : 4) void k_accumulate_segments(const ims_render_params_t P, const ims_photons_t pool,
                                                                                const int64_t* __restrict__ pool_start)
{
    const int64_t per = (P.n_segments + N_XCD - 1) / N_XCD;
    const int64_t b = blockIdx.x;
    if ((b / N_XCD) >= per) return;
    const int64_t seg = xcd_segment(b, P.n_segments);
    if (seg >= P.n_segments) return;
    const int64_t oi = P.seg_object ? (int64_t)P.seg_object[seg] : find_object(P.seg_prefix, P.n_objects, seg);
    accumulate_segment<NV>(P, pool, pool_start, oi, (seg - P.seg_prefix[oi]) * P.seg_size, P.objects[oi].n_phot);
}

// The same for objects of at most a wavefront or two of photons (the per-batch shares of photon-pooling mode: median 19):
// ONE WAVEFRONT per object, four independent objects per workgroup, no LDS tile and no barrier -- a share of a few dozen
// photons lands in as many different pixels, so the tile would merge nothing, and a million workgroups of one live
// wavefront each are bound by the dispatch rate (9 ms per batch of C4 for 2 ms of work).  The object index is made
// wave-uniform with readfirstlane so that the row still comes through scalar loads.
template <int NV>
__global__ __launch_bounds__(256, (NV == 8) ? 2 : 5) void k_accumulate_small(const ims_render_params_t P, const ims_photons_t pool,
                                                                             const int64_t* __restrict__ pool_start)
{
    const int64_t first = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6));
    const int64_t oi = ((int64_t)__builtin_amdgcn_readfirstlane((int)(first >> 32)) << 32) |
                       (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)first);
    if (oi >= P.n_objects) return;
    const ims_object_t& o = P.objects[oi];
    const bool silicon = (P.sensor != nullptr) && (P.sensor->kind == IMS_SENSOR_SILICON);
    ChargeTile ct;
    ct.x0 = 0; ct.y0 = 0;
    ct.track = silicon && !(o.flags & IMS_OBJ_FAINT) && (o.bf_state > 0 || P.track_static_delta);
    if (ct.track) ct.slot = P.sensor->bf_slots[slot_index(P, o)];
    const int64_t base = pool_start[oi], n = o.n_phot;
    double added = 0.0;
#pragma unroll 1
    for (int64_t j = threadIdx.x & 63; j < n; j += 64) {
        const int64_t i = base + j;
        const double x0 = pool.x[i], y0 = pool.y[i], flux = pool.flux[i], zs = pool.dxdz[i];
        if (flux == 0.0) continue;
        int ix, iy;
        bool ok;
        if (!silicon || (o.flags & IMS_OBJ_FAINT)) {
            ix = (int)floor(x0 + 0.5); iy = (int)floor(y0 + 0.5);
            ok = !(ix < o.stamp_xmin || ix > o.stamp_xmax || iy < o.stamp_ymin || iy > o.stamp_ymax);
        } else {
            ok = land_search<NV, true>(P, o, x0, y0, fabs(zs), __double_as_longlong(zs) < 0, ix, iy);
        }
        if (ok) {
            added += flux;
            deposit_global(P, ct, ix, iy, flux);
        }
    }
    if (P.realized_flux != nullptr) {
        const double tot = wave_sum(added);
        if ((threadIdx.x & 63) == 0 && tot != 0.0) unsafeAtomicAdd(P.realized_flux + oi, tot);
    }
}

// The same for round `round` of a chain class whose table holds the objects' FULL photon counts: the round covers the
// photons [round * nrecalc, (round + 1) * nrecalc) of every object, `segs` = ceil(nrecalc / 256) workgroups per object;
// the first n_active rows (sorted by photon count, brightest first) are the objects that reach this round.
// At most 128 VGPRs (4 and 0 vertices; 8 needs the whole file): the rounds share the GPU with the 128-VGPR photon kernels,
// and a wave that needs 136 registers only starts where TWO of those have left a SIMD (measured: the rounds 4 x slower).
template <int NV, int WG = 256>
__global__ __launch_bounds__(WG, (NV == 8) ? 2 : 4) void k_accumulate_round(const ims_render_params_t P, const ims_photons_t pool,
                                                          const int64_t* __restrict__ pool_start, int64_t round_first, int32_t nrecalc,
                                                          int32_t segs)
{
    PROBE(0);
    PROBE_WG(0, 0);
    const int64_t oi = blockIdx.x / segs;
    const int64_t j0 = round_first + (int64_t)(blockIdx.x % segs) * WG;
    int64_t j_end = round_first + nrecalc;
    const int64_t n = P.objects[oi].n_phot;
    if (j_end > n) j_end = n;
    if (j0 >= j_end) return;
    accumulate_segment<NV, WG>(P, pool, pool_start, oi, j0, j_end);
}

// The same with a COMPACT argument block: the pixel search reads a dozen fields of the 1.4-KB launch parameters (object table,
// sensor, image and its bounds, the realized-flux array, the tile tag and slot shift) and four pointers of the pool.  A chain is
// thousands of launches, and HIP's kernel-argument pool holds only so many 1.4-KB blocks: with it full the host enqueues at the
// GPU's pace (25 us of host time per round of a long chain measured) -- which bounds a focal plane of CCDs with long chains.
// The kernel rebuilds the two descriptors in registers (the unused fields vanish), so the body is the one above.
struct RoundArgs {
    const ims_object_t* objects;
    const ims_sensor_t* sensor;
    double* image;
    double* realized_flux;
    const double *px, *py, *pflux, *pz;
    int32_t nx, ny, xmin, ymin;
    uint32_t bf_tag, bf_slot_shift;
    int32_t track_static_delta, pad;
};

template <int NV, int WG = 256>
__global__ __launch_bounds__(WG, (NV == 8) ? 2 : 4) void k_accumulate_round_c(const RoundArgs a, const int64_t* __restrict__ pool_start,
                                                          int64_t round_first, int32_t nrecalc, int32_t segs)
{
    ims_render_params_t P;
    P.objects = a.objects; P.sensor = a.sensor; P.image = a.image; P.realized_flux = a.realized_flux;
    P.nx = a.nx; P.ny = a.ny; P.xmin = a.xmin; P.ymin = a.ymin;
    P.bf_tag = a.bf_tag; P.bf_slot_shift = a.bf_slot_shift; P.track_static_delta = a.track_static_delta;
    ims_photons_t pool;
    pool.x = const_cast<double*>(a.px); pool.y = const_cast<double*>(a.py); pool.flux = const_cast<double*>(a.pflux);
    pool.dxdz = const_cast<double*>(a.pz);
    // (the object index through readfirstlane: the division runs on the vector unit, and the loads of the row are to be scalar loads)
    const int64_t oi = (int64_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x / (unsigned)segs));
    const int64_t j0 = round_first + (int64_t)(blockIdx.x % segs) * WG;
    int64_t j_end = round_first + nrecalc;
    const int64_t n = P.objects[oi].n_phot;
    if (j_end > n) j_end = n;
    if (j0 >= j_end) return;
    accumulate_segment<NV, WG>(P, pool, pool_start, oi, j0, j_end);
}

template <int NV>
__global__ __launch_bounds__(256, 4) void k_accumulate_round_c2(const RoundArgs a, const int64_t* __restrict__ pool_start,
                                                                int64_t round_first, int32_t nrecalc, int32_t segs2)
{
    ims_render_params_t P;
    P.objects = a.objects; P.sensor = a.sensor; P.image = a.image; P.realized_flux = a.realized_flux;
    P.nx = a.nx; P.ny = a.ny; P.xmin = a.xmin; P.ymin = a.ymin;
    P.bf_tag = a.bf_tag; P.bf_slot_shift = a.bf_slot_shift; P.track_static_delta = a.track_static_delta;
    ims_photons_t pool;
    pool.x = const_cast<double*>(a.px); pool.y = const_cast<double*>(a.py); pool.flux = const_cast<double*>(a.pflux);
    pool.dxdz = const_cast<double*>(a.pz);
    const int64_t oi = blockIdx.x / segs2;
    const int64_t j0 = round_first + (int64_t)(blockIdx.x % segs2) * 512;
    int64_t j_end = round_first + nrecalc;
    const int64_t n = P.objects[oi].n_phot;
    if (j_end > n) j_end = n;
    if (j0 >= j_end) return;
    accumulate_segment2<NV>(P, pool, pool_start, oi, j0, j_end);
}

// ---- JOINT rounds: the same round of the top chains of several CCDs in one launch ----
// A focal plane's CCDs are independent renders (own sensor state, image, pool, object table), and with a bright tail each is
// bound by the dependent rounds of its brightest star -- hundreds of launches of a few dozen workgroups.  Chains of different
// CCDs on different streams barely overlap on this device (four side by side 108 - 150 ms against 158 ms
// one after the other), while one launch that holds the round of MANY objects costs little more than the round of one (C3: 41
// objects 2.8 x one star).  So the launch takes the argument blocks of up to IMS_JOINT_MAX chains by value and a workgroup
// finds its chain from the ascending workgroup ends (scalar compares on kernel arguments); the body is the one above.
constexpr int IMS_JOINT_MAX = 64;

struct JointEnds { int32_t v[IMS_JOINT_MAX]; };      // ascending workgroup ends of the chains of one launch (a chain that sits out: zero width)

struct JointAcc {                           // per chain, constant over the rounds: lives in device memory (ims_plans_run_joint)
    RoundArgs a[IMS_JOINT_MAX];
    const int64_t* pool_start[IMS_JOINT_MAX];
    const int64_t* tile_prefix[IMS_JOINT_MAX];      // search-side lists (TileLister): tiles before each slot of the class,
    unsigned int* words[IMS_JOINT_MAX];             // the chain's tile words,
    int32_t first_slot[IMS_JOINT_MAX];              // and the class's first slot
};

// The tables of ONE round of a joint run -- workgroup ends of the pixel search (ea), tile ends (eu), regions that go on (ns),
// workgroup ends of the list builder (ewg), tiles per chain (tof) -- live in DEVICE memory, all rounds of the run uploaded at
// once (ims_plans_run_joint), and a launch takes a pointer to its round's block.  (By value they were 128 .. 768 bytes of
// kernel arguments per launch, four launches per round, thousands of rounds per batch, and would have doubled with 64 chains
// per run.)  Read through the constant address space: uniform addresses, scalar loads.
struct JointRound { JointEnds ea, eu, ns, ewg, tof; };
typedef const JointEnds __attribute__((address_space(4))) * ConstEnds;

// which chain workgroup b belongs to: the number of ends <= b.  One entry per lane (a single 256-byte load), a ballot and a population
// count: ONE round trip at the head of every round kernel, where 63 scalar compares on four batches of scalar loads were three
// (the registers do not hold 64 ends at once) -- behind a kernel boundary every trip of the prologue is a cold one (round 6).
// All 64 lanes must be active.
__device__ __forceinline__ int joint_chain(const JointEnds* e_global, int& b)
{
    const int lane = (int)(threadIdx.x & 63);
    const int v = e_global->v[lane];
    const int c = (int)__popcll(__builtin_amdgcn_ballot_w64(lane < IMS_JOINT_MAX - 1 && b >= v));
    if (c > 0) b -= __builtin_amdgcn_readlane(v, c - 1);
    return c;
}
__device__ __forceinline__ int joint_entry(const JointEnds* e_global, int c) { return ((ConstEnds)(uintptr_t)e_global)->v[c]; }

__device__ __forceinline__ TileListerChain lister_chain(const TileLister& L, int slot_idx)
{
    typedef const JointAcc __attribute__((address_space(4))) * ConstAcc;
    ConstAcc Jc = (ConstAcc)(uintptr_t)L.J;
    TileListerChain C;
    C.lo = slot_idx - Jc->first_slot[L.chain];
    C.tile_base = Jc->tile_prefix[L.chain][C.lo];
    C.words = Jc->words[L.chain];
    C.on = C.tile_base < joint_entry(&L.R->tof, L.chain);          // (regions are in the order of their chains' lengths)
    return C;
}

template <int NV, int WG = 256>
__global__ __launch_bounds__(WG, (NV == 8) ? 2 : 4) void k_accumulate_round_j(const JointAcc* __restrict__ J, const JointRound* __restrict__ R,
                                                                              uint32_t tag, int64_t round_first, int32_t nrecalc, int32_t segs,
                                                                              unsigned long long* list, int* list_count, unsigned int round_word)
{
    int b = (int)blockIdx.x;
    const int c = joint_chain(&R->ea, b);
    // the table through the CONSTANT address space: its loads at uniform addresses are scalar loads whatever the kernel stores
    // (through a plain pointer the twelve fields sit in vector registers for the whole body: 128 + spills instead of 92)
    typedef const JointAcc __attribute__((address_space(4))) * ConstAcc;
    ConstAcc Jc = (ConstAcc)(uintptr_t)J;
    ims_render_params_t P;
    P.objects = Jc->a[c].objects; P.sensor = Jc->a[c].sensor; P.image = Jc->a[c].image; P.realized_flux = Jc->a[c].realized_flux;
    P.nx = Jc->a[c].nx; P.ny = Jc->a[c].ny; P.xmin = Jc->a[c].xmin; P.ymin = Jc->a[c].ymin;
    P.bf_tag = tag; P.bf_slot_shift = 0; P.track_static_delta = Jc->a[c].track_static_delta;
    ims_photons_t pool;
    pool.x = const_cast<double*>(Jc->a[c].px); pool.y = const_cast<double*>(Jc->a[c].py); pool.flux = const_cast<double*>(Jc->a[c].pflux);
    pool.dxdz = const_cast<double*>(Jc->a[c].pz);
    const int64_t* pool_start = Jc->pool_start[c];
    const int64_t oi = b / segs;
    const int64_t j0 = round_first + (int64_t)(b % segs) * WG;
    int64_t j_end = round_first + nrecalc;
    const int64_t n = P.objects[oi].n_phot;
    if (j_end > n) j_end = n;
    if (j0 >= j_end) return;
    TileLister tl;
    tl.list = list; tl.count = list_count; tl.round_word = round_word; tl.chain = c; tl.J = J; tl.R = R;
    accumulate_segment<NV, WG>(P, pool, pool_start, oi, j0, j_end, &tl);
}

// Which slot of a range does block b work on: the last k with prefix[k] <= b (prefix = ascending tile offsets of the slots,
// prefix[0] = 0).  A bisection is log2(n) DEPENDENT loads at the head of every tile kernel -- 6 for the 41 regions of the
// long chains, 11 for the 1 550 of the middle class, each a cold-L2 round trip of ~1.4 us in a kernel that runs for 6 - 30 us.
// Here the wavefront looks at 64 entries at a time (one load per lane, a ballot, a population count): one load for up to 64
// slots, two up to 4 096.  Every wavefront of the workgroup computes the same (uniform) answer; all 64 lanes must be active.
__device__ __forceinline__ int find_slot(const int64_t* __restrict__ prefix, int n_slots, int64_t b)
{
    const int lane = threadIdx.x & 63;
    int base = 0, n = n_slots;
    while (n > 64) {
        const int stride = (n + 63) >> 6;
        const int k = lane * stride;
        const bool le = k < n && prefix[base + k] <= b;
        const int cnt = __popcll(__builtin_amdgcn_ballot_w64(le));       // entries 0, stride, 2 stride, ... <= b form a prefix
        const int first = (cnt - 1) * stride;
        base += first;
        n = (n - first < stride) ? n - first : stride;
    }
    const bool le = lane < n && prefix[base + lane] <= b;
    return base + __popcll(__builtin_amdgcn_ballot_w64(le)) - 1;
}

// ---------------- Silicon boundary state ----------------
__device__ __forceinline__ void empty_owned(const ims_sensor_t& s, int n, double& x, double& y)
{
    const int nV = s.num_vertices;
    if (n == 0) { x = 0.0; y = 0.0; return; }
    if (n <= nV) { x = s.emptypoly[2 * n]; y = 0.0; return; }
    if (n == nV + 1) { x = 1.0; y = 0.0; return; }
    const int m = n - nV - 2;
    x = 0.0; y = s.emptypoly[2 * (1 + m)];
}

// What the tree-ring closed form needs of the sensor block, read ONCE per thread.  (Read through `s` at every evaluation, the
// stores in between -- the state arrays hang off the same block -- made the compiler reload the block and redo everything that
// depends on tr_dr alone: the Newton steps of 1 / tr_dr and the IEEE division tr_dr^2 / 6, ~30 of an evaluation's ~115 VALU
// instructions, twenty evaluations per cell in k_init_tiles; SQ_INSTS_VALU of round 5: the initial state was 24 % of a focal
// plane's VALU work.)  Same operations on the same operands: same bits.
struct TreeRing {
    double dr, dr_y, h2, cx, cy;
    const double* table;
    const double* table2;
    int n;
};

__device__ __forceinline__ TreeRing treering_of(const ims_sensor_t& s)
{
    TreeRing T;
    T.n = s.n_tr; T.dr = s.tr_dr; T.cx = s.tr_cx; T.cy = s.tr_cy; T.table = s.tr_table; T.table2 = s.tr_table2;
    T.dr_y = 0.0; T.h2 = 0.0;
    if (T.n > 0) {
        // the part of ddiv(r, dr) that does not depend on r (ims_math.h: two Newton steps on the reciprocal seed)
        double y = __builtin_amdgcn_rcp(T.dr);
        double e = fma(-T.dr, y, 1.0);
        y = fma(y, e, y);
        e = fma(-T.dr, y, 1.0);
        T.dr_y = fma(y, e, y);
        T.h2 = T.dr * T.dr / 6.0;
    }
    return T;
}

__device__ __forceinline__ double treering_shift(const TreeRing& T, double r)
{
    if (T.n <= 0) return 0.0;
    // ddiv(r, dr) with the reciprocal prepared above
    const double q = r * T.dr_y;
    const double rem = fma(-T.dr, q, r);
    const double f = fma(rem, T.dr_y, q);
    if (!(f > 0.0) || f >= (double)(T.n - 1)) return 0.0;
    const int i = (int)f;
    const double b = f - (double)i;
    const double v0 = T.table[i], v1 = T.table[i + 1];
    if (T.table2 == nullptr) return v0 + b * (v1 - v0);
    const double a = 1.0 - b;
    return a * v0 + b * v1 + ((a * a * a - a) * T.table2[i] + (b * b * b - b) * T.table2[i + 1]) * T.h2;
}

// The boundary kernels run over a RANGE of slots in one launch: thread -> global owner cell ->
// slot by binary search over the packed cell offsets (wave-uniform except at slot seams).
struct CellRef { SlotView sl; int i, j; int64_t c; bool valid; };

__device__ __forceinline__ CellRef locate_cell(const ims_sensor_t& s, int first_slot, int n_slots, int64_t cell_begin,
                                               int64_t cell_count)
{
    CellRef r;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    r.valid = t < cell_count;
    if (!r.valid) return r;
    const int64_t g = cell_begin + t;
    int lo = first_slot, hi = first_slot + n_slots;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (s.bf_slots[mid].offset <= g) lo = mid; else hi = mid;
    }
    const ims_bf_slot_t bs = s.bf_slots[lo];
    r.sl = { bs.xmin, bs.ymin, bs.nx, bs.ny, bs.offset };
    r.c = g - bs.offset;
    r.i = (int)(r.c % (bs.nx + 1));
    r.j = (int)(r.c / (bs.nx + 1));
    return r;
}

// owned boundary point n of owner cell (ci, cj) in its undistorted-plus-tree-ring state (pixel-local coordinates)
__device__ __forceinline__ void init_point(const ims_sensor_t& s, const TreeRing& T, const SlotView& sl, int ci, int cj, int n,
                                           double& px, double& py)
{
    double ex, ey;
    empty_owned(s, n, ex, ey);
    const double tx = ((double)(sl.xmin + ci) - 0.5 + ex) - T.cx;
    const double ty = ((double)(sl.ymin + cj) - 0.5 + ey) - T.cy;
    const double rr = dsqrt0(tx * tx + ty * ty);
    const double sh = treering_shift(T, rr);
    px = ex; py = ey;
    if (rr > 0.0 && sh != 0.0) { px = ex + ddiv(sh * tx, rr); py = ey + ddiv(sh * ty, rr); }
}

// Initial state of a range of slots: boundary points, zero delta charge AND the bounds line of every pixel.
// The bounds need the points of the right and upper neighbour cells; they are recomputed here from the same
// closed form (bit-identical to what those cells store) instead of being read back in a second pass.
__global__ __launch_bounds__(256) void k_init_boundaries(const ims_sensor_t* __restrict__ sp, int first_slot, int n_slots,
                                                         int64_t cell_begin, int64_t cell_count)
{
    const ims_sensor_t& s = *sp;
    const CellRef r = locate_cell(s, first_slot, n_slots, cell_begin, cell_count);
    if (!r.valid) return;
    const SlotView& sl = r.sl;
    const TreeRing T = treering_of(s);
    const int nV = s.num_vertices, npo = 2 * nV + 2, nv = 4 * nV + 4;
    double* pts = s.bf_boundary + (sl.offset + r.c) * npo * 2;
    for (int n = 0; n < npo; ++n) {
        double px, py;
        init_point(s, T, sl, r.i, r.j, n, px, py);
        pts[2 * n] = px; pts[2 * n + 1] = py;
    }
    s.bf_delta[sl.offset + r.c] = 0.0;
    const int i = r.i, j = r.j;
    if (i >= sl.nx || j >= sl.ny) return;
    double ixmin = 0.0, ixmax = 1.0, iymin = 0.0, iymax = 1.0;
    double oxmin = 0.0, oxmax = 1.0, oymin = 0.0, oymax = 1.0;
    double v0x = 0.0;
    for (int k = 0; k < nv; ++k) {
        // same vertex -> (cell, owned point) map as polygon_vertex
        int ci = i, cj = j, q;
        double ax = 0.0, ay = 0.0;
        if (k <= nV + 1) { q = k; }
        else if (k <= 2 * nV + 1) { ci = i + 1; ax = 1.0; q = nV + 2 + (k - nV - 2); }
        else if (k <= 3 * nV + 3) { cj = j + 1; ay = 1.0; q = nV + 1 - (k - 2 * nV - 2); }
        else { q = nV + 2 + (nV - 1 - (k - 3 * nV - 4)); }
        double vx, vy;
        init_point(s, T, sl, ci, cj, q, vx, vy);
        vx = vx + ax; vy = vy + ay;
        if (k == 0) v0x = vx;
        if (vx < oxmin) oxmin = vx;
        if (vx > oxmax) oxmax = vx;
        if (vy < oymin) oymin = vy;
        if (vy > oymax) oymax = vy;
        if (k <= nV + 1) { if (vy > iymin) iymin = vy; }
        if (k >= nV + 1 && k <= 2 * nV + 2) { if (vx < ixmax) ixmax = vx; }
        if (k >= 2 * nV + 2 && k <= 3 * nV + 3) { if (vy < iymax) iymax = vy; }
        if (k >= 3 * nV + 3) { if (vx > ixmin) ixmin = vx; }
    }
    if (v0x > ixmin) ixmin = v0x;
    double* b = s.bf_bounds + (sl.offset + r.c) * 8;
    b[0] = ixmin; b[1] = ixmax; b[2] = iymin; b[3] = iymax;
    b[4] = oxmin; b[5] = oxmax; b[6] = oymin; b[7] = oymax;
}

// The same initial state, one 16 x 16 tile of owner cells per workgroup (grid.y = slot of the range, grid.x = tile; 4
// vertices per edge): every lane evaluates the ten owned points of ITS cell once, hands the left-edge and bottom-row
// points its left / lower neighbours need through LDS, and only the lanes at the right / upper rim of the tile evaluate
// their neighbour's points themselves -- ~12 evaluations of the tree-ring closed form per cell instead of 30 (each is a
// sqrt, two divisions and a spline).  Same function of (cell, point), so the same bits as k_init_boundaries.
constexpr int UT = 16;            // tile edge of the boundary kernels (mark_tile_charge assumes 16)
constexpr int IT_NV = 4, IT_NPO = 2 * IT_NV + 2, IT_NVT = 4 * IT_NV + 4;

// tile_prefix (device, prefix sum of the tiles of the slots first_slot ..): a 1-D grid over exactly the tiles of the
// range, block -> (slot, tile) by binary search as in the update kernels; NULL: grid.y = slot, grid.x = tile of the slot
// (blocks beyond the slot's tiles leave).
__global__ __launch_bounds__(256) void k_init_tiles(const ims_sensor_t* __restrict__ sp, int first_slot, int n_slots,
                                                    const int64_t* __restrict__ tile_prefix)
{
    // the owned points of the tile's cells in the order of the global array -- [row][cell][point], ten double2 per cell --
    // so that they go out as contiguous runs (a lane that stored its own ten points wrote 16 bytes every 160: 64 requests
    // per store instruction; 2.16 ms for the 16.8 M cells of a CCD, 1.8 TB/s of the bytes written).  The bounds lines reuse
    // the buffer for the same reason.
    __shared__ double2 P[UT * UT * IT_NPO];
    const ims_sensor_t& s = *sp;
    int slot = first_slot + (int)blockIdx.y, t = (int)blockIdx.x;
    if (tile_prefix != nullptr) {
        const int64_t b = blockIdx.x;
        const int lo = find_slot(tile_prefix, n_slots, b);
        slot = first_slot + lo;
        t = (int)(b - tile_prefix[lo]);
    }
    const ims_bf_slot_t bs = s.bf_slots[slot];
    const SlotView sl = { bs.xmin, bs.ymin, bs.nx, bs.ny, bs.offset };
    const int tiles_x = (sl.nx + 1 + UT - 1) / UT, tiles_y = (sl.ny + 1 + UT - 1) / UT;
    if (t >= tiles_x * tiles_y) return;
    const TreeRing T = treering_of(s);
    const int lx = threadIdx.x % UT, ly = threadIdx.x / UT;
    const int tx0 = (t % tiles_x) * UT, ty0 = (t / tiles_x) * UT;
    const int i = tx0 + lx, j = ty0 + ly;
    const bool owner = (i <= sl.nx && j <= sl.ny);
    double2 own[IT_NPO];
    if (owner) {
#pragma unroll
        for (int n = 0; n < IT_NPO; ++n) {
            init_point(s, T, sl, i, j, n, own[n].x, own[n].y);
            P[(ly * UT + lx) * IT_NPO + n] = own[n];
        }
        s.bf_delta[cell_index(sl, i, j)] = 0.0;
    }
    __syncthreads();
    // points out: row ly of the tile is one contiguous run of (cells of the slot in that row) x ten double2
    {
        const int ncx = (sl.nx + 1 - tx0 < UT) ? sl.nx + 1 - tx0 : UT;           // owner cells per row of this tile
        const int nrow = (sl.ny + 1 - ty0 < UT) ? sl.ny + 1 - ty0 : UT;
#pragma unroll
        for (int e = (int)threadIdx.x; e < UT * UT * IT_NPO; e += 256) {
            const int r = e / (UT * IT_NPO), k = e - r * (UT * IT_NPO);
            if (r < nrow && k < ncx * IT_NPO)
                ((double2*)(s.bf_boundary + cell_index(sl, tx0, ty0 + r) * IT_NPO * 2))[k] = P[e];
        }
    }
    const bool inner = owner && i < sl.nx && j < sl.ny;                           // a pixel: has a bounds line
    // The points of the cells just outside the tile, which the pixels at its right / upper rim need, evaluated by the wavefront
    // TOGETHER (a wavefront is four rows of sixteen cells): the four left-edge points of the cell right of a row by the row's
    // lanes 12 .. 15, one each, handed to lane 15 by shuffles; the six bottom-row points of the cells above the tile by the
    // four rows of its last wavefront, two rounds.  (Each rim lane evaluating its own -- four and six evaluations that the whole
    // wavefront sits through -- made 14 / 20 evaluations per wavefront where 11 / 13 do: the closed form is most of this kernel.)
    static_assert(IT_NV == 4 && UT == 16, "the rim exchange below is written for 4 vertices per edge and 16 x 16 tiles");
    double2 rgt_rim[IT_NV], upp_rim[IT_NV + 2];
    {
        const int lane = (int)(threadIdx.x & 63), row_base = lane & ~15;
        double2 rr;
        init_point(s, T, sl, tx0 + UT, j, IT_NV + 2 + (lx & 3), rr.x, rr.y);
#pragma unroll
        for (int m = 0; m < IT_NV; ++m) { rgt_rim[m].x = __shfl(rr.x, row_base + 12 + m, 64); rgt_rim[m].y = __shfl(rr.y, row_base + 12 + m, 64); }
#pragma unroll
        for (int q = 0; q <= IT_NV + 1; ++q) upp_rim[q] = make_double2(0.0, 0.0);
        if (threadIdx.x >= 192) {                                                    // rows 12 .. 15: wave-uniform
            const int r = ly - 12;
            double2 ua, ub;
            init_point(s, T, sl, i, ty0 + UT, r, ua.x, ua.y);
            init_point(s, T, sl, i, ty0 + UT, 4 + (r & 1), ub.x, ub.y);
#pragma unroll
            for (int q = 0; q < 4; ++q) { upp_rim[q].x = __shfl(ua.x, lx + 16 * q, 64); upp_rim[q].y = __shfl(ua.y, lx + 16 * q, 64); }
#pragma unroll
            for (int q = 0; q < 2; ++q) { upp_rim[4 + q].x = __shfl(ub.x, lx + 16 * q, 64); upp_rim[4 + q].y = __shfl(ub.y, lx + 16 * q, 64); }
        }
    }
    double bnd[8] = { 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0 };
    if (inner) {
        // the right neighbour's left edge and the upper neighbour's bottom row: from LDS inside the tile, from the exchange at its rim
        double2 rgt[IT_NV], upp[IT_NV + 2];
        if (lx + 1 < UT) {
#pragma unroll
            for (int m = 0; m < IT_NV; ++m) rgt[m] = P[(ly * UT + lx + 1) * IT_NPO + IT_NV + 2 + m];
        } else {
#pragma unroll
            for (int m = 0; m < IT_NV; ++m) rgt[m] = rgt_rim[m];
        }
        if (ly + 1 < UT) {
#pragma unroll
            for (int q = 0; q <= IT_NV + 1; ++q) upp[q] = P[((ly + 1) * UT + lx) * IT_NPO + q];
        } else {
#pragma unroll
            for (int q = 0; q <= IT_NV + 1; ++q) upp[q] = upp_rim[q];
        }
        double ixmin = 0.0, ixmax = 1.0, iymin = 0.0, iymax = 1.0;
        double oxmin = 0.0, oxmax = 1.0, oymin = 0.0, oymax = 1.0;
        double v0x = 0.0;
#pragma unroll
        for (int k = 0; k < IT_NVT; ++k) {
            // same vertex -> (cell, owned point) map as polygon_vertex
            double vx, vy;
            if (k <= IT_NV + 1) { vx = own[k].x; vy = own[k].y; }
            else if (k <= 2 * IT_NV + 1) { vx = rgt[k - IT_NV - 2].x + 1.0; vy = rgt[k - IT_NV - 2].y; }
            else if (k <= 3 * IT_NV + 3) { vx = upp[IT_NV + 1 - (k - 2 * IT_NV - 2)].x; vy = upp[IT_NV + 1 - (k - 2 * IT_NV - 2)].y + 1.0; }
            else { vx = own[IT_NV + 2 + (IT_NV - 1 - (k - 3 * IT_NV - 4))].x; vy = own[IT_NV + 2 + (IT_NV - 1 - (k - 3 * IT_NV - 4))].y; }
            if (k == 0) v0x = vx;
            if (vx < oxmin) oxmin = vx;
            if (vx > oxmax) oxmax = vx;
            if (vy < oymin) oymin = vy;
            if (vy > oymax) oymax = vy;
            if (k <= IT_NV + 1) { if (vy > iymin) iymin = vy; }
            if (k >= IT_NV + 1 && k <= 2 * IT_NV + 2) { if (vx < ixmax) ixmax = vx; }
            if (k >= 2 * IT_NV + 2 && k <= 3 * IT_NV + 3) { if (vy < iymax) iymax = vy; }
            if (k >= 3 * IT_NV + 3) { if (vx > ixmin) ixmin = vx; }
        }
        if (v0x > ixmin) ixmin = v0x;
        bnd[0] = ixmin; bnd[1] = ixmax; bnd[2] = iymin; bnd[3] = iymax;
        bnd[4] = oxmin; bnd[5] = oxmax; bnd[6] = oymin; bnd[7] = oymax;
    }
    __syncthreads();                                   // everybody has read its neighbours' points: the buffer is free
    if (inner) {
#pragma unroll
        for (int q = 0; q < 4; ++q) P[(ly * UT + lx) * 4 + q] = make_double2(bnd[2 * q], bnd[2 * q + 1]);
    }
    __syncthreads();
    {
        const int ncx = (sl.nx - tx0 < UT) ? sl.nx - tx0 : UT;                   // pixels per row of this tile (may be <= 0)
        const int nrow = (sl.ny - ty0 < UT) ? sl.ny - ty0 : UT;
#pragma unroll
        for (int e = (int)threadIdx.x; e < UT * UT * 4; e += 256) {
            const int r = e / (UT * 4), k = e - r * (UT * 4);
            if (r < nrow && k < ncx * 4)
                ((double2*)(s.bf_bounds + cell_index(sl, tx0, ty0 + r) * 8))[k] = P[e];
        }
    }
}

// ---- slot 0 without stored state (ims_render_params_t.lazy_static): the photons the fused launch set aside ----
// Pixel (i, j) of slot 0 in its initial state, evaluated where it is needed: the ten owned points of its cell, the four left-edge
// points of the cell to the right, the six bottom-row points of the cell above -- init_point, the function k_init_tiles stores --
// and from them the bounds line as k_init_tiles forms it.  4 vertices per edge.
struct LazyPixel { double2 own[IT_NPO], rgt[IT_NV], upp[IT_NV + 2]; double b[8]; };

__device__ __forceinline__ void lazy_pixel(const ims_sensor_t& s, const TreeRing& T, const SlotView& sl, int i, int j, LazyPixel& px)
{
#pragma unroll
    for (int n = 0; n < IT_NPO; ++n) init_point(s, T, sl, i, j, n, px.own[n].x, px.own[n].y);
#pragma unroll
    for (int m = 0; m < IT_NV; ++m) init_point(s, T, sl, i + 1, j, IT_NV + 2 + m, px.rgt[m].x, px.rgt[m].y);
#pragma unroll
    for (int q = 0; q <= IT_NV + 1; ++q) init_point(s, T, sl, i, j + 1, q, px.upp[q].x, px.upp[q].y);
    double ixmin = 0.0, ixmax = 1.0, iymin = 0.0, iymax = 1.0;
    double oxmin = 0.0, oxmax = 1.0, oymin = 0.0, oymax = 1.0;
    double v0x = 0.0;
#pragma unroll
    for (int k = 0; k < IT_NVT; ++k) {
        double vx, vy;
        if (k <= IT_NV + 1) { vx = px.own[k].x; vy = px.own[k].y; }
        else if (k <= 2 * IT_NV + 1) { vx = px.rgt[k - IT_NV - 2].x + 1.0; vy = px.rgt[k - IT_NV - 2].y; }
        else if (k <= 3 * IT_NV + 3) { vx = px.upp[IT_NV + 1 - (k - 2 * IT_NV - 2)].x; vy = px.upp[IT_NV + 1 - (k - 2 * IT_NV - 2)].y + 1.0; }
        else { vx = px.own[IT_NV + 2 + (IT_NV - 1 - (k - 3 * IT_NV - 4))].x; vy = px.own[IT_NV + 2 + (IT_NV - 1 - (k - 3 * IT_NV - 4))].y; }
        if (k == 0) v0x = vx;
        if (vx < oxmin) oxmin = vx;
        if (vx > oxmax) oxmax = vx;
        if (vy < oymin) oymin = vy;
        if (vy > oymax) oymax = vy;
        if (k <= IT_NV + 1) { if (vy > iymin) iymin = vy; }
        if (k >= IT_NV + 1 && k <= 2 * IT_NV + 2) { if (vx < ixmax) ixmax = vx; }
        if (k >= 2 * IT_NV + 2 && k <= 3 * IT_NV + 3) { if (vy < iymax) iymax = vy; }
        if (k >= 3 * IT_NV + 3) { if (vx > ixmin) ixmin = vx; }
    }
    if (v0x > ixmin) ixmin = v0x;
    px.b[0] = ixmin; px.b[1] = ixmax; px.b[2] = iymin; px.b[3] = iymax;
    px.b[4] = oxmin; px.b[5] = oxmax; px.b[6] = oymin; px.b[7] = oymax;
}

// polygon_test (ims_photon.h) on the evaluated points: the same vertices in the same order through the same arithmetic
__device__ __forceinline__ bool lazy_polygon_test(const ims_sensor_t& s, const LazyPixel& px, double x, double y, double zfactor)
{
    const bool scaled = (zfactor != 1.0);
    bool inside = false;
    double lx, ly;
    {
        constexpr int k = IT_NVT - 1;                                                    // closing vertex: own left edge, first point
        lx = px.own[5 * IT_NV + 5 - k].x; ly = px.own[5 * IT_NV + 5 - k].y;
        if (scaled) {
            const double ex = s.emptypoly[2 * k], ey = s.emptypoly[2 * k + 1];
            lx = ex + (lx - ex) * zfactor;
            ly = ey + (ly - ey) * zfactor;
        }
    }
#pragma unroll
    for (int k = 0; k < IT_NVT; ++k) {
        double kx, ky;
        if (k <= IT_NV + 1) { kx = px.own[k].x; ky = px.own[k].y; }
        else if (k <= 2 * IT_NV + 1) { kx = px.rgt[k - IT_NV - 2].x + 1.0; ky = px.rgt[k - IT_NV - 2].y + 0.0; }
        else if (k <= 3 * IT_NV + 3) { kx = px.upp[3 * IT_NV + 3 - k].x + 0.0; ky = px.upp[3 * IT_NV + 3 - k].y + 1.0; }
        else { kx = px.own[5 * IT_NV + 5 - k].x; ky = px.own[5 * IT_NV + 5 - k].y; }
        if (scaled) {
            const double ex = s.emptypoly[2 * k], ey = s.emptypoly[2 * k + 1];
            kx = ex + (kx - ex) * zfactor;
            ky = ey + (ky - ey) * zfactor;
        }
        if ((ky > y) != (ly > y)) {
            const double dy = ly - ky;
            const double lhs = (x - kx) * dy, rhs = (lx - kx) * (y - ky);
            if ((dy > 0.0) ? (lhs < rhs) : (lhs > rhs)) inside = !inside;
        }
        lx = kx; ly = ky;
    }
    return inside;
}

// inside_pixel (ims_photon.h) for slot 0 without stored state; z: the conversion depth
__device__ __forceinline__ bool lazy_inside_pixel(const ims_sensor_t& s, const TreeRing& T, const SlotView& sl, int ix, int iy, double x, double y,
                                                  double z, bool want_edge, bool& off_edge)
{
    const int i = ix - sl.xmin, j = iy - sl.ymin;
    if (i < 0 || i >= sl.nx || j < 0 || j >= sl.ny) {
        if (want_edge) off_edge = true;
        return false;
    }
    LazyPixel px;
    lazy_pixel(s, T, sl, i, j, px);
    bool inside;
    if (x > px.b[0] && x < px.b[1] && y > px.b[2] && y < px.b[3]) inside = true;
    else if (!(x >= px.b[4] && x <= px.b[5] && y >= px.b[6] && y <= px.b[7])) inside = false;
    else inside = lazy_polygon_test(s, px, x, y, dtanh_pos(ddiv(z, 12.0)));
    if (!inside && want_edge) {
        off_edge = false;
        if (i == 0 && x < px.b[0]) off_edge = true;
        if (i == sl.nx - 1 && x > px.b[1]) off_edge = true;
        if (j == 0 && y < px.b[2]) off_edge = true;
        if (j == sl.ny - 1 && y > px.b[3]) off_edge = true;
    }
    return inside;
}

// The photons ims_shoot_accumulate set aside (margin_append): land_search from the point where it would have read the state.
// A neighbour whose outer bounds cannot hold the point is not evaluated: no vertex of a pristine polygon lies further than
// pristine_margin from its nominal place, so outer bounds lie within [-m, 1 + m] -- the candidates of the walk are the same.
__device__ __forceinline__ void margin_photon(const ims_render_params_t& P, const ims_sensor_t& s, const TreeRing& T, const SlotView& sl, double mm,
                                              int64_t r)
{
    const double* rec = P.margin_list + 5 * (size_t)r;
    const double x0 = rec[0], y0 = rec[1], zs = rec[2], flux = rec[3];
    const int64_t oi = __double_as_longlong(rec[4]);
    const ims_object_t& o = P.objects[oi];
    const double z = fabs(zs);
    const bool coin = __double_as_longlong(zs) < 0;
    int ix = (int)floor(x0 + 0.5), iy = (int)floor(y0 + 0.5);
    const double x = x0 - (double)ix + 0.5, y = y0 - (double)iy + 0.5;
    bool off_edge = false;
    bool found = lazy_inside_pixel(s, T, sl, ix, iy, x, y, z, true, off_edge);
    if (!found && off_edge) return;
    int step = 0;
    if (!found) {
        step = search_step(x, y);
        for (int m = 1; m < 9; ++m) {
            const int nb = ((m * step - 1) & 7) + 1;
            const int jx = ix + xoff(nb), jy = iy + yoff(nb);
            const int i = jx - sl.xmin, j = jy - sl.ymin;
            if (i < 0 || i >= sl.nx || j < 0 || j >= sl.ny) continue;
            const double xb = x - (double)xoff(nb), yb = y - (double)yoff(nb);
            if (xb < -mm || xb > 1.0 + mm || yb < -mm || yb > 1.0 + mm) continue;
            LazyPixel px;
            lazy_pixel(s, T, sl, i, j, px);
            if (!(xb >= px.b[4] && xb <= px.b[5] && yb >= px.b[6] && yb <= px.b[7])) continue;      // not a candidate of the walk
            bool in = (xb > px.b[0] && xb < px.b[1] && yb > px.b[2] && yb < px.b[3]);
            if (!in) in = lazy_polygon_test(s, px, xb, yb, dtanh_pos(ddiv(z, 12.0)));
            if (in) { ix = jx; iy = jy; found = true; break; }
        }
    }
    if (!found) {
        const int nb = coin ? 0 : step;
        ix = ix + xoff(nb); iy = iy + yoff(nb);
    }
    if (ix < o.stamp_xmin || ix > o.stamp_xmax || iy < o.stamp_ymin || iy > o.stamp_ymax) return;
    const int pxl = ix - P.xmin, pyl = iy - P.ymin;
    if (pxl >= 0 && pxl < P.nx && pyl >= 0 && pyl < P.ny) unsafeAtomicAdd(P.image + ((int64_t)pyl * P.nx + pxl), flux);
    if (P.realized_flux != nullptr) unsafeAtomicAdd(P.realized_flux + oi, flux);
}

// A workgroup takes 2 048 places of the list at a time (the wavefronts' octets, then the overflow region), collects the filled ones
// in LDS -- one place in six is -- and then works on them with every lane busy (a lane per place sat through the search with a sixth
// of its lanes: 0.97 ms for C3's launch where 0.25 do).  The order of the photons does not matter: integer counts, atomics.
constexpr int MARGIN_CHUNK = 2048;
__global__ __launch_bounds__(256, 2) void k_margin_photons(const ims_render_params_t P)
{
    __shared__ int q[MARGIN_CHUNK];
    __shared__ int qn;
    const ims_sensor_t& s = *P.sensor;
    const TreeRing T = treering_of(s);
    const ims_bf_slot_t bs = s.bf_slots[0];
    const SlotView sl = { bs.xmin, bs.ymin, bs.nx, bs.ny, bs.offset };
    const double mm = s.pristine_margin;
    unsigned int n_over = (unsigned int)P.margin_count[0];
    if (n_over > P.margin_cap) n_over = P.margin_cap;
    const int64_t n_oct = P.margin_waves * 8, n = n_oct + (int64_t)n_over;
    for (int64_t c0 = (int64_t)blockIdx.x * MARGIN_CHUNK; c0 < n; c0 += (int64_t)gridDim.x * MARGIN_CHUNK) {
        if (threadIdx.x == 0) qn = 0;
        __syncthreads();
        for (int k = (int)threadIdx.x; k < MARGIN_CHUNK; k += 256) {
            const int64_t r = c0 + k;
            const bool filled = r < n && (r >= n_oct || (int)(r & 7) < (int)P.margin_wave_count[r >> 3]);
            if (filled) q[atomicAdd(&qn, 1)] = k;
        }
        __syncthreads();
        const int m = qn;
        for (int k = (int)threadIdx.x; k < m; k += 256) margin_photon(P, s, T, sl, mm, c0 + q[k]);
        __syncthreads();
    }
}

__device__ __forceinline__ int owned_to_vertex(int nV, int n)
{
    if (n <= nV + 1) return n;
    const int m = n - nV - 2;
    return 3 * nV + 4 + (nV - 1 - m);
}

// Silicon::updatePixelDistortions.  One 16x16 workgroup per tile of owner cells of one slot: the
// delta-charge halo tile ((16+2q+1)^2 doubles) is staged in LDS once, then every thread gathers its
// charged neighbours from LDS in a FIXED order (so the result is bit-reproducible) and adds the
// scaled tabulated displacements to the boundary points it owns.  A per-cell `changed` byte lets
// k_refresh_changed skip pixels whose polygon did not move.
constexpr int UQMAX = 4;          // largest supported qdist
constexpr int UH = UT + 2 * UQMAX + 1;

// no charge was deposited (with this tag) within reach of tile (tx, ty): nothing can move there
__device__ __forceinline__ bool tile_out_of_reach(const ims_sensor_t& s, const SlotView& sl, int tx, int ty, unsigned int tag)
{
    if (tag == 0u || s.bf_tile_charge == nullptr) return false;
    const int tiles_x = (sl.nx + 1 + UT - 1) / UT, tiles_y = (sl.ny + 1 + UT - 1) / UT;
    // (the nine marks read together -- a neighbour that does not exist reads the tile's own mark -- not one after the other behind
    // `any ||`: up to nine dependent round trips at the head of a tile that turns out to be out of reach)
    const unsigned char* __restrict__ marks = s.bf_tile_charge;
    unsigned char m[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        int ux = tx + k % 3 - 1, uy = ty + k / 3 - 1;
        if (ux < 0 || uy < 0 || ux >= tiles_x || uy >= tiles_y) { ux = tx; uy = ty; }
        m[k] = marks[cell_index(sl, ux * UT, uy * UT)];
    }
    bool any = false;
#pragma unroll
    for (int k = 0; k < 9; ++k) any = any || (m[k] == (unsigned char)tag);
    return !any;
}

__global__ __launch_bounds__(256) void k_update_distortions(const ims_sensor_t* __restrict__ sp, int first_slot, int n_slots,
                                                            const int64_t* __restrict__ tile_prefix,
                                                            unsigned char* __restrict__ changed, unsigned int tag)
{
    __shared__ double tile[UH * UH];
    const ims_sensor_t& s = *sp;
    // block -> (slot, tile)
    const int64_t b = blockIdx.x;
    const int lo = find_slot(tile_prefix, n_slots, b);
    const ims_bf_slot_t bs = s.bf_slots[first_slot + lo];
    const SlotView sl = { bs.xmin, bs.ymin, bs.nx, bs.ny, bs.offset };
    const int tiles_x = (sl.nx + 1 + UT - 1) / UT;
    const int t = (int)(b - tile_prefix[lo]);
    const int tx0 = (t % tiles_x) * UT, ty0 = (t / tiles_x) * UT;
    if (tile_out_of_reach(s, sl, tx0 / UT, ty0 / UT, tag)) return;
    const int q = s.qdist;
    const int hw = UT + 2 * q + 1;                 // halo tile edge
    const int sx0 = tx0 - (q + 1), sy0 = ty0 - (q + 1);
    for (int e = threadIdx.x; e < hw * hw; e += 256) {
        const int hx = e % hw, hy = e / hw;
        const int si = sx0 + hx, sj = sy0 + hy;
        double v = 0.0;
        if (si >= 0 && si < sl.nx && sj >= 0 && sj < sl.ny) v = s.bf_delta[cell_index(sl, si, sj)];
        tile[hy * hw + hx] = v;
    }
    __syncthreads();
    const int lx = threadIdx.x % UT, ly = threadIdx.x / UT;
    const int i = tx0 + lx, j = ty0 + ly;
    if (i > sl.nx || j > sl.ny) return;
    const int nV = s.num_vertices, npo = 2 * nV + 2, nv = 4 * nV + 4;
    const int cx = (s.nx - 1) / 2, cy = (s.ny - 1) / 2;
    const int64_t c = cell_index(sl, i, j);
    double* pts = s.bf_boundary + c * npo * 2;
    bool any = false;
    for (int dj = -q; dj <= q + 1; ++dj) {
        const int sj = j - dj;
        if (sj < 0 || sj >= sl.ny) continue;
        for (int di = -q; di <= q + 1; ++di) {
            const int si = i - di;
            if (si < 0 || si >= sl.nx) continue;
            const double charge = tile[(sj - sy0) * hw + (si - sx0)];
            if (charge == 0.0) continue;
            any = true;
            const double w = ddiv(charge, s.num_elec);
            const double* dist = s.distortions + ((int64_t)(di + cx) * s.ny + (dj + cy)) * nv * 2;
            for (int n = 0; n < npo; ++n) {
                if (n <= nV + 1 && di == q + 1) continue;
                if (n > nV + 1 && dj == q + 1) continue;
                const int vtx = owned_to_vertex(nV, n);
                pts[2 * n] = fma(dist[2 * vtx], w, pts[2 * n]);
                pts[2 * n + 1] = fma(dist[2 * vtx + 1], w, pts[2 * n + 1]);
            }
        }
    }
    changed[c] = any ? 1 : 0;
    if (any && tag != 0u && s.bf_tile_changed != nullptr) s.bf_tile_changed[cell_index(sl, tx0, ty0)] = (unsigned char)tag;
}

// LDS of one q3 tile update: scaled charges of the halo, per-row occupancy bitmaps, displacement table
template <int NV>
struct UpdateLds {
    static constexpr int Q = 3, HW = UT + 2 * Q + 1, NPO = 2 * NV + 2;
    double wt[HW * HW];
    double dl[8 * 8 * NPO * 2];           // [dj+Q][di+Q][owned point][x,y]
    unsigned int occ[HW];
    int any_charge;
};

template <int NV>
__device__ __forceinline__ void load_displacements(const ims_sensor_t& s, UpdateLds<NV>& L)
{
    constexpr int Q = 3, NPO = 2 * NV + 2, NVV = 4 * NV + 4;
    const int cx = (s.nx - 1) / 2, cy = (s.ny - 1) / 2;
    for (int e = threadIdx.x; e < 8 * 8 * NPO * 2; e += 256) {
        const int comp = e & 1, n = (e >> 1) % NPO, cell = (e >> 1) / NPO;
        const int di = (cell & 7) - Q, dj = (cell >> 3) - Q;
        const int vtx = owned_to_vertex(NV, n);
        L.dl[e] = s.distortions[(((int64_t)(di + cx) * s.ny + (dj + cy)) * NVV + vtx) * 2 + comp];
    }
}

// Silicon::updatePixelDistortions for the 16x16 owner cells of tile (tx0, ty0) of one slot (qdist 3).
// dl_loaded: the displacement table is already in L.dl (persistent kernels load it once).  Returns, per thread,
// whether its cell moved; `tile_moved` is set when any cell of the tile did.
// DlPtr: the type of the global table's pointer -- a kernel's own restrict-qualified argument, or (joint launches, where the
// pointer comes out of an argument block and carries no aliasing information) a pointer into the constant address space, whose
// loads at uniform addresses are scalar loads by definition.
typedef const double __attribute__((address_space(4))) * ConstTable;

template <int NV, bool GLOBAL_DL = false, class DlPtr = const double*>
__device__ __forceinline__ void update_tile_q3(const ims_sensor_t& s, const SlotView& sl, int tx0, int ty0,
                                               unsigned char* __restrict__ changed, UpdateLds<NV>& L, bool dl_loaded,
                                               unsigned int tag, DlPtr __restrict__ dl_global = nullptr)
{
    constexpr int Q = 3, HW = UT + 2 * Q + 1, NPO = 2 * NV + 2;
    const int sx0 = tx0 - (Q + 1), sy0 = ty0 - (Q + 1);
    PROBE(9);
    if (threadIdx.x < HW) L.occ[threadIdx.x] = 0u;
    if (threadIdx.x == 0) L.any_charge = 0;
    // the halo's 529 cells, up to three per thread, requested TOGETHER: as a rolled loop (load, wait, scale, store, next) they were three
    // dependent memory round trips at the head of every tile -- behind a kernel boundary ~1.5 us each under load (round 6)
    constexpr int HE = (HW * HW + 255) / 256;
    double charge[HE];
    const double num_elec = s.num_elec;
#pragma unroll
    for (int u = 0; u < HE; ++u) {
        const int e = (int)threadIdx.x + 256 * u;
        const int hx = e % HW, hy = e / HW;
        const int si = sx0 + hx, sj = sy0 + hy;
        const bool in = e < HW * HW && si >= 0 && si < sl.nx && sj >= 0 && sj < sl.ny;
        charge[u] = in ? (double)s.bf_delta[cell_index(sl, si, sj)] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < HE; ++u) {
        const int e = (int)threadIdx.x + 256 * u;
        if (e < HW * HW) {
            double w = 0.0;
            if (charge[u] != 0.0) { w = ddiv(charge[u], num_elec); atomicOr(&L.occ[e / HW], 1u << (e % HW)); L.any_charge = 1; }
            L.wt[e] = w;
        }
    }
    __syncthreads();
    PROBE(10);
    const int lx = threadIdx.x % UT, ly = threadIdx.x / UT;
    const int i = tx0 + lx, j = ty0 + ly;
    if (!L.any_charge) {                     // nothing landed near this tile: nothing moves
        if (i <= sl.nx && j <= sl.ny) changed[cell_index(sl, i, j)] = 0;
        return;
    }
    if (!GLOBAL_DL && !dl_loaded) {
        load_displacements<NV>(s, L);
        __syncthreads();
    }
    PROBE(11);
    if (i > sl.nx || j > sl.ny) return;
    // 64-bit window: byte a <-> dj = -Q + a (row hy = ly + 2Q + 1 - a); inside a byte bit bb <-> di = -Q + bb
    // (column hx = lx + 2Q + 1 - bb), i.e. the row bitmap reversed.
    unsigned long long mask = 0ull;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        const unsigned int row = (L.occ[ly + 2 * Q + 1 - a] >> lx) & 0xFFu;
        const unsigned int rev = __brev(row) >> 24;
        mask |= (unsigned long long)rev << (8 * a);
    }
    const int64_t c = cell_index(sl, i, j);
    changed[c] = mask ? 1 : 0;
    if (!mask) return;
    if (tag != 0u && s.bf_tile_changed != nullptr) s.bf_tile_changed[cell_index(sl, tx0, ty0)] = (unsigned char)tag;
    double* pts = s.bf_boundary + c * NPO * 2;
    double acc[NPO * 2];
#pragma unroll
    for (int n = 0; n < NPO * 2; ++n) acc[n] = pts[n];
    PROBE(12);
    // The 8 x 8 source window with STATIC offsets, in the spec's order (dj ascending, then di ascending).  A bright
    // star's region is a few hundred tiles, so a SIMD holds ONE wavefront and nothing hides latency: the bit-walking loop
    // (address of the displacement row from the next set bit) waited ~850 cycles per neighbour.  Here every row of the
    // window is straight-line code: the displacement rows sit at compile-time addresses (SGPRs through the scalar cache
    // with the global table, LDS broadcasts otherwise), the scaled charge w of a neighbour without charge is exactly 0
    // and fma(d, +0, acc) leaves acc unchanged bit for bit (acc is never -0: the initial points are +0 / positive plus a
    // tree-ring shift, and an exact cancellation rounds to +0), so no lane needs a branch; a row of the window that no
    // lane of the wavefront needs is skipped as a whole.
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        const unsigned int rowbits = (unsigned int)(mask >> (8 * a)) & 0xFFu;
        if (__builtin_amdgcn_ballot_w64(rowbits != 0u) == 0ull) continue;
#pragma unroll
        for (int bb = 0; bb < 8; ++bb) {
            const double w = L.wt[(ly + 2 * Q + 1 - a) * HW + (lx + 2 * Q + 1 - bb)];
            // GLOBAL_DL: the row comes through the scalar cache into SGPRs (uniform, read-only table of the launch)
            auto row = [&](auto d) {
                if (bb != 7) {                     // not the extra column: bottom-row points
#pragma unroll
                    for (int n = 0; n <= NV + 1; ++n) {
                        acc[2 * n] = fma(d[2 * n], w, acc[2 * n]);
                        acc[2 * n + 1] = fma(d[2 * n + 1], w, acc[2 * n + 1]);
                    }
                }
                if (a != 7) {                      // not the extra row: left-edge points
#pragma unroll
                    for (int n = NV + 2; n < NPO; ++n) {
                        acc[2 * n] = fma(d[2 * n], w, acc[2 * n]);
                        acc[2 * n + 1] = fma(d[2 * n + 1], w, acc[2 * n + 1]);
                    }
                }
            };
            if constexpr (GLOBAL_DL) row(dl_global + (a * 8 + bb) * NPO * 2);
            else row((const double*)L.dl + (a * 8 + bb) * NPO * 2);
        }
    }
    PROBE(13);
#pragma unroll
    for (int n = 0; n < NPO * 2; ++n) pts[n] = acc[n];
    PROBE(14);
}

// acc = fma(d of lane SEL of this lane's row of 16, w, acc): the DPP form of v_fmac_f64 (gfx90a and later accept
// row_newbcast on the 64-bit ALU).  The operand that is uniform over the wavefront -- an entry of the displacement table --
// sits in ONE lane of every row of a VGPR pair and is broadcast by the instruction itself: no scalar load, no LDS read and
// no wait between the FMAs.  All 64 lanes must be active (the source lane is read through EXEC).
__device__ __forceinline__ void fmac_bcast(int sel, double& acc, double d, double w)
{
#define IMS_FB(n) case n: asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #n " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(d), "v"(w)); break;
    switch (sel) {
        IMS_FB(0) IMS_FB(1) IMS_FB(2) IMS_FB(3) IMS_FB(4) IMS_FB(5) IMS_FB(6) IMS_FB(7)
        IMS_FB(8) IMS_FB(9) IMS_FB(10) IMS_FB(11) IMS_FB(12) IMS_FB(13) IMS_FB(14) IMS_FB(15)
    }
#undef IMS_FB
}

// update_tile_q3 with the displacement table delivered by DPP broadcasts.  In the form above a tap costs two scalar
// loads, a wait for them and twenty FMAs on SGPR operands: the 102 SGPRs hold two taps at most, so nothing is fetched ahead
// and a cell's 64 taps are 128 scalar-cache round trips -- 10.8 of the kernel's 14 us for a star's core tile.  Here the
// table is staged in LDS once per workgroup (beside the charge halo, one round trip for both), a row of the window (8 taps
// x 2 NPO values) is read as NPO VGPR pairs, lane l holding entry 16 r + (l & 15), and every FMA names the lane it wants.
// Same operands, same order, same bits.  Lanes without a cell or without charge in their window run along (their w are
// +0, which leaves acc unchanged bit for bit) because a broadcast reads its source lane through EXEC.
// early_table: the table's loads are issued before the halo's (one round trip for both: a star's few tiles, where the latency of
// a round counts); otherwise only the tiles that found charge fetch it (a wide launch is mostly tiles without: 10 KB each).
template <int NV>
__device__ __forceinline__ void update_tile_q3_dpp(const ims_sensor_t& s, const SlotView& sl, int tx0, int ty0,
                                                   unsigned char* __restrict__ changed, UpdateLds<NV>& L, unsigned int tag,
                                                   const double* __restrict__ dl_global, bool early_table)
{
    constexpr int Q = 3, HW = UT + 2 * Q + 1, NPO = 2 * NV + 2, NPT = 2 * NPO;
    const int sx0 = tx0 - (Q + 1), sy0 = ty0 - (Q + 1);
    if (threadIdx.x < HW) L.occ[threadIdx.x] = 0u;
    if (threadIdx.x == 0) L.any_charge = 0;
    // the table first: its loads fly while the halo is gathered
    constexpr int DLN = 8 * 8 * NPT, DLP = (DLN + 255) / 256;
    double dreg[DLP];
    if (early_table) {
#pragma unroll
        for (int u = 0; u < DLP; ++u) {
            const int e = threadIdx.x + 256 * u;
            dreg[u] = (e < DLN) ? dl_global[e] : 0.0;
        }
    }
    constexpr int HE = (HW * HW + 255) / 256;
    double charge[HE];                               // (requested together, as in update_tile_q3)
    const double num_elec = s.num_elec;
#pragma unroll
    for (int u = 0; u < HE; ++u) {
        const int e = (int)threadIdx.x + 256 * u;
        const int hx = e % HW, hy = e / HW;
        const int si = sx0 + hx, sj = sy0 + hy;
        const bool in = e < HW * HW && si >= 0 && si < sl.nx && sj >= 0 && sj < sl.ny;
        charge[u] = in ? (double)s.bf_delta[cell_index(sl, si, sj)] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < HE; ++u) {
        const int e = (int)threadIdx.x + 256 * u;
        if (e < HW * HW) {
            double w = 0.0;
            if (charge[u] != 0.0) { w = ddiv(charge[u], num_elec); atomicOr(&L.occ[e / HW], 1u << (e % HW)); L.any_charge = 1; }
            L.wt[e] = w;
        }
    }
    if (early_table) {
#pragma unroll
        for (int u = 0; u < DLP; ++u) {
            const int e = threadIdx.x + 256 * u;
            if (e < DLN) L.dl[e] = dreg[u];
        }
    }
    __syncthreads();
    const int lx = threadIdx.x % UT, ly = threadIdx.x / UT;
    const int i = tx0 + lx, j = ty0 + ly;
    const bool cell = (i <= sl.nx && j <= sl.ny);
    if (!L.any_charge) {                     // nothing landed near this tile: nothing moves
        if (cell) changed[cell_index(sl, i, j)] = 0;
        return;
    }
    if (!early_table) {
        for (int e = threadIdx.x; e < DLN; e += 256) L.dl[e] = dl_global[e];
        __syncthreads();
    }
    unsigned long long mask = 0ull;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        const unsigned int row = (L.occ[ly + 2 * Q + 1 - a] >> lx) & 0xFFu;
        const unsigned int rev = __brev(row) >> 24;
        mask |= (unsigned long long)rev << (8 * a);
    }
    if (!cell) mask = 0ull;
    const int64_t c = cell ? cell_index(sl, i, j) : sl.offset;
    if (cell) changed[c] = mask ? 1 : 0;
    if (mask && tag != 0u && s.bf_tile_changed != nullptr) s.bf_tile_changed[cell_index(sl, tx0, ty0)] = (unsigned char)tag;
    double* pts = s.bf_boundary + c * NPT;
    double acc[NPT];
#pragma unroll
    for (int n = 0; n < NPT; ++n) acc[n] = mask ? pts[n] : 0.0;
    const int l16 = threadIdx.x & 15;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        const unsigned int rowbits = (unsigned int)(mask >> (8 * a)) & 0xFFu;
        if (__builtin_amdgcn_ballot_w64(rowbits != 0u) == 0ull) continue;
        double d[NPO], w[8];
#pragma unroll
        for (int r = 0; r < NPO; ++r) d[r] = L.dl[a * 8 * NPT + r * 16 + l16];
#pragma unroll
        for (int bb = 0; bb < 8; ++bb) {
            const double ww = L.wt[(ly + 2 * Q + 1 - a) * HW + (lx + 2 * Q + 1 - bb)];
            w[bb] = mask ? ww : 0.0;
        }
#pragma unroll
        for (int bb = 0; bb < 8; ++bb) {
            if (__builtin_amdgcn_ballot_w64(w[bb] != 0.0) == 0ull) continue;       // a tap no lane of the wavefront has charge under
#pragma unroll
            for (int n2 = 0; n2 < NPT; ++n2) {
                const bool bottom = n2 < 2 * (NV + 2);
                if (bottom ? (bb == 7) : (a == 7)) continue;                    // the extra column / row of the window
                const int v = bb * NPT + n2;
                fmac_bcast(v & 15, acc[n2], d[v >> 4], w[bb]);
            }
        }
    }
    if (mask) {
#pragma unroll
        for (int n = 0; n < NPT; ++n) pts[n] = acc[n];
    }
}

// Fast path for qdist == 3 (the GalSim default): the 8x8 source window of a cell is a 64-bit
// occupancy mask cut out of per-row LDS bitmaps, so a lane only iterates over its OWN charged
// neighbours (in the spec's order: dj ascending, then di ascending) instead of testing all 64.
// Sparse stamp wings cost max-over-lanes(nnz) iterations per wave; the dense core stays dense.
// Boundary points are accumulated in registers (NV is a template parameter) and the displacement
// table and the scaled charges w = delta / num_elec live in LDS.
template <int NV, bool DPP = false>
__global__ __launch_bounds__(256) void k_update_distortions_q3(const ims_sensor_t* __restrict__ sp, int first_slot, int n_slots,
                                                               const int64_t* __restrict__ tile_prefix,
                                                               unsigned char* __restrict__ changed, unsigned int tag,
                                                               const double* __restrict__ dl_global)
{
    __shared__ UpdateLds<NV> L;
    PROBE(8);
    const ims_sensor_t& s = *sp;
    const int64_t b = blockIdx.x;
    const int lo = find_slot(tile_prefix, n_slots, b);
    const ims_bf_slot_t bs = s.bf_slots[first_slot + lo];
    const SlotView sl = { bs.xmin, bs.ymin, bs.nx, bs.ny, bs.offset };
    const int tiles_x = (sl.nx + 1 + UT - 1) / UT;
    const int t = (int)(b - tile_prefix[lo]);
    const int tx0 = (t % tiles_x) * UT, ty0 = (t / tiles_x) * UT;
    if (tile_out_of_reach(s, sl, tx0 / UT, ty0 / UT, tag)) return;
#ifdef IMS_UPD_FORCE_LDS
    dl_global = nullptr;
#endif
    if (DPP) update_tile_q3_dpp<NV>(s, sl, tx0, ty0, changed, L, tag, dl_global, gridDim.x <= 64u);
    else if (dl_global != nullptr) update_tile_q3<NV, true>(s, sl, tx0, ty0, changed, L, false, tag, dl_global);
    else update_tile_q3<NV, false>(s, sl, tx0, ty0, changed, L, false, tag);
}

// the update of several CCDs' regions in one launch (k_accumulate_round_j): per chain the sensor, its slots, tile prefix and flags
struct JointUpd {                           // constant over the rounds: device memory
    const ims_sensor_t* sp[IMS_JOINT_MAX];
    const int64_t* tile_prefix[IMS_JOINT_MAX];
    unsigned char* changed[IMS_JOINT_MAX];
    const double* dl[IMS_JOINT_MAX];
    int32_t first_slot[IMS_JOINT_MAX];
};

struct JointTables { JointAcc acc; JointUpd upd; };

// the tables of one joint run go to device memory as the by-value arguments of small launches, eight chains at a time (a kernel's
// arguments are limited to 4 KB): stream-ordered, no staging
constexpr int JOINT_CHUNK = 8;
struct JointChunk {
    RoundArgs a[JOINT_CHUNK];
    const int64_t* pool_start[JOINT_CHUNK];
    const ims_sensor_t* sp[JOINT_CHUNK];
    const int64_t* tile_prefix[JOINT_CHUNK];
    unsigned char* changed[JOINT_CHUNK];
    const double* dl[JOINT_CHUNK];
    unsigned int* words[JOINT_CHUNK];
    int32_t first_slot[JOINT_CHUNK];
};

__global__ __launch_bounds__(64) void k_store_joint_tables(const JointChunk C, JointTables* __restrict__ dst, int k0)
{
    const int k = (int)threadIdx.x;
    if (k >= JOINT_CHUNK) return;
    dst->acc.a[k0 + k] = C.a[k];
    dst->acc.pool_start[k0 + k] = C.pool_start[k];
    dst->upd.sp[k0 + k] = C.sp[k];
    dst->upd.tile_prefix[k0 + k] = C.tile_prefix[k];
    dst->upd.changed[k0 + k] = C.changed[k];
    dst->upd.dl[k0 + k] = C.dl[k];
    dst->upd.first_slot[k0 + k] = C.first_slot[k];
    dst->acc.tile_prefix[k0 + k] = C.tile_prefix[k];
    dst->acc.words[k0 + k] = C.words[k];
    dst->acc.first_slot[k0 + k] = C.first_slot[k];
}

// (the body as a function of restrict-qualified pointers: loaded from the argument block they would carry no aliasing
// information, and the SGPR form of the update then needs 122 registers instead of 72)
template <int NV, bool DPP>
__device__ __forceinline__ void update_block_q3(const ims_sensor_t* __restrict__ sp, int first_slot, int n_slots,
                                                const int64_t* __restrict__ tile_prefix, unsigned char* __restrict__ changed,
                                                unsigned int tag, const double* __restrict__ dl_global, int64_t b, bool small,
                                                UpdateLds<NV>& L)
{
    const ims_sensor_t& s = *sp;
    const int lo = find_slot(tile_prefix, n_slots, b);
    const ims_bf_slot_t bs = s.bf_slots[first_slot + lo];
    const SlotView sl = { bs.xmin, bs.ymin, bs.nx, bs.ny, bs.offset };
    const int tiles_x = (sl.nx + 1 + UT - 1) / UT;
    const int t = (int)(b - tile_prefix[lo]);
    const int tx0 = (t % tiles_x) * UT, ty0 = (t / tiles_x) * UT;
    if (tile_out_of_reach(s, sl, tx0 / UT, ty0 / UT, tag)) return;
    if (DPP) update_tile_q3_dpp<NV>(s, sl, tx0, ty0, changed, L, tag, dl_global, small);
    else if (dl_global != nullptr) update_tile_q3<NV, true, ConstTable>(s, sl, tx0, ty0, changed, L, false, tag, (ConstTable)(uintptr_t)dl_global);
    else update_tile_q3<NV, false>(s, sl, tx0, ty0, changed, L, false, tag);
}

template <int NV, bool DPP = false>
__global__ __launch_bounds__(256) void k_update_distortions_q3_j(const JointUpd* __restrict__ U, const JointRound* __restrict__ R,
                                                                 unsigned int tag)
{
    __shared__ UpdateLds<NV> L;
    int bb = (int)blockIdx.x;
    const int c = joint_chain(&R->eu, bb);
    update_block_q3<NV, DPP>(U->sp[c], U->first_slot[c], joint_entry(&R->ns, c), U->tile_prefix[c], U->changed[c], tag, U->dl[c], (int64_t)bb,
                             gridDim.x <= 64u, L);
}


// bounds of the pixels whose polygon moved (own cell, right cell or upper cell changed); one 16x16
// tile of owner cells per workgroup, same grid as the update kernel.  With a tag, tiles that saw
// neither charge nor movement (own, right and upper tile) leave after three byte loads.
// NV > 0: the owned points of the cell and of its right / upper neighbours are fetched with independent loads into
// registers and the vertex loop is unrolled (the generic loop issues one dependent load per vertex: 12 us per wave).
// fold_image: the f64 image of which this slot is the pixel grid (row length sl.nx): the consumed delta charge is added to it --
// Silicon's `target += delta` at a recalculation (deposits of ims_render_params_t.track_static_delta 2 went to the delta image only)
template <int NV>
__device__ __forceinline__ void refresh_tile(const ims_sensor_t& s, const SlotView& sl, int tx, int ty, bool own, bool right,
                                             bool up, bool charged, const unsigned char* __restrict__ changed,
                                             double* __restrict__ fold_image = nullptr)
{
    const int lx = threadIdx.x % UT, ly = threadIdx.x / UT;
    const int i = tx * UT + lx, j = ty * UT + ly;
    if (i > sl.nx || j > sl.ny) return;
    const int64_t c = cell_index(sl, i, j);
    if (charged) {                            // the update has consumed the delta charge
        if (fold_image != nullptr && i < sl.nx && j < sl.ny) {
            const double d = s.bf_delta[c];
            if (d != 0.0) fold_image[(int64_t)j * sl.nx + i] += d;
        }
        s.bf_delta[c] = 0.0;
    }
    if (i >= sl.nx || j >= sl.ny) return;
    // per-cell flags are only meaningful in tiles the update worked on this round
    // (the three bytes are read whether or not their tile's flag asks for them -- (i + 1, j) and (i, j + 1) are owner cells of the
    // region, i < nx and j < ny here -- so that they are ONE round trip, not up to three behind short-circuit tests)
    const unsigned char c_own = changed[c], c_right = changed[cell_index(sl, i + 1, j)], c_up = changed[cell_index(sl, i, j + 1)];
    const bool f_own = own && c_own;
    const bool f_right = ((lx + 1 < UT) ? own : right) && c_right;
    const bool f_up = ((ly + 1 < UT) ? own : up) && c_up;
    if (!(f_own || f_right || f_up)) return;
    const int nV = (NV > 0) ? NV : s.num_vertices, nv = 4 * nV + 4;
    double ixmin = 0.0, ixmax = 1.0, iymin = 0.0, iymax = 1.0;
    double oxmin = 0.0, oxmax = 1.0, oymin = 0.0, oymax = 1.0;
    double v0x = 0.0;
    if (NV > 0) {
        constexpr int NPO = 2 * NV + 2;
        constexpr int NVX = (NV > 0) ? NV : 1;
        double2 ownp[(NV > 0) ? NPO : 1], rgt[NVX], upp[(NV > 0) ? NV + 2 : 1];
        const double2* po = (const double2*)(s.bf_boundary + c * NPO * 2);
        const double2* pr = (const double2*)(s.bf_boundary + cell_index(sl, i + 1, j) * NPO * 2);
        const double2* pu = (const double2*)(s.bf_boundary + cell_index(sl, i, j + 1) * NPO * 2);
#pragma unroll
        for (int q = 0; q < NPO; ++q) ownp[q] = po[q];
#pragma unroll
        for (int m = 0; m < NV; ++m) rgt[m] = pr[NV + 2 + m];
#pragma unroll
        for (int q = 0; q < NV + 2; ++q) upp[q] = pu[q];
#pragma unroll
        for (int k = 0; k < 4 * NV + 4; ++k) {
            double vx, vy;
            if (k <= NV + 1) { vx = ownp[k].x; vy = ownp[k].y; }
            else if (k <= 2 * NV + 1) { vx = rgt[k - NV - 2].x + 1.0; vy = rgt[k - NV - 2].y; }
            else if (k <= 3 * NV + 3) { vx = upp[NV + 1 - (k - 2 * NV - 2)].x; vy = upp[NV + 1 - (k - 2 * NV - 2)].y + 1.0; }
            else { vx = ownp[NV + 2 + (NV - 1 - (k - 3 * NV - 4))].x; vy = ownp[NV + 2 + (NV - 1 - (k - 3 * NV - 4))].y; }
            if (k == 0) v0x = vx;
            if (vx < oxmin) oxmin = vx;
            if (vx > oxmax) oxmax = vx;
            if (vy < oymin) oymin = vy;
            if (vy > oymax) oymax = vy;
            if (k <= NV + 1) { if (vy > iymin) iymin = vy; }
            if (k >= NV + 1 && k <= 2 * NV + 2) { if (vx < ixmax) ixmax = vx; }
            if (k >= 2 * NV + 2 && k <= 3 * NV + 3) { if (vy < iymax) iymax = vy; }
            if (k >= 3 * NV + 3) { if (vx > ixmin) ixmin = vx; }
        }
    } else {
        for (int k = 0; k < nv; ++k) {
            double vx, vy;
            polygon_vertex(s, sl, i, j, k, 1.0, vx, vy);
            if (k == 0) v0x = vx;
            if (vx < oxmin) oxmin = vx;
            if (vx > oxmax) oxmax = vx;
            if (vy < oymin) oymin = vy;
            if (vy > oymax) oymax = vy;
            if (k <= nV + 1) { if (vy > iymin) iymin = vy; }
            if (k >= nV + 1 && k <= 2 * nV + 2) { if (vx < ixmax) ixmax = vx; }
            if (k >= 2 * nV + 2 && k <= 3 * nV + 3) { if (vy < iymax) iymax = vy; }
            if (k >= 3 * nV + 3) { if (vx > ixmin) ixmin = vx; }
        }
    }
    if (v0x > ixmin) ixmin = v0x;
    double* bb = s.bf_bounds + c * 8;
    bb[0] = ixmin; bb[1] = ixmax; bb[2] = iymin; bb[3] = iymax;
    bb[4] = oxmin; bb[5] = oxmax; bb[6] = oymin; bb[7] = oymax;
}

template <int NV>
__global__ __launch_bounds__(256) void k_refresh_changed(const ims_sensor_t* __restrict__ sp, int first_slot, int n_slots,
                                                         const int64_t* __restrict__ tile_prefix,
                                                         const unsigned char* __restrict__ changed, unsigned int tag,
                                                         double* __restrict__ fold_image)
{
    PROBE(16);
    const ims_sensor_t& s = *sp;
    const int64_t b = blockIdx.x;
    const int lo = find_slot(tile_prefix, n_slots, b);
    const ims_bf_slot_t bs = s.bf_slots[first_slot + lo];
    const SlotView sl = { bs.xmin, bs.ymin, bs.nx, bs.ny, bs.offset };
    const int tiles_x = (sl.nx + 1 + UT - 1) / UT, tiles_y = (sl.ny + 1 + UT - 1) / UT;
    const int t = (int)(b - tile_prefix[lo]);
    const int tx = t % tiles_x, ty = t / tiles_x;
    const bool flags = (tag != 0u) && s.bf_tile_charge != nullptr && s.bf_tile_changed != nullptr;
    bool own = true, right = true, up = true, charged = true;
    if (flags) {
        const unsigned char tg = (unsigned char)tag;
        // (the four bytes in ONE round trip: the right / upper tile's at the tile's own address where there is none)
        const bool has_r = tx + 1 < tiles_x, has_u = ty + 1 < tiles_y;
        const unsigned char t_own = s.bf_tile_changed[cell_index(sl, tx * UT, ty * UT)],
                            t_right = s.bf_tile_changed[cell_index(sl, (has_r ? tx + 1 : tx) * UT, ty * UT)],
                            t_up = s.bf_tile_changed[cell_index(sl, tx * UT, (has_u ? ty + 1 : ty) * UT)],
                            t_charge = s.bf_tile_charge[cell_index(sl, tx * UT, ty * UT)];
        own = t_own == tg; right = has_r && t_right == tg; up = has_u && t_up == tg; charged = t_charge == tg;
        if (!(own || right || up || charged)) return;
    }
    PROBE(17);
    refresh_tile<NV>(s, sl, tx, ty, own, right, up, charged, changed, fold_image);
    PROBE(18);
}

template <int NV>
__global__ __launch_bounds__(256) void k_refresh_changed_j(const JointUpd* __restrict__ U, const JointRound* __restrict__ R,
                                                           unsigned int tag)
{
    int bb = (int)blockIdx.x;
    const int c = joint_chain(&R->eu, bb);
    const ims_sensor_t& s = *U->sp[c];
    const int64_t* __restrict__ tile_prefix = U->tile_prefix[c];
    const unsigned char* changed = U->changed[c];
    const int64_t b = bb;
    const int lo = find_slot(tile_prefix, joint_entry(&R->ns, c), b);
    const ims_bf_slot_t bs = s.bf_slots[U->first_slot[c] + lo];
    const SlotView sl = { bs.xmin, bs.ymin, bs.nx, bs.ny, bs.offset };
    const int tiles_x = (sl.nx + 1 + UT - 1) / UT, tiles_y = (sl.ny + 1 + UT - 1) / UT;
    const int t = (int)(b - tile_prefix[lo]);
    const int tx = t % tiles_x, ty = t / tiles_x;
    const bool flags = (tag != 0u) && s.bf_tile_charge != nullptr && s.bf_tile_changed != nullptr;
    bool own = true, right = true, up = true, charged = true;
    if (flags) {
        const unsigned char tg = (unsigned char)tag;
        // (the four bytes in ONE round trip: the right / upper tile's at the tile's own address where there is none)
        const bool has_r = tx + 1 < tiles_x, has_u = ty + 1 < tiles_y;
        const unsigned char t_own = s.bf_tile_changed[cell_index(sl, tx * UT, ty * UT)],
                            t_right = s.bf_tile_changed[cell_index(sl, (has_r ? tx + 1 : tx) * UT, ty * UT)],
                            t_up = s.bf_tile_changed[cell_index(sl, tx * UT, (has_u ? ty + 1 : ty) * UT)],
                            t_charge = s.bf_tile_charge[cell_index(sl, tx * UT, ty * UT)];
        own = t_own == tg; right = has_r && t_right == tg; up = has_u && t_up == tg; charged = t_charge == tg;
        if (!(own || right || up || charged)) return;
    }
    refresh_tile<NV>(s, sl, tx, ty, own, right, up, charged, changed);
}

// ---- joint rounds over the ACTIVE tiles only ----
// A round's update and refresh launches held one workgroup per 16 x 16 tile of every region of every chain -- 24 000 workgroups
// for the brightest stars of eight CCDs (678 x 678-pixel stamps: 1 849 tiles each), of which a few thousand find charge within
// reach; the rest leave at once, but their dispatch is what the launch costs (4 ns per workgroup alone, twice that beside the
// photon kernels: 50 - 90 us of a 110-us round).  Here one thread per tile reads the charge marks the pixel search left
// (ims_sensor_t.bf_tile_charge, this round's tag) and appends the tiles within reach to a list -- and those whose bounds can
// change (the tile itself, its right or its upper neighbour within reach) to a second one; the update and refresh launches
// then hold a fraction of the workgroups and walk the lists.  Same decisions as the tagged kernels, tile by tile: same bits.
struct JointLists {
    unsigned long long* upd;            // entries: chain << 58 | slot of the class << 32 | tile of the slot
    unsigned long long* ref;
    int* count;                         // [parity][2]: entries of upd / ref
};

// fine: a tile that passes the test on the tile marks is looked at block by block (mark_tile_charge: 4 x 4 pixels).  The update
// of tile [tx0, tx0 + 15]^2 reads the delta charge of [tx0 - 4, tx0 + 18]^2 (update_tile_q3: Q = 3) and changes nothing without
// charge there; those are the blocks -1 .. 4 of the tile in x and y.  The bounds of the tile's pixels also hang on the first
// column of the right tile and the first row of the upper one, which move with charge in [tx0 + 12, tx0 + 19] x [ty0 - 4, ty0 + 18]
// (and transposed): blocks 3 .. 4, inside the same window.  So ONE test decides both lists, and a lone electron in the wings of
// a star lists the one or two tiles within four pixels of it instead of nine.  Tiles that are listed do what they did; tiles
// that are no longer listed would have found no charge in their halo and left (update_tile_q3: !any_charge) -- their `changed`
// bytes stay as they are, which can make the refresh of a LATER round recompute bounds that did not move (same bits).
__global__ __launch_bounds__(256) void k_build_active_j(const JointUpd* __restrict__ U, const JointRound* __restrict__ R,
                                                        unsigned int tag, const JointLists Ls, int parity, int fine)
{
    int bb = (int)blockIdx.x;
    const int c = joint_chain(&R->ewg, bb);
    const ims_sensor_t& s = *U->sp[c];
    const int64_t* __restrict__ tile_prefix = U->tile_prefix[c];
    const int n_tiles = joint_entry(&R->tof, c), ns = joint_entry(&R->ns, c);
    if (blockIdx.x == 0 && threadIdx.x == 0) { Ls.count[2 * (parity ^ 1)] = 0; Ls.count[2 * (parity ^ 1) + 1] = 0; }   // the next round's
    const int lane = threadIdx.x & 63;
    const int64_t g = (int64_t)bb * 256 + threadIdx.x;
    bool todo = g < n_tiles;
    int my_lo = 0, my_t = 0;
    // slot of every lane's tile: the wavefront resolves one slot per turn (find_slot wants all 64 lanes)
    unsigned long long pending = __builtin_amdgcn_ballot_w64(todo);
    while (pending != 0ull) {
        const int src = __builtin_ctzll(pending);
        const int64_t gf = __shfl(g, src, 64);
        const int lo = find_slot(tile_prefix, ns, gf);
        const int64_t p0 = tile_prefix[lo], p1 = tile_prefix[lo + 1];
        if (todo && g >= p0 && g < p1) { my_lo = lo; my_t = (int)(g - p0); todo = false; }
        pending = __builtin_amdgcn_ballot_w64(todo);
    }
    bool in_reach = false, bounds = false;
    if (g < n_tiles) {
        const ims_bf_slot_t bs = s.bf_slots[U->first_slot[c] + my_lo];
        const SlotView sl = { bs.xmin, bs.ymin, bs.nx, bs.ny, bs.offset };
        const int tiles_x = (sl.nx + 1 + UT - 1) / UT, tiles_y = (sl.ny + 1 + UT - 1) / UT;
        const int tx = my_t % tiles_x, ty = my_t / tiles_x;
        const unsigned char tg = (unsigned char)tag;
        // charge marks of the 4 x 4 window (-1 .. 2)^2 around the tile, as bits
        unsigned int m = 0u;
#pragma unroll
        for (int dy = -1; dy <= 2; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 2; ++dx) {
                const int ux = tx + dx, uy = ty + dy;
                if (ux < 0 || uy < 0 || ux >= tiles_x || uy >= tiles_y) continue;
                if (s.bf_tile_charge[cell_index(sl, ux * UT, uy * UT)] == tg) m |= 1u << ((dy + 1) * 4 + (dx + 1));
            }
        // 3 x 3 windows around the tile, its right and its upper neighbour (bit = row * 4 + column of the 4 x 4 window)
        const unsigned int own_w = 0x0777u, right_w = 0x0EEEu, up_w = 0x7770u;
        in_reach = (m & own_w) != 0u;
        const bool right_in = tx + 1 < tiles_x && (m & right_w) != 0u;
        const bool up_in = ty + 1 < tiles_y && (m & up_w) != 0u;
        bounds = in_reach || right_in || up_in;
        if (fine && bounds) {
            // charge in the tile itself is in reach; otherwise only the blocks of the marked neighbours that touch this tile are
            // looked at (an edge neighbour's four, a corner neighbour's one): 1 .. 20 bytes instead of the window's 36
            bool any = (m & (1u << 5)) != 0u;
            if (!any) {
#pragma unroll
                for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                    for (int dx = -1; dx <= 1; ++dx) {
                        if ((dx == 0 && dy == 0) || !(m & (1u << ((dy + 1) * 4 + (dx + 1))))) continue;
                        const int bx0 = dx < 0 ? -1 : (dx > 0 ? 4 : 0), bx1 = dx < 0 ? -1 : (dx > 0 ? 4 : 3);
                        const int by0 = dy < 0 ? -1 : (dy > 0 ? 4 : 0), by1 = dy < 0 ? -1 : (dy > 0 ? 4 : 3);
                        for (int by = by0; by <= by1; ++by) {
                            const int cy = ty * UT + by * 4 + 1;
                            if (cy < 0 || cy > sl.ny) continue;
                            for (int bx = bx0; bx <= bx1; ++bx) {
                                const int cx = tx * UT + bx * 4 + 1;
                                if (cx < 0 || cx > sl.nx) continue;
                                any = any || (s.bf_tile_charge[cell_index(sl, cx, cy)] == tg);
                            }
                        }
                    }
            }
            in_reach = any; bounds = any;
        }
    }
    const unsigned long long entry = ((unsigned long long)c << 58) | ((unsigned long long)(unsigned int)my_lo << 32) | (unsigned int)my_t;
    {
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(in_reach);
        if (mask != 0ull) {
            int base = 0;
            if (lane == 0) base = atomicAdd(Ls.count + 2 * parity, (int)__popcll(mask));
            base = __shfl(base, 0, 64);
            if (in_reach) Ls.upd[base + (int)__popcll(mask & ((1ull << lane) - 1ull))] = entry;
        }
    }
    {
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(bounds);
        if (mask != 0ull) {
            int base = 0;
            if (lane == 0) base = atomicAdd(Ls.count + 2 * parity + 1, (int)__popcll(mask));
            base = __shfl(base, 0, 64);
            if (bounds) Ls.ref[base + (int)__popcll(mask & ((1ull << lane) - 1ull))] = entry;
        }
    }
}

// one listed tile
template <int NV>
__device__ __forceinline__ void update_listed_tile(const JointUpd* __restrict__ U, unsigned long long e, unsigned int tag, UpdateLds<NV>& L)
{
    const int c = (int)(e >> 58), lo = (int)((e >> 32) & 0x3FFFFFFu), t = (int)(e & 0xFFFFFFFFu);
    const ims_sensor_t& s = *U->sp[c];
    const ims_bf_slot_t bs = s.bf_slots[U->first_slot[c] + lo];
    const SlotView sl = { bs.xmin, bs.ymin, bs.nx, bs.ny, bs.offset };
    const int tiles_x = (sl.nx + 1 + UT - 1) / UT;
    const int tx0 = (t % tiles_x) * UT, ty0 = (t / tiles_x) * UT;
    const double* dl_global = U->dl[c];
#ifdef IMS_HIST
    const unsigned long long hist_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    if (dl_global != nullptr) update_tile_q3<NV, true, ConstTable>(s, sl, tx0, ty0, U->changed[c], L, false, tag, (ConstTable)(uintptr_t)dl_global);
    else update_tile_q3<NV, false>(s, sl, tx0, ty0, U->changed[c], L, false, tag);
#ifdef IMS_HIST
    __syncthreads();
    if (threadIdx.x == 0) {
        int halo = 0, own = 0;
        for (int hy = 0; hy < UpdateLds<NV>::HW; ++hy) {
            halo += __popc(L.occ[hy]);
            if (hy >= 4 && hy < 4 + UT) own += __popc((L.occ[hy] >> 4) & 0xFFFFu);
        }
        const int bin = halo < 63 ? halo : 63;
        atomicAdd(&g_hist[0][bin], 1ull);
        atomicAdd(&g_hist[1][bin], __builtin_amdgcn_s_memrealtime() - hist_t0);
        atomicAdd(&g_hist[2][bin], (unsigned long long)own);
    }
#endif
}

template <int NV, bool DPP = false>
__global__ __launch_bounds__(256, 5) void k_update_list_j(const JointUpd* __restrict__ U, const JointLists Ls, int parity, unsigned int tag,
                                                       int reset_next)
{
    __shared__ UpdateLds<NV> L0;
    const int n = Ls.count[2 * parity];
    // (search-side lists: no builder launch to empty the next round's list -- this launch sits between this round's search and the next's)
    if (reset_next && blockIdx.x == 0 && threadIdx.x == 0) { Ls.count[2 * (parity ^ 1)] = 0; Ls.count[2 * (parity ^ 1) + 1] = 0; }
    for (int i = (int)blockIdx.x; i < n; i += (int)gridDim.x) {
        if (i != (int)blockIdx.x) __syncthreads();                  // the tile before is through with the shared buffers
        // (the address of the shared buffers through a register the compiler cannot look through: inside a loop it otherwise
        // keeps every one of the ~180 constant LDS addresses of the unrolled window in a register of its own -- 256 against 80)
        UpdateLds<NV>* Lp = &L0;
        asm volatile("" : "+v"(Lp));
        update_listed_tile<NV>(U, Ls.upd[i], tag, *Lp);
    }
}

template <int NV>
__global__ __launch_bounds__(256) void k_refresh_list_j(const JointUpd* __restrict__ U, const JointLists Ls, int parity, unsigned int tag,
                                                        int from_upd)
{
    // from_upd: ONE list for both (search-side lists: a tile is listed when charge lies within [t0 - 4, t0 + 19] of it, which covers
    // the update's halo and the right / upper neighbour's first column / row)
    const int n = Ls.count[2 * parity + (from_upd ? 0 : 1)];
    const unsigned long long* __restrict__ list = from_upd ? Ls.upd : Ls.ref;
    for (int i = (int)blockIdx.x; i < n; i += (int)gridDim.x) {
        const unsigned long long e = list[i];
        const int c = (int)(e >> 58), lo = (int)((e >> 32) & 0x3FFFFFFu), t = (int)(e & 0xFFFFFFFFu);
        const ims_sensor_t& s = *U->sp[c];
        const ims_bf_slot_t bs = s.bf_slots[U->first_slot[c] + lo];
        const SlotView sl = { bs.xmin, bs.ymin, bs.nx, bs.ny, bs.offset };
        const int tiles_x = (sl.nx + 1 + UT - 1) / UT, tiles_y = (sl.ny + 1 + UT - 1) / UT;
        const int tx = t % tiles_x, ty = t / tiles_x;
        const unsigned char tg = (unsigned char)tag;
        // (the four bytes in ONE round trip: the right / upper tile's at the tile's own address where there is none)
        const bool has_r = tx + 1 < tiles_x, has_u = ty + 1 < tiles_y;
        const unsigned char t_own = s.bf_tile_changed[cell_index(sl, tx * UT, ty * UT)],
                            t_right = s.bf_tile_changed[cell_index(sl, (has_r ? tx + 1 : tx) * UT, ty * UT)],
                            t_up = s.bf_tile_changed[cell_index(sl, tx * UT, (has_u ? ty + 1 : ty) * UT)],
                            t_charge = s.bf_tile_charge[cell_index(sl, tx * UT, ty * UT)];
        const bool own = t_own == tg, right = has_r && t_right == tg, up = has_u && t_up == tg, charged = t_charge == tg;
        if (!(own || right || up || charged)) continue;
        refresh_tile<NV>(s, sl, tx, ty, own, right, up, charged, (const unsigned char*)U->changed[c]);
    }
}

// ---------------- LSST_Flat ----------------
constexpr long long FLAT_ID_BASE = 0x7F00000000ll;      // object id space of the flat builder's Poisson streams

// Silicon::fillWithPixelAreas: shoelace area of the (distorted) polygon of every pixel of one slot
__global__ __launch_bounds__(256) void k_pixel_areas(const ims_sensor_t* __restrict__ sp, int slot, double* __restrict__ area,
                                                     long long* __restrict__ sum_q32)
{
    const ims_sensor_t& s = *sp;
    const ims_bf_slot_t bs = s.bf_slots[slot];
    const SlotView sl = { bs.xmin, bs.ymin, bs.nx, bs.ny, bs.offset };
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (int64_t)sl.nx * sl.ny) return;
    const int i = (int)(p % sl.nx), j = (int)(p / sl.nx);
    const int nv = 4 * s.num_vertices + 4;
    double x0, y0, xp, yp, a2 = 0.0;
    polygon_vertex(s, sl, i, j, 0, 1.0, x0, y0);
    xp = x0; yp = y0;
    for (int k = 1; k <= nv; ++k) {
        double xk = x0, yk = y0;
        if (k < nv) polygon_vertex(s, sl, i, j, k, 1.0, xk, yk);
        a2 = a2 + (xp * yk - xk * yp);
        xp = xk; yp = yk;
    }
    const double a = 0.5 * a2;
    area[p] = a;
    atomicAdd((unsigned long long*)sum_q32, (unsigned long long)(long long)floor(a * 0x1.0p32 + 0.5));
}

__global__ __launch_bounds__(256) void k_flat_add(const double* __restrict__ area, const double* __restrict__ base, double level,
                                                  double inv_mean_area, uint64_t seed, int64_t iteration, int nx, int ny,
                                                  double* __restrict__ image, double* __restrict__ delta)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (int64_t)nx * ny) return;
    double mean = level;
    if (base != nullptr) mean = mean * base[p];
    if (area != nullptr) mean = mean * (area[p] * inv_mean_area);
    const double v = poisson(mean, seed, FLAT_ID_BASE + iteration, p);
    image[p] = image[p] + v;
    if (delta != nullptr) {
        const int i = (int)(p % nx), j = (int)(p / nx);
        delta[(int64_t)j * (nx + 1) + i] = delta[(int64_t)j * (nx + 1) + i] + v;
    }
}

__global__ void k_clear_pristine(ims_sensor_t* sp) { sp->pristine_margin = -1.0; }

__global__ __launch_bounds__(256) void k_zero_delta(const ims_sensor_t* __restrict__ sp, int64_t cell_begin, int64_t cell_count)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= cell_count) return;
    sp->bf_delta[cell_begin + t] = 0.0;
}

__global__ __launch_bounds__(256) void k_image_add(double* __restrict__ dst, const double* __restrict__ src, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] += src[i];
}

__global__ __launch_bounds__(256) void k_image_to_float(const double* __restrict__ src, float* __restrict__ dst, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = (float)src[i];
}


// ---------------- FFT branch ----------------
__device__ __forceinline__ int64_t find_prefix(const int64_t* __restrict__ prefix, int64_t n, int64_t e)
{
    int64_t lo = 0, hi = n;
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (prefix[mid] <= e) lo = mid; else hi = mid;
    }
    return lo;
}

// row and column of element `local` of a grid of row length `len` (FFT sizes are powers of two: a shift; half spectra -- len =
// nfft / 2 + 1 -- and anything else: a 32-bit division, the grids hold fewer than 2^31 elements)
__device__ __forceinline__ void row_col(int64_t local, int len, int& row, int& col)
{
    const uint32_t l = (uint32_t)local, n = (uint32_t)len;
    if ((n & (n - 1u)) == 0u) { const int sh = 31 - __clz((int)n); row = (int)(l >> sh); col = (int)(l & (n - 1u)); }
    else { row = (int)(l / n); col = (int)(l - (uint32_t)row * n); }
}

// The elementwise kernels of the FFT branch walk the back-to-back grids of their objects in SPANS: a workgroup owns `span` consecutive
// elements (a multiple of 256, fft_span below), a wavefront 64 consecutive ones per pass.  The object of a wavefront's elements is
// found ONCE per span (a bisection with scalar operands) and then only moved along when the wavefront's first element passes the
// object's end; a wavefront that lies inside one object -- all of them but the few on a boundary -- calls the body with a UNIFORM object
// number, so that the object's row and its prefix entry come by scalar loads.  (A grid-stride loop used to bisect in every pass: 13
// dependent loads before the first useful one, `k_fft_finish` waited 67 % of its wave-cycles and moved 0.26 TB/s, round 6.)
template <class Body>
__device__ __forceinline__ void walk_span(const int64_t* __restrict__ prefix, int64_t n_objects, int64_t n_elems, int64_t span, Body&& body)
{
    const int64_t wg_begin = (int64_t)blockIdx.x * span;
    int64_t wg_end = wg_begin + span;
    if (wg_end > n_elems) wg_end = n_elems;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)(threadIdx.x & 63u);
    int64_t oi = -1, o_end = 0;
    for (int64_t base = wg_begin + 64 * wave; base < wg_end; base += 256) {
        if (oi < 0) { oi = find_prefix(prefix, n_objects, base); o_end = prefix[oi + 1]; }
        else while (base >= o_end) { ++oi; o_end = prefix[oi + 1]; }
        const int64_t el = base + lane;
        if (base + 64 <= o_end) {
            if (el < wg_end) body(el, oi);
        } else if (el < wg_end) {
            body(el, el < o_end ? oi : find_prefix(prefix, n_objects, el));
        }
    }
}

// 1 / (nfft * nfft) as ims_fft_inverse's scaling pass forms it on the host (powers of two: the exponent written directly)
__device__ __forceinline__ double inv_n2(int nfft)
{
    const uint32_t n = (uint32_t)nfft;
    if ((n & (n - 1u)) == 0u) return __longlong_as_double((long long)(1023 - 2 * (31 - __clz((int)n))) << 52);
    return 1.0 / ((double)nfft * (double)nfft);
}
// a value of the real-space buffer as the image holds it (ims_fft_params_t.rbuf_raw)
__device__ __forceinline__ double rbuf_value(const double* __restrict__ rbuf, int64_t at, bool raw, double scale)
{
    const double v = rbuf[at];
    return raw ? v * scale : v;
}

// LDS tables of the pixel response for a workgroup whose span lies inside ONE object's half spectrum (every workgroup of a grid of
// 1 024 or more): sinc along x for every column of the row, sinc along y (for ky and for -ky) for the few rows the span covers.
constexpr int FILL_NH_MAX = 2049, FILL_ROWS_MAX = 264;      // grids up to 4096 (larger ones: no tables); 20 KB of LDS
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_fft_kspace_fill(const ims_fft_params_t P, const ims_fft_object_t* __restrict__ objs,
                                                         int64_t n_objects, const int64_t* __restrict__ prefix, int64_t n_elems, int64_t span,
                                                         double* __restrict__ kbuf)
{
    __shared__ double sinc_x[FILL_NH_MAX];
    __shared__ double sinc_y[2 * FILL_ROWS_MAX];
    // (uniform over the workgroup: formed from the block's number and scalar loads)
    const int64_t wg_begin = (int64_t)blockIdx.x * span;
    int64_t wg_end = wg_begin + span;
    if (wg_end > n_elems) wg_end = n_elems;
    bool tables = false;
    int row_first = 0;
    if (P.n_alias <= 0 && wg_begin < wg_end) {
        const int64_t o0 = find_prefix(prefix, n_objects, wg_begin);
        const int64_t ob = prefix[o0];
        if (wg_end <= prefix[o0 + 1]) {
            const int n = objs[o0].nfft, nh = n / 2 + 1, half = n / 2;
            row_first = (int)((wg_begin - ob) / nh);
            const int row_last = (int)((wg_end - 1 - ob) / nh);
            if (row_first > half) return;              // rows written by the threads of their mirror rows
            if (nh <= FILL_NH_MAX && row_last - row_first + 1 <= FILL_ROWS_MAX) {
                const double dk = TWO_PI / ((double)n * P.pixel_scale);
                for (int j = threadIdx.x; j < nh; j += 256) sinc_x[j] = pixel_response((double)j * dk, P.pixel_scale);
                for (int r = threadIdx.x; r < 2 * (row_last - row_first + 1); r += 256) {
                    const double ky = (double)(row_first + (r >> 1)) * dk;
                    sinc_y[r] = pixel_response((r & 1) ? -ky : ky, P.pixel_scale);
                }
                tables = true;
                __syncthreads();
            }
        }
    }
    walk_span(prefix, n_objects, n_elems, span, [&](int64_t el, int64_t oi) {
        const ims_fft_object_t& o = objs[oi];
        const int64_t local = el - prefix[oi];
        const int nh = o.nfft / 2 + 1;
        int i, j;
        row_col(local, nh, i, j);
        double re, im;
        if (P.n_alias <= 0) {
            // rows i and n - i share most of their arithmetic (kspace_pair): the thread of row 0 < i < n / 2 writes both, the
            // threads of the rows beyond n / 2 have nothing to do (whole wavefronts of them: a row is nfft / 2 + 1 elements)
            const int n = o.nfft, half = n / 2;
            if (i > half) return;
            if (i > 0 && i < half) {
                const double dk = TWO_PI / ((double)n * P.pixel_scale);
                double re2, im2;
                if (tables) {
                    const double tab[3] = { sinc_x[j], sinc_y[2 * (i - row_first)], sinc_y[2 * (i - row_first) + 1] };
                    kspace_pair(P, o, (double)j * dk, (double)i * dk, re, im, re2, im2, tab);
                } else {
                    kspace_pair(P, o, (double)j * dk, (double)i * dk, re, im, re2, im2);
                }
                const int64_t at2 = o.k_offset + (int64_t)(n - i) * nh + j;
                kbuf[2 * (o.k_offset + local)] = re;
                kbuf[2 * (o.k_offset + local) + 1] = im;
                kbuf[2 * at2] = re2;
                kbuf[2 * at2 + 1] = im2;
                return;
            }
        }
        kspace_value(P, o, i, j, re, im);
        kbuf[2 * (o.k_offset + local)] = re;
        kbuf[2 * (o.k_offset + local) + 1] = im;
    });
}

// saturated bounding box of every object (saturated_region, imsim/diffraction_fft.py:211-227)
__global__ __launch_bounds__(256) void k_fft_bbox(const ims_fft_params_t P, const ims_fft_object_t* __restrict__ objs,
                                                  int64_t n_objects, const int64_t* __restrict__ prefix, int64_t n_pix, int64_t span,
                                                  const double* __restrict__ rbuf, int32_t* __restrict__ bbox)
{
    walk_span(prefix, n_objects, n_pix, span, [&](int64_t el, int64_t oi) {
        const ims_fft_object_t& o = objs[oi];
        const int64_t local = el - prefix[oi];
        int iy, ix;
        row_col(local, o.nfft, iy, ix);
        const int px = o.x0 + ix, py = o.y0 + iy;
        if (px < o.stamp_xmin || px > o.stamp_xmax || py < o.stamp_ymin || py > o.stamp_ymax) return;
        if (rbuf_value(rbuf, o.r_offset + local, P.rbuf_raw != 0, inv_n2(o.nfft)) > P.spikes.threshold) {
            atomicMin(&bbox[4 * oi + 0], iy); atomicMax(&bbox[4 * oi + 1], iy);
            atomicMin(&bbox[4 * oi + 2], ix); atomicMax(&bbox[4 * oi + 3], ix);
        }
    });
}

// ims_spikes_t.tab_*: one thread per row a of the stencil; b from +cutoff down to -cutoff.  row_ptr == NULL: count only.
__global__ __launch_bounds__(64) void k_fft_spike_table(const ims_spikes_t k, const int32_t* __restrict__ row_ptr, int32_t* __restrict__ row_count,
                                                        int32_t* __restrict__ col, double* __restrict__ val)
{
    const int row = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (row > 2 * k.cutoff) return;
    const int a = row - k.cutoff;
    int n = 0;
    const int base = row_ptr ? row_ptr[row] : 0;
    for (int b = k.cutoff; b >= -k.cutoff; --b) {
        const double sv = spike_stencil(k, a, b);
        if (sv == 0.0) continue;
        if (row_ptr) { col[base + n] = b; val[base + n] = sv / k.norm; }
        ++n;
    }
    if (!row_ptr) row_count[row] = n;
}

// convolve_region (imsim/diffraction_fft.py:170-208): clipped image with the box zeroed + box (x) stencil
// Far from the arms of EVERY source pixel's cross the whole spike sum of a pixel is exact zeros: the offsets from the pixel to the
// sources differ from the offset to the box centre by at most half the box, so if even the nearer arm of the centre's cross is
// further away than that (plus the stencil's own margins, ims_fft.h spike_stencil), no term can be non-zero.  97 % of a 4096^2 stamp.
__device__ __forceinline__ bool spike_sum_is_zero(const ims_spikes_t& k, int iy, int ix, int r0, int r1, int c0, int c1)
{
    const double ac = (double)iy - 0.5 * (double)(r0 + r1), bc = (double)ix - 0.5 * (double)(c0 + c1);
    const double ha = 0.5 * (double)(r1 - r0), hb = 0.5 * (double)(c1 - c0);
    const double e = fabs(k.cos0) * ha + fabs(k.sin0) * hb + fabs(k.sin0) * ha + fabs(k.cos0) * hb;
    const double xc = k.cos0 * ac + k.sin0 * bc, yc = -k.sin0 * ac + k.cos0 * bc;
    const double mc = (fabs(xc) < fabs(yc) ? fabs(xc) : fabs(yc)) - e;
    const double rmax = sqrt(ac * ac + bc * bc) + sqrt(ha * ha + hb * hb) + 1.0;
    const double lim = 0.5 * fabs(k.d_alpha) + 1.0e-6;
    return mc > 1.0 + 1.0e-3 && mc - 1.0e-3 > lim * rmax;
}

// the spike sum of pixel (iy, ix) of object o over the saturated box rows r0 .. r1, columns c0 .. c1 (DiffractionFFT.apply's convolution)
__device__ __forceinline__ double spike_sum(const ims_fft_params_t& P, const ims_fft_object_t& o, int iy, int ix, int r0, int r1, int c0,
                                            int c1, const double* __restrict__ rin, bool raw, double scale)
{
    double acc = 0.0;
    if (P.spikes.tab_row != nullptr) {
        // The stencil's non-zero entries come out of the visit's table (ims_spikes_t.tab_*): of source row ry the entries of
        // stencil row a = iy - ry whose column offset b puts the source column ix - b inside the box, in ascending source
        // column -- the terms the loops below find among their candidates, in the same order, each formed by the same
        // operations (stencil / norm in the table's kernel, times the source here), without the three arctangents per term
        // and the ~16 candidates per row: the same sum.  (A star's spike stencil was 1.1 ms of a CCD's front, round 6.)
        // The walk is a chain of dependent loads (row pointer -> column offsets -> source pixels) on a few lanes of a wavefront, so
        // the loads are asked for in batches: the NEXT row's pointer while this row is worked on (rows are consecutive in the
        // table: row a - 1 ends where row a begins), four entries' offsets and values at once, then their four source pixels, then
        // the four terms in the table's order -- the entries with blo <= b <= bhi, a contiguous run of the descending columns, as before.
        const ims_spikes_t& k = P.spikes;
        const int blo = ix - c1, bhi = ix - c0;
        const int ry_lo = r0 > iy - k.cutoff ? r0 : iy - k.cutoff, ry_hi = r1 < iy + k.cutoff ? r1 : iy + k.cutoff;
        if (ry_lo <= ry_hi) {
            int a = iy - ry_lo;
            int e0 = k.tab_row[a + k.cutoff], e1 = k.tab_row[a + k.cutoff + 1];
            for (int ry = ry_lo; ry <= ry_hi; ++ry, --a) {
                const int e_next = ry < ry_hi ? k.tab_row[a - 1 + k.cutoff] : 0;
                int e = e0;
                if (e1 - e > 16) {                         // an arm along this row: the first entry with b <= bhi by bisection
                    int lo = e, hi = e1;
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if (k.tab_col[mid] > bhi) lo = mid + 1; else hi = mid; }
                    e = lo;
                }
                const int64_t row_at = o.r_offset + (int64_t)ry * o.nfft + ix;
                for (; e < e1; e += 4) {
                    int b[4];
                    double tv[4], sv[4];
                    bool use[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int eq = e + q < e1 ? e + q : e1 - 1;
                        b[q] = k.tab_col[eq]; tv[q] = k.tab_val[eq];
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        use[q] = e + q < e1 && b[q] <= bhi && b[q] >= blo;
                        sv[q] = rbuf_value(rin, row_at - (use[q] ? b[q] : bhi), raw, scale);      // (not used: the box's first column)
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (use[q]) { const double src = sv[q] < 0.0 ? 0.0 : sv[q]; acc = acc + tv[q] * src; }
                    if (b[3] < blo || b[0] < blo) break;       // descending columns: nothing further down is inside the box
                }
                e1 = e0; e0 = e_next;
            }
        }
    } else {
        // Of a source row only the columns near the two arms through this pixel can contribute (|xr| or |yr| within the
        // stencil's reach T: one interval of columns each, ims_fft.h spike_stencil); everything else in the row is an exact
        // zero.  The columns are visited in ascending order as before, so the sum is the same sum -- a saturated star's
        // box is 100 x 100 source pixels, of which a target pixel on an arm needs ~16 per row.
        const ims_spikes_t& k = P.spikes;
        const double lim = 0.5 * fabs(k.d_alpha) + 1.0e-6;
        const double bfar = fabs((double)(ix - c0)) > fabs((double)(ix - c1)) ? fabs((double)(ix - c0)) : fabs((double)(ix - c1));
        const bool has_s = fabs(k.sin0) > 1.0e-12, has_c = fabs(k.cos0) > 1.0e-12;
        const double inv_s = has_s ? 1.0 / k.sin0 : 0.0, inv_c = has_c ? 1.0 / k.cos0 : 0.0;
        for (int ry = r0; ry <= r1; ++ry) {
            const int a = iy - ry;
            if (a < -k.cutoff || a > k.cutoff) continue;
            const double da = (double)a;
            const double T = 1.0 + 1.0e-3 + lim * sqrt(da * da + bfar * bfar);
            int lo[2], hi[2];
            // |cos0 a + sin0 b| <= T
            if (has_s) {
                const double b1 = (-T - k.cos0 * da) * inv_s, b2 = (T - k.cos0 * da) * inv_s;
                const double bl = b1 < b2 ? b1 : b2, bh = b1 < b2 ? b2 : b1;
                lo[0] = (int)floor((double)ix - bh) - 1; hi[0] = (int)ceil((double)ix - bl) + 1;
            } else if (fabs(k.cos0 * da) <= T) { lo[0] = c0; hi[0] = c1; }
            else { lo[0] = 1; hi[0] = 0; }
            // |-sin0 a + cos0 b| <= T
            if (has_c) {
                const double b1 = (-T + k.sin0 * da) * inv_c, b2 = (T + k.sin0 * da) * inv_c;
                const double bl = b1 < b2 ? b1 : b2, bh = b1 < b2 ? b2 : b1;
                lo[1] = (int)floor((double)ix - bh) - 1; hi[1] = (int)ceil((double)ix - bl) + 1;
            } else if (fabs(k.sin0 * da) <= T) { lo[1] = c0; hi[1] = c1; }
            else { lo[1] = 1; hi[1] = 0; }
            for (int q = 0; q < 2; ++q) { if (lo[q] < c0) lo[q] = c0; if (hi[q] > c1) hi[q] = c1; }
            if (lo[1] < lo[0]) { const int tl = lo[0], th = hi[0]; lo[0] = lo[1]; hi[0] = hi[1]; lo[1] = tl; hi[1] = th; }
            if (hi[0] >= lo[0] && hi[1] >= lo[1] && lo[1] <= hi[0] + 1) { if (hi[1] > hi[0]) hi[0] = hi[1]; lo[1] = 1; hi[1] = 0; }   // one run
            for (int q = 0; q < 2; ++q)
                for (int rx = lo[q]; rx <= hi[q]; ++rx) {
                    const int b = ix - rx;
                    if (b < -k.cutoff || b > k.cutoff) continue;
                    double src = rbuf_value(rin, o.r_offset + (int64_t)ry * o.nfft + rx, raw, scale);
                    if (src < 0.0) src = 0.0;
                    acc = acc + spike_stencil(k, a, b) / k.norm * src;
                }
        }
    }
    return acc;
}

__global__ __launch_bounds__(256) void k_fft_spikes(const ims_fft_params_t P, const ims_fft_object_t* __restrict__ objs,
                                                    int64_t n_objects, const int64_t* __restrict__ prefix, int64_t n_pix, int64_t span,
                                                    const double* __restrict__ rin, double* __restrict__ rout,
                                                    const int32_t* __restrict__ bbox, const unsigned int* __restrict__ only_if_over,
                                                    unsigned int over_cap)
{
    // (the fallback of the listed form below: runs only when the list ran over)
    if (only_if_over != nullptr && only_if_over[0] <= over_cap) return;
    walk_span(prefix, n_objects, n_pix, span, [&](int64_t el, int64_t oi) {
        const ims_fft_object_t& o = objs[oi];
        const int64_t local = el - prefix[oi];
        int iy, ix;
        row_col(local, o.nfft, iy, ix);
        const bool raw = P.rbuf_raw != 0;
        const double scale = inv_n2(o.nfft);
        double v = rbuf_value(rin, o.r_offset + local, raw, scale);
        if (v < 0.0) v = 0.0;
        const int px = o.x0 + ix, py = o.y0 + iy;
        const bool in_stamp = !(px < o.stamp_xmin || px > o.stamp_xmax || py < o.stamp_ymin || py > o.stamp_ymax);
        const int r0 = bbox[4 * oi + 0], r1 = bbox[4 * oi + 1], c0 = bbox[4 * oi + 2], c1 = bbox[4 * oi + 3];
        if (P.spikes.enabled && in_stamp && r1 >= r0) {
            if (iy >= r0 && iy <= r1 && ix >= c0 && ix <= c1) v = 0.0;
            double acc = 0.0;
            if (!spike_sum_is_zero(P.spikes, iy, ix, r0, r1, c0, c1)) acc = spike_sum(P, o, iy, ix, r0, r1, c0, c1, rin, raw, scale);
            v = v + acc;
        }
        rout[o.r_offset + local] = v;
    });
}

// The spike step in two launches (ims_fft_spikes_listed).  This one streams: clip, the saturated box to zero, the image written --
// and the pixels whose spike sum is NOT known to be zero (3 % of a bright star's stamp: the arms) appended to a list, object << 40 |
// element of the object.  Light (no table walk in it: 40 registers instead of 98), so that its 16 bytes per pixel move at the
// rate of a copy; the arms' sums are then formed by k_fft_spikes_arms with every lane at work, not three in a wavefront.
__global__ __launch_bounds__(256) void k_fft_spikes_stream(const ims_fft_params_t P, const ims_fft_object_t* __restrict__ objs,
                                                           int64_t n_objects, const int64_t* __restrict__ prefix, int64_t n_pix, int64_t span,
                                                           const double* __restrict__ rin, double* __restrict__ rout,
                                                           const int32_t* __restrict__ bbox, unsigned long long* __restrict__ list,
                                                           unsigned int cap, unsigned int* __restrict__ count)
{
    walk_span(prefix, n_objects, n_pix, span, [&](int64_t el, int64_t oi) {
        const ims_fft_object_t& o = objs[oi];
        const int64_t local = el - prefix[oi];
        int iy, ix;
        row_col(local, o.nfft, iy, ix);
        double v = rbuf_value(rin, o.r_offset + local, P.rbuf_raw != 0, inv_n2(o.nfft));
        if (v < 0.0) v = 0.0;
        const int px = o.x0 + ix, py = o.y0 + iy;
        const bool in_stamp = !(px < o.stamp_xmin || px > o.stamp_xmax || py < o.stamp_ymin || py > o.stamp_ymax);
        const int r0 = bbox[4 * oi + 0], r1 = bbox[4 * oi + 1], c0 = bbox[4 * oi + 2], c1 = bbox[4 * oi + 3];
        bool want = false;
        if (P.spikes.enabled && in_stamp && r1 >= r0) {
            if (iy >= r0 && iy <= r1 && ix >= c0 && ix <= c1) v = 0.0;
            want = !spike_sum_is_zero(P.spikes, iy, ix, r0, r1, c0, c1);
        }
        rout[o.r_offset + local] = v;
        // one addition to the counter per wavefront
        const unsigned long long m = __ballot(want);
        if (m != 0ull) {
            const int lane = (int)(threadIdx.x & 63u), lead = __ffsll((long long)m) - 1;
            unsigned int base = 0u;
            if (lane == lead) base = atomicAdd(count, (unsigned int)__popcll(m));
            base = __shfl(base, lead, 64);
            const unsigned int at = base + (unsigned int)__popcll(m & ((1ull << lane) - 1ull));
            if (want && at < cap) list[at] = ((unsigned long long)oi << 40) | (unsigned long long)local;
        }
    });
}

__global__ __launch_bounds__(256) void k_fft_spikes_arms(const ims_fft_params_t P, const ims_fft_object_t* __restrict__ objs,
                                                         const double* __restrict__ rin, double* __restrict__ rout,
                                                         const int32_t* __restrict__ bbox, const unsigned long long* __restrict__ list,
                                                         unsigned int cap, const unsigned int* __restrict__ count)
{
    const unsigned int n = count[0];
    if (n > cap) return;                          // the list ran over: k_fft_spikes does the whole step again
    const unsigned int stride = gridDim.x * blockDim.x;
    for (unsigned int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) {
        const unsigned long long e = list[t];
        const int64_t oi = (int64_t)(e >> 40), local = (int64_t)(e & ((1ull << 40) - 1ull));
        const ims_fft_object_t& o = objs[oi];
        int iy, ix;
        row_col(local, o.nfft, iy, ix);
        const int r0 = bbox[4 * oi + 0], r1 = bbox[4 * oi + 1], c0 = bbox[4 * oi + 2], c1 = bbox[4 * oi + 3];
        const double acc = spike_sum(P, o, iy, ix, r0, r1, c0, c1, rin, P.rbuf_raw != 0, inv_n2(o.nfft));
        rout[o.r_offset + local] = rout[o.r_offset + local] + acc;
    }
}

// clip, Poisson noise, stamp -> CCD add (stamp.py:519-524); realized flux = noise-free sum inside the stamp
__global__ __launch_bounds__(256) void k_fft_finish(const ims_fft_params_t P, const ims_fft_object_t* __restrict__ objs,
                                                    int64_t n_objects, const int64_t* __restrict__ prefix, int64_t n_pix, int64_t span,
                                                    const double* __restrict__ rbuf)
{
    walk_span(prefix, n_objects, n_pix, span, [&](int64_t el, int64_t oi) {
        const ims_fft_object_t& o = objs[oi];
        const int64_t local = el - prefix[oi];
        int iy, ix;
        row_col(local, o.nfft, iy, ix);
        const int px = o.x0 + ix, py = o.y0 + iy;
        if (px < o.stamp_xmin || px > o.stamp_xmax || py < o.stamp_ymin || py > o.stamp_ymax) return;
        double v = rbuf_value(rbuf, o.r_offset + local, P.rbuf_raw != 0, inv_n2(o.nfft));
        if (v < 0.0) v = 0.0;
        if (P.realized_flux != nullptr && v != 0.0) unsafeAtomicAdd(P.realized_flux + oi, v);
        if (P.add_noise) v = poisson(v, P.seed, o.obj_id, local);
        const int cxp = px - P.xmin, cyp = py - P.ymin;
        if (cxp < 0 || cxp >= P.nx || cyp < 0 || cyp >= P.ny || v == 0.0) return;
        unsafeAtomicAdd(P.image + ((int64_t)cyp * P.nx + cxp), v);
    });
}

// ---------------- CCD readout (imsim/readout.py:413-478, imsim/bleed_trails.py) ----------------
constexpr long long READOUT_ID_BASE = 0x7E00000000ll;   // object id space of the read-noise streams (+ amp index)

// span[4 x + 2 half + {0,1}]: first / last saturated row of the channel (relative to the channel's first row)
__global__ __launch_bounds__(256) void k_readout_span_init(int* __restrict__ span, int nx)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x < nx) { span[4 * x] = 0x7FFFFFFF; span[4 * x + 1] = -1; span[4 * x + 2] = 0x7FFFFFFF; span[4 * x + 3] = -1; }
}

__global__ __launch_bounds__(256) void k_readout_flags(const double* __restrict__ image, unsigned char* __restrict__ flags,
                                                       int* __restrict__ span, int nx, int ny, double full_well, int midline_stop)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (int64_t)nx * ny) return;
    const bool sat = image[p] > full_well;
    flags[p] = sat ? 1 : 0;
    if (sat) {                                   // rare: a few hundred pixels of a CCD
        const int x = (int)(p % nx), y = (int)(p / nx);
        const int ymid = ny / 2;
        const int half = (midline_stop && y >= ymid) ? 1 : 0;
        const int yl = half ? y - ymid : y;
        atomicMin(&span[4 * x + 2 * half], yl);
        atomicMax(&span[4 * x + 2 * half + 1], yl);
    }
}

// BleedCharge.__call__ (bleed_trails.py:117-152): returns true once the excess is used up
__device__ __forceinline__ bool bleed_into(double* __restrict__ c, int64_t stride, int n, int ypix, double full_well, double& excess)
{
    if (ypix >= 0 && ypix < n) {
        const double room = full_well - c[(int64_t)ypix * stride];
        const double b = room < excess ? room : excess;
        c[(int64_t)ypix * stride] = c[(int64_t)ypix * stride] + b;
        excess = excess - b;
    } else if (ypix < 0) {
        excess = excess - (full_well < excess ? full_well : excess);      // leaves through the serial register
    }
    return excess == 0.0;
}

// one thread per channel (a column, or half a column with the midline stop); neighbouring threads walk neighbouring
// columns, so every step of the walk is one coalesced row access
__global__ __launch_bounds__(256) void k_readout_bleed(double* __restrict__ image, const unsigned char* __restrict__ flags,
                                                       const int* __restrict__ span, int nx, int ny, double full_well,
                                                       int midline_stop)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_half = midline_stop ? 2 : 1;
    if (t >= nx * n_half) return;
    const int x = t % nx, half = t / nx;
    const int ymid = ny / 2;
    const int ylo = (midline_stop && half == 1) ? ymid : 0;
    const int n = midline_stop ? (half == 0 ? ymid : ny - ymid) : ny;
    double* c = image + (int64_t)ylo * nx + x;
    const unsigned char* f = flags + (int64_t)ylo * nx + x;
    // runs are found on the ORIGINAL flags: only the rows between the first and the last saturated pixel need a look
    int y = span[4 * x + 2 * half];
    const int y_last = span[4 * x + 2 * half + 1];
    while (y <= y_last) {
        // the way to the next saturated row, eight rows' flags at a time (independent loads: the walk used to wait for every
        // single byte before it asked for the next)
        unsigned char fl[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) fl[q] = (y + q <= y_last) ? f[(int64_t)(y + q) * nx] : (unsigned char)0;
        int skip = 0;
        while (skip < 8 && !fl[skip]) ++skip;
        y += skip;
        if (skip == 8) continue;
        const int y0 = y;
        while (y < n && f[(int64_t)y * nx]) ++y;
        const int y1 = y;
        double excess = 0.0;
        for (int k = y0; k < y1; ++k) excess = excess + c[(int64_t)k * nx];
        excess = excess - (double)(y1 - y0) * full_well;
        for (int k = y0; k < y1; ++k) c[(int64_t)k * nx] = full_well;
        const int reach = y0 > n - y1 ? y0 : n - y1;
        for (int dy = 0; dy < reach; ++dy) {
            if (bleed_into(c, nx, n, y0 - dy - 1, full_well, excess)) break;
            if (bleed_into(c, nx, n, y1 + dy, full_well, excess)) break;
        }
    }
}

// e-image pixel read out at position (u, v) of amp a's imaging section, in ADU (float32 like the reference's arrays)
__device__ __forceinline__ float amp_adu(const double* __restrict__ image, int nx, const ims_readout_t& ro, int a, int u, int v)
{
    const ims_amp_t& A = ro.amps[a];
    const int sx = A.flip_x ? ro.seg_w - 1 - u : u;
    const int sy = A.flip_y ? ro.seg_h - 1 - v : v;
    const float e = (float)image[(int64_t)(A.y0 + sy) * nx + (A.x0 + sx)];
    return e / A.gain;
}

// grid (raw_w / 64, raw_h / 4, n_amps), block (64, 4)
// grid (raw_w / 64, raw_h / 4), block (64, 4): a thread forms position (rx, ry) of ALL amplifiers' raw segments -- crosstalk makes every
// amplifier's pixel there a sum over the sixteen pixels at that position, which were read sixteen times (once per output amplifier:
// 2.1 GB for a 131-MB image) and are read once now; the sums are the same sums (ascending source amplifier, zero coefficients skipped)
__global__ __launch_bounds__(256) void k_readout_segments(const double* __restrict__ image, int nx, const ims_readout_t ro,
                                                          float* __restrict__ seg)
{
    const int rx = blockIdx.x * 64 + threadIdx.x, ry = blockIdx.y * 4 + threadIdx.y;
    if (rx >= ro.raw_w || ry >= ro.raw_h) return;
    const int u = rx - ro.data_x0, v = ry - ro.data_y0;
    const bool inside = u >= 0 && u < ro.seg_w && v >= 0 && v < ro.seg_h;
    float e[IMS_MAX_AMPS];
#pragma unroll
    for (int a = 0; a < IMS_MAX_AMPS; ++a) e[a] = (inside && a < ro.n_amps) ? amp_adu(image, nx, ro, a, u, v) : 0.0f;
#pragma unroll
    for (int a = 0; a < IMS_MAX_AMPS; ++a) {
        if (a >= ro.n_amps) break;
        float out = 0.0f;
        if (inside) {
            out = e[a];
            if (ro.has_xtalk) {
                float sum = 0.0f;
#pragma unroll
                for (int j = 0; j < IMS_MAX_AMPS; ++j) {
                    if (j >= ro.n_amps) break;
                    const float x = ro.xtalk[a * IMS_MAX_AMPS + j];
                    if (x != 0.0f) sum = sum + x * e[j];          // a zero coefficient adds exactly nothing
                }
                out = out + sum;
            }
        }
        seg[((int64_t)a * ro.raw_h + ry) * ro.raw_w + rx] = out;
    }
}

// grid (raw_w / 64, raw_h / 4, n_amps), block (64, 4): a wavefront is 64 neighbouring columns of one row, so every tap
// of the parallel direction is one coalesced 256-B row access and the four rows of a workgroup share their taps in L1;
// no integer division anywhere.
// NB > 0: the band width is known at compile time (the reference's ntransfers = 20: 21 taps); pixels with a full
// set of taps take an unrolled path whose 2 x 21 loads are all in flight before the first use.
template <int NB>
__global__ __launch_bounds__(256) void k_readout_cte(const float* __restrict__ src, float* __restrict__ dst, int raw_w, int raw_h,
                                                     const double* __restrict__ band, int n_band, int axis)
{
    const int rx = blockIdx.x * 64 + threadIdx.x, ry = blockIdx.y * 4 + threadIdx.y;
    // serial direction: every lane has a row of weights of its own (its column's) -- the workgroup's 64 rows of weights are one
    // contiguous piece of the band matrix, staged through LDS once instead of 21 scattered 8-byte loads per lane and row
    __shared__ double wl[64 * (NB > 0 ? NB : 1)];
    const bool staged = NB > 0 && axis == 1;
    if (staged) {
        const int i0 = blockIdx.x * 64;
        for (int e = threadIdx.y * 64 + threadIdx.x; e < 64 * NB; e += 256)
            wl[e] = (i0 + e / NB < raw_w) ? band[(int64_t)i0 * NB + e] : 0.0;
        __syncthreads();
    }
    if (rx >= raw_w || ry >= raw_h) return;
    const int p = ((int)blockIdx.z * raw_h + ry) * raw_w + rx;
    const int i = axis == 0 ? ry : rx;
    const int step = axis == 0 ? raw_w : 1;
    const int dmax = i < n_band - 1 ? i : n_band - 1;
    const double* w = band + i * n_band;
    const double* ws = wl + threadIdx.x * (NB > 0 ? NB : 1);
    double acc = 0.0;
    if (NB > 0 && i >= NB - 1) {
        float v[NB > 0 ? NB : 1];
        double ww[NB > 0 ? NB : 1];
#pragma unroll
        for (int d = 0; d < NB; ++d) v[d] = src[p - d * step];
        if (staged) {
#pragma unroll
            for (int d = 0; d < NB; ++d) ww[d] = ws[d];
        } else {
#pragma unroll
            for (int d = 0; d < NB; ++d) ww[d] = w[d];
        }
#pragma unroll
        for (int d = NB - 1; d >= 0; --d) acc = acc + ww[d] * (double)v[d];
    } else if (staged) {
        for (int d = dmax; d >= 0; --d) acc = acc + ws[d] * (double)src[p - d * step];
    } else {
        for (int d = dmax; d >= 0; --d) acc = acc + w[d] * (double)src[p - d * step];
    }
    dst[p] = (float)acc;
}

// The read-noise deviates of pixels 2 m and 2 m + 1 of an amplifier are the two Gaussians of ONE counter block: a thread forms the
// block once and finishes both pixels (segments of an even number of pixels; the single-pixel kernel below did the block twice).
__global__ __launch_bounds__(256) void k_readout_finish_pairs(const float* __restrict__ seg, const ims_readout_t ro, uint64_t seed,
                                                              int32_t* __restrict__ out)
{
    const int64_t per = (int64_t)ro.raw_w * ro.raw_h;
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // pair number over all amplifiers
    if (2 * m >= per * ro.n_amps) return;
    const int64_t p = 2 * m;
    const int a = (int)(p / per);
    const int64_t q = p - (int64_t)a * per;                                 // even
    typedef float fvec2 __attribute__((ext_vector_type(2)));
    typedef int ivec2 __attribute__((ext_vector_type(2)));
    const fvec2 sv = *(const fvec2*)(seg + p);
    Rng r;
    rng_reset(r);
    rng_block(r, seed, READOUT_ID_BASE + a, q >> 1, 0u);
    double g0, g1;
    gauss_words(r.w[0], r.w[1], g0, g1);
    float v0 = sv.x + ro.amps[a].bias_level, v1 = sv.y + ro.amps[a].bias_level;
    v0 = v0 + (float)((double)ro.amps[a].read_noise * g0);
    v1 = v1 + (float)((double)ro.amps[a].read_noise * g1);
    ivec2 o;
    o.x = (int32_t)v0; o.y = (int32_t)v1;
    *(ivec2*)(out + p) = o;
}

__global__ __launch_bounds__(256) void k_readout_finish(const float* __restrict__ seg, const ims_readout_t ro, uint64_t seed,
                                                        int32_t* __restrict__ out)
{
    const int64_t per = (int64_t)ro.raw_w * ro.raw_h;
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= per * ro.n_amps) return;
    const int a = (int)(p / per);
    const int64_t q = p - (int64_t)a * per;
    float v = seg[p] + ro.amps[a].bias_level;
    Rng r;
    rng_reset(r);
    rng_block(r, seed, READOUT_ID_BASE + a, q >> 1, 0u);
    double g0, g1;
    gauss_words(r.w[0], r.w[1], g0, g1);
    v = v + (float)((double)ro.amps[a].read_noise * ((q & 1) ? g1 : g0));
    out[p] = (int32_t)v;
}

// ---------------- phase-screen pre-pass (include/imsim_hip.h: ims_screen_prepass) ----------------
struct ScreenSlices { int64_t first[9]; int64_t seg_first[9]; };     // first object / first segment of every slice
constexpr int SCR_MAXB = 256;
constexpr int SCR_CH = 16;            // segments (of 256 photons) one workgroup of the sort kernels works through

// MODE 0: count the photons of every (slice, time bucket); MODE 1: scatter (object, photon index) into slice-major, time-minor
// order (bins = the exclusive prefix of the counts; the order inside a bin is that of the atomics and does not matter).
// Block b works on slice b mod 8, on SCR_CH consecutive segments of it: its 4 096 photons are counted in LDS first, so a bin
// costs ONE global atomic per workgroup (one per 256-photon segment was 75 M atomics on 1 024 counters: 4 ms), and the
// entries a workgroup adds to a bin are neighbours in memory.
template <int MODE>
__global__ __launch_bounds__(256) void k_screen_sort(const ims_render_params_t P, int comp, int n_buckets, const ScreenSlices S,
                                                     unsigned long long* __restrict__ bins, int64_t* __restrict__ entries)
{
    __shared__ unsigned int hist[SCR_MAXB];
    __shared__ unsigned long long base[SCR_MAXB];
    const int x = blockIdx.x & 7;
    const int64_t seg0 = S.seg_first[x] + (int64_t)(blockIdx.x >> 3) * SCR_CH;
    const int64_t seg_end = S.seg_first[x + 1];
    if (seg0 >= seg_end) return;
    if ((int)threadIdx.x < n_buckets) hist[threadIdx.x] = 0u;
    __syncthreads();
    unsigned char bucket[SCR_CH];
    unsigned int valid = 0u;
#pragma unroll
    for (int c = 0; c < SCR_CH; ++c) {
        const int64_t seg = seg0 + c;
        bucket[c] = 0;
        if (seg >= seg_end) continue;
        const int64_t oi = P.seg_object ? (int64_t)P.seg_object[seg] : find_object(P.seg_prefix, P.n_objects, seg);
        const ims_object_t& o = P.objects[oi];
        const int64_t j = (seg - P.seg_prefix[oi]) * P.seg_size + threadIdx.x;
        if (j >= o.n_phot) continue;
        Rng rng;
        rng_reset(rng);
        rng_block(rng, P.seed, o.obj_id, o.phot_first + j, SLOT_PSF_TIME + (uint32_t)comp);
        const int bk = (int)(((unsigned long long)rng.w[0] * (unsigned long long)n_buckets) >> 32);   // floor(u n): monotone in the time
        bucket[c] = (unsigned char)bk;
        valid |= 1u << c;
        atomicAdd(&hist[bk], 1u);
    }
    __syncthreads();
    if ((int)threadIdx.x < n_buckets) {
        const unsigned int cnt = hist[threadIdx.x];
        if (cnt != 0u) {
            const unsigned long long got = atomicAdd(&bins[x * n_buckets + threadIdx.x], (unsigned long long)cnt);
            if (MODE == 1) base[threadIdx.x] = got;
        }
        if (MODE == 1) hist[threadIdx.x] = 0u;           // becomes the cursor inside the workgroup's share of the bin
    }
    if (MODE == 1) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < SCR_CH; ++c) {
            if (!(valid & (1u << c))) continue;
            const int64_t seg = seg0 + c;
            const int64_t oi = P.seg_object ? (int64_t)P.seg_object[seg] : find_object(P.seg_prefix, P.n_objects, seg);
            const ims_object_t& o = P.objects[oi];
            const int64_t k = o.phot_first + (seg - P.seg_prefix[oi]) * P.seg_size + threadIdx.x;
            const unsigned int rank = atomicAdd(&hist[bucket[c]], 1u);
            entries[base[bucket[c]] + rank] = (int64_t)(((unsigned long long)oi << 32) | (unsigned long long)(uint32_t)k);
        }
    }
}

// exclusive prefix of the counts in place (slice-major, bucket-minor) + the slice boundaries behind them
__global__ void k_screen_scan(unsigned long long* __restrict__ bins, int n_bins, int n_buckets)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    unsigned long long run = 0ull;
    for (int i = 0; i < n_bins; ++i) {
        if (i % n_buckets == 0) bins[n_bins + i / n_buckets] = run;          // start of slice i / n_buckets
        const unsigned long long c = bins[i];
        bins[i] = run;
        run += c;
    }
    bins[n_bins + 8] = run;
}

// what the gather needs of an object, 32 bytes instead of a 256-byte row (3 MB for 100 000 objects: the rows of a slice would
// take as much of the XCD's L2 as the screen windows the pre-pass is there to keep)
struct ScreenObject { int64_t obj_id, screen_base; double tan_x, tan_y; };

__global__ __launch_bounds__(256) void k_screen_objects(const ims_render_params_t P, ScreenObject* __restrict__ slim)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= P.n_objects) return;
    const ims_object_t& o = P.objects[i];
    ScreenObject q = { o.obj_id, o.screen_base, o.atm_tan_x, o.atm_tan_y };
    slim[i] = q;
}

// Slice x in time order on XCD x: block b works on the x = b mod 8 slice, chunk b / 8.  The entries stream in and the kicks
// stream out past the cache (non-temporal): what stays in the XCD's L2 are the screen windows and the 32-byte object records.
__global__ __launch_bounds__(256) void k_screen_gather(const ims_render_params_t P, int comp, const unsigned long long* __restrict__ slice_start,
                                                       const int64_t* __restrict__ entries, const ScreenObject* __restrict__ slim,
                                                       double* __restrict__ kick)
{
    const int x = blockIdx.x & 7;
    const unsigned long long pos = slice_start[x] + (unsigned long long)(blockIdx.x >> 3) * 256ull + threadIdx.x;
    if (pos >= slice_start[x + 1]) return;
    const unsigned long long e = (unsigned long long)__builtin_nontemporal_load(entries + pos);
    const int64_t oi = (int64_t)(e >> 32), k = (int64_t)(e & 0xFFFFFFFFull);
    const ScreenObject o = slim[oi];
    const ims_atmosphere_t& A = *P.atm;
    Rng rng;
    rng_reset(rng);
    rng_block(rng, P.seed, o.obj_id, k, SLOT_PSF + ((uint32_t)comp >> 1));
    const uint32_t wa = (comp & 1) ? rng.w[2] : rng.w[0], wb = (comp & 1) ? rng.w[3] : rng.w[1];
    const double r = dsqrt0(A.aper_ri2 + w01(wa) * A.aper_dr2);
    double s, cc;
    sincos2pi_w(wb, s, cc);
    const double pu = r * cc, pv = r * s;
    rng_block(rng, P.seed, o.obj_id, k, SLOT_PSF_TIME + (uint32_t)comp);
    const double t = A.t0 + w01(rng.w[0]) * A.exptime;
    double gx, gy;
    screen_gradient<true>(A, pu, pv, t, o.tan_x, o.tan_y, gx, gy);
    double* dst = kick + 2 * (o.screen_base + k);
    __builtin_nontemporal_store(gx, dst);
    __builtin_nontemporal_store(gy, dst + 1);
}

// ---------------- object table on the device (include/imsim_hip.h: ims_build_object_table) ----------------
// One thread per catalog source.  The test-side CPU restatement repeats this arithmetic; the formulas are those of
// imsim_amd/catalog.py (the numpy builder, which stays the portable host path).
__device__ __forceinline__ int good_image_size(double stepk, double pixel_scale)
{
    // GSObject.getGoodImageSize: N = ceil(2 pi / (stepk * scale)) rounded up to even
    const double nn = ceil(ddiv(TWO_PI, stepk * pixel_scale));
    const long long n = (long long)nn;
    return (int)(2 * ((n + 1) / 2));
}

// get_good_phot_stamp_size1 (imsim/stamp_utils.py:293-354) for a transformed Sersic profile: grow the square of N pixels by 10 %
// until xValue on its four edge midpoints and four corners is below `keep`, cap at nmax, shrink while the next smaller square
// still is (not below 64).  j0 .. j3: the profile's affine (sky = J profile); amp = flux I(0) / |det J|.
__device__ __forceinline__ double sersic_edge_max(double h, double j0, double j1, double j2, double j3, double det, double hlr,
                                                  double amp, double b, double inv_n)
{
    // the eight points: (h, 0) (-h, 0) (0, h) (0, -h) (h, h) (h, -h) (-h, h) (-h, -h); the profile is point-symmetric, so four suffice
    const double px[4] = { h, 0.0, h, h }, py[4] = { 0.0, h, h, -h };
    double best = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double u = ddiv(j3 * px[k] - j1 * py[k], det), v = ddiv(-j2 * px[k] + j0 * py[k], det);
        const double r = ddiv(dsqrt0(u * u + v * v), hlr);
        const double val = amp * dexp(-b * dpow(r, inv_n));
        best = val > best ? val : best;
    }
    return best;
}

__device__ __forceinline__ long long phot_stamp_size1(int own, double keep, int nmax, double pixel_scale, double j0, double j1, double j2,
                                                      double j3, double hlr, double flux, double norm, double b, double inv_n)
{
    const double det = j0 * j3 - j1 * j2;
    const double amp = ddiv(flux * norm, hlr * hlr * fabs(det));
    double N = (double)own;
    bool active = N < (double)nmax;
    for (int it = 0; it < 200 && active; ++it) {
        const double mv = sersic_edge_max(N * 0.5 * pixel_scale, j0, j1, j2, j3, det, hlr, amp, b, inv_n);
        if (mv < keep) break;
        N = N * 1.1;
        active = N < (double)nmax;
    }
    if (N > (double)nmax) N = (double)nmax;
    active = N >= 64.0 * 1.1;
    for (int it = 0; it < 200 && active; ++it) {
        const double mv = sersic_edge_max(ddiv(N, 2.0 * 1.1) * pixel_scale, j0, j1, j2, j3, det, hlr, amp, b, inv_n);
        if (mv > keep) break;
        N = ddiv(N, 1.1);
        active = N >= 64.0 * 1.1;
    }
    return (long long)N;
}

__global__ __launch_bounds__(256) void k_build_object_table(const ims_catalog_t C, const ims_optics_t* __restrict__ optics,
                                                            ims_object_t* __restrict__ rows, ims_object_meta_t* __restrict__ meta)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= C.n) return;
    ims_object_t o;
    memset(&o, 0, sizeof(o));
    ims_object_meta_t m = { 0, 0, 0 };
    const int kind = C.kind[i];
    const double nominal = C.nominal_flux[i];
    const int64_t id = C.obj_id ? C.obj_id[i] : i;
    const int64_t phot = C.phot_flux ? C.phot_flux[i] : (int64_t)poisson(nominal, C.seed, id, (int64_t)IMS_FLUX_PIXEL);
    if (kind < 0 || kind > 2) {                       // knots, streaks, FITS stamps: the host writes these rows -- with the photon
        m.flags = IMS_META_HOST_ROW;                  // count realised HERE, by the rule of every other object (same stream, same seed)
        m.n_phot = phot;
        rows[i] = o; meta[i] = m;
        return;
    }
    const double x = C.x[i], y = C.y[i];
    o.obj_id = id; o.phot_first = 0; o.n_phot = phot;
    o.x0 = x; o.y0 = y; o.flux_per_photon = 1.0;
    // profile and its affine
    double j0 = 1.0, j1 = 0.0, j2 = 0.0, j3 = 1.0;
    if (kind == 0) {
        o.prof_table = IMS_PROF_POINT; o.prof_scale = 0.0;
    } else {
        o.prof_table = C.prof_table[i]; o.prof_scale = C.hlr[i];
        const double q = C.q[i];
        const double g = ddiv(1.0 - q, 1.0 + q);
        double s2, c2;
        dsincos(2.0 * ((90.0 - C.pa[i]) * 0.017453292519943295), s2, c2);         // beta = 90 deg - pa (flip_g2, instcat.py:503-508)
        const double f = ddiv(1.0, dsqrt0(1.0 - g * g));
        const double sg1 = g * c2, sg2 = g * s2;
        j0 = f * (1.0 + sg1); j1 = f * sg2; j2 = f * sg2; j3 = f * (1.0 - sg1);
        if (C.g1 != nullptr) {
            const double l1 = C.g1[i], l2 = C.g2[i];
            const double lg2 = l1 * l1 + l2 * l2;
            const double lf = ddiv(dsqrt0(C.mu[i]), dsqrt0(1.0 - lg2));
            const double a0 = lf * (1.0 + l1), a1 = lf * l2, a2 = lf * l2, a3 = lf * (1.0 - l1);
            const double b0 = j0, b1 = j1, b2 = j2, b3 = j3;
            j0 = a0 * b0 + a1 * b2; j1 = a0 * b1 + a1 * b3; j2 = a2 * b0 + a3 * b2; j3 = a2 * b1 + a3 * b3;
        }
    }
    o.jac[0] = j0; o.jac[1] = j1; o.jac[2] = j2; o.jac[3] = j3;
    // local WCS: analytic jacobian of pixel -> (u west, v north) [arcsec] at image_pos, inverted
    const ims_tansip_t& w = optics->img_wcs;
    double p[3];
    {
        double u = x - w.crpix[0], v = y - w.crpix[1];
        double fu = 0.0, fv = 0.0, gu = 0.0, gv = 0.0;
        if (w.order > 0) {
            double f, g;
            sip_value_grad(w.a, u, v, f, fu, fv);
            sip_value_grad(w.b, u, v, g, gu, gv);
            u = u + f; v = v + g;
        }
        const double Ux = 1.0 + fu, Uy = fv, Vx = gu, Vy = 1.0 + gv;
        const double xi = w.cd[0] * u + w.cd[1] * v, eta = w.cd[2] * u + w.cd[3] * v;
        const double xi_x = w.cd[0] * Ux + w.cd[1] * Vx, xi_y = w.cd[0] * Uy + w.cd[1] * Vy;
        const double et_x = w.cd[2] * Ux + w.cd[3] * Vx, et_y = w.cd[2] * Uy + w.cd[3] * Vy;
        double px[3], py[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            p[k] = w.rot[k] + w.rot[3 + k] * xi + w.rot[6 + k] * eta;
            px[k] = w.rot[3 + k] * xi_x + w.rot[6 + k] * et_x;
            py[k] = w.rot[3 + k] * xi_y + w.rot[6 + k] * et_y;
        }
        const double inv = ddiv(1.0, dsqrt0(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]));
        p[0] = p[0] * inv; p[1] = p[1] * inv; p[2] = p[2] * inv;                   // unit vector of the object
        const double cd = dsqrt0(p[0] * p[0] + p[1] * p[1]), icd = ddiv(1.0, cd);
        const double e0 = -p[1] * icd, e1 = p[0] * icd;                             // east = pole x p, normalised
        const double n0 = -p[2] * e1, n1 = p[2] * e0, n2 = p[0] * e1 - p[1] * e0;   // north = p x east
        const double k = 206264.80624709636 * inv;                                  // arcsec per radian over |P|
        const double dudx = -(px[0] * e0 + px[1] * e1) * k, dudy = -(py[0] * e0 + py[1] * e1) * k;
        const double dvdx = (px[0] * n0 + px[1] * n1 + px[2] * n2) * k, dvdy = (py[0] * n0 + py[1] * n1 + py[2] * n2) * k;
        const double idet = ddiv(1.0, dudx * dvdy - dudy * dvdx);
        o.winv[0] = dvdy * idet; o.winv[1] = -dudy * idet; o.winv[2] = -dvdx * idet; o.winv[3] = dudx * idet;
        // PhotonDCR: zenith distance and parallactic angle of this object, from the zenith's components along east / north
        const double cz = p[0] * C.zenith[0] + p[1] * C.zenith[1] + p[2] * C.zenith[2];
        const double ez = e0 * C.zenith[0] + e1 * C.zenith[1];
        const double nz = n0 * C.zenith[0] + n1 * C.zenith[1] + n2 * C.zenith[2];
        const double hz = dsqrt0(ez * ez + nz * nz);
        o.dcr_tanz = ddiv(hz, cz);
        o.dcr_sinp = hz > 0.0 ? ddiv(ez, hz) : 0.0;
        o.dcr_cosp = hz > 0.0 ? ddiv(nz, hz) : 1.0;
        if (C.has_field) {
            double thx, thy;
            wcs_vec_to_pix(optics->icrf_to_field, p, thx, thy);
            o.atm_tan_x = thx; o.atm_tan_y = thy;
        }
    }
    o.sed_table = C.sed_table ? C.sed_table[i] : C.sed_table_all;
    o.sed_wave = 0.0;
    o.flags = nominal < C.max_flux_simple ? IMS_OBJ_FAINT : 0;
    o.bf_state = 0;
    // stamp size (stamp.py:205-232)
    int size = C.stamp_size ? C.stamp_size[i] : 0;
    if (size <= 0) {
        if (nominal < C.tiny_flux) size = 32;
        else if (kind == 0) {
            const double ft = ddiv(C.noise_var, nominal);
            int k = 0;                                                   // index 0: the default folding threshold
            if (ft < 5.0e-3 && ft != 0.0) k = (int)(-floor(dlog(ft)));
            if (k >= C.n_star_size) k = C.n_star_size - 1;
            size = C.star_size[k];
        } else {
            const double s1 = j0 * j0 + j1 * j1 + j2 * j2 + j3 * j3;
            const double dd = j0 * j0 + j1 * j1 - j2 * j2 - j3 * j3, od = j0 * j2 + j1 * j3;
            const double s2 = dsqrt0(fmax(dd * dd + 4.0 * od * od, 0.0));
            const double smax = dsqrt0(0.5 * (s1 + s2));
            int t = o.prof_table;
            if (t < 0) t = 0;
            if (t >= C.n_gal_radius) t = C.n_gal_radius - 1;
            const double rr = C.gal_radius[t] * C.hlr[i] * smax;
            constexpr double PI_ = 3.14159265358979323846;
            const double stepk = ddiv(1.0, dsqrt0(ddiv(rr * rr, PI_ * PI_) + ddiv(1.0, C.dg_stepk * C.dg_stepk)));
            size = good_image_size(stepk, C.pixel_scale);
            if (nominal > 10.0 * (double)size * (double)size || size > C.nmax) {
                if (C.sb_tables) {
                    // bright or oversized: the size follows from the surface brightness at the stamp's edge (stamp_utils.py:196-220)
                    const int own = good_image_size(ddiv(PI_, rr), C.pixel_scale);
                    const double flux = C.sb_flux ? C.sb_flux[i] : nominal;
                    const double hl = C.hlr[i];
                    const long long g1 = phot_stamp_size1(own, C.keep_sb, C.nmax, C.pixel_scale, j0, j1, j2, j3, hl, flux, C.sersic_norm[t],
                                                          C.sersic_b[t], C.sersic_inv_n[t]);
                    long long sz = (long long)dsqrt0((double)g1 * (double)g1 + (double)C.psf_size_keep * (double)C.psf_size_keep);
                    if (sz > C.nmax) {
                        const long long g3 = phot_stamp_size1(own, 3.0 * C.keep_sb, C.nmax, C.pixel_scale, j0, j1, j2, j3, hl, flux,
                                                              C.sersic_norm[t], C.sersic_b[t], C.sersic_inv_n[t]);
                        sz = (long long)dsqrt0((double)g3 * (double)g3 + (double)C.psf_size_keep3 * (double)C.psf_size_keep3);
                    }
                    size = (int)(sz > C.nmax ? C.nmax : sz);
                } else {
                    m.flags |= IMS_META_SIZE_PENDING;
                }
            }
            if (size > C.nmax) size = C.nmax;
        }
    }
    const long long icx = (long long)floor(x + 0.5), icy = (long long)floor(y + 0.5);
    o.stamp_xmin = (int)(icx - size / 2); o.stamp_xmax = (int)(icx - size / 2 + size - 1);
    o.stamp_ymin = (int)(icy - size / 2); o.stamp_ymax = (int)(icy - size / 2 + size - 1);
    m.n_phot = phot; m.size = size;
    rows[i] = o; meta[i] = m;
}

__global__ __launch_bounds__(256) void k_patch_stamp_sizes(ims_object_t* __restrict__ rows, ims_object_meta_t* __restrict__ meta,
                                                           const int64_t* __restrict__ index, const int32_t* __restrict__ size, int64_t n)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    ims_object_t& o = rows[index[k]];
    const int sz = size[k];
    const long long icx = (long long)floor(o.x0 + 0.5), icy = (long long)floor(o.y0 + 0.5);
    o.stamp_xmin = (int)(icx - sz / 2); o.stamp_xmax = (int)(icx - sz / 2 + sz - 1);
    o.stamp_ymin = (int)(icy - sz / 2); o.stamp_ymax = (int)(icy - sz / 2 + sz - 1);
    meta[index[k]].size = sz;
    meta[index[k]].flags &= ~IMS_META_SIZE_PENDING;
}

// dst[k] = rows[index[k]] with the launch's photon range and boundary slot: 256-byte rows move as 16 uint4 per row, one row per
// 16 lanes
__global__ __launch_bounds__(256) void k_gather_rows(const ims_object_t* __restrict__ rows, const int64_t* __restrict__ index,
                                                     const int64_t* __restrict__ first, const int64_t* __restrict__ count,
                                                     const int32_t* __restrict__ bf_state, int clear_flags,
                                                     ims_object_t* __restrict__ dst, int64_t n)
{
    static_assert(sizeof(ims_object_t) == 256, "row size");
    static_assert(offsetof(ims_object_t, phot_first) == 8 && offsetof(ims_object_t, n_phot) == 16 &&
                  offsetof(ims_object_t, bf_state) == 172 && offsetof(ims_object_t, flags) == 152, "row layout");
    const int64_t k = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    const int part = threadIdx.x & 15;
    if (k >= n) return;
    uint4 v = ((const uint4*)(rows + index[k]))[part];
    if (part == 0 && first != nullptr) {                       // bytes 8 .. 15: phot_first
        const long long pf = (long long)(((unsigned long long)v.w << 32) | v.z) + (long long)first[k];
        v.z = (unsigned)(unsigned long long)pf; v.w = (unsigned)((unsigned long long)pf >> 32);
    }
    if (part == 1 && count != nullptr) {                       // bytes 16 .. 23: n_phot
        const unsigned long long c = (unsigned long long)count[k];
        v.x = (unsigned)c; v.y = (unsigned)(c >> 32);
    }
    if (part == 10) v.w = bf_state ? (unsigned)bf_state[k] : 0u;   // bytes 172 .. 175: bf_state
    if (part == 9) v.z &= ~(unsigned)clear_flags;                  // bytes 152 .. 155: flags
    ((uint4*)(dst + k))[part] = v;
}

// device math probe for the parity tests (which: 0 log,1 exp,2 sincos2pi,3 atan,4 sincos,5 tanh,6 gauss,7 dsqrt_n,8 ddiv of pairs,9 dsqrt0)
__global__ void k_test_math(int which, const double* __restrict__ in, double* __restrict__ out, int64_t n,
                            uint64_t seed, int64_t obj, uint32_t slot)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s, c;
    switch (which) {
    case 0: out[i] = dlog(in[i]); break;
    case 1: out[i] = dexp(in[i]); break;
    case 2: sincos2pi(in[i], s, c); out[2 * i] = s; out[2 * i + 1] = c; break;
    case 3: out[i] = datan(in[i]); break;
    case 4: dsincos(in[i], s, c); out[2 * i] = s; out[2 * i + 1] = c; break;
    case 5: out[i] = dtanh_pos(in[i]); break;
    case 7: out[i] = dsqrt_n(in[i]); break;
    case 8: out[i] = ddiv(in[2 * i], in[2 * i + 1]); break;
    case 9: out[i] = dsqrt0(in[i]); break;
    case 6: { Rng r; rng_reset(r); rng_block(r, seed, obj, i, slot); gauss_words(r.w[0], r.w[1], s, c);
              out[2 * i] = s; out[2 * i + 1] = c; break; }
    // the deviate functions of spec v6; the words come as integer-valued doubles
    case 10: sincos2pi_w((uint32_t)in[i], s, c); out[2 * i] = s; out[2 * i + 1] = c; break;
    case 11: out[i] = dlog_w((uint32_t)in[i]); break;
    case 12: out[i] = gauss_word_cos((uint32_t)in[2 * i], (uint32_t)in[2 * i + 1]); break;
    }
}

// ---------------- host side of the C-ABI ----------------
static int check_params(const ims_render_params_t* p)
{
    if (!p) return set_err(IMS_ERR_ARG, "params is NULL");
    if (p->n_objects < 0 || p->n_segments < 0) return set_err(IMS_ERR_ARG, "negative object/segment count");
    if (p->n_objects > 0 && (!p->objects || !p->seg_prefix)) return set_err(IMS_ERR_ARG, "objects/seg_prefix is NULL");
    if (p->seg_size != 256) return set_err(IMS_ERR_ARG, "seg_size must be 256 (one photon per thread of a 256-thread workgroup)");
    if (p->n_psf < 0 || p->n_psf > IMS_MAX_PSF) return set_err(IMS_ERR_ARG, "n_psf out of range");
    if (p->n_ops < 0 || p->n_ops > IMS_MAX_OPS) return set_err(IMS_ERR_ARG, "n_ops out of range");
    for (int k = 0; k < p->n_psf; ++k)
        if (p->psf[k].kind == IMS_PSF_SCREENS && !p->atm) return set_err(IMS_ERR_ARG, "phase-screen PSF without atmosphere descriptor");
    for (int k = 0; k < p->n_ops; ++k) {
        const int kind = p->ops[k].kind;
        if ((kind == IMS_OP_RUBIN_OPTICS || kind == IMS_OP_RUBIN_DIFFRACTION || kind == IMS_OP_RUBIN_DIFFRACTION_OPTICS) && !p->optics)
            return set_err(IMS_ERR_ARG, "Rubin optics op without optics descriptor");
    }
    return IMS_OK;
}

static unsigned grid_for_segments(int64_t n_segments)
{
    const int64_t per = (n_segments + N_XCD - 1) / N_XCD;
    return (unsigned)(per * N_XCD);
}

extern "C" {

int ims_abi_version(void) { return IMS_ABI_VERSION; }

int ims_tuning_defaults(ims_tuning_t* out)
{
    if (!out) return set_err(IMS_ERR_ARG, "out is NULL");
    *out = tuning_defaults();
    return IMS_OK;
}
int ims_get_tuning(ims_tuning_t* out)
{
    if (!out) return set_err(IMS_ERR_ARG, "out is NULL");
    std::lock_guard<std::mutex> lock(g_state_mutex);
    *out = g_tune;
    return IMS_OK;
}
int ims_set_tuning(const ims_tuning_t* t)
{
    if (!t) return set_err(IMS_ERR_ARG, "tuning is NULL");
    if (t->photon_lds > 65536 - 4096) return set_err(IMS_ERR_ARG, "tuning: photon_lds beyond what a workgroup may ask for");
    if (t->upd_dpp_max < 0 || t->joint_list_min < 0 || !(t->active_fraction > 0.0) || t->active_fraction > 1.0)
        return set_err(IMS_ERR_ARG, "tuning: upd_dpp_max / joint_list_min / active_fraction out of range");
    std::lock_guard<std::mutex> lock(g_state_mutex);
    g_tune = *t;
    return IMS_OK;
}
const char* ims_last_error(void) { return g_err; }

int ims_device_count(int* count)
{
    if (!count) return set_err(IMS_ERR_ARG, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return hip_err(e, "hipGetDeviceCount"); }
    *count = n;
    return IMS_OK;
}

int ims_device_info(int device, int* n_cu, int* n_xcd, int64_t* lds_bytes, int64_t* hbm_bytes)
{
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (n_xcd) *n_xcd = N_XCD;
    if (lds_bytes) *lds_bytes = (int64_t)prop.sharedMemPerMultiprocessor;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return IMS_OK;
}

int ims_enable_timing(int which) { g_timing = which; g_events_used = 0; return IMS_OK; }

int ims_last_kernel_ms(float* ms, int* n_launches)
{
    if (!ms) return set_err(IMS_ERR_ARG, "ms is NULL");
    if (g_events_used == 0) return set_err(IMS_ERR_ARG, "no timed launch recorded");
    float total = 0.0f;
    for (size_t k = 0; k < g_events_used; k += 2) {
        float t = 0.0f;
        HIP_TRY(hipEventSynchronize(g_events[k + 1]));
        HIP_TRY(hipEventElapsedTime(&t, g_events[k], g_events[k + 1]));
        total += t;
    }
    *ms = total;
    if (n_launches) *n_launches = (int)(g_events_used / 2);
    g_events_used = 0;
    return IMS_OK;
}

// the descriptor lists exactly imSim's default photon-op chain (run_ops<1>)
static bool is_default_chain(const ims_render_params_t* p)
{
    static const int32_t kinds[IMS_DEFAULT_CHAIN_LEN] = { IMS_OP_TIME_SAMPLER, IMS_OP_PUPIL_ANNULUS_SAMPLER, IMS_OP_PHOTON_DCR,
                                                          IMS_OP_RUBIN_DIFFRACTION_OPTICS, IMS_OP_FOCUS_DEPTH, IMS_OP_REFRACTION };
    if (p->n_ops != IMS_DEFAULT_CHAIN_LEN || !g_tune.chain_kernels) return false;
    for (int k = 0; k < IMS_DEFAULT_CHAIN_LEN; ++k)
        if (p->ops[k].kind != kinds[k]) return false;
    return true;
}

// 1: radial table then Gaussian (run_psf<1>); 2: phase screens, second-kick table, Gaussian (run_psf<2>: imSim's default
// AtmosphericPSF); 0: anything else.  (Variant 2 is slower than the component loop at four workgroups per CU -- C3b 36.9 -> 38.6 ms
// in round 2, 35.2 -> 35.9 ms now -- and faster at the three that launches with phase screens run with (photon_lds_pad):
// 33.7 -> 33.2 ms.  ims_tuning_t.psf_screens_kernel = 0 takes the loop.)
static int psf_variant(const ims_render_params_t* p)
{
    if (!g_tune.chain_kernels) return 0;
    if (p->n_psf == 2 && p->psf[0].kind == IMS_PSF_RADIAL && p->psf[1].kind == IMS_PSF_GAUSSIAN) return 1;
    if (g_tune.psf_screens_kernel && p->n_psf == 3 && p->psf[0].kind == IMS_PSF_SCREENS && p->psf[1].kind == IMS_PSF_RADIAL &&
        p->psf[2].kind == IMS_PSF_GAUSSIAN && p->atm != nullptr) return 2;
    return 0;
}

int ims_known_optics_layout(uint64_t layout) { return layout == IMS_LAYOUT_RUBIN_LIKE ? 1 : 0; }

// Dynamic LDS a photon-kernel launch asks for without using it: with more than a quarter of a CU's 160 KB per workgroup only
// three photon workgroups are resident per CU.  Launches whose PSF gathers phase screens are bound by those scattered loads, not
// by instruction issue, and run 2.5 % faster that way (C3b 34.7 -> 33.8 ms: fewer wavefronts thrash the caches less); the
// analytic-PSF kernels are issue-bound and lose 9 % (C4 237 -> 258 ms; the free fourth slot does not buy the brighter-fatter
// chains enough: C3 23.6 -> 25.4 ms), so they ask for none.  ims_tuning_t.photon_lds >= 0 (bytes) overrides both.
static unsigned photon_lds_pad(const ims_render_params_t* p)
{
    if (g_tune.photon_lds >= 0 && g_tune.photon_lds <= 65536 - 4096) return (unsigned)g_tune.photon_lds;
    for (int c = 0; c < p->n_psf; ++c)
        if (p->psf[c].kind == IMS_PSF_SCREENS && p->atm != nullptr && p->screen_kick == nullptr) return 41984u;
    return 0u;
}

// the list of the photons a lazy_static launch sets aside: one buffer per (device, stream), sized for every photon of the launch
// (launches on a stream run in order, so the next launch finds the second pass of the one before through with it)
struct MarginBuf { double* list = nullptr; int32_t* count = nullptr; unsigned char* wave_count = nullptr; int64_t waves_cap = 0; uint32_t cap = 0; };
static std::map<std::pair<int, void*>, MarginBuf> g_margin;

int ims_shoot_accumulate(const ims_render_params_t* params_in, void* stream)
{
    int rc = check_params(params_in);
    if (rc) return rc;
    if (!params_in->image) return set_err(IMS_ERR_ARG, "image is NULL");
    if (params_in->n_segments == 0) return IMS_OK;
    hipStream_t st = (hipStream_t)stream;
    ims_render_params_t lazy_copy;
    const ims_render_params_t* params = params_in;
    if (params_in->lazy_static) {
        if (!params_in->sensor || params_in->track_static_delta)
            return set_err(IMS_ERR_ARG, "lazy_static needs a Silicon sensor whose slot 0 is not a live region (track_static_delta 0)");
        const int64_t want = params_in->n_segments * (int64_t)params_in->seg_size;
        if (want > (int64_t)1 << 26) return set_err(IMS_ERR_UNSUPPORTED, "lazy_static: more than 2^26 photons in one launch");
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        MarginBuf mb;
        {
            std::lock_guard<std::mutex> lock(g_state_mutex);
            MarginBuf& m = g_margin[std::make_pair(dev, stream)];
            const int64_t waves = params_in->n_segments * 4;
            if (m.cap < (uint32_t)want || m.waves_cap < waves) {
                // grown into temporaries and committed only when every allocation has succeeded: an entry left with freed
                // pointers and the new capacities would pass this very check on the next call
                if (!m.count) HIP_TRY(hipMalloc((void**)&m.count, 2 * sizeof(int32_t)));
                const uint32_t cap = (uint32_t)(want + want / 4 + 65536);
                const int64_t waves_cap = waves + waves / 4 + 1024;
                double* list = nullptr;
                unsigned char* wave_count = nullptr;
                // an octet of records per wavefront, then an overflow region that could take every photon of the launch
                HIP_TRY(hipMalloc((void**)&list, ((size_t)waves_cap * 8 + (size_t)cap) * 5 * sizeof(double)));
                const hipError_t e2 = hipMalloc((void**)&wave_count, (size_t)waves_cap);
                if (e2 != hipSuccess) { (void)hipFree(list); return hip_err(e2, "hipMalloc(margin wave counts)"); }
                if (m.list) { (void)hipFree(m.list); (void)hipFree(m.wave_count); }
                m.list = list; m.wave_count = wave_count; m.cap = cap; m.waves_cap = waves_cap;
            }
            mb = m;
        }
        lazy_copy = *params_in;
        lazy_copy.margin_list = mb.list; lazy_copy.margin_count = mb.count; lazy_copy.margin_cap = mb.cap;
        lazy_copy.margin_wave_count = mb.wave_count; lazy_copy.margin_waves = params_in->n_segments * 4;
        params = &lazy_copy;
        HIP_TRY(hipMemsetAsync(mb.count, 0, 2 * sizeof(int32_t), st));
        HIP_TRY(hipMemsetAsync(mb.wave_count, 0, (size_t)lazy_copy.margin_waves, st));
    }
    {
        LaunchTimer tm(st, 1);
        const dim3 grid(grid_for_segments(params->n_segments));
        const int pv = is_default_chain(params) ? psf_variant(params) : -1;
        const bool lay = pv >= 0 && params->optics_layout == IMS_LAYOUT_RUBIN_LIKE && g_tune.layout_kernels != 0;
        if (pv == 2 && lay) hipLaunchKernelGGL((k_shoot_accumulate<1, 2, IMS_LAYOUT_RUBIN_LIKE>), grid, dim3(256), photon_lds_pad(params), st, *params);
        else if (pv == 2) hipLaunchKernelGGL((k_shoot_accumulate<1, 0>), grid, dim3(256), photon_lds_pad(params), st, *params);
        else if (pv == 1 && lay) hipLaunchKernelGGL((k_shoot_accumulate<1, 1, IMS_LAYOUT_RUBIN_LIKE>), grid, dim3(256), photon_lds_pad(params), st, *params);
        else if (pv == 0 && lay) hipLaunchKernelGGL((k_shoot_accumulate<1, 0, IMS_LAYOUT_RUBIN_LIKE>), grid, dim3(256), photon_lds_pad(params), st, *params);
        else if (pv == 1) hipLaunchKernelGGL((k_shoot_accumulate<1, 1>), grid, dim3(256), photon_lds_pad(params), st, *params);
        else if (pv == 0) hipLaunchKernelGGL((k_shoot_accumulate<1, 0>), grid, dim3(256), photon_lds_pad(params), st, *params);
        else hipLaunchKernelGGL((k_shoot_accumulate<0, 0>), grid, dim3(256), photon_lds_pad(params), st, *params);
    }
    HIP_TRY(hipGetLastError());
    if (params->lazy_static) {
        // the photons set aside (~2 % of the launch's): one lane each, the polygons evaluated on the way
        hipLaunchKernelGGL(k_margin_photons, dim3(1024), dim3(256), 0, st, *params);
        HIP_TRY(hipGetLastError());
    }
    return IMS_OK;
}

int ims_shoot_photons(const ims_render_params_t* params, const int64_t* photon_offset,
                      const ims_photons_t* pool, void* stream)
{
    int rc = check_params(params);
    if (rc) return rc;
    if (!photon_offset || !pool) return set_err(IMS_ERR_ARG, "photon_offset/pool is NULL");
    if (params->n_segments == 0) return IMS_OK;
    hipStream_t st = (hipStream_t)stream;
    {
        hipLaunchKernelGGL((k_shoot_photons<0, 0>), dim3(grid_for_segments(params->n_segments)), dim3(256), 0, st,
                           *params, photon_offset, *pool);
    }
    HIP_TRY(hipGetLastError());
    return IMS_OK;
}

int ims_shoot_ops_photons(const ims_render_params_t* params, const int64_t* photon_offset,
                          const ims_photons_t* pool, void* stream)
{
    int rc = check_params(params);
    if (rc) return rc;
    if (!photon_offset || !pool) return set_err(IMS_ERR_ARG, "photon_offset/pool is NULL");
    if (!pool->x || !pool->y || !pool->flux || !pool->dxdz) return set_err(IMS_ERR_ARG, "pool: x / y / flux / dxdz is NULL");
    if (!pool->converted && (!pool->dydz || !pool->wavelength)) return set_err(IMS_ERR_ARG, "pool: dydz / wavelength is NULL");
    if (params->n_segments == 0) return IMS_OK;
    hipStream_t st = (hipStream_t)stream;
    {
        LaunchTimer tm(st, 2);
        const dim3 grid(grid_for_segments(params->n_segments));
        const int pv = is_default_chain(params) ? psf_variant(params) : -1;
        const bool lay = pv >= 0 && params->optics_layout == IMS_LAYOUT_RUBIN_LIKE && g_tune.layout_kernels != 0;
        if (pool->converted && pv == 2 && lay)
            hipLaunchKernelGGL((k_shoot_photons<2, 1, 2, IMS_LAYOUT_RUBIN_LIKE>), grid, dim3(256), photon_lds_pad(params), st, *params, photon_offset, *pool);
        else if (pool->converted && pv == 2)
            hipLaunchKernelGGL((k_shoot_photons<2, 1, 0>), grid, dim3(256), photon_lds_pad(params), st, *params, photon_offset, *pool);
        else if (pool->converted && pv == 1 && lay)
            hipLaunchKernelGGL((k_shoot_photons<2, 1, 1, IMS_LAYOUT_RUBIN_LIKE>), grid, dim3(256), photon_lds_pad(params), st, *params, photon_offset, *pool);
        else if (pool->converted && pv == 0 && lay)
            hipLaunchKernelGGL((k_shoot_photons<2, 1, 0, IMS_LAYOUT_RUBIN_LIKE>), grid, dim3(256), photon_lds_pad(params), st, *params, photon_offset, *pool);
        else if (pool->converted && pv == 1) hipLaunchKernelGGL((k_shoot_photons<2, 1, 1>), grid, dim3(256), photon_lds_pad(params), st, *params, photon_offset, *pool);
        else if (pool->converted && pv == 0) hipLaunchKernelGGL((k_shoot_photons<2, 1, 0>), grid, dim3(256), photon_lds_pad(params), st, *params, photon_offset, *pool);
        else if (pool->converted) hipLaunchKernelGGL((k_shoot_photons<2, 0>), grid, dim3(256), photon_lds_pad(params), st, *params, photon_offset, *pool);
        else hipLaunchKernelGGL((k_shoot_photons<1, 0>), grid, dim3(256), photon_lds_pad(params), st, *params, photon_offset, *pool);
    }
    HIP_TRY(hipGetLastError());
    return IMS_OK;
}

int ims_accumulate_segments(const ims_render_params_t* params, const ims_photons_t* pool, const int64_t* pool_start,
                            int32_t num_vertices, void* stream)
{
    int rc = check_params(params);
    if (rc) return rc;
    if (!pool || !pool_start) return set_err(IMS_ERR_ARG, "pool/pool_start is NULL");
    if (!pool->converted) return set_err(IMS_ERR_ARG, "pool must hold converted photons (ims_shoot_ops_photons with pool->converted = 1)");
    if (!params->image) return set_err(IMS_ERR_ARG, "image is NULL");
    if (params->lazy_static) return set_err(IMS_ERR_ARG, "lazy_static parameters (slot 0 holds no state) belong to ims_shoot_accumulate only");
    if (params->n_segments == 0) return IMS_OK;
    hipStream_t st = (hipStream_t)stream;
    {
        const dim3 grid(grid_for_segments(params->n_segments));
        if (num_vertices == 4)
            hipLaunchKernelGGL(k_accumulate_segments<4>, grid, dim3(256), 0, st, *params, *pool, pool_start);
        else if (num_vertices == 8)
            hipLaunchKernelGGL(k_accumulate_segments<8>, grid, dim3(256), 0, st, *params, *pool, pool_start);
        else
            hipLaunchKernelGGL(k_accumulate_segments<0>, grid, dim3(256), 0, st, *params, *pool, pool_start);
    }
    HIP_TRY(hipGetLastError());
    return IMS_OK;
}

int ims_accumulate_small(const ims_render_params_t* params, const ims_photons_t* pool, const int64_t* pool_start,
                         int32_t num_vertices, void* stream)
{
    if (!params) return set_err(IMS_ERR_ARG, "params is NULL");
    if (!params->objects || !params->image) return set_err(IMS_ERR_ARG, "objects/image is NULL");
    if (!pool || !pool_start) return set_err(IMS_ERR_ARG, "pool/pool_start is NULL");
    if (!pool->converted) return set_err(IMS_ERR_ARG, "pool must hold converted photons (ims_shoot_ops_photons with pool->converted = 1)");
    if (params->lazy_static) return set_err(IMS_ERR_ARG, "lazy_static parameters (slot 0 holds no state) belong to ims_shoot_accumulate only");
    if (params->n_objects <= 0) return IMS_OK;
    const int64_t blocks = (params->n_objects + 3) / 4;
    if (blocks > 0x7fffffffLL) return set_err(IMS_ERR_ARG, "too many objects for one launch");
    const dim3 grid((unsigned)blocks);
    hipStream_t st = (hipStream_t)stream;
    if (num_vertices == 4) hipLaunchKernelGGL(k_accumulate_small<4>, grid, dim3(256), 0, st, *params, *pool, pool_start);
    else if (num_vertices == 8) hipLaunchKernelGGL(k_accumulate_small<8>, grid, dim3(256), 0, st, *params, *pool, pool_start);
    else hipLaunchKernelGGL(k_accumulate_small<0>, grid, dim3(256), 0, st, *params, *pool, pool_start);
    HIP_TRY(hipGetLastError());
    return IMS_OK;
}

int ims_accumulate_round(const ims_render_params_t* params, const ims_photons_t* pool, const int64_t* pool_start,
                         int32_t round, int32_t nrecalc, int32_t n_active, int32_t num_vertices, void* stream)
{
    if (!params) return set_err(IMS_ERR_ARG, "params is NULL");
    if (!params->objects || !params->image) return set_err(IMS_ERR_ARG, "objects/image is NULL");
    if (!pool || !pool_start) return set_err(IMS_ERR_ARG, "pool/pool_start is NULL");
    if (!pool->converted) return set_err(IMS_ERR_ARG, "pool must hold converted photons (ims_shoot_ops_photons with pool->converted = 1)");
    if (round < 0 || nrecalc <= 0) return set_err(IMS_ERR_ARG, "round must be >= 0 and nrecalc positive");
    if (n_active < 0 || n_active > params->n_objects) return set_err(IMS_ERR_ARG, "n_active out of range");
    if (n_active == 0) return IMS_OK;
    // (One-wavefront workgroups for this launch -- a 256-thread workgroup starts only where a compute unit has a free register slot on
    // each of its four SIMDs at once, and the pixel search of a round waited ~220 us for that beside the photon kernels -- were built
    // and measured in round 4: the wait disappears and the photon kernels lose what the chain gains, C3 24.0 -> 25.8 ms; removed.)
    const int32_t segs = (nrecalc + 255) / 256;
    if ((int64_t)n_active * segs > 0x7fffffffLL) return set_err(IMS_ERR_ARG, "too many workgroups for one round");
    const dim3 grid((unsigned)(n_active * segs));
    const int64_t first = (int64_t)round * nrecalc;
    LaunchTimer tm((hipStream_t)stream, 4);
    if (num_vertices == 4)
        hipLaunchKernelGGL(k_accumulate_round<4>, grid, dim3(256), 0, (hipStream_t)stream, *params, *pool, pool_start, first, nrecalc, segs);
    else if (num_vertices == 8)
        hipLaunchKernelGGL(k_accumulate_round<8>, grid, dim3(256), 0, (hipStream_t)stream, *params, *pool, pool_start, first, nrecalc, segs);
    else
        hipLaunchKernelGGL(k_accumulate_round<0>, grid, dim3(256), 0, (hipStream_t)stream, *params, *pool, pool_start, first, nrecalc, segs);
    HIP_TRY(hipGetLastError());
    return IMS_OK;
}

// ims_accumulate_round launched with the compact argument block (4 vertices per edge, 256-thread workgroups)
static int accumulate_round_compact(const ims_render_params_t* params, const ims_photons_t* pool, const int64_t* pool_start, int32_t round,
                                    int32_t nrecalc, int32_t n_active, int32_t num_vertices, void* stream)
{
    if (n_active == 0) return IMS_OK;
    if (num_vertices != 4 || !g_tune.round_compact)
        return ims_accumulate_round(params, pool, pool_start, round, nrecalc, n_active, num_vertices, stream);
    if (!params || !params->objects || !params->image || !pool || !pool_start || !pool->converted) return set_err(IMS_ERR_ARG, "NULL argument");
    const int32_t segs = (nrecalc + 255) / 256;
    if ((int64_t)n_active * segs > 0x7fffffffLL) return set_err(IMS_ERR_ARG, "too many workgroups for one round");
    RoundArgs a;
    a.objects = params->objects; a.sensor = params->sensor; a.image = params->image; a.realized_flux = params->realized_flux;
    a.px = pool->x; a.py = pool->y; a.pflux = pool->flux; a.pz = pool->dxdz;
    a.nx = params->nx; a.ny = params->ny; a.xmin = params->xmin; a.ymin = params->ymin;
    a.bf_tag = params->bf_tag; a.bf_slot_shift = params->bf_slot_shift; a.track_static_delta = params->track_static_delta; a.pad = 0;
    LaunchTimer tm((hipStream_t)stream, 4);
    if (g_tune.round_two_segments) {
        const int32_t segs2 = (nrecalc + 511) / 512;
        hipLaunchKernelGGL((k_accumulate_round_c2<4>), dim3((unsigned)(n_active * segs2)), dim3(256), 0, (hipStream_t)stream, a, pool_start,
                           (int64_t)round * nrecalc, nrecalc, segs2);
    } else
    hipLaunchKernelGGL((k_accumulate_round_c<4, 256>), dim3((unsigned)(n_active * segs)), dim3(256), 0, (hipStream_t)stream, a, pool_start,
                       (int64_t)round * nrecalc, nrecalc, segs);
    HIP_TRY(hipGetLastError());
    return IMS_OK;
}

static unsigned grid_for_pool(int64_t n)
{
    int64_t blocks = (n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

int ims_apply_ops(const ims_render_params_t* params, const int64_t* photon_offset, const ims_photons_t* pool, void* stream)
{
    int rc = check_params(params);
    if (rc) return rc;
    if (!photon_offset || !pool) return set_err(IMS_ERR_ARG, "photon_offset/pool is NULL");
    if (pool->n == 0 || params->n_ops == 0) return IMS_OK;
    hipStream_t st = (hipStream_t)stream;
    {
        hipLaunchKernelGGL(k_apply_ops, dim3(grid_for_pool(pool->n)), dim3(256), 0, st, *params, photon_offset, *pool);
    }
    HIP_TRY(hipGetLastError());
    return IMS_OK;
}

int ims_accumulate(const ims_render_params_t* params, const int64_t* photon_offset, const ims_photons_t* pool,
                   int32_t* pixel_index_out, void* stream)
{
    int rc = check_params(params);
    if (rc) return rc;
    if (!photon_offset || !pool) return set_err(IMS_ERR_ARG, "photon_offset/pool is NULL");
    if (!params->image) return set_err(IMS_ERR_ARG, "image is NULL");
    if (params->lazy_static) return set_err(IMS_ERR_ARG, "lazy_static parameters (slot 0 holds no state) belong to ims_shoot_accumulate only");
    if (pool->n == 0) return IMS_OK;
    hipStream_t st = (hipStream_t)stream;
    {
        hipLaunchKernelGGL(k_accumulate, dim3(grid_for_pool(pool->n)), dim3(256), 0, st, *params, photon_offset, *pool,
                           pixel_index_out);
    }
    HIP_TRY(hipGetLastError());
    return IMS_OK;
}

static int slot_range_cells(const ims_sensor_t* host, int first, int n, int64_t* begin, int64_t* count)
{
    if (!host || !host->bf_slots) return set_err(IMS_ERR_ARG, "sensor_host/bf_slots is NULL (host copy of the slot table required)");
    if (n <= 0 || first < 0 || first + n > host->n_bf_slots) return set_err(IMS_ERR_ARG, "slot range out of bounds");
    const ims_bf_slot_t& a = host->bf_slots[first];
    const ims_bf_slot_t& z = host->bf_slots[first + n - 1];
    *begin = a.offset;
    *count = z.offset + (int64_t)(z.nx + 1) * (z.ny + 1) - a.offset;
    return IMS_OK;
}

int ims_sensor_init_boundaries(const ims_sensor_t* sensor_dev, const ims_sensor_t* sensor_host,
                               int32_t first_slot, int32_t n_slots, const int64_t* tile_prefix_dev, int64_t n_tiles, void* stream)
{
    if (!sensor_dev) return set_err(IMS_ERR_ARG, "sensor_dev is NULL");
    if (n_slots == 0) return IMS_OK;
    int64_t begin, count;
    int rc = slot_range_cells(sensor_host, first_slot, n_slots, &begin, &count);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (sensor_host->num_vertices == IT_NV && g_tune.init_tiles && tile_prefix_dev != nullptr) {
        if (n_tiles <= 0 || n_tiles > 0x7fffffffLL) return set_err(IMS_ERR_ARG, "n_tiles out of range");
        hipLaunchKernelGGL(k_init_tiles, dim3((unsigned)n_tiles), dim3(256), 0, st, sensor_dev, first_slot, n_slots, tile_prefix_dev);
        HIP_TRY(hipGetLastError());
        return IMS_OK;
    }
    if (sensor_host->num_vertices == IT_NV && g_tune.init_tiles) {
        // tiled kernel: grid.x = the largest tile count of a run of consecutive slots, grid.y = the slots of the run (at most
        // 65 535); a run ends where the workgroups that find no tile would outnumber the working ones four to one
        int a = first_slot;
        const int end = first_slot + n_slots;
        while (a < end) {
            auto tiles_of = [&](int k) {
                const ims_bf_slot_t& b = sensor_host->bf_slots[k];
                return (int64_t)((b.nx + 1 + UT - 1) / UT) * ((b.ny + 1 + UT - 1) / UT);
            };
            int64_t hi = tiles_of(a), sum = hi;
            int z = a + 1;
            while (z < end && z - a < 65535) {
                const int64_t tz = tiles_of(z);
                const int64_t nhi = tz > hi ? tz : hi;
                if (nhi * (z - a + 1) > 4 * (sum + tz) + 4096) break;
                hi = nhi; sum += tz; ++z;
            }
            if (hi > 0x7fffffffLL) return set_err(IMS_ERR_ARG, "slot too large");
            hipLaunchKernelGGL(k_init_tiles, dim3((unsigned)hi, (unsigned)(z - a)), dim3(256), 0, st, sensor_dev, a, z - a,
                               (const int64_t*)nullptr);
            a = z;
        }
        HIP_TRY(hipGetLastError());
        return IMS_OK;
    }
    const unsigned g = (unsigned)((count + 255) / 256);
    hipLaunchKernelGGL(k_init_boundaries, dim3(g), dim3(256), 0, st, sensor_dev, first_slot, n_slots, begin, count);
    HIP_TRY(hipGetLastError());
    return IMS_OK;
}

static int update_distortions_impl(const ims_sensor_t* sensor_dev, const ims_sensor_t* sensor_host,
                                   int32_t first_slot, int32_t n_slots, const int64_t* tile_prefix_dev,
                                   int64_t n_tiles, unsigned char* changed_dev, uint32_t tag, double* fold_image, void* stream);

int ims_sensor_update_distortions(const ims_sensor_t* sensor_dev, const ims_sensor_t* sensor_host,
                                  int32_t first_slot, int32_t n_slots, const int64_t* tile_prefix_dev,
                                  int64_t n_tiles, unsigned char* changed_dev, uint32_t tag, void* stream)
{
    return update_distortions_impl(sensor_dev, sensor_host, first_slot, n_slots, tile_prefix_dev, n_tiles, changed_dev, tag, nullptr, stream);
}

// slot `slot` must be the pixel grid of the image (nx x ny): what the fold forms index with
static int fold_geometry(const ims_sensor_t* sensor_host, int32_t slot, double* image_dev, int32_t nx, int32_t ny)
{
    if (!image_dev) return set_err(IMS_ERR_ARG, "image is NULL");
    if (!sensor_host || !sensor_host->bf_slots || slot < 0 || slot >= sensor_host->n_bf_slots)
        return set_err(IMS_ERR_ARG, "sensor_host / its slot table is NULL or the slot is out of range");
    const ims_bf_slot_t& b = sensor_host->bf_slots[slot];
    if (b.nx != nx || b.ny != ny) return set_err(IMS_ERR_ARG, "the slot is not the pixel grid of the image (nx / ny differ)");
    return IMS_OK;
}

int ims_sensor_update_distortions_fold(const ims_sensor_t* sensor_dev, const ims_sensor_t* sensor_host, const int64_t* tile_prefix_dev,
                                       int64_t n_tiles, unsigned char* changed_dev, uint32_t tag, double* image_dev, int32_t nx, int32_t ny,
                                       void* stream)
{
    const int rc = fold_geometry(sensor_host, 0, image_dev, nx, ny);
    if (rc) return rc;
    return update_distortions_impl(sensor_dev, sensor_host, 0, 1, tile_prefix_dev, n_tiles, changed_dev, tag, image_dev, stream);
}

__global__ __launch_bounds__(256) void k_fold_delta(const ims_sensor_t* __restrict__ sp, int slot, double* __restrict__ image)
{
    const ims_sensor_t& s = *sp;
    const ims_bf_slot_t bs = s.bf_slots[slot];
    const int64_t n = (int64_t)bs.nx * bs.ny, stride = (int64_t)gridDim.x * 256;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n; p += stride) {
        const int i = (int)(p % bs.nx), j = (int)(p / bs.nx);
        const int64_t c = bs.offset + (int64_t)j * (bs.nx + 1) + i;
        const double d = s.bf_delta[c];
        if (d != 0.0) { image[p] += d; s.bf_delta[c] = 0.0; }
    }
}

int ims_sensor_fold_delta(const ims_sensor_t* sensor_dev, const ims_sensor_t* sensor_host, int32_t slot, double* image_dev, int32_t nx,
                          int32_t ny, void* stream)
{
    if (!sensor_dev) return set_err(IMS_ERR_ARG, "sensor_dev is NULL");
    const int rc = fold_geometry(sensor_host, slot, image_dev, nx, ny);
    if (rc) return rc;
    hipLaunchKernelGGL(k_fold_delta, dim3(4096), dim3(256), 0, (hipStream_t)stream, sensor_dev, slot, image_dev);
    HIP_TRY(hipGetLastError());
    return IMS_OK;
}

static int update_distortions_impl(const ims_sensor_t* sensor_dev, const ims_sensor_t* sensor_host,
                                   int32_t first_slot, int32_t n_slots, const int64_t* tile_prefix_dev,
                                   int64_t n_tiles, unsigned char* changed_dev, uint32_t tag, double* fold_image, void* stream)
{
    if (!sensor_dev) return set_err(IMS_ERR_ARG, "sensor_dev is NULL");
    if (n_slots == 0) return IMS_OK;
    if (!tile_prefix_dev || !changed_dev) return set_err(IMS_ERR_ARG, "tile_prefix/changed is NULL");
    if (sensor_host && sensor_host->qdist > UQMAX) return set_err(IMS_ERR_UNSUPPORTED, "qdist > 4 not supported");
    int64_t begin, count;
    int rc = slot_range_cells(sensor_host, first_slot, n_slots, &begin, &count);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (first_slot == 0)                      // slot 0 leaves its pristine state: photons must consult the boundaries again
        hipLaunchKernelGGL(k_clear_pristine, dim3(1), dim3(1), 0, st, const_cast<ims_sensor_t*>(sensor_dev));
    const unsigned g = (unsigned)((count + 255) / 256);
    const int nV = sensor_host ? sensor_host->num_vertices : 0;
    const int q = sensor_host ? sensor_host->qdist : 0;
    // The DPP form of the update is the faster one where a round's latency counts (a few tiles: 14.2 -> 11.0 us per launch for
    // one star), the SGPR form where a launch is throughput work beside the photon kernels (its waves sleep on the scalar
    // cache instead of pulling 10 KB of table through LDS per tile: C3 25.0 against 25.9 ms): the tile count decides.
    const bool dpp = sensor_host && sensor_host->bf_dl != nullptr && g_tune.upd_dpp && n_tiles <= g_tune.upd_dpp_max;
    if (q == 3 && nV == 4 && dpp)
        hipLaunchKernelGGL((k_update_distortions_q3<4, true>), dim3((unsigned)n_tiles), dim3(256), 0, st, sensor_dev, first_slot,
                           n_slots, tile_prefix_dev, changed_dev, tag, sensor_host->bf_dl);
    else if (q == 3 && nV == 8 && dpp)
        hipLaunchKernelGGL((k_update_distortions_q3<8, true>), dim3((unsigned)n_tiles), dim3(256), 0, st, sensor_dev, first_slot,
                           n_slots, tile_prefix_dev, changed_dev, tag, sensor_host->bf_dl);
    else if (q == 3 && nV == 4)
        hipLaunchKernelGGL(k_update_distortions_q3<4>, dim3((unsigned)n_tiles), dim3(256), 0, st, sensor_dev, first_slot,
                           n_slots, tile_prefix_dev, changed_dev, tag, sensor_host->bf_dl);
    else if (q == 3 && nV == 8)
        hipLaunchKernelGGL(k_update_distortions_q3<8>, dim3((unsigned)n_tiles), dim3(256), 0, st, sensor_dev, first_slot,
                           n_slots, tile_prefix_dev, changed_dev, tag, sensor_host->bf_dl);
    else
        hipLaunchKernelGGL(k_update_distortions, dim3((unsigned)n_tiles), dim3(256), 0, st, sensor_dev, first_slot, n_slots,
                           tile_prefix_dev, changed_dev, tag);
    if (nV == 4)
        hipLaunchKernelGGL(k_refresh_changed<4>, dim3((unsigned)n_tiles), dim3(256), 0, st, sensor_dev, first_slot, n_slots,
                           tile_prefix_dev, (const unsigned char*)changed_dev, tag, fold_image);
    else if (nV == 8)
        hipLaunchKernelGGL(k_refresh_changed<8>, dim3((unsigned)n_tiles), dim3(256), 0, st, sensor_dev, first_slot, n_slots,
                           tile_prefix_dev, (const unsigned char*)changed_dev, tag, fold_image);
    else
        hipLaunchKernelGGL(k_refresh_changed<0>, dim3((unsigned)n_tiles), dim3(256), 0, st, sensor_dev, first_slot, n_slots,
                           tile_prefix_dev, (const unsigned char*)changed_dev, tag, fold_image);
    HIP_TRY(hipGetLastError());
    return IMS_OK;
}


// library events of RECORD / WAIT items (one process per GPU)
static int plan_event(int number, hipEvent_t* out)
{
    static std::unordered_map<int, hipEvent_t> evs;
    if (number < 0 || number > 65535) return set_err(IMS_ERR_ARG, "plan event number out of range");
    std::lock_guard<std::mutex> lock(g_state_mutex);
    auto it = evs.find(number);
    if (it == evs.end()) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        it = evs.emplace(number, e).first;
    }
    *out = it->second;
    return IMS_OK;
}

// objects of a table sorted by photon count (descending) that have more than `threshold` photons
static int32_t count_above(const int64_t* n_phot, int32_t n, int64_t threshold)
{
    int32_t lo = 0, hi = n;
    while (lo < hi) {
        const int32_t mid = (lo + hi) / 2;
        if (n_phot[mid] > threshold) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// IMS_PLAN_ROUNDS: the rounds of several chain classes, enqueued round by round so that every chain advances at the same pace
static int run_rounds(const ims_chain_t* chains, int32_t n_chains, const ims_sensor_t* sensor_dev, const ims_sensor_t* sensor_host,
                      unsigned char* changed_dev, void* const* streams, int32_t n_streams)
{
    if (!chains || n_chains < 0 || n_chains > IMS_MAX_CHAINS) return set_err(IMS_ERR_ARG, "chains is NULL or too many chains");
    int32_t max_rounds = 0;
    for (int32_t c = 0; c < n_chains; ++c) {
        const ims_chain_t& ch = chains[c];
        if (!ch.params || !ch.pool || !ch.pool_start || !ch.n_phot || !ch.tile_prefix || !ch.tile_prefix_host)
            return set_err(IMS_ERR_ARG, "chain: NULL pointer");
        if (ch.stream < 0 || ch.stream >= n_streams) return set_err(IMS_ERR_ARG, "chain stream index out of range");
        if (ch.nrecalc <= 0 || ch.n_rounds < 0 || ch.n_objects < 0 || ch.n_objects > ch.params->n_objects)
            return set_err(IMS_ERR_ARG, "chain: nrecalc / n_rounds / n_objects out of range");
        if (ch.n_edges < 0 || ch.n_edges > IMS_MAX_CHAIN_EDGES) return set_err(IMS_ERR_ARG, "chain: too many edges");
        for (int32_t k = 1; k < ch.n_objects; ++k)
            if (ch.n_phot[k] > ch.n_phot[k - 1]) return set_err(IMS_ERR_ARG, "chain: objects must be sorted by photon count, brightest first");
        if (ch.n_rounds > max_rounds) max_rounds = ch.n_rounds;
    }
    for (int32_t r = 0; r < max_rounds; ++r) {
        for (int32_t c = 0; c < n_chains; ++c) {
            const ims_chain_t& ch = chains[c];
            if (r >= ch.n_rounds) continue;
            void* st = streams[ch.stream];
            for (int32_t j = 1; j < ch.n_edges; ++j)
                if (ch.edges[j] == r) {
                    hipEvent_t e;
                    const int rc = plan_event(ch.ev_base + j, &e);
                    if (rc) return rc;
                    HIP_TRY(hipStreamWaitEvent((hipStream_t)st, e, 0));
                }
            const int32_t n_act = count_above(ch.n_phot, ch.n_objects, (int64_t)r * ch.nrecalc);
            if (n_act == 0) continue;
            const uint32_t tag = ch.use_tags ? (uint32_t)(r % 255 + 1) : 0u;       // marks the tiles this round's charge lands in
            // (the varying fields are set in a copy that is only read on the host: the launch takes the compact argument block)
            ims_render_params_t P = *ch.params;
            P.bf_tag = tag;
            P.bf_slot_shift = 0u;
            int rc = accumulate_round_compact(&P, ch.pool, ch.pool_start, r, ch.nrecalc, n_act, sensor_host ? sensor_host->num_vertices : 0, st);
            if (rc) return rc;
            const int32_t n_cont = count_above(ch.n_phot, ch.n_objects, (int64_t)(r + 1) * ch.nrecalc);
            if (n_cont > 0) {
                rc = ims_sensor_update_distortions(sensor_dev, sensor_host, ch.first_slot, n_cont, ch.tile_prefix,
                                                   ch.tile_prefix_host[n_cont], changed_dev, tag, st);
                if (rc) return rc;
            }
        }
    }
    return IMS_OK;
}

int ims_run_plan(const ims_plan_item_t* items, int64_t n_items, const ims_sensor_t* sensor_dev,
                 const ims_sensor_t* sensor_host, unsigned char* changed_dev, void* const* streams, int32_t n_streams)
{
    if (!items && n_items > 0) return set_err(IMS_ERR_ARG, "items is NULL");
    if (!streams || n_streams <= 0) return set_err(IMS_ERR_ARG, "streams is NULL");
    for (int64_t k = 0; k < n_items; ++k)
        if (items[k].stream < 0 || items[k].stream >= n_streams) return set_err(IMS_ERR_ARG, "plan item stream index out of range");
    for (int64_t k = 0; k < n_items; ++k) {
        const ims_plan_item_t& it = items[k];
        void* st = streams[it.stream];
        int rc = IMS_OK;
        switch (it.kind) {
        case IMS_PLAN_RENDER:     rc = ims_shoot_accumulate(it.params, st); break;
        case IMS_PLAN_SHOOT_POOL: rc = ims_shoot_ops_photons(it.params, it.aux, it.pool, st); break;
        case IMS_PLAN_ACC_POOL:   rc = ims_accumulate_segments(it.params, it.pool, it.aux, sensor_host ? sensor_host->num_vertices : 0, st); break;
        case IMS_PLAN_UPDATE:     rc = ims_sensor_update_distortions(sensor_dev, sensor_host, it.first_slot, it.n_slots, it.aux,
                                                                     it.n_tiles, changed_dev, it.tag, st); break;
        case IMS_PLAN_INIT:       rc = ims_sensor_init_boundaries(sensor_dev, sensor_host, it.first_slot, it.n_slots, it.aux, it.n_tiles, st); break;
        case IMS_PLAN_ROUNDS:     rc = run_rounds((const ims_chain_t*)it.aux2, it.n_slots, sensor_dev, sensor_host, changed_dev, streams,
                                                  n_streams); break;
        case IMS_PLAN_RECORD:
        case IMS_PLAN_WAIT: {
            hipEvent_t e;
            rc = plan_event(it.n_slots, &e);
            if (rc) return rc;
            if (it.kind == IMS_PLAN_RECORD) HIP_TRY(hipEventRecord(e, (hipStream_t)st));
            else HIP_TRY(hipStreamWaitEvent((hipStream_t)st, e, 0));
            break; }
        default: return set_err(IMS_ERR_ARG, "unknown plan item kind");
        }
        if (rc) return rc;
    }
    return IMS_OK;
}

// ---- LSST_Image launch planner (csrc/ims_plan.h builds the plan; here: binding to memory, upload, run) ----
}  // extern "C"
#include "ims_plan.h"

__global__ __launch_bounds__(256) void k_scatter_add(const int64_t* __restrict__ where, const double* __restrict__ src,
                                                     double* __restrict__ dst, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && src[i] != 0.0) unsafeAtomicAdd(dst + where[i], src[i]);
}

extern "C" {

int ims_plan_lsst_image(const ims_plan_input_t* in, void** plan_out, ims_plan_sizes_t* sizes)
{
    if (!in || !plan_out || !sizes) return set_err(IMS_ERR_ARG, "NULL argument");
    ims_planner::Plan* pl = new ims_planner::Plan();
    pl->in = *in;
    const int rc = ims_planner::build(*pl);
    if (rc) { delete pl; return rc; }
    // the plan keeps nothing of the caller's arrays
    pl->in.row = nullptr; pl->in.n_phot = nullptr; pl->in.stamp = nullptr; pl->in.faint = nullptr;
    *sizes = pl->sizes;
    *plan_out = pl;
    return IMS_OK;
}

int ims_plan_destroy(void* plan)
{
    ims_planner::Plan* pl = (ims_planner::Plan*)plan;
    if (!pl) return IMS_OK;
    for (hipEvent_t e : pl->events) (void)hipEventDestroy(e);
    for (hipEvent_t e : pl->d_events) (void)hipEventDestroy(e);
    delete pl;
    return IMS_OK;
}

int ims_plan_bind(void* plan, const ims_render_params_t* base, void* arena_host, void* arena_dev, void* rows_dev,
                  const ims_object_t* master_dev, double* pool_dev, double* realized_dev)
{
    using namespace ims_planner;
    Plan* pl = (Plan*)plan;
    if (!pl || !base || !arena_host || !arena_dev || !rows_dev || !master_dev) return set_err(IMS_ERR_ARG, "NULL argument");
    if (pl->sizes.pool_photons > 0 && !pool_dev) return set_err(IMS_ERR_ARG, "pool_dev is NULL");
    if (pl->sizes.realized_count > 0 && !realized_dev) return set_err(IMS_ERR_ARG, "realized_dev is NULL");
    pl->arena_host = (uint8_t*)arena_host; pl->arena_dev = (uint8_t*)arena_dev; pl->rows_dev = (uint8_t*)rows_dev;
    pl->master_dev = master_dev; pl->pool_dev = pool_dev; pl->realized_dev = realized_dev;
    std::memcpy(pl->arena_host, pl->arena.data(), pl->arena.size());
    auto dev = [&](int64_t off) { return off < 0 ? (uint8_t*)nullptr : pl->arena_dev + off; };
    for (Launch& L : pl->launches) {
        L.P = *base;
        L.P.objects = (const ims_object_t*)(pl->rows_dev + L.rows_off);
        L.P.n_objects = L.n;
        L.P.seg_prefix = (const int64_t*)dev(L.off_prefix);
        L.P.n_segments = L.n_segments;
        L.P.seg_object = (const int32_t*)dev(L.off_segobj);
        L.P.realized_flux = (L.realized_off >= 0) ? pl->realized_dev + L.realized_off : nullptr;
        L.P.bf_tag = 0; L.P.bf_slot_shift = 0;
        L.P.lazy_static = 0;                 // (set again below for the launches that are fused renders: nothing else may carry it)
    }
    for (Group& g : pl->groups) {
        std::memset(&g.pool, 0, sizeof(g.pool));
        g.pool.n = g.pool_photons;
        g.pool.converted = 1;
        const int64_t np = pl->sizes.pool_photons;
        g.pool.x = pool_dev; g.pool.y = pool_dev ? pool_dev + np : nullptr;
        g.pool.flux = pool_dev ? pool_dev + 2 * np : nullptr; g.pool.dxdz = pool_dev ? pool_dev + 3 * np : nullptr;
        g.chain_structs.assign(g.chains.size(), ims_chain_t());
        for (size_t c = 0; c < g.chains.size(); ++c) {
            const ChainDesc& ch = g.chains[c];
            ims_chain_t& cs = g.chain_structs[c];
            std::memset(&cs, 0, sizeof(cs));
            const Launch& L = pl->launches[ch.launch];
            cs.params = &L.P; cs.pool = &g.pool;

            cs.pool_start = (const int64_t*)dev(L.off_pool);
            cs.n_phot = ch.n_phot.data();
            cs.tile_prefix = (const int64_t*)dev(ch.off_tile_prefix);
            cs.tile_prefix_host = ch.tile_prefix_host.data();
            cs.n_objects = (int32_t)ch.n_phot.size(); cs.first_slot = ch.first_slot; cs.stream = ch.stream; cs.nrecalc = pl->in.nrecalc;
            cs.n_rounds = ch.n_rounds; cs.use_tags = pl->in.use_tags; cs.ev_base = ch.ev_base; cs.n_edges = (int32_t)ch.edges.size();
            for (size_t k = 0; k < ch.edges.size(); ++k) cs.edges[k] = ch.edges[k];
        }
        g.items.assign(g.steps.size(), ims_plan_item_t());
        for (size_t k = 0; k < g.steps.size(); ++k) {
            const Step& s = g.steps[k];
            ims_plan_item_t& it = g.items[k];
            std::memset(&it, 0, sizeof(it));
            it.kind = s.kind; it.stream = s.stream;
            switch (s.kind) {
            case IMS_PLAN_RENDER: pl->launches[s.launch].P.lazy_static = base->lazy_static; it.params = &pl->launches[s.launch].P; break;
            case IMS_PLAN_SHOOT_POOL:
                it.params = &pl->launches[s.launch].P; it.pool = &g.pool; it.aux = (const int64_t*)dev(pl->launches[s.launch].off_pool); break;
            case IMS_PLAN_INIT:
                it.first_slot = s.first_slot; it.n_slots = s.n_slots; it.aux = (const int64_t*)dev(s.off_tile_prefix); it.n_tiles = s.n_tiles; break;
            case IMS_PLAN_RECORD: it.n_slots = s.event; break;
            case IMS_PLAN_WAIT: it.n_slots = s.event; break;
            case IMS_PLAN_ROUNDS: it.n_slots = s.n_chains; it.aux2 = g.chain_structs.data() + s.chain_begin; break;
            default: return set_err(IMS_ERR_ARG, "planner: unexpected step kind");
            }
        }
    }
    pl->bound = true; pl->uploaded = false;
    return IMS_OK;
}

int ims_plan_upload(void* plan, void* stream)
{
    using namespace ims_planner;
    Plan* pl = (Plan*)plan;
    if (!pl || !pl->bound) return set_err(IMS_ERR_ARG, "plan is NULL or not bound");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(pl->arena_dev, pl->arena_host, pl->arena.size(), hipMemcpyHostToDevice, st));
    for (const Launch& L : pl->launches) {
        if (L.n <= 0) continue;
        const int rc = ims_gather_rows(pl->master_dev, (const int64_t*)(pl->arena_dev + L.off_index), (const int64_t*)(pl->arena_dev + L.off_first),
                                       (const int64_t*)(pl->arena_dev + L.off_count), (const int32_t*)(pl->arena_dev + L.off_bf), 0,
                                       (ims_object_t*)(pl->rows_dev + L.rows_off), L.n, stream);
        if (rc) return rc;
    }
    pl->uploaded = true;
    return IMS_OK;
}

static int plan_enqueue(ims_planner::Plan* pl, ims_sensor_t* sensor_dev, ims_sensor_t* sensor_host, ims_bf_slot_t* slots_dev,
                        unsigned char* changed_dev, void* main_stream, void* const* streams, int32_t n_streams, int32_t own_work_queued,
                        bool defer_top = false);

// (Two other forms of this call were built, measured in round 4 and removed in round 5: the whole enqueue captured into a
// hipGraph -- capture + instantiate + launch of a CCD's ~3 000-node plan cost the host 51 ms against 24 ms for enqueueing it, and a
// graph does not run on the caller's streams -- and one plan's rounds through the joint runner with its tile lists -- a fourth
// dependent launch per round on a chain of 185 rounds: C3 24.2 -> 25.2 ms without lists, 33 ms with them.)
int ims_plan_run(void* plan, ims_sensor_t* sensor_dev, ims_sensor_t* sensor_host, ims_bf_slot_t* slots_dev, unsigned char* changed_dev,
                 void* main_stream, void* const* streams, int32_t n_streams, int32_t own_work_queued)
{
    using namespace ims_planner;
    Plan* pl = (Plan*)plan;
    if (!pl || !pl->uploaded) return set_err(IMS_ERR_ARG, "plan is NULL or not uploaded");
    return plan_enqueue(pl, sensor_dev, sensor_host, slots_dev, changed_dev, main_stream, streams, n_streams, own_work_queued);
}

static int plan_enqueue(ims_planner::Plan* pl, ims_sensor_t* sensor_dev, ims_sensor_t* sensor_host, ims_bf_slot_t* slots_dev,
                        unsigned char* changed_dev, void* main_stream, void* const* streams, int32_t n_streams, int32_t own_work_queued,
                        bool defer_top)
{
    using namespace ims_planner;
    if (!streams || n_streams < 5) return set_err(IMS_ERR_ARG, "streams: five plan streams by role are required");
    bool any_slots = false;
    for (const Group& g : pl->groups) any_slots = any_slots || g.n_slots > 0;
    if (any_slots && (!sensor_dev || !sensor_host || !slots_dev || !changed_dev || !sensor_host->bf_slots))
        return set_err(IMS_ERR_ARG, "sensor / slot table / changed is NULL");
    // the rounds of the top chain can be left to a joint run when the plan is ONE group of regions (a later group rewrites the
    // slot table behind everything queued) in the form the joint kernels take: 4 vertices per edge, qdist 3, regions in place
    pl->deferred = false;
    bool defer = defer_top && pl->groups.size() == 1 && pl->groups[0].n_slots > 0 && !pl->groups[0].chain_structs.empty() &&
                 sensor_host->num_vertices == IT_NV && sensor_host->qdist == 3;
    if (defer)
        for (const ims_chain_t& cs : pl->groups[0].chain_structs)
            defer = defer && cs.first_slot > 0;
    // distinct streams
    std::vector<hipStream_t> uniq;
    for (int k = 0; k < n_streams; ++k)
        if (std::find(uniq.begin(), uniq.end(), (hipStream_t)streams[k]) == uniq.end()) uniq.push_back((hipStream_t)streams[k]);
    const size_t need = 2 + 2 * uniq.size();
    while (pl->events.size() < need) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        pl->events.push_back(e);
    }
    hipStream_t main = (hipStream_t)main_stream, chain = (hipStream_t)streams[ROLE_CHAIN];
    if (pl->realized_dev && pl->sizes.realized_count > 0)
        HIP_TRY(hipMemsetAsync(pl->realized_dev, 0, (size_t)pl->sizes.realized_count * sizeof(double), main));
    HIP_TRY(hipEventRecord(pl->events[0], main));
    for (hipStream_t s : uniq) HIP_TRY(hipStreamWaitEvent(s, pl->events[0], 0));
    bool queued = own_work_queued != 0;
    for (size_t gi = 0; gi < pl->groups.size(); ++gi) {
        Group& g = pl->groups[gi];
        if (g.n_slots > 0) {
            const int n0 = pl->in.n_static_slots;
            if (n0 + g.n_slots > pl->in.slot_capacity) return set_err(IMS_ERR_ARG, "too many brighter-fatter slots");
            // The slot table is rewritten by a copy on the chain stream: behind everything this renderer has queued on every
            // stream (a later group's table must not change under the launches of the one before) and ahead of what follows
            if (queued)
                for (size_t k = 0; k < uniq.size(); ++k)
                    if (uniq[k] != chain) {
                        HIP_TRY(hipEventRecord(pl->events[2 + k], uniq[k]));
                        HIP_TRY(hipStreamWaitEvent(chain, pl->events[2 + k], 0));
                    }
            HIP_TRY(hipMemcpyAsync(slots_dev + n0, pl->arena_host + g.off_slots, (size_t)g.n_slots * sizeof(ims_bf_slot_t),
                                   hipMemcpyHostToDevice, chain));
            HIP_TRY(hipMemcpyAsync(&sensor_dev->n_bf_slots, pl->arena_host + pl->off_nslots[gi], sizeof(int32_t), hipMemcpyHostToDevice, chain));
            HIP_TRY(hipEventRecord(pl->events[1], chain));
            for (hipStream_t s : uniq) if (s != chain) HIP_TRY(hipStreamWaitEvent(s, pl->events[1], 0));
            std::memcpy((ims_bf_slot_t*)(uintptr_t)sensor_host->bf_slots + n0, pl->arena_host + g.off_slots, (size_t)g.n_slots * sizeof(ims_bf_slot_t));
            sensor_host->n_bf_slots = n0 + g.n_slots;
        }
        if (defer) {
            // without the rounds item: the rounds of ALL chain classes are left to joint runs
            std::vector<ims_plan_item_t> items;
            for (const ims_plan_item_t& it : g.items)
                if (it.kind != IMS_PLAN_ROUNDS) items.push_back(it);
            const int rc = ims_run_plan(items.data(), (int64_t)items.size(), sensor_dev, sensor_host, changed_dev, streams, n_streams);
            if (rc) return rc;
        } else {
            const int rc = ims_run_plan(g.items.data(), (int64_t)g.items.size(), sensor_dev, sensor_host, changed_dev, streams, n_streams);
            if (rc) return rc;
        }
        queued = true;
    }
    if (defer_top) {
        // no join yet (ims_plan_join): where this plan's work ends on every stream is recorded NOW -- on streams shared by role the
        // next CCD's work follows, and a join must not wait for that
        for (size_t k = 0; k < uniq.size(); ++k) HIP_TRY(hipEventRecord(pl->events[2 + uniq.size() + k], uniq[k]));
        pl->unjoined = (int)uniq.size();
        pl->deferred = defer;
        pl->left = defer ? ((1u << pl->groups[0].chain_structs.size()) - 1u) : 0u;
        pl->d_done_used = 0;
        pl->d_sensor_dev = sensor_dev; pl->d_sensor_host = sensor_host; pl->d_changed = changed_dev; pl->d_main = main_stream;
        pl->d_streams.assign(streams, streams + n_streams);
        return IMS_OK;
    }
    for (size_t k = 0; k < uniq.size(); ++k) {
        HIP_TRY(hipEventRecord(pl->events[2 + uniq.size() + k], uniq[k]));
        HIP_TRY(hipStreamWaitEvent(main, pl->events[2 + uniq.size() + k], 0));
    }
    return IMS_OK;
}

int ims_plan_join(void* plan, void* stream)
{
    using namespace ims_planner;
    Plan* pl = (Plan*)plan;
    if (!pl) return set_err(IMS_ERR_ARG, "plan is NULL");
    if (pl->deferred) return set_err(IMS_ERR_ARG, "ims_plan_join: the rounds left by ims_plan_run_deferred have not been run (ims_plans_run_joint)");
    hipStream_t st = (hipStream_t)stream;
    for (int k = 0; k < pl->unjoined; ++k) HIP_TRY(hipStreamWaitEvent(st, pl->events[2 + (size_t)pl->unjoined + k], 0));
    for (size_t k = 0; k < pl->d_done_used; ++k) HIP_TRY(hipStreamWaitEvent(st, pl->d_events[4 + k], 0));
    pl->unjoined = 0; pl->d_done_used = 0;
    return IMS_OK;
}

int ims_plan_run_deferred(void* plan, ims_sensor_t* sensor_dev, ims_sensor_t* sensor_host, ims_bf_slot_t* slots_dev, unsigned char* changed_dev,
                          void* main_stream, void* const* streams, int32_t n_streams, int32_t own_work_queued, int32_t* deferred)
{
    using namespace ims_planner;
    Plan* pl = (Plan*)plan;
    if (!pl || !pl->uploaded) return set_err(IMS_ERR_ARG, "plan is NULL or not uploaded");
    if (!deferred) return set_err(IMS_ERR_ARG, "deferred is NULL");
    const int rc = plan_enqueue(pl, sensor_dev, sensor_host, slots_dev, changed_dev, main_stream, streams, n_streams, own_work_queued, true);
    *deferred = (rc == IMS_OK && pl->deferred) ? (int32_t)pl->groups[0].chain_structs.size() : 0;
    return rc;
}

int ims_plans_run_joint(void* const* plans, int32_t n_plans, void* joint_stream, int32_t first_chain, int32_t n_chains)
{
    using namespace ims_planner;
    if (n_plans < 0 || (n_plans > 0 && !plans)) return set_err(IMS_ERR_ARG, "plans is NULL");
    if (first_chain < 0 || n_chains < 1) return set_err(IMS_ERR_ARG, "joint run: chain range");
    struct Act { Plan* pl; const ims_chain_t* ch; int chain; hipEvent_t done; };
    std::vector<Act> act;
    hipStream_t js = (hipStream_t)joint_stream;
    // what an error return has to undo: the done events handed out below (a plan's join must not wait for a record that
    // never comes) and the ring entry held while this call enqueues (its event is recorded all the same: launches that
    // did go out read the entry's tables)
    struct Undo {
        std::vector<Plan*> counted; bool* busy = nullptr; hipEvent_t table_event = nullptr; hipStream_t js = nullptr; bool ok = false;
        ~Undo() {
            if (!ok) for (Plan* p : counted) if (p->d_done_used > 0) --p->d_done_used;
            if (busy) {
                if (!ok && table_event) (void)hipEventRecord(table_event, js);
                std::lock_guard<std::mutex> lock(g_state_mutex);
                *busy = false;
            }
        }
    } undo;
    undo.js = js;
    for (int32_t k = 0; k < n_plans; ++k) {
        Plan* pl = (Plan*)plans[k];
        if (!pl) return set_err(IMS_ERR_ARG, "plans: NULL entry");
        if (!pl->deferred) continue;                       // ran whole (no bright object, several groups): nothing left to do
        const int nc = (int)pl->groups[0].chain_structs.size();
        bool any = false;
        for (int c = first_chain; c < first_chain + n_chains && c < nc; ++c)
            if (pl->left & (1u << c)) { act.push_back({ pl, &pl->groups[0].chain_structs[c], c, nullptr }); any = true; }
        if (!any) continue;
        // own events: four "chain stream ready", then one per joint run since the last join
        while (pl->d_events.size() < 4 + pl->d_done_used + 1) {
            hipEvent_t e;
            HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            pl->d_events.push_back(e);
        }
        const hipEvent_t done = pl->d_events[4 + pl->d_done_used++];
        undo.counted.push_back(pl);
        for (Act& a : act) if (a.pl == pl) a.done = done;
    }
    if (act.empty()) { undo.ok = true; return IMS_OK; }
    if ((int)act.size() > IMS_JOINT_MAX) return set_err(IMS_ERR_ARG, "at most 64 chains per joint run");
    const int32_t nrecalc = act[0].ch->nrecalc, use_tags = act[0].ch->use_tags;
    int32_t max_rounds = 0;
    for (const Act& a : act) {
        const ims_chain_t& ch = *a.ch;
        if (ch.nrecalc != nrecalc || ch.use_tags != use_tags) return set_err(IMS_ERR_ARG, "joint run: the plans differ in nrecalc / tile tags");
        if (!ch.params || !ch.pool || !ch.pool->converted || !ch.pool_start || !ch.params->objects || !ch.params->image)
            return set_err(IMS_ERR_ARG, "joint run: chain without table / pool / image");
        if (ch.n_rounds > max_rounds) max_rounds = ch.n_rounds;
        // the joint stream takes over behind the chain's own stream (the regions' initial state, the first pool slice)
        // (behind the END OF THE CCD'S OWN FRONT on that stream -- the event ims_plan_run_deferred recorded there -- not behind
        // whatever the stream holds by now: the joint run may be enqueued from a second host thread while the fronts of the
        // next batch of CCDs are already being queued on the shared streams, and the chain must not wait for those)
        hipStream_t cs = (hipStream_t)a.pl->d_streams[ch.stream];
        if (cs != js) {
            std::vector<hipStream_t> uniq;
            for (void* st : a.pl->d_streams)
                if (std::find(uniq.begin(), uniq.end(), (hipStream_t)st) == uniq.end()) uniq.push_back((hipStream_t)st);
            const size_t k = (size_t)(std::find(uniq.begin(), uniq.end(), cs) - uniq.begin());
            if ((int)uniq.size() != a.pl->unjoined || k >= uniq.size())
                return set_err(IMS_ERR_ARG, "joint run: the plan was not enqueued by ims_plan_run_deferred on these streams");
            HIP_TRY(hipStreamWaitEvent(js, a.pl->events[2 + uniq.size() + k], 0));
        }
    }
    const int32_t segs = (nrecalc + 255) / 256;
    const long long dpp_max_tiles = g_tune.upd_dpp_max;
    // the per-chain argument blocks in device memory, one table per joint run out of a ring (a table is read until the run's
    // last round: the ring's event says when)
    // active-tile lists (k_build_active_j): on for rounds of more than list_min_tiles tiles; joint_lists = 0: the full sweeps
    const bool lists_on = g_tune.joint_lists != 0;
    const long long list_min_tiles = g_tune.joint_list_min;
    const double list_fraction = g_tune.active_fraction;
    int64_t tiles_max = 0;                                   // round 0 has them all
    for (const Act& a : act) {
        const int32_t n_cont = count_above(a.ch->n_phot, a.ch->n_objects, (int64_t)nrecalc);
        tiles_max += n_cont > 0 ? a.ch->tile_prefix_host[n_cont] : 0;
    }
    const bool lists = lists_on && tiles_max > list_min_tiles;
    struct Ring { JointTables* dev; hipEvent_t free_after; bool used; bool busy; unsigned long long* upd; unsigned long long* ref; int* count; int64_t cap;
                  JointRound* rounds_dev; JointRound* rounds_pin; int64_t rounds_cap; unsigned int* words; };
    // one ring per DEVICE (its tables live in that device's memory: one process per GPU is the rule, but a process that holds
    // two devices must not hand the second one the first one's tables)
    static std::map<int, std::pair<std::vector<Ring>, size_t>> rings;
    int device_now = 0;
    HIP_TRY(hipGetDevice(&device_now));
    JointTables* tables_dev = nullptr;
    JointRound* rounds_dev = nullptr;
    JointRound* rounds_pin = nullptr;
    hipEvent_t table_event = nullptr;
    JointLists Ls{ nullptr, nullptr, nullptr };
    unsigned int* words_dev = nullptr;
    // search-side lists (TileLister): the pixel search appends the tiles, no builder launch
    const bool search_lists = lists && g_tune.joint_search_lists != 0;
    {
        std::lock_guard<std::mutex> lock(g_state_mutex);
        std::vector<Ring>& ring = rings[device_now].first;
        size_t& ring_next = rings[device_now].second;
        if (ring.empty()) {
            JointTables* block = nullptr;
            int* counts = nullptr;
            HIP_TRY(hipMalloc((void**)&block, 64 * sizeof(JointTables)));
            HIP_TRY(hipMalloc((void**)&counts, 64 * 4 * sizeof(int)));
            for (int k = 0; k < 64; ++k) {
                Ring r{ block + k, nullptr, false, false, nullptr, nullptr, counts + 4 * k, 0, nullptr, nullptr, 0, nullptr };
                HIP_TRY(hipEventCreateWithFlags(&r.free_after, hipEventDisableTiming));
                ring.push_back(r);
            }
        }
        // an entry another thread is still enqueueing with (its event not yet recorded) is passed over
        size_t tries = 0;
        while (ring[ring_next % ring.size()].busy && tries++ < ring.size()) ++ring_next;
        Ring& r = ring[ring_next++ % ring.size()];
        if (r.busy) return set_err(IMS_ERR_ARG, "joint run: all 64 table entries are being enqueued with");
        if (r.used) HIP_TRY(hipEventSynchronize(r.free_after));
        r.used = true; r.busy = true;
        undo.busy = &r.busy; undo.table_event = r.free_after;
        if (lists && r.cap < tiles_max) {
            // (a growth is a device-wide synchronisation: the capacity is kept and rounded up generously)
            if (r.upd) { HIP_TRY(hipFree(r.upd)); HIP_TRY(hipFree(r.ref)); HIP_TRY(hipFree(r.words)); }
            r.cap = tiles_max + tiles_max / 2 + 65536;
            HIP_TRY(hipMalloc((void**)&r.upd, (size_t)r.cap * sizeof(unsigned long long)));
            HIP_TRY(hipMalloc((void**)&r.ref, (size_t)r.cap * sizeof(unsigned long long)));
            HIP_TRY(hipMalloc((void**)&r.words, (size_t)r.cap * sizeof(unsigned int)));
        }
        if (r.rounds_cap < max_rounds) {
            // the round tables of the run: a page-locked staging buffer and its device copy (grown rarely: capacity kept)
            if (r.rounds_dev) { HIP_TRY(hipFree(r.rounds_dev)); HIP_TRY(hipHostFree(r.rounds_pin)); }
            r.rounds_cap = max_rounds + max_rounds / 2 + 64;
            HIP_TRY(hipMalloc((void**)&r.rounds_dev, (size_t)r.rounds_cap * sizeof(JointRound)));
            HIP_TRY(hipHostMalloc((void**)&r.rounds_pin, (size_t)r.rounds_cap * sizeof(JointRound), hipHostMallocDefault));
        }
        tables_dev = r.dev; table_event = r.free_after;
        Ls.upd = r.upd; Ls.ref = r.ref; Ls.count = r.count;
        words_dev = r.words;
        rounds_dev = r.rounds_dev; rounds_pin = r.rounds_pin;
    }
    if (lists) HIP_TRY(hipMemsetAsync(Ls.count, 0, 4 * sizeof(int), js));
    if (search_lists) HIP_TRY(hipMemsetAsync(words_dev, 0, (size_t)tiles_max * sizeof(unsigned int), js));
    // a chain's tile words: behind those of the chains before it (round 0 lists from all the regions that go on)
    std::vector<int64_t> word_off(act.size() + 1, 0);
    for (size_t k = 0; k < act.size(); ++k) {
        const int32_t n_cont = count_above(act[k].ch->n_phot, act[k].ch->n_objects, (int64_t)nrecalc);
        word_off[k + 1] = word_off[k] + (n_cont > 0 ? act[k].ch->tile_prefix_host[n_cont] : 0);
    }
    bool dpp_ok = true;
    for (size_t k0 = 0; k0 < act.size(); k0 += JOINT_CHUNK) {
        JointChunk Ck;
        std::memset(&Ck, 0, sizeof(Ck));
        for (size_t k = 0; k < (size_t)JOINT_CHUNK; ++k) {
            const Act& a = act[k0 + k < act.size() ? k0 + k : 0];         // entries beyond the last repeat chain 0 (never selected: zero width)
            const ims_chain_t& ch = *a.ch;
            const ims_render_params_t& P = *ch.params;
            RoundArgs& ra = Ck.a[k];
            ra.objects = P.objects; ra.sensor = P.sensor; ra.image = P.image; ra.realized_flux = P.realized_flux;
            ra.px = ch.pool->x; ra.py = ch.pool->y; ra.pflux = ch.pool->flux; ra.pz = ch.pool->dxdz;
            ra.nx = P.nx; ra.ny = P.ny; ra.xmin = P.xmin; ra.ymin = P.ymin;
            ra.bf_tag = 0; ra.bf_slot_shift = 0; ra.track_static_delta = P.track_static_delta; ra.pad = 0;
            Ck.pool_start[k] = ch.pool_start;
            Ck.sp[k] = a.pl->d_sensor_dev; Ck.tile_prefix[k] = ch.tile_prefix; Ck.changed[k] = a.pl->d_changed;
            Ck.dl[k] = a.pl->d_sensor_host->bf_dl; Ck.first_slot[k] = ch.first_slot;
            Ck.words[k] = (search_lists && k0 + k < act.size()) ? words_dev + word_off[k0 + k] : nullptr;
            dpp_ok = dpp_ok && a.pl->d_sensor_host->bf_dl != nullptr;
        }
        hipLaunchKernelGGL(k_store_joint_tables, dim3(1), dim3(64), 0, js, Ck, tables_dev, (int)k0);
        HIP_TRY(hipGetLastError());
    }
    // every round's tables (workgroup / tile ends per chain) in one pass on the host, ONE copy to the device; per round the
    // launches then take a pointer
    struct RoundTotals { int64_t wgs, tiles, build_wgs; };
    std::vector<RoundTotals> totals((size_t)max_rounds);
    for (int32_t r = 0; r < max_rounds; ++r) {
        JointRound& R = rounds_pin[r];
        int64_t wgs = 0, tiles = 0, build_wgs = 0;
        for (int k = 0; k < IMS_JOINT_MAX; ++k) {
            int32_t n_act = 0, n_cont = 0;
            int64_t tk = 0;
            if (k < (int)act.size() && r < act[k].ch->n_rounds) {
                const ims_chain_t& ch = *act[k].ch;
                n_act = count_above(ch.n_phot, ch.n_objects, (int64_t)r * nrecalc);
                n_cont = count_above(ch.n_phot, ch.n_objects, (int64_t)(r + 1) * nrecalc);
                tk = n_cont > 0 ? ch.tile_prefix_host[n_cont] : 0;
            }
            tiles += tk;
            build_wgs += (tk + 255) / 256;
            wgs += (int64_t)n_act * segs;
            if (wgs > 0x7fffffffLL || tiles > 0x7fffffffLL) return set_err(IMS_ERR_ARG, "joint run: too many workgroups for one round");
            R.ea.v[k] = (int32_t)wgs; R.eu.v[k] = (int32_t)tiles; R.ns.v[k] = n_cont > 0 ? n_cont : 1;
            R.ewg.v[k] = (int32_t)build_wgs; R.tof.v[k] = (int32_t)tk;
        }
        totals[(size_t)r] = { wgs, tiles, build_wgs };
    }
    if (max_rounds > 0)
        HIP_TRY(hipMemcpyAsync(rounds_dev, rounds_pin, (size_t)max_rounds * sizeof(JointRound), hipMemcpyHostToDevice, js));
    // the round after which a CCD's chains of this run are through (its longest chain)
    std::vector<std::pair<int32_t, hipEvent_t>> done_at;
    for (size_t k = 0; k < act.size(); ++k) {
        bool first_of_plan = true;
        int32_t last = 0;
        for (size_t j = 0; j < act.size(); ++j)
            if (act[j].pl == act[k].pl) { if (j < k) first_of_plan = false; if (act[j].ch->n_rounds > last) last = act[j].ch->n_rounds; }
        if (first_of_plan) done_at.push_back({ last - 1, act[k].done });
    }
    for (int32_t r = 0; r < max_rounds; ++r) {
        for (const Act& a : act) {
            const ims_chain_t& ch = *a.ch;
            if (r >= ch.n_rounds) continue;
            for (int32_t j = 1; j < ch.n_edges; ++j)
                if (ch.edges[j] == r) {
                    hipEvent_t e;
                    const int rc = plan_event(ch.ev_base + j, &e);
                    if (rc) return rc;
                    HIP_TRY(hipStreamWaitEvent(js, e, 0));
                }
        }
        const uint32_t tag = (use_tags || lists) ? (uint32_t)(r % 255 + 1) : 0u;
        const int64_t wgs = totals[(size_t)r].wgs, tiles = totals[(size_t)r].tiles, build_wgs = totals[(size_t)r].build_wgs;
        const JointRound* R = rounds_dev + r;
        const bool list_round = tiles > 0 && lists && tiles > list_min_tiles;
        const int parity_r = r & 1;
        if (wgs > 0) {
            LaunchTimer tm(js, 4);
            const bool sl = search_lists && list_round;
            hipLaunchKernelGGL((k_accumulate_round_j<4, 256>), dim3((unsigned)wgs), dim3(256), 0, js, (const JointAcc*)&tables_dev->acc, R, tag,
                               (int64_t)r * nrecalc, nrecalc, segs, sl ? Ls.upd : (unsigned long long*)nullptr,
                               sl ? Ls.count + 2 * parity_r : (int*)nullptr, (unsigned int)(r + 1));
        }
        if (list_round) {
            // marks -> lists -> the update and the refresh over the listed tiles (a launch of a fraction of the tiles walks them)
            const JointUpd* U = &tables_dev->upd;
            const int parity = r & 1;
            int64_t grid = (int64_t)((double)tiles * list_fraction);
            if (grid < 256 && list_fraction >= 0.05) grid = 256;
            if (grid < 1) grid = 1;
            if (grid > tiles) grid = tiles;
            if (!search_lists)
                hipLaunchKernelGGL(k_build_active_j, dim3((unsigned)build_wgs), dim3(256), 0, js, U, R, tag, Ls, parity, (int)g_tune.joint_fine_marks);
            hipLaunchKernelGGL((k_update_list_j<4, false>), dim3((unsigned)grid), dim3(256), 0, js, U, Ls, parity, tag, search_lists ? 1 : 0);
            hipLaunchKernelGGL(k_refresh_list_j<4>, dim3((unsigned)grid), dim3(256), 0, js, U, Ls, parity, tag, search_lists ? 1 : 0);
        } else if (tiles > 0) {
            const JointUpd* U = &tables_dev->upd;
            const bool dpp = dpp_ok && g_tune.upd_dpp && tiles <= dpp_max_tiles;
            if (dpp) hipLaunchKernelGGL((k_update_distortions_q3_j<4, true>), dim3((unsigned)tiles), dim3(256), 0, js, U, R, tag);
            else hipLaunchKernelGGL((k_update_distortions_q3_j<4, false>), dim3((unsigned)tiles), dim3(256), 0, js, U, R, tag);
            hipLaunchKernelGGL(k_refresh_changed_j<4>, dim3((unsigned)tiles), dim3(256), 0, js, U, R, tag);
        }
#ifdef IMS_EXTRA_LAUNCHES
        for (int xl = 0; xl < IMS_EXTRA_LAUNCHES; ++xl) hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, js, (int*)nullptr);
#endif
        // a CCD's chains of this run are through with the longest of them
        for (const auto& d : done_at)
            if (d.first == r) HIP_TRY(hipEventRecord(d.second, js));
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(table_event, js));
    undo.ok = true;
    for (const Act& a : act) {
        a.pl->left &= ~(1u << a.chain);
        if (a.pl->left == 0u) a.pl->deferred = false;
    }
    return IMS_OK;
}

int ims_plan_add_realized(void* plan, double* out_dev, void* stream)
{
    using namespace ims_planner;
    Plan* pl = (Plan*)plan;
    if (!pl || !pl->uploaded) return set_err(IMS_ERR_ARG, "plan is NULL or not uploaded");
    const int64_t n = pl->sizes.realized_count;
    if (n <= 0) return IMS_OK;
    if (!out_dev) return set_err(IMS_ERR_ARG, "out_dev is NULL");
    hipLaunchKernelGGL(k_scatter_add, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const int64_t*)(pl->arena_dev + pl->off_realized_where), (const double*)pl->realized_dev, out_dev, n);
    HIP_TRY(hipGetLastError());
    return IMS_OK;
}

// ---- the inverse transforms of the FFT branch (hipFFT, plans cached) and the exchanges between GPUs (RCCL) ----
}  // extern "C"
#include "ims_libs.h"

__global__ __launch_bounds__(256) void k_scale(double* __restrict__ v, int64_t n, double s)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) v[i] = v[i] * s;
}

__global__ __launch_bounds__(256) void k_f64_to_i32(const double* __restrict__ src, int32_t* __restrict__ dst, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = (int32_t)src[i];
}

__global__ __launch_bounds__(256) void k_i32_to_f64(const int32_t* __restrict__ src, double* __restrict__ dst, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = (double)src[i];
}

// not an integer count below 2^31 / world: the int32 exchange would not be exact for it
__global__ __launch_bounds__(256) void k_count_inexact(const double* __restrict__ src, int64_t n, double limit, unsigned long long* __restrict__ bad)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    unsigned long long c = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const double v = src[i];
        if (!(v >= 0.0 && v < limit && v == floor(v))) ++c;
    }
    if (c) atomicAdd(bad, c);
}

extern "C" {

static int fft_err(hipfftResult r, const char* what)
{
    if (r == HIPFFT_SUCCESS) return IMS_OK;
    char buf[256];
    snprintf(buf, sizeof(buf), "%s: hipfft error %d", what, (int)r);
    return set_err(IMS_ERR_HIP, buf);
}

// Plans are kept per (size, stream) and transform ONE stamp: a batch is a loop over it.  The FIRST plan of a process costs seconds
// (measured in round 5 on a fresh box: 2.4 s for the first call -- hipFFT / rocFFT start up and compile their kernels at run time
// --, 0.5 s for the next new size, nothing after that; with a plan per size AND batch count every new combination paid again: 2.8
// of the 3.4 s a fresh process spent on its first 48 CCDs), a CCD holds one to three FFT-drawn objects of a handful of sizes, and
// a transform of 1024^2 or more fills the device by itself, so nothing is lost by not batching.  Per STREAM because a plan owns
// its work buffer (the transposes of the large sizes go through it): the same plan executing on two streams at once -- the FFT
// objects of two CCDs of a focal plane on the two top-chain streams -- would share it.  Small transforms (below 1024^2:
// launch-bound) keep batched plans.  ims_fft_warm makes the plans ahead of time (from another host thread, while the host
// still reads catalogs).
static std::mutex g_fft_mutex;
static std::map<std::tuple<int32_t, int64_t, void*>, hipfftHandle> g_fft_plans;

// the plan of (nfft, per_plan, stream): looked up under the lock, MADE outside it (seconds for a size's first plan: plans of
// different sizes are made side by side by ims_fft_warm's callers), entered under the lock again
static int fft_plan_get(const ims_libs::Fft* F, int32_t nfft, int64_t per_plan, void* stream, hipfftHandle* out)
{
    const std::tuple<int32_t, int64_t, void*> key(nfft, per_plan, stream);
    {
        std::lock_guard<std::mutex> lock(g_fft_mutex);
        auto it = g_fft_plans.find(key);
        if (it != g_fft_plans.end()) { *out = it->second; return IMS_OK; }
    }
    int n[2] = { nfft, nfft };
    int inembed[2] = { nfft, nfft / 2 + 1 }, onembed[2] = { nfft, nfft };
    hipfftHandle p;
    int rc = fft_err(F->plan_many(&p, 2, n, inembed, 1, nfft * (nfft / 2 + 1), onembed, 1, nfft * nfft, HIPFFT_Z2D, (int)per_plan),
                     "hipfftPlanMany");
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(g_fft_mutex);
    auto ins = g_fft_plans.emplace(key, p);
    if (!ins.second) (void)F->destroy(p);            // another thread was faster
    *out = ins.first->second;
    return IMS_OK;
}

int ims_fft_warm(int32_t nfft, void* stream)
{
    if (nfft < 2 || (nfft & 1)) return set_err(IMS_ERR_ARG, "nfft must be even and >= 2");
    const ims_libs::Fft* F = ims_libs::fft();
    if (!F) return set_err(IMS_ERR_UNSUPPORTED, "hipFFT is not loadable (libhipfft.so; IMS_HIPFFT_LIB names a file)");
    hipfftHandle plan;
    return fft_plan_get(F, nfft, 1, stream, &plan);
}

static int fft_inverse_impl(double* kbuf_dev, double* rbuf_dev, int32_t nfft, int64_t batch, void* stream, bool normalise)
{
    if (batch <= 0) return IMS_OK;
    if (!kbuf_dev || !rbuf_dev) return set_err(IMS_ERR_ARG, "NULL buffer");
    if (nfft < 2 || (nfft & 1) || batch > 0x7fffffffLL) return set_err(IMS_ERR_ARG, "nfft must be even and >= 2");
    const ims_libs::Fft* F = ims_libs::fft();
    if (!F) return set_err(IMS_ERR_UNSUPPORTED, "hipFFT is not loadable (libhipfft.so; IMS_HIPFFT_LIB names a file)");
    int64_t per_plan = nfft >= 1024 ? 1 : batch;
    {
        if (per_plan != batch) {
            // a (size, batch) pair that comes again -- a replayed step, the same counts on many CCDs -- gets its batched plan (a few
            // milliseconds once the size's kernels exist: only a size's FIRST plan is expensive)
            static std::map<std::tuple<int32_t, int64_t, void*>, int> asked;
            std::lock_guard<std::mutex> lock(g_fft_mutex);
            if (++asked[std::make_tuple(nfft, batch, stream)] >= 2) per_plan = batch;
        }
        hipfftHandle plan;
        int rc = fft_plan_get(F, nfft, per_plan, stream, &plan);
        if (rc) return rc;
        // (set-stream + execute of one plan must not interleave between host threads: a plan carries its stream)
        static std::mutex exec_mutex;
        std::lock_guard<std::mutex> lock(exec_mutex);
        rc = fft_err(F->set_stream(plan, (hipStream_t)stream), "hipfftSetStream");
        if (rc) return rc;
        for (int64_t b = 0; b < batch; b += per_plan) {
            rc = fft_err(F->exec_z2d(plan, (hipfftDoubleComplex*)kbuf_dev + b * (int64_t)nfft * (nfft / 2 + 1),
                                     (hipfftDoubleReal*)rbuf_dev + b * (int64_t)nfft * nfft), "hipfftExecZ2D");
            if (rc) return rc;
        }
    }
    if (!normalise) return IMS_OK;          // the readers multiply (ims_fft_params_t.rbuf_raw)
    // numpy / GalSim normalisation of the inverse ("backward": 1 / N^2): exact for the even sizes used (powers of two)
    const int64_t total = batch * (int64_t)nfft * nfft;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_scale, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, rbuf_dev, total, 1.0 / ((double)nfft * (double)nfft));
    HIP_TRY(hipGetLastError());
    return IMS_OK;
}

int ims_fft_inverse(double* kbuf_dev, double* rbuf_dev, int32_t nfft, int64_t batch, void* stream)
{
    return fft_inverse_impl(kbuf_dev, rbuf_dev, nfft, batch, stream, true);
}

int ims_fft_inverse_raw(double* kbuf_dev, double* rbuf_dev, int32_t nfft, int64_t batch, void* stream)
{
    return fft_inverse_impl(kbuf_dev, rbuf_dev, nfft, batch, stream, false);
}

// ---- exchanges between the GPUs of a node (RCCL over xGMI) ----
static int nccl_err(ncclResult_t r, const char* what)
{
    if (r == ncclSuccess) return IMS_OK;
    const ims_libs::Rccl* R = ims_libs::rccl();
    char buf[384];
    snprintf(buf, sizeof(buf), "%s: %s", what, R ? R->error_string(r) : "rccl error");
    return set_err(IMS_ERR_HIP, buf);
}

struct ImsComm { ncclComm_t comm; int32_t rank, world; };

int ims_comm_unique_id(void* id128)
{
    if (!id128) return set_err(IMS_ERR_ARG, "id128 is NULL");
    const ims_libs::Rccl* R = ims_libs::rccl();
    if (!R) return set_err(IMS_ERR_UNSUPPORTED, "RCCL is not loadable (librccl.so; IMS_RCCL_LIB names a file)");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId");
    return nccl_err(R->get_unique_id((ncclUniqueId*)id128), "ncclGetUniqueId");
}

int ims_comm_init(const void* id128, int32_t rank, int32_t world, void** comm_out)
{
    if (!id128 || !comm_out || world < 1 || rank < 0 || rank >= world) return set_err(IMS_ERR_ARG, "id / comm_out / rank / world");
    const ims_libs::Rccl* R = ims_libs::rccl();
    if (!R) return set_err(IMS_ERR_UNSUPPORTED, "RCCL is not loadable (librccl.so; IMS_RCCL_LIB names a file)");
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    const int rc = nccl_err(R->comm_init_rank(&c, world, id, rank), "ncclCommInitRank");
    if (rc) return rc;
    *comm_out = new ImsComm{ c, rank, world };
    return IMS_OK;
}

int ims_comm_destroy(void* comm)
{
    if (!comm) return IMS_OK;
    const ims_libs::Rccl* R = ims_libs::rccl();
    if (!R) return set_err(IMS_ERR_UNSUPPORTED, "RCCL is not loadable");
    ImsComm* ic = (ImsComm*)comm;
    const int rc = nccl_err(R->comm_destroy(ic->comm), "ncclCommDestroy");
    delete ic;
    return rc;
}

static unsigned exchange_grid(int64_t n)
{
    int64_t b = (n + 255) / 256;
    return (unsigned)(b > 16384 ? 16384 : (b < 1 ? 1 : b));
}

int ims_count_inexact(const double* image_dev, int64_t n, int32_t world, unsigned long long* bad_dev, void* stream)
{
    if (!image_dev || !bad_dev || n < 0 || world < 1) return set_err(IMS_ERR_ARG, "NULL / negative argument");
    if (n == 0) return IMS_OK;
    hipLaunchKernelGGL(k_count_inexact, dim3(exchange_grid(n)), dim3(256), 0, (hipStream_t)stream, image_dev, n,
                       2147483648.0 / (double)world, bad_dev);
    HIP_TRY(hipGetLastError());
    return IMS_OK;
}

// sum of the ranks' images onto `root` (in place there; the other ranks' images are left as they are), or onto every rank
static int exchange(void* comm, double* image_dev, int32_t* scratch_i32_dev, int64_t n, int32_t root, int32_t integer_counts, void* stream)
{
    if (!comm || !image_dev || n < 0) return set_err(IMS_ERR_ARG, "comm / image is NULL");
    if (integer_counts && !scratch_i32_dev) return set_err(IMS_ERR_ARG, "scratch_i32_dev is NULL");
    if (n == 0) return IMS_OK;
    const ims_libs::Rccl* R = ims_libs::rccl();
    if (!R) return set_err(IMS_ERR_UNSUPPORTED, "RCCL is not loadable");
    hipStream_t st = (hipStream_t)stream;
    const ImsComm* ic = (const ImsComm*)comm;
    ncclComm_t c = ic->comm;
    if (root >= ic->world) return set_err(IMS_ERR_ARG, "root out of range");
    if (integer_counts) {
        // unit photon fluxes: every pixel is an integer count, exchanged as int32 -- half the bytes on the per-link-bound ring, and
        // still exact (the caller checks the counts against 2^31 / world once: ims_count_inexact)
        hipLaunchKernelGGL(k_f64_to_i32, dim3(exchange_grid(n)), dim3(256), 0, st, (const double*)image_dev, scratch_i32_dev, n);
        const int rc = root >= 0 ? nccl_err(R->reduce(scratch_i32_dev, scratch_i32_dev, (size_t)n, ncclInt32, ncclSum, root, c, st), "ncclReduce")
                                 : nccl_err(R->all_reduce(scratch_i32_dev, scratch_i32_dev, (size_t)n, ncclInt32, ncclSum, c, st), "ncclAllReduce");
        if (rc) return rc;
        if (root < 0 || root == ic->rank)          // a reduce leaves the other ranks' images as they were
            hipLaunchKernelGGL(k_i32_to_f64, dim3(exchange_grid(n)), dim3(256), 0, st, (const int32_t*)scratch_i32_dev, image_dev, n);
        HIP_TRY(hipGetLastError());
        return IMS_OK;
    }
    return root >= 0 ? nccl_err(R->reduce(image_dev, image_dev, (size_t)n, ncclFloat64, ncclSum, root, c, st), "ncclReduce")
                     : nccl_err(R->all_reduce(image_dev, image_dev, (size_t)n, ncclFloat64, ncclSum, c, st), "ncclAllReduce");
}

int ims_reduce_image(void* comm, double* image_dev, int32_t* scratch_i32_dev, int64_t n, int32_t root, int32_t integer_counts, void* stream)
{
    if (root < 0) return set_err(IMS_ERR_ARG, "root must be >= 0");
    return exchange(comm, image_dev, scratch_i32_dev, n, root, integer_counts, stream);
}

int ims_allreduce_delta(void* comm, double* delta_dev, int32_t* scratch_i32_dev, int64_t n, int32_t integer_counts, void* stream)
{
    return exchange(comm, delta_dev, scratch_i32_dev, n, -1, integer_counts, stream);
}

// a workgroup's share of the elements of an elementwise FFT kernel (walk_span): at least 4 096 workgroups where the elements allow it
// (256 CUs x 16), at most 8 192 elements each (32 passes per lookup of the object; short enough that a star's core rows, where a
// pixel costs many times a pixel of the wings, spread over many workgroups)
struct FftSpan { int64_t span; unsigned grid; };
static FftSpan fft_span(int64_t n)
{
    int64_t span = ((n + 4095) / 4096 + 255) / 256 * 256;
    if (span < 256) span = 256;
    if (span > 8192) span = 8192;
    return { span, (unsigned)((n + span - 1) / span) };
}

int ims_fft_kspace_fill(const ims_fft_params_t* params, const ims_fft_object_t* objects_dev, int64_t n_objects,
                        const int64_t* elem_prefix_dev, int64_t n_elems, double* kbuf, void* stream)
{
    if (!params || !objects_dev || !elem_prefix_dev || !kbuf) return set_err(IMS_ERR_ARG, "NULL argument");
    if (params->n_kpsf < 0 || params->n_kpsf > IMS_MAX_PSF) return set_err(IMS_ERR_ARG, "n_kpsf out of range");
    if (n_objects <= 0 || n_elems <= 0) return IMS_OK;
    {
        LaunchTimer tm((hipStream_t)stream, 3);
        const FftSpan sp = fft_span(n_elems);
        hipLaunchKernelGGL(k_fft_kspace_fill, dim3(sp.grid), dim3(256), 0, (hipStream_t)stream, *params, objects_dev,
                           n_objects, elem_prefix_dev, n_elems, sp.span, kbuf);
    }
    HIP_TRY(hipGetLastError());
    return IMS_OK;
}

__global__ void k_fill_bbox(int32_t* bbox, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { bbox[4 * i + 0] = 0x7fffffff; bbox[4 * i + 1] = -1; bbox[4 * i + 2] = 0x7fffffff; bbox[4 * i + 3] = -1; }
}

int ims_fft_spike_table(const ims_spikes_t* spikes, const int32_t* row_ptr_dev, int32_t* row_count_dev, int32_t* col_dev, double* val_dev,
                        void* stream)
{
    if (!spikes) return set_err(IMS_ERR_ARG, "spikes is NULL");
    if (spikes->cutoff < 0 || spikes->cutoff > 32767 || !(spikes->norm > 0.0)) return set_err(IMS_ERR_ARG, "spikes: cutoff / norm out of range");
    if (row_ptr_dev ? (!col_dev || !val_dev) : !row_count_dev) return set_err(IMS_ERR_ARG, "spike table: NULL output");
    ims_spikes_t k = *spikes;
    k.tab_row = nullptr; k.tab_col = nullptr; k.tab_val = nullptr;
    const unsigned rows = 2u * (unsigned)k.cutoff + 1u;
    hipLaunchKernelGGL(k_fft_spike_table, dim3((rows + 63u) / 64u), dim3(64), 0, (hipStream_t)stream, k, row_ptr_dev, row_count_dev, col_dev, val_dev);
    HIP_TRY(hipGetLastError());
    return IMS_OK;
}

int ims_fft_spikes(const ims_fft_params_t* params, const ims_fft_object_t* objects_dev, int64_t n_objects,
                   const int64_t* pix_prefix_dev, int64_t n_pix, const double* rbuf_in, double* rbuf_out,
                   int32_t* bbox_dev, void* stream)
{
    if (!params || !objects_dev || !pix_prefix_dev || !rbuf_in || !rbuf_out || !bbox_dev) return set_err(IMS_ERR_ARG, "NULL argument");
    if (n_objects <= 0 || n_pix <= 0) return IMS_OK;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_fill_bbox, dim3((unsigned)((n_objects + 255) / 256)), dim3(256), 0, st, bbox_dev, n_objects);
    const FftSpan sp = fft_span(n_pix);
    if (params->spikes.enabled)
        hipLaunchKernelGGL(k_fft_bbox, dim3(sp.grid), dim3(256), 0, st, *params, objects_dev, n_objects,
                           pix_prefix_dev, n_pix, sp.span, rbuf_in, bbox_dev);
    hipLaunchKernelGGL(k_fft_spikes, dim3(sp.grid), dim3(256), 0, st, *params, objects_dev, n_objects,
                       pix_prefix_dev, n_pix, sp.span, rbuf_in, rbuf_out, (const int32_t*)bbox_dev, (const unsigned int*)nullptr, 0u);
    HIP_TRY(hipGetLastError());
    return IMS_OK;
}

int ims_fft_spikes_listed(const ims_fft_params_t* params, const ims_fft_object_t* objects_dev, int64_t n_objects,
                          const int64_t* pix_prefix_dev, int64_t n_pix, const double* rbuf_in, double* rbuf_out,
                          int32_t* bbox_dev, uint64_t* list_dev, int64_t list_cap, uint32_t* count_dev, void* stream)
{
    if (!params || !objects_dev || !pix_prefix_dev || !rbuf_in || !rbuf_out || !bbox_dev || !list_dev || !count_dev)
        return set_err(IMS_ERR_ARG, "NULL argument");
    if (list_cap < 1 || list_cap > 0xffffffffLL) return set_err(IMS_ERR_ARG, "list_cap out of range");
    if (n_objects >= (1ll << 24)) return set_err(IMS_ERR_ARG, "the listed spike step holds at most 2^24 objects");
    if (n_objects <= 0 || n_pix <= 0) return IMS_OK;
    if (!params->spikes.enabled) return ims_fft_spikes(params, objects_dev, n_objects, pix_prefix_dev, n_pix, rbuf_in, rbuf_out, bbox_dev, stream);
    hipStream_t st = (hipStream_t)stream;
    const unsigned int cap = (unsigned int)list_cap;
    HIP_TRY(hipMemsetAsync(count_dev, 0, sizeof(uint32_t), st));
    hipLaunchKernelGGL(k_fill_bbox, dim3((unsigned)((n_objects + 255) / 256)), dim3(256), 0, st, bbox_dev, n_objects);
    const FftSpan sp = fft_span(n_pix);
    hipLaunchKernelGGL(k_fft_bbox, dim3(sp.grid), dim3(256), 0, st, *params, objects_dev, n_objects,
                       pix_prefix_dev, n_pix, sp.span, rbuf_in, bbox_dev);
    hipLaunchKernelGGL(k_fft_spikes_stream, dim3(sp.grid), dim3(256), 0, st, *params, objects_dev, n_objects, pix_prefix_dev, n_pix,
                       sp.span, rbuf_in, rbuf_out, (const int32_t*)bbox_dev, (unsigned long long*)list_dev, cap, (unsigned int*)count_dev);
    // the arms: a grid that covers the list at 256 entries per workgroup, at most 16 384 workgroups (they stride)
    int64_t arms = (list_cap + 255) / 256;
    if (arms > 16384) arms = 16384;
    hipLaunchKernelGGL(k_fft_spikes_arms, dim3((unsigned)arms), dim3(256), 0, st, *params, objects_dev, rbuf_in, rbuf_out,
                       (const int32_t*)bbox_dev, (const unsigned long long*)list_dev, cap, (const unsigned int*)count_dev);
    // more arm pixels than the list holds: the whole step once more in one launch (returns at once otherwise)
    hipLaunchKernelGGL(k_fft_spikes, dim3(sp.grid), dim3(256), 0, st, *params, objects_dev, n_objects, pix_prefix_dev, n_pix, sp.span,
                       rbuf_in, rbuf_out, (const int32_t*)bbox_dev, (const unsigned int*)count_dev, cap);
    HIP_TRY(hipGetLastError());
    return IMS_OK;
}

int ims_fft_finish(const ims_fft_params_t* params, const ims_fft_object_t* objects_dev, int64_t n_objects,
                   const int64_t* pix_prefix_dev, int64_t n_pix, const double* rbuf, void* stream)
{
    if (!params || !objects_dev || !pix_prefix_dev || !rbuf) return set_err(IMS_ERR_ARG, "NULL argument");
    if (!params->image) return set_err(IMS_ERR_ARG, "image is NULL");
    if (n_objects <= 0 || n_pix <= 0) return IMS_OK;
    const FftSpan sp = fft_span(n_pix);
    hipLaunchKernelGGL(k_fft_finish, dim3(sp.grid), dim3(256), 0, (hipStream_t)stream, *params, objects_dev,
                       n_objects, pix_prefix_dev, n_pix, sp.span, rbuf);
    HIP_TRY(hipGetLastError());
    return IMS_OK;
}

int ims_sensor_pixel_areas(const ims_sensor_t* sensor_dev, const ims_sensor_t* sensor_host, int32_t slot,
                           double* area_dev, long long* sum_q32_dev, void* stream)
{
    if (!sensor_dev || !area_dev || !sum_q32_dev) return set_err(IMS_ERR_ARG, "NULL argument");
    if (!sensor_host || !sensor_host->bf_slots) return set_err(IMS_ERR_ARG, "sensor_host/bf_slots is NULL (host copy of the slot table required)");
    if (slot < 0 || slot >= sensor_host->n_bf_slots) return set_err(IMS_ERR_ARG, "slot out of range");
    const ims_bf_slot_t& b = sensor_host->bf_slots[slot];
    const int64_t n = (int64_t)b.nx * b.ny;
    if (n <= 0) return IMS_OK;
    hipLaunchKernelGGL(k_pixel_areas, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sensor_dev, slot,
                       area_dev, sum_q32_dev);
    HIP_TRY(hipGetLastError());
    return IMS_OK;
}

int ims_flat_add(const double* area_dev, const double* base_dev, double level, double inv_mean_area, uint64_t seed,
                 int64_t iteration, int32_t nx, int32_t ny, double* image_dev, double* delta_dev, void* stream)
{
    if (!image_dev) return set_err(IMS_ERR_ARG, "image is NULL");
    if (nx <= 0 || ny <= 0) return IMS_OK;
    const int64_t n = (int64_t)nx * ny;
    hipLaunchKernelGGL(k_flat_add, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, area_dev, base_dev,
                       level, inv_mean_area, seed, iteration, nx, ny, image_dev, delta_dev);
    HIP_TRY(hipGetLastError());
    return IMS_OK;
}

int ims_image_add(double* dst, const double* src, int64_t n, void* stream)
{
    if (!dst || !src) return set_err(IMS_ERR_ARG, "dst/src is NULL");
    if (n <= 0) return IMS_OK;
    hipLaunchKernelGGL(k_image_add, dim3(grid_for_pool(n)), dim3(256), 0, (hipStream_t)stream, dst, src, n);
    HIP_TRY(hipGetLastError());
    return IMS_OK;
}

// Launch-wide constants of the air index (host arithmetic, IEEE binary64, same operation order as the
// oracle's restatement): n - 1 = air_p * dispersion(wavelength) - air_w * water(wavelength).
static void air_factors(double p_kpa, double t_k, double h2o_kpa, double* air_p, double* air_w)
{
    const double Pm = p_kpa * 7.50061683;
    const double T = t_k - 273.15;
    const double W = h2o_kpa * 7.50061683;
    const double tf = 1.0 + 0.003661 * T;
    *air_p = 1.0e-6 * (Pm * (1.0 + (1.049 - 0.0157 * T) * 1.0e-6 * Pm) / (720.883 * tf));
    *air_w = W * 1.0e-6 / tf;
}
static double host_air_n_minus_one(double wave_nm, double air_p, double air_w)
{
    const double wm = wave_nm * 1.0e-3;
    const double w2 = wm * wm;
    const double d1 = fma(146.0, w2, -1.0), d2 = fma(41.0, w2, -1.0);
    const double den = d1 * d2;
    const double num = fma(29498.1, d2, 255.4 * d1);
    const double disp = fma(64.328, den, w2 * num);
    const double wat = fma(0.0624, w2, -0.000680) * den;
    return (air_p * (disp * w2) - air_w * wat) / (den * w2);
}

int ims_fill_derived_op(ims_op_t* op)
{
    if (!op) return set_err(IMS_ERR_ARG, "op is NULL");
    if (op->kind == IMS_OP_PHOTON_DCR) {
        air_factors(op->p[1], op->p[2], op->p[3], &op->p[5], &op->p[6]);
        const double nm1 = host_air_n_minus_one(op->p[0], op->p[5], op->p[6]);
        op->p[7] = nm1 * (nm1 + 2.0) / 2.0 / (nm1 * nm1 + 2.0 * nm1 + 1.0);
    }
    if (op->kind == IMS_OP_PUPIL_ANNULUS_SAMPLER) {
        const double ro2 = op->p[0] * op->p[0], ri2 = op->p[1] * op->p[1];
        op->p[2] = ri2; op->p[3] = ro2 - ri2;
    }
    if (op->kind == IMS_OP_REFRACTION) {
        const double nn = op->p[0] * op->p[0];
        op->p[1] = nn; op->p[2] = nn - 1.0;
    }
    return IMS_OK;
}

// The derived constants below are single IEEE operations (this file is compiled with -ffp-contract=off), so the kernels
// compute with exactly the values they would have formed themselves.
int ims_fill_derived_optics(ims_optics_t* o)
{
    if (!o) return set_err(IMS_ERR_ARG, "optics is NULL");
    if (o->n_surfaces < 0 || o->n_surfaces > IMS_MAX_SURFACES) return set_err(IMS_ERR_ARG, "n_surfaces out of range");
    for (int k = 0; k < o->n_surfaces; ++k) {
        ims_surface_t& S = o->surf[k];
        S.k1 = 1.0 + S.conic;
        S.k1c = S.k1 * S.inv_R;
        S.m2R = -2.0 * S.R;
        S.cc = S.inv_R * S.inv_R;
        S.obsc_i2 = S.obsc_inner * S.obsc_inner;
        S.obsc_o2 = S.obsc_outer * S.obsc_outer;
        for (int a = 0; a < 4; ++a) S.asph_d[a] = S.asph[a] * (double)(a + 2);
    }
    const double* ef = o->e_focal;
    const double* z0 = o->e_z0;
    o->rot_g[0] = ef[1] * z0[2] - ef[2] * z0[1];
    o->rot_g[1] = ef[2] * z0[0] - ef[0] * z0[2];
    o->rot_g[2] = ef[0] * z0[1] - ef[1] * z0[0];
    o->rot_gnorm = sqrt(o->rot_g[0] * o->rot_g[0] + o->rot_g[1] * o->rot_g[1] + o->rot_g[2] * o->rot_g[2]);
    return IMS_OK;
}

int ims_fill_derived_atmosphere(ims_atmosphere_t* A)
{
    if (!A) return set_err(IMS_ERR_ARG, "atmosphere is NULL");
    if (A->npix <= 0 || !(A->scale > 0.0)) return set_err(IMS_ERR_ARG, "atmosphere: npix and scale must be positive");
    A->dn = (double)A->npix;
    A->inv_n = 1.0 / A->dn;
    A->inv_scale = 1.0 / A->scale;
    const double ro2 = A->aper_r_outer * A->aper_r_outer, ri2 = A->aper_r_inner * A->aper_r_inner;
    A->aper_ri2 = ri2;
    A->aper_dr2 = ro2 - ri2;
    return IMS_OK;
}

int ims_fill_derived_sensor(ims_sensor_t* S)
{
    if (!S) return set_err(IMS_ERR_ARG, "sensor is NULL");
    if (S->kind == IMS_SENSOR_SILICON) {
        if (!(S->thickness > 0.0) || !(S->pixel_size > 0.0)) return set_err(IMS_ERR_ARG, "sensor: thickness and pixel_size must be positive");
        S->diff_coef = S->diff_step / (S->thickness * S->pixel_size);
        S->thick_m1 = S->thickness - 1.0;
    }
    return IMS_OK;
}

int ims_fill_derived_medium(int32_t kind, double* c6)
{
    if (!c6) return set_err(IMS_ERR_ARG, "c6 is NULL");
    if (kind == IMS_MEDIUM_AIR) air_factors(c6[0], c6[1], c6[2], &c6[3], &c6[4]);
    return IMS_OK;
}

int ims_image_to_float(const double* src, float* dst, int64_t n, void* stream)
{
    if (!dst || !src) return set_err(IMS_ERR_ARG, "dst/src is NULL");
    if (n <= 0) return IMS_OK;
    hipLaunchKernelGGL(k_image_to_float, dim3(grid_for_pool(n)), dim3(256), 0, (hipStream_t)stream, src, dst, n);
    HIP_TRY(hipGetLastError());
    return IMS_OK;
}

static int check_readout(const ims_readout_t* ro)
{
    if (!ro) return set_err(IMS_ERR_ARG, "readout descriptor is NULL");
    if (ro->n_amps < 1 || ro->n_amps > IMS_MAX_AMPS) return set_err(IMS_ERR_ARG, "n_amps out of range");
    if (ro->seg_w < 1 || ro->seg_h < 1 || ro->data_x0 < 0 || ro->data_y0 < 0 || ro->data_x0 + ro->seg_w > ro->raw_w ||
        ro->data_y0 + ro->seg_h > ro->raw_h)
        return set_err(IMS_ERR_ARG, "imaging section does not fit the raw segment");
    for (int a = 0; a < ro->n_amps; ++a)
        if (!(ro->amps[a].gain > 0.0f)) return set_err(IMS_ERR_ARG, "amplifier gain must be positive");
    return IMS_OK;
}

int ims_readout_bleed(double* image_dev, unsigned char* flags_dev, int32_t nx, int32_t ny, double full_well,
                      int32_t midline_stop, void* stream)
{
    if (!image_dev || !flags_dev) return set_err(IMS_ERR_ARG, "image / flags is NULL");
    if (nx < 1 || ny < 1) return set_err(IMS_ERR_ARG, "empty image");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (int64_t)nx * ny;
    int* span = (int*)(flags_dev + ((n + 15) / 16) * 16);        // the tail of the scratch: 4 ints per column
    hipLaunchKernelGGL(k_readout_span_init, dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, st, span, nx);
    hipLaunchKernelGGL(k_readout_flags, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, image_dev, flags_dev, span, nx, ny,
                       full_well, midline_stop ? 1 : 0);
    const int threads = nx * (midline_stop ? 2 : 1);
    hipLaunchKernelGGL(k_readout_bleed, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, image_dev, flags_dev, span, nx, ny,
                       full_well, midline_stop ? 1 : 0);
    HIP_TRY(hipGetLastError());
    return IMS_OK;
}

int ims_readout_segments(const double* image_dev, int32_t nx, int32_t ny, const ims_readout_t* ro, float* seg_dev, void* stream)
{
    if (int e = check_readout(ro)) return e;
    if (!image_dev || !seg_dev) return set_err(IMS_ERR_ARG, "image / segments is NULL");
    for (int a = 0; a < ro->n_amps; ++a)
        if (ro->amps[a].x0 < 0 || ro->amps[a].y0 < 0 || ro->amps[a].x0 + ro->seg_w > nx || ro->amps[a].y0 + ro->seg_h > ny)
            return set_err(IMS_ERR_ARG, "amplifier section outside the e-image");
    hipLaunchKernelGGL(k_readout_segments, dim3((unsigned)((ro->raw_w + 63) / 64), (unsigned)((ro->raw_h + 3) / 4), 1u),
                       dim3(64, 4), 0, (hipStream_t)stream, image_dev, nx, *ro, seg_dev);
    HIP_TRY(hipGetLastError());
    return IMS_OK;
}

int ims_readout_cte(const float* src_dev, float* dst_dev, const ims_readout_t* ro, const double* band_dev, int32_t n_band,
                    int32_t axis, void* stream)
{
    if (int e = check_readout(ro)) return e;
    if (!src_dev || !dst_dev || !band_dev) return set_err(IMS_ERR_ARG, "src / dst / band is NULL");
    if (src_dev == dst_dev) return set_err(IMS_ERR_ARG, "ims_readout_cte is out of place: src and dst must differ");
    if (n_band < 1 || (axis != 0 && axis != 1)) return set_err(IMS_ERR_ARG, "n_band / axis out of range");
    if ((int64_t)ro->raw_w * ro->raw_h * ro->n_amps >= (1ll << 31)) return set_err(IMS_ERR_ARG, "raw segments too large");
    const dim3 grid((unsigned)((ro->raw_w + 63) / 64), (unsigned)((ro->raw_h + 3) / 4), (unsigned)ro->n_amps);
    if (n_band == 21)
        hipLaunchKernelGGL(k_readout_cte<21>, grid, dim3(64, 4), 0, (hipStream_t)stream, src_dev, dst_dev, ro->raw_w, ro->raw_h,
                           band_dev, n_band, axis);
    else
        hipLaunchKernelGGL(k_readout_cte<0>, grid, dim3(64, 4), 0, (hipStream_t)stream, src_dev, dst_dev, ro->raw_w, ro->raw_h,
                           band_dev, n_band, axis);
    HIP_TRY(hipGetLastError());
    return IMS_OK;
}

int ims_readout_finish(const float* seg_dev, const ims_readout_t* ro, uint64_t seed, int32_t* out_dev, void* stream)
{
    if (int e = check_readout(ro)) return e;
    if (!seg_dev || !out_dev) return set_err(IMS_ERR_ARG, "segments / output is NULL");
    const int64_t n = (int64_t)ro->raw_w * ro->raw_h * ro->n_amps;
    if ((((int64_t)ro->raw_w * ro->raw_h) & 1) == 0 && (((uintptr_t)seg_dev | (uintptr_t)out_dev) & 7u) == 0)
        hipLaunchKernelGGL(k_readout_finish_pairs, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, seg_dev, *ro,
                           seed, out_dev);
    else
        hipLaunchKernelGGL(k_readout_finish, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, seg_dev, *ro, seed,
                           out_dev);
    HIP_TRY(hipGetLastError());
    return IMS_OK;
}

int ims_screen_prepass(const ims_render_params_t* params, int32_t comp, int32_t n_buckets, const int64_t* slice_first_host,
                       int64_t max_slice_photons, int64_t* scratch_dev, int64_t* entries_dev, double* kick_dev, void* stream)
{
    if (!params || !slice_first_host || !scratch_dev || !entries_dev || !kick_dev) return set_err(IMS_ERR_ARG, "NULL argument");
    if (!params->objects || !params->seg_prefix || !params->atm) return set_err(IMS_ERR_ARG, "objects / seg_prefix / atm is NULL");
    if (comp < 0 || comp >= params->n_psf || params->psf[comp].kind != IMS_PSF_SCREENS)
        return set_err(IMS_ERR_ARG, "comp is not a phase-screen PSF component");
    if (n_buckets < 1 || n_buckets > SCR_MAXB || (n_buckets & (n_buckets - 1)) != 0) return set_err(IMS_ERR_ARG, "n_buckets must be a power of two <= 256");
    if (params->seg_size != 256) return set_err(IMS_ERR_ARG, "seg_size must be 256");
    if (params->n_segments <= 0) return IMS_OK;
    if (params->n_segments > 0x7fffffffLL) return set_err(IMS_ERR_ARG, "too many segments");
    ScreenSlices S;
    int64_t max_segs = 0;
    for (int q = 0; q < 9; ++q) {
        S.first[q] = slice_first_host[q];
        S.seg_first[q] = slice_first_host[9 + q];
        if (q > 0 && S.seg_first[q] - S.seg_first[q - 1] > max_segs) max_segs = S.seg_first[q] - S.seg_first[q - 1];
    }
    if (S.seg_first[0] != 0 || S.seg_first[8] != params->n_segments) return set_err(IMS_ERR_ARG, "slice segment boundaries do not cover the table");
    // (every argument is checked before the first launch: a refused call leaves nothing enqueued)
    if (max_slice_photons <= 0 || (max_slice_photons + 255) / 256 * 8 > 0x7fffffffLL) return set_err(IMS_ERR_ARG, "max_slice_photons out of range");
    hipStream_t st = (hipStream_t)stream;
    const int n_bins = 8 * n_buckets;
    unsigned long long* bins = (unsigned long long*)scratch_dev;
    HIP_TRY(hipMemsetAsync(bins, 0, sizeof(unsigned long long) * (size_t)(n_bins + 16), st));
    const dim3 grid((unsigned)(8 * ((max_segs + SCR_CH - 1) / SCR_CH)));
    hipLaunchKernelGGL(k_screen_sort<0>, grid, dim3(256), 0, st, *params, comp, n_buckets, S, bins, (int64_t*)nullptr);
    hipLaunchKernelGGL(k_screen_scan, dim3(1), dim3(64), 0, st, bins, n_bins, n_buckets);
    hipLaunchKernelGGL(k_screen_sort<1>, grid, dim3(256), 0, st, *params, comp, n_buckets, S, bins, entries_dev);
    // every slice gets the blocks of the largest one (a block beyond its slice's end leaves at once); the slice boundaries sit
    // behind the bins, whose cursors the scatter has advanced to the bin ends
    ScreenObject* slim = (ScreenObject*)(scratch_dev + n_bins + 16);
    hipLaunchKernelGGL(k_screen_objects, dim3((unsigned)((params->n_objects + 255) / 256)), dim3(256), 0, st, *params, slim);
    hipLaunchKernelGGL(k_screen_gather, dim3((unsigned)(8 * ((max_slice_photons + 255) / 256))), dim3(256), 0, st,
                       *params, comp, bins + n_bins, entries_dev, slim, kick_dev);
    HIP_TRY(hipGetLastError());
    return IMS_OK;
}

int ims_build_object_table(const ims_catalog_t* cat, const ims_optics_t* optics_dev, ims_object_t* rows_dev,
                           ims_object_meta_t* meta_dev, void* stream)
{
    if (!cat || !optics_dev || !rows_dev || !meta_dev) return set_err(IMS_ERR_ARG, "NULL argument");
    if (cat->n < 0) return set_err(IMS_ERR_ARG, "negative catalog length");
    if (cat->n == 0) return IMS_OK;
    if (!cat->x || !cat->y || !cat->nominal_flux || !cat->kind || !cat->hlr || !cat->q || !cat->pa || !cat->prof_table)
        return set_err(IMS_ERR_ARG, "catalog column is NULL");
    if ((cat->g1 != nullptr) != (cat->g2 != nullptr) || (cat->g1 != nullptr) != (cat->mu != nullptr))
        return set_err(IMS_ERR_ARG, "g1, g2 and mu come together");
    if (!cat->star_size || cat->n_star_size <= 0 || !cat->gal_radius || cat->n_gal_radius <= 0)
        return set_err(IMS_ERR_ARG, "star_size / gal_radius table missing");
    if (!(cat->pixel_scale > 0.0) || !(cat->dg_stepk > 0.0) || cat->nmax <= 0) return set_err(IMS_ERR_ARG, "pixel_scale / dg_stepk / nmax");
    hipLaunchKernelGGL(k_build_object_table, dim3((unsigned)((cat->n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *cat, optics_dev,
                       rows_dev, meta_dev);
    HIP_TRY(hipGetLastError());
    return IMS_OK;
}

int ims_patch_stamp_sizes(ims_object_t* rows_dev, ims_object_meta_t* meta_dev, const int64_t* index_dev, const int32_t* size_dev,
                          int64_t n, void* stream)
{
    if (n <= 0) return IMS_OK;
    if (!rows_dev || !meta_dev || !index_dev || !size_dev) return set_err(IMS_ERR_ARG, "NULL argument");
    hipLaunchKernelGGL(k_patch_stamp_sizes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rows_dev, meta_dev, index_dev,
                       size_dev, n);
    HIP_TRY(hipGetLastError());
    return IMS_OK;
}

int ims_gather_rows(const ims_object_t* rows_dev, const int64_t* index_dev, const int64_t* first_dev, const int64_t* count_dev,
                    const int32_t* bf_state_dev, int32_t clear_flags, ims_object_t* dst_dev, int64_t n, void* stream)
{
    if (n <= 0) return IMS_OK;
    if (!rows_dev || !index_dev || !dst_dev) return set_err(IMS_ERR_ARG, "NULL argument");
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((n * 16 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rows_dev, index_dev, first_dev,
                       count_dev, bf_state_dev, clear_flags, dst_dev, n);
    HIP_TRY(hipGetLastError());
    return IMS_OK;
}

// ---- instance-catalog tokenizer: plain host code (the catalog reader is per-line Python in the reference) ----
static bool tok_double(const char* p, int len, double* out)
{
    char buf[64];
    if (len <= 0 || len >= (int)sizeof(buf)) return false;
    memcpy(buf, p, len);
    buf[len] = 0;
    char* end = nullptr;
    *out = strtod(buf, &end);
    return end == buf + len;
}
static bool tok_is(const char* p, int len, const char* word)
{
    const int n = (int)strlen(word);
    if (len != n) return false;
    for (int i = 0; i < n; ++i) {
        char c = p[i];
        if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a');
        if (c != word[i]) return false;
    }
    return true;
}
static bool tok_ends(const char* p, int len, const char* suffix)
{
    const int n = (int)strlen(suffix);
    return len >= n && memcmp(p + len - n, suffix, n) == 0;
}

int64_t ims_parse_instcat_objects(const char* text, int64_t n_bytes, int64_t max_objects, double* num, int32_t* kind, int64_t* span)
{
    if (!text || !num || !kind || !span || n_bytes < 0 || max_objects < 0) { set_err(IMS_ERR_ARG, "NULL argument"); return -1; }
    constexpr int MAXT = 32;
    int64_t n_out = 0, n_line = 0;
    int64_t pos = 0;
    while (pos < n_bytes && n_out < max_objects) {
        int64_t eol = pos;
        while (eol < n_bytes && text[eol] != '\n') ++eol;
        const char* ln = text + pos;
        const int64_t len = eol - pos;
        pos = eol + 1;
        if (len < 6 || memcmp(ln, "object", 6) != 0) continue;
        const int64_t this_line = n_line++;
        bool has_inf = false;
        for (int64_t i = 0; i + 5 <= len; ++i)
            if (ln[i] == ' ' && ln[i + 1] == 'i' && ln[i + 2] == 'n' && ln[i + 3] == 'f' && ln[i + 4] == ' ') { has_inf = true; break; }
        if (has_inf) continue;
        const char* tp[MAXT]; int tl[MAXT]; int nt = 0;
        for (int64_t i = 0; i < len && nt < MAXT;) {
            while (i < len && (ln[i] == ' ' || ln[i] == '\t' || ln[i] == '\r' || ln[i] == '\v' || ln[i] == '\f')) ++i;
            if (i >= len) break;
            const int64_t a = i;
            while (i < len && !(ln[i] == ' ' || ln[i] == '\t' || ln[i] == '\r' || ln[i] == '\v' || ln[i] == '\f')) ++i;
            tp[nt] = ln + a; tl[nt] = (int)(i - a); ++nt;
        }
        if (nt < 13) return -(this_line + 1);
        double v[16] = { 0.0 };
        int k = 5, di = 15;
        double a = 0.0, b = 0.0, pa = 0.0, nn = 0.0;
        auto need = [&](int t, double* out) { return t < nt && tok_double(tp[t], tl[t], out); };
        if (!need(4, &v[2])) return -(this_line + 1);
        if (tok_is(tp[12], tl[12], "point")) { k = 0; di = 13; }
        else if (tok_is(tp[12], tl[12], "sersic2d")) {
            k = 1; di = 17;
            double raw;
            if (!need(13, &a) || !need(14, &b) || !need(15, &pa) || !need(16, &raw)) return -(this_line + 1);
            nn = nearbyint(raw * 20.0) / 20.0;                    // Python's round(): half to even, as nearbyint in the default mode
        } else if (tok_is(tp[12], tl[12], "knots")) {
            k = 2; di = 17;
            if (!need(13, &a) || !need(14, &b) || !need(15, &pa)) return -(this_line + 1);
            // int(token): a decimal integer literal only
            if (16 >= nt) return -(this_line + 1);
            char buf[32];
            if (tl[16] <= 0 || tl[16] >= (int)sizeof(buf)) return -(this_line + 1);
            memcpy(buf, tp[16], tl[16]); buf[tl[16]] = 0;
            char* end = nullptr;
            const long long iv = strtoll(buf, &end, 10);
            if (end != buf + tl[16]) return -(this_line + 1);
            nn = (double)iv;
        } else if (tok_is(tp[12], tl[12], "streak")) {
            k = 3; di = 16;
            if (!need(13, &a) || !need(14, &b) || !need(15, &pa)) return -(this_line + 1);
        } else if (tok_ends(tp[12], tl[12], ".fits") || tok_ends(tp[12], tl[12], ".fits.gz")) {
            k = 4;
            if (!need(13, &a) || !need(14, &pa)) return -(this_line + 1);
        }
        const bool valid = v[2] < 50.0 && !((k == 1 || k == 2) && a < b) && !(k == 2 && nn <= 0.0);
        if (!valid) continue;
        if (!need(2, &v[0]) || !need(3, &v[1]) || !need(6, &v[3]) || !need(7, &v[4]) || !need(8, &v[5]) || !need(9, &v[6]))
            return -(this_line + 1);
        v[7] = a; v[8] = b; v[9] = pa; v[10] = nn;
        v[11] = 0.0; v[12] = 3.1; v[13] = 0.0; v[14] = 3.1;
        int d = di;
        if (d < nt && !tok_is(tp[d], tl[d], "none")) {
            if (!need(d + 1, &v[11]) || !need(d + 2, &v[12])) return -(this_line + 1);
            d += 3;
        } else d += 1;
        if (d < nt && !tok_is(tp[d], tl[d], "none")) {
            if (!need(d + 1, &v[13]) || !need(d + 2, &v[14])) return -(this_line + 1);
        }
        memcpy(num + n_out * 16, v, sizeof(v));
        kind[n_out] = k;
        int64_t* sp = span + n_out * 6;
        sp[0] = tp[1] - text; sp[1] = tl[1];
        sp[2] = tp[5] - text; sp[3] = tl[5];
        sp[4] = tp[12] - text; sp[5] = tl[12];
        ++n_out;
    }
    return n_out;
}

int ims_struct_size(int which)
{
    switch (which) {
    case 0: return (int)sizeof(ims_object_t);
    case 1: return (int)sizeof(ims_radial_tables_t);
    case 2: return (int)sizeof(ims_lin_tables_t);
    case 3: return (int)sizeof(ims_psf_component_t);
    case 4: return (int)sizeof(ims_op_t);
    case 5: return (int)sizeof(ims_surface_t);
    case 6: return (int)sizeof(ims_tansip_t);
    case 7: return (int)sizeof(ims_optics_t);
    case 8: return (int)sizeof(ims_bf_slot_t);
    case 9: return (int)sizeof(ims_sensor_t);
    case 10: return (int)sizeof(ims_photons_t);
    case 11: return (int)sizeof(ims_render_params_t);
    case 12: return (int)sizeof(ims_plan_item_t);
    case 13: return (int)sizeof(ims_atmosphere_t);
    case 14: return (int)sizeof(ims_fft_object_t);
    case 15: return (int)sizeof(ims_fft_params_t);
    case 16: return (int)sizeof(ims_readout_t);
    case 17: return (int)sizeof(ims_chain_t);
    case 18: return (int)sizeof(ims_catalog_t);
    case 19: return (int)sizeof(ims_object_meta_t);
    case 20: return (int)sizeof(ims_plan_input_t);
    case 21: return (int)sizeof(ims_plan_sizes_t);
    case 22: return (int)sizeof(ims_tuning_t);
    }
    return -1;
}

int ims_test_math(int which, const double* in_dev, double* out_dev, int64_t n, uint64_t seed, int64_t obj,
                  uint32_t slot, void* stream)
{
    if (n <= 0) return IMS_OK;
    hipLaunchKernelGGL(k_test_math, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       which, in_dev, out_dev, n, seed, obj, slot);
    HIP_TRY(hipGetLastError());
    return IMS_OK;
}

}  // extern "C"
