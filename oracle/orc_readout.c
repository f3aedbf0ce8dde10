/* ORACLE -- test infrastructure only (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).
 *
 * CPU restatement of the CCD readout chain, array by array like the reference runs it:
 *   bleed trails         imsim/bleed_trails.py:28-152 (bleed_eimage, bleed_channel, BleedCharge)
 *   amplifier segments   imsim/readout.py:413-478 (CcdReadout.build_amp_images: gain, flips, crosstalk :403-411,
 *                        prescan/overscan :455-460, CTE :391-401 with the matrix of :163-205, bias + read noise :465-478)
 * Pinned by golden vectors generated from the reference's own bleed_trails.py / cte_matrix
 * (tests/golden/make_readout_golden.py).  Arrays that are float32 in the reference (ImageF e-image and segments)
 * are float32 here; the CTE product is binary64, summed with ascending source index. */
#include <stdint.h>
#include <string.h>
#include "orc.h"
#include "orc_math.h"

#define ORC_READOUT_ID_BASE 0x7E00000000ll

/* one BleedCharge call: pixel `ypix` of channel c (length n, element stride s) takes what it can */
static int bleed_call(double* c, int64_t s, int n, int ypix, double fw, double* excess)
{
    if (0 <= ypix && ypix < n) {
        double room = fw - c[ypix * s];
        double bled = room < *excess ? room : *excess;
        c[ypix * s] += bled;
        *excess -= bled;
    } else if (ypix < 0) {
        *excess -= (fw < *excess ? fw : *excess);
    }
    return *excess == 0.0;
}

/* bleed_channel on a strided channel; `sat` holds the ORIGINAL above-full-well flags of the channel */
static void bleed_channel(double* c, const unsigned char* sat, int64_t s, int n, double fw)
{
    int y = 0;
    while (y < n) {
        if (!sat[y * s]) { y++; continue; }
        int y0 = y;
        while (y < n && sat[y * s]) y++;
        int y1 = y;
        double excess = 0.0;
        for (int k = y0; k < y1; ++k) excess += c[k * s];
        excess -= (double)(y1 - y0) * fw;
        for (int k = y0; k < y1; ++k) c[k * s] = fw;
        int reach = y0 > n - y1 ? y0 : n - y1;
        for (int dy = 0; dy < reach; ++dy)
            if (bleed_call(c, s, n, y0 - dy - 1, fw, &excess) || bleed_call(c, s, n, y1 + dy, fw, &excess)) break;
    }
}

void orc_readout_bleed(double* image, int32_t nx, int32_t ny, double full_well, int32_t midline_stop, unsigned char* flags)
{
    for (int64_t p = 0; p < (int64_t)nx * ny; ++p) flags[p] = image[p] > full_well;
    int ymid = ny / 2;
    for (int x = 0; x < nx; ++x) {
        if (midline_stop) {
            bleed_channel(image + x, flags + x, nx, ymid, full_well);
            bleed_channel(image + (int64_t)ymid * nx + x, flags + (int64_t)ymid * nx + x, nx, ny - ymid, full_well);
        } else {
            bleed_channel(image + x, flags + x, nx, ny, full_well);
        }
    }
}

/* amp_data of every amplifier in readout order (flips applied), float32 */
static void amp_arrays(const double* image, int32_t nx, const ims_readout_t* ro, float* arr)
{
    for (int a = 0; a < ro->n_amps; ++a) {
        const ims_amp_t* A = &ro->amps[a];
        float* out = arr + (int64_t)a * ro->seg_w * ro->seg_h;
        for (int v = 0; v < ro->seg_h; ++v)
            for (int u = 0; u < ro->seg_w; ++u) {
                int sx = A->flip_x ? ro->seg_w - 1 - u : u;
                int sy = A->flip_y ? ro->seg_h - 1 - v : v;
                float e = (float)image[(int64_t)(A->y0 + sy) * nx + (A->x0 + sx)];
                out[(int64_t)v * ro->seg_w + u] = e / A->gain;
            }
    }
}

/* scratch: n_amps * seg_w * seg_h floats */
void orc_readout_segments(const double* image, int32_t nx, int32_t ny, const ims_readout_t* ro, float* seg, float* scratch)
{
    (void)ny;
    amp_arrays(image, nx, ro, scratch);
    const int64_t per = (int64_t)ro->raw_w * ro->raw_h, sec = (int64_t)ro->seg_w * ro->seg_h;
    memset(seg, 0, sizeof(float) * per * ro->n_amps);
    for (int a = 0; a < ro->n_amps; ++a)
        for (int v = 0; v < ro->seg_h; ++v)
            for (int u = 0; u < ro->seg_w; ++u) {
                int64_t k = (int64_t)v * ro->seg_w + u;
                float out = scratch[a * sec + k];
                if (ro->has_xtalk) {
                    float sum = 0.0f;
                    for (int j = 0; j < ro->n_amps; ++j) sum = sum + ro->xtalk[a * IMS_MAX_AMPS + j] * scratch[j * sec + k];
                    out = out + sum;
                }
                seg[a * per + (int64_t)(v + ro->data_y0) * ro->raw_w + (u + ro->data_x0)] = out;
            }
}

void orc_readout_cte(const float* src, float* dst, const ims_readout_t* ro, const double* band, int32_t n_band, int32_t axis)
{
    const int64_t per = (int64_t)ro->raw_w * ro->raw_h;
    for (int a = 0; a < ro->n_amps; ++a)
        for (int ry = 0; ry < ro->raw_h; ++ry)
            for (int rx = 0; rx < ro->raw_w; ++rx) {
                int i = axis == 0 ? ry : rx;
                int64_t step = axis == 0 ? ro->raw_w : 1;
                int64_t p = a * per + (int64_t)ry * ro->raw_w + rx;
                int jmin = i - (n_band - 1) < 0 ? 0 : i - (n_band - 1);
                double acc = 0.0;
                for (int j = jmin; j <= i; ++j) acc = acc + band[(int64_t)i * n_band + (i - j)] * (double)src[p - (int64_t)(i - j) * step];
                dst[p] = (float)acc;
            }
}

void orc_readout_finish(const float* seg, const ims_readout_t* ro, uint64_t seed, int32_t* out)
{
    const int64_t per = (int64_t)ro->raw_w * ro->raw_h;
    for (int a = 0; a < ro->n_amps; ++a)
        for (int64_t q = 0; q < per; ++q) {
            float v = seg[a * per + q] + ro->amps[a].bias_level;
            orc_words_t w = orc_words(seed, ORC_READOUT_ID_BASE + a, q >> 1, 0u);
            double g0, g1;
            orc_gauss_words(w.w[0], w.w[1], &g0, &g1);
            v = v + (float)((double)ro->amps[a].read_noise * ((q & 1) ? g1 : g0));
            out[a * per + q] = (int32_t)v;
        }
}
