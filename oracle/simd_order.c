/*
 * simd_order.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Scalar restatement of the SUMMATION ORDER of the reference's SIMD paths (opt = 1 / 2 / 3), which are not bit-equal to
 * the opt = 0 path:
 *   /root/reference/src/resize_plane_sse41.cpp:41-90    4 lanes,  multiply then add (no FMA, :49)
 *   /root/reference/src/resize_plane_avx2.cpp:45-98     8 lanes,  fused multiply-add (:53, :72, :91)
 *   /root/reference/src/resize_plane_avx512.cpp:45-103  16 lanes, fused multiply-add (:53, :73, :93)
 * Per output sample: lane l accumulates the taps lx = l, l + W, l + 2W, ... of every kernel row, rows in order; then the
 * horizontal sum  512 -> 256 (lane i + lane i+8, avx512:60), 256 -> 128 (lane i + lane i+4, avx2:60),
 * (h0 + h1) + (h2 + h3) (hadd of hadd, sse41:56).  Integer store: cvtps_epi32 (round-half-even; NaN / out of range ->
 * INT_MIN) then packus_epi32 (saturate to 0..65535, NOT to the clip's peak) and for 8-bit packus_epi16 (signed 16-bit
 * -> 0..255: a 16-bit pattern >= 0x8000 is negative, i.e. 0) -- sse41:57,:75, avx2:62,:81.  Float source samples are
 * first clamped from below: max_ps(src, min_val) with min_val = -0.5 for planes 1.. of YUV clips, else 0 (sse41:20,:83);
 * max_ps returns its SECOND operand when the first is NaN.
 *
 * Lanes whose lx lies past filter_size multiply whatever the reference reads to the right of the window by the zero
 * padding of the coefficient row (coeff_stride, ref JincResize.cpp:290,:476): a no-op for finite samples, which is what
 * is restated here (for non-finite floats right of the window the reference's own result depends on memory outside
 * the window and is not defined by the algorithm).
 */
#include <math.h>
#include <stdint.h>
#include <limits.h>

#include "jinc_oracle.h"

static float fetch_f32(float v, float min_val) { return v > min_val ? v : min_val; } /* _mm_max_ps(src, min_val) */

static int32_t cvtps_epi32(float v)
{
    if (!(v >= -2147483648.0f && v < 2147483648.0f))
        return INT32_MIN; /* "integer indefinite" */
    return (int32_t)lrintf(v);
}

static uint16_t packus_epi32(int32_t v) { return (uint16_t)(v < 0 ? 0 : (v > 65535 ? 65535 : v)); }

static uint8_t packus_epi16(uint16_t w)
{
    const int16_t s = (int16_t)w;
    return (uint8_t)(s < 0 ? 0 : (s > 255 ? 255 : s));
}

static float tree_sum(const float *part, int lanes)
{
    float q[8], h[4];
    int i;
    if (lanes == 16)
        for (i = 0; i < 8; ++i)
            q[i] = part[i] + part[i + 8];
    else
        for (i = 0; i < 8; ++i)
            q[i] = part[i];
    if (lanes >= 8)
        for (i = 0; i < 4; ++i)
            h[i] = q[i] + q[i + 4];
    else
        for (i = 0; i < 4; ++i)
            h[i] = q[i];
    return (h[0] + h[1]) + (h[2] + h[3]);
}

#define DEFINE_SIMD_ROW(NAME, T, FETCH)                                                                          \
    static float NAME(const T *sp, int src_stride, const float *cp, int fs, int cs, int lanes, int fused, float min_val) \
    {                                                                                                            \
        float part[16] = {0};                                                                                    \
        int ly, lx;                                                                                              \
        (void)min_val;                                                                                           \
        for (ly = 0; ly < fs; ++ly) {                                                                            \
            for (lx = 0; lx < fs; ++lx) {                                                                        \
                const float s = FETCH;                                                                           \
                float *p = &part[lx % lanes];                                                                    \
                if (fused)                                                                                       \
                    *p = fmaf(s, cp[lx], *p);                                                                    \
                else                                                                                             \
                    *p = *p + s * cp[lx];                                                                        \
            }                                                                                                    \
            cp += cs;                                                                                            \
            sp += src_stride;                                                                                    \
        }                                                                                                        \
        return tree_sum(part, lanes);                                                                            \
    }

DEFINE_SIMD_ROW(sum_u8, uint8_t, (float)sp[lx])
DEFINE_SIMD_ROW(sum_u16, uint16_t, (float)sp[lx])
DEFINE_SIMD_ROW(sum_f32, float, fetch_f32(sp[lx], min_val))

/* order: 1 = SSE4.1 (4 lanes, un-fused), 2 = AVX2 (8 lanes, FMA), 3 = AVX-512 (16 lanes, FMA) */
void oracle_resize_plane_simd(int order, const void *src, int src_pitch, void *dst, int dst_pitch, const oracle_table *t,
                              int sample_bytes, float min_val, int threads)
{
    const int lanes = order == 1 ? 4 : (order == 2 ? 8 : 16), fused = order != 1;
    const int fs = t->filter_size, cs = t->coeff_stride, w = t->dst_width, h = t->dst_height;
    int y;
#ifdef _OPENMP
    if (threads < 1)
        threads = 1;
#pragma omp parallel for num_threads(threads) schedule(static) if (threads > 1)
#else
    (void)threads;
#endif
    for (y = 0; y < h; ++y) {
        int x;
        for (x = 0; x < w; ++x) {
            const oracle_meta *m = t->meta + (int64_t)y * w + x;
            const float *cp = t->factor + m->coeff_meta;
            if (sample_bytes == 1) {
                const uint8_t *sp = (const uint8_t *)src + m->start_y * (int64_t)src_pitch + m->start_x;
                const float r = sum_u8(sp, src_pitch, cp, fs, cs, lanes, fused, min_val);
                ((uint8_t *)((char *)dst + (int64_t)y * dst_pitch))[x] = packus_epi16(packus_epi32(cvtps_epi32(r)));
            } else if (sample_bytes == 2) {
                const uint16_t *sp = (const uint16_t *)src + m->start_y * (int64_t)(src_pitch / 2) + m->start_x;
                const float r = sum_u16(sp, src_pitch / 2, cp, fs, cs, lanes, fused, min_val);
                ((uint16_t *)((char *)dst + (int64_t)y * dst_pitch))[x] = packus_epi32(cvtps_epi32(r));
            } else {
                const float *sp = (const float *)src + m->start_y * (int64_t)(src_pitch / 4) + m->start_x;
                ((float *)((char *)dst + (int64_t)y * dst_pitch))[x] = sum_f32(sp, src_pitch / 4, cp, fs, cs, lanes, fused, min_val);
            }
        }
    }
}
