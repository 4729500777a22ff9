/*
 * simd_avx2.c -- TEST / BENCH INFRASTRUCTURE ONLY: an own-written AVX2 + FMA implementation of the per-frame path in
 * the summation order of the reference's opt = 2 code (/root/reference/src/resize_plane_avx2.cpp:45-98 is the order to
 * match: 8 lanes across lx, fused multiply-add, 256 -> 128 fold, hadd of hadd).  It is NOT the reference and NOT
 * bit-equal to opt = 0; it is bit-equal to oracle_resize_plane_simd(order = 2) (tests/test_simd_order.py) and serves as
 * the fast CPU baseline SURVEY.md 8(d) allows next to the GPU number (bench.py cpu_baseline), because the reference
 * itself cannot be built on the GPU box.  Compile with -mavx2 -mfma (this file only).
 *
 * Unlike the reference (avx2:51: a 16-byte load per 8 samples, reading up to 9 bytes past the window) the loads here stay
 * inside the window's rows: the last group of a row is assembled from exactly the samples the window has.
 */
#include <immintrin.h>
#include <stdint.h>
#include <string.h>

#include "jinc_oracle.h"

/* Groups of 8 taps.  A row's last group holds n < 8 samples of the window; the reference loads past the window and lets
 * the zero padding of the coefficient row cancel the surplus lanes.  Here the full-width load is used only while it stays
 * inside the plane's allocation (`end`), i.e. everywhere but in the last bytes of the last row; integer surplus lanes are
 * finite, so the zero coefficients cancel them exactly.  Float surplus lanes are masked off (an infinity there would
 * turn 0 * inf into NaN). */
static inline __m256 load8_u8(const uint8_t *p, int n, const uint8_t *end)
{
    if (n >= 8 || p + 8 <= end)
        return _mm256_cvtepi32_ps(_mm256_cvtepu8_epi32(_mm_loadl_epi64((const __m128i *)p)));
    uint8_t tmp[8] = {0};
    memcpy(tmp, p, (size_t)n);
    return _mm256_cvtepi32_ps(_mm256_cvtepu8_epi32(_mm_loadl_epi64((const __m128i *)tmp)));
}

static inline __m256 load8_u16(const uint16_t *p, int n, const uint8_t *end)
{
    if (n >= 8 || (const uint8_t *)(p + 8) <= end)
        return _mm256_cvtepi32_ps(_mm256_cvtepu16_epi32(_mm_loadu_si128((const __m128i *)p)));
    uint16_t tmp[8] = {0};
    memcpy(tmp, p, (size_t)n * 2);
    return _mm256_cvtepi32_ps(_mm256_cvtepu16_epi32(_mm_loadu_si128((const __m128i *)tmp)));
}

static inline __m256 load8_f32(const float *p, int n, __m256 min_val)
{
    static const int32_t mask[16] = {-1, -1, -1, -1, -1, -1, -1, -1, 0, 0, 0, 0, 0, 0, 0, 0};
    if (n >= 8)
        return _mm256_max_ps(_mm256_loadu_ps(p), min_val);
    return _mm256_max_ps(_mm256_maskload_ps(p, _mm256_loadu_si256((const __m256i *)(mask + 8 - n))), min_val);
}

static inline __m128 hsum(__m256 r)
{
    __m128 h = _mm_add_ps(_mm256_castps256_ps128(r), _mm256_extractf128_ps(r, 1));
    return _mm_hadd_ps(_mm_hadd_ps(h, h), _mm_hadd_ps(h, h));
}

int oracle_avx2_available(void) { return __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma"); }

void oracle_resize_plane_avx2(const void *src, int src_pitch, size_t src_bytes, void *dst, int dst_pitch, const oracle_table *t,
                              int sample_bytes, float min_val_f, int threads)
{
    const uint8_t *end = (const uint8_t *)src + src_bytes;
    const int fs = t->filter_size, cs = t->coeff_stride, w = t->dst_width, h = t->dst_height;
    const __m256 min_val = _mm256_set1_ps(min_val_f);
    int y;
#ifdef _OPENMP
    if (threads < 1)
        threads = 1;
#pragma omp parallel for num_threads(threads) schedule(static) if (threads > 1)
#else
    (void)threads;
#endif
    for (y = 0; y < h; ++y) {
        char *drow = (char *)dst + (int64_t)y * dst_pitch;
        int x;
        for (x = 0; x < w; ++x) {
            const oracle_meta *m = t->meta + (int64_t)y * w + x;
            const float *cp = t->factor + m->coeff_meta; /* 64-byte aligned rows, zero padded (ref JincResize.cpp:290,:476) */
            __m256 acc = _mm256_setzero_ps();
            int ly, lx;
            if (sample_bytes == 1) {
                const uint8_t *sp = (const uint8_t *)src + m->start_y * (int64_t)src_pitch + m->start_x;
                for (ly = 0; ly < fs; ++ly, cp += cs, sp += src_pitch)
                    for (lx = 0; lx < fs; lx += 8)
                        acc = _mm256_fmadd_ps(load8_u8(sp + lx, fs - lx, end), _mm256_loadu_ps(cp + lx), acc);
                ((uint8_t *)drow)[x] = (uint8_t)_mm_cvtsi128_si32(_mm_packus_epi16(
                    _mm_packus_epi32(_mm_cvtps_epi32(hsum(acc)), _mm_setzero_si128()), _mm_setzero_si128()));
            } else if (sample_bytes == 2) {
                const uint16_t *sp = (const uint16_t *)src + m->start_y * (int64_t)(src_pitch / 2) + m->start_x;
                for (ly = 0; ly < fs; ++ly, cp += cs, sp += src_pitch / 2)
                    for (lx = 0; lx < fs; lx += 8)
                        acc = _mm256_fmadd_ps(load8_u16(sp + lx, fs - lx, end), _mm256_loadu_ps(cp + lx), acc);
                ((uint16_t *)drow)[x] =
                    (uint16_t)_mm_cvtsi128_si32(_mm_packus_epi32(_mm_cvtps_epi32(hsum(acc)), _mm_setzero_si128()));
            } else {
                const float *sp = (const float *)src + m->start_y * (int64_t)(src_pitch / 4) + m->start_x;
                for (ly = 0; ly < fs; ++ly, cp += cs, sp += src_pitch / 4)
                    for (lx = 0; lx < fs; lx += 8)
                        acc = _mm256_fmadd_ps(load8_f32(sp + lx, fs - lx, min_val), _mm256_loadu_ps(cp + lx), acc);
                ((float *)drow)[x] = _mm_cvtss_f32(hsum(acc));
            }
        }
    }
}
