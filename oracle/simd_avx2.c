/*
 * simd_avx2.c -- TEST / BENCH INFRASTRUCTURE ONLY: an own-written AVX2 + FMA implementation of the per-frame path in
 * the summation order of the reference's opt = 2 code (/root/reference/src/resize_plane_avx2.cpp:45-98 is the order to
 * match: 8 lanes across lx, fused multiply-add, 256 -> 128 fold, hadd of hadd).  It is NOT the reference and NOT
 * bit-equal to opt = 0; it is bit-equal to oracle_resize_plane_simd(order = 2) (tests/test_simd_order.py) and serves as
 * the fast CPU baseline SURVEY.md 8(d) allows next to the GPU number (bench.py cpu_baseline), because the reference
 * itself cannot be built on the GPU box.  Compile with -O3 -mavx2 -mfma (this file only).
 *
 * Round 6 (the port ran at 0.43 of the reference's own AVX2 code per thread): the pixel loop is now
 * one plain function per sample type over a range of rows -- every loop invariant a const local, the destination row
 * `restrict`, nothing captured by an OpenMP-outlined body (whose captured scalars the compiler had to reload after
 * every store through a char pointer) -- and the question "may this row's last group be fetched with a full-width
 * load?" is answered once per pixel from its window's last row instead of once per group of taps.  Two neighbouring
 * pixels are computed side by side: each keeps its own accumulator and its own summation order (the results are those of
 * oracle_resize_plane_simd(order = 2), bit for bit), the two dependency chains of fused multiply-adds merely overlap in
 * the pipeline -- which the reference's one-pixel loop leaves to the out-of-order window, so this code runs FASTER than
 * the reference's own (BASELINE.md section 3 has both figures from the build container); as a baseline beside the GPU
 * number that errs on the safe side.
 *
 * Unlike the reference (avx2:51: a 16-byte load per 8 samples, reading up to 9 bytes past the window and, in the
 * plane's last row, past the plane) the loads here stay inside the plane: wherever a full-width load of a row's last
 * group could leave the plane's allocation (only in windows that touch the plane's last row), that group is assembled
 * from exactly the samples the window has.  Integer surplus lanes are finite, so the zero padding of the coefficient
 * row cancels them exactly; float surplus lanes are masked off everywhere (an infinity there would turn 0 * inf into
 * NaN, which oracle_resize_plane_simd -- the definition this file is held to -- does not produce).
 */
#include <immintrin.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "jinc_oracle.h"

#define LIKELY(x) __builtin_expect(!!(x), 1)

static inline __m128 hsum(__m256 r)
{
    __m128 h = _mm_add_ps(_mm256_castps256_ps128(r), _mm256_extractf128_ps(r, 1));
    return _mm_hadd_ps(_mm_hadd_ps(h, h), _mm_hadd_ps(h, h));
}

static inline __m256 full_u8(const uint8_t *p) { return _mm256_cvtepi32_ps(_mm256_cvtepu8_epi32(_mm_loadl_epi64((const __m128i *)p))); }
static inline __m256 full_u16(const uint16_t *p) { return _mm256_cvtepi32_ps(_mm256_cvtepu16_epi32(_mm_loadu_si128((const __m128i *)p))); }

/* the last group of a kernel row, `n` < 8 samples, without touching anything past them */
static inline __m256 part_u8(const uint8_t *p, int n)
{
    uint8_t tmp[8] = {0};
    memcpy(tmp, p, (size_t)n);
    return full_u8(tmp);
}

static inline __m256 part_u16(const uint16_t *p, int n)
{
    uint16_t tmp[8] = {0};
    memcpy(tmp, p, (size_t)n * 2);
    return full_u16(tmp);
}

static inline uint8_t out_u8(__m256 acc)
{
    const __m128i zero = _mm_setzero_si128();
    return (uint8_t)_mm_cvtsi128_si32(_mm_packus_epi16(_mm_packus_epi32(_mm_cvtps_epi32(hsum(acc)), zero), zero));
}

static inline uint16_t out_u16(__m256 acc)
{
    return (uint16_t)_mm_cvtsi128_si32(_mm_packus_epi32(_mm_cvtps_epi32(hsum(acc)), _mm_setzero_si128()));
}

int oracle_avx2_available(void) { return __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma"); }

/* Rows [y0, y1) of one integer plane; `pitch` / `dpitch` in samples.  `safe_y`: the largest window origin row whose windows
 * may use full-width loads in every kernel row (the surplus lanes of a row's last group then lie in the same or the next
 * row of the plane); windows that reach further down assemble their last group from the samples they have. */
#define INT_ROWS_FN(NAME, T, FULL, PART, OUT)                                                                                          \
    static __attribute__((noinline)) void NAME(const T *restrict src, ptrdiff_t pitch, int safe_y, T *restrict dst, ptrdiff_t dpitch,   \
                                               const oracle_meta *restrict meta, const float *restrict factor, int fs, int cs, int w,  \
                                               int y0, int y1)                                                                         \
    {                                                                                                                                  \
        const int whole = fs & ~7, rem = fs & 7; /* taps in whole groups of 8; taps of the last group */                               \
        for (int y = y0; y < y1; ++y) {                                                                                                \
            const oracle_meta *restrict m = meta + (int64_t)y * w;                                                                     \
            T *restrict drow = dst + y * dpitch;                                                                                       \
            int x = 0;                                                                                                                 \
            for (; x + 1 < w && LIKELY(m[x].start_y <= safe_y && m[x + 1].start_y <= safe_y); x += 2) {                                \
                /* coefficient rows: 64-byte aligned, zero padded to coeff_stride (ref JincResize.cpp:290, :476) */                    \
                const float *c0 = factor + m[x].coeff_meta, *c1 = factor + m[x + 1].coeff_meta;                                        \
                const T *s0 = src + m[x].start_y * pitch + m[x].start_x, *s1 = src + m[x + 1].start_y * pitch + m[x + 1].start_x;      \
                __m256 a0 = _mm256_setzero_ps(), a1 = _mm256_setzero_ps();                                                             \
                if (fs <= 8) { /* one group per kernel row (taps 1 ... 4 at >= 1x) */                                                  \
                    for (int ly = 0; ly < fs; ++ly, c0 += cs, c1 += cs, s0 += pitch, s1 += pitch) {                                    \
                        a0 = _mm256_fmadd_ps(FULL(s0), _mm256_load_ps(c0), a0);                                                        \
                        a1 = _mm256_fmadd_ps(FULL(s1), _mm256_load_ps(c1), a1);                                                        \
                    }                                                                                                                  \
                } else {                                                                                                               \
                    for (int ly = 0; ly < fs; ++ly, c0 += cs, c1 += cs, s0 += pitch, s1 += pitch)                                      \
                        for (int lx = 0; lx < fs; lx += 8) {                                                                           \
                            a0 = _mm256_fmadd_ps(FULL(s0 + lx), _mm256_load_ps(c0 + lx), a0);                                          \
                            a1 = _mm256_fmadd_ps(FULL(s1 + lx), _mm256_load_ps(c1 + lx), a1);                                          \
                        }                                                                                                              \
                }                                                                                                                      \
                drow[x] = OUT(a0);                                                                                                     \
                drow[x + 1] = OUT(a1);                                                                                                 \
            }                                                                                                                          \
            for (; x < w; ++x) { /* windows on the plane's last row, and an odd last pixel */                                          \
                const float *cp = factor + m[x].coeff_meta;                                                                            \
                const T *sp = src + m[x].start_y * pitch + m[x].start_x;                                                               \
                __m256 acc = _mm256_setzero_ps();                                                                                      \
                for (int ly = 0; ly < fs; ++ly, cp += cs, sp += pitch) {                                                               \
                    int lx = 0;                                                                                                        \
                    for (; lx < whole; lx += 8)                                                                                        \
                        acc = _mm256_fmadd_ps(FULL(sp + lx), _mm256_load_ps(cp + lx), acc);                                            \
                    if (rem)                                                                                                           \
                        acc = _mm256_fmadd_ps(PART(sp + lx, rem), _mm256_load_ps(cp + lx), acc);                                       \
                }                                                                                                                      \
                drow[x] = OUT(acc);                                                                                                    \
            }                                                                                                                          \
        }                                                                                                                              \
    }

INT_ROWS_FN(rows_u8, uint8_t, full_u8, part_u8, out_u8)
INT_ROWS_FN(rows_u16, uint16_t, full_u16, part_u16, out_u16)

/* Float planes: the lower clamp first (avx2:90), the last group of every kernel row under a lane mask (see the header). */
static __attribute__((noinline)) void rows_f32(const float *restrict src, ptrdiff_t pitch, float *restrict dst, ptrdiff_t dpitch,
                                               const oracle_meta *restrict meta, const float *restrict factor, int fs, int cs, int w,
                                               float min_val_f, int y0, int y1)
{
    static const int32_t mask_src[16] = {-1, -1, -1, -1, -1, -1, -1, -1, 0, 0, 0, 0, 0, 0, 0, 0};
    const int whole = fs & ~7, rem = fs & 7;
    const __m256 min_val = _mm256_set1_ps(min_val_f);
    const __m256i part = _mm256_loadu_si256((const __m256i *)(mask_src + 8 - rem)); /* the first `rem` lanes */
    for (int y = y0; y < y1; ++y) {
        const oracle_meta *restrict m = meta + (int64_t)y * w;
        float *restrict drow = dst + y * dpitch;
        int x = 0;
        for (; x + 1 < w; x += 2) {
            const float *c0 = factor + m[x].coeff_meta, *c1 = factor + m[x + 1].coeff_meta;
            const float *s0 = src + m[x].start_y * pitch + m[x].start_x, *s1 = src + m[x + 1].start_y * pitch + m[x + 1].start_x;
            __m256 a0 = _mm256_setzero_ps(), a1 = _mm256_setzero_ps();
            for (int ly = 0; ly < fs; ++ly, c0 += cs, c1 += cs, s0 += pitch, s1 += pitch) {
                int lx = 0;
                for (; lx < whole; lx += 8) {
                    a0 = _mm256_fmadd_ps(_mm256_max_ps(_mm256_loadu_ps(s0 + lx), min_val), _mm256_load_ps(c0 + lx), a0);
                    a1 = _mm256_fmadd_ps(_mm256_max_ps(_mm256_loadu_ps(s1 + lx), min_val), _mm256_load_ps(c1 + lx), a1);
                }
                if (rem) { /* masked-off lanes read nothing and are +0 before the clamp: max(0, min_val) = 0 for min_val <= 0 */
                    a0 = _mm256_fmadd_ps(_mm256_max_ps(_mm256_maskload_ps(s0 + lx, part), min_val), _mm256_load_ps(c0 + lx), a0);
                    a1 = _mm256_fmadd_ps(_mm256_max_ps(_mm256_maskload_ps(s1 + lx, part), min_val), _mm256_load_ps(c1 + lx), a1);
                }
            }
            drow[x] = _mm_cvtss_f32(hsum(a0));
            drow[x + 1] = _mm_cvtss_f32(hsum(a1));
        }
        for (; x < w; ++x) {
            const float *cp = factor + m[x].coeff_meta;
            const float *sp = src + m[x].start_y * pitch + m[x].start_x;
            __m256 acc = _mm256_setzero_ps();
            for (int ly = 0; ly < fs; ++ly, cp += cs, sp += pitch) {
                int lx = 0;
                for (; lx < whole; lx += 8)
                    acc = _mm256_fmadd_ps(_mm256_max_ps(_mm256_loadu_ps(sp + lx), min_val), _mm256_load_ps(cp + lx), acc);
                if (rem)
                    acc = _mm256_fmadd_ps(_mm256_max_ps(_mm256_maskload_ps(sp + lx, part), min_val), _mm256_load_ps(cp + lx), acc);
            }
            drow[x] = _mm_cvtss_f32(hsum(acc));
        }
    }
}

void oracle_resize_plane_avx2(const void *src, int src_pitch, size_t src_bytes, void *dst, int dst_pitch, const oracle_table *t,
                              int sample_bytes, float min_val_f, int threads)
{
    const int fs = t->filter_size, cs = t->coeff_stride, w = t->dst_width, h = t->dst_height;
    /* A window whose last row is not the plane's last row keeps the surplus lanes of a full-width load (< 8 samples past
     * the window's right edge) inside the plane: they fall into the row's own tail or the start of the next row. */
    const int src_rows = (int)(src_bytes / (size_t)src_pitch);
    const int safe_y = src_rows - fs - 1;
    if (threads < 1)
        threads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(threads) if (threads > 1)
    {
        const int nt = omp_get_num_threads(), id = omp_get_thread_num();
#else
    {
        const int nt = 1, id = 0;
#endif
        const int y0 = (int)((int64_t)h * id / nt), y1 = (int)((int64_t)h * (id + 1) / nt);
        if (sample_bytes == 1)
            rows_u8((const uint8_t *)src, src_pitch, safe_y, (uint8_t *)dst, dst_pitch, t->meta, t->factor, fs, cs, w, y0, y1);
        else if (sample_bytes == 2)
            rows_u16((const uint16_t *)src, src_pitch / 2, safe_y, (uint16_t *)dst, dst_pitch / 2, t->meta, t->factor, fs, cs, w, y0, y1);
        else
            rows_f32((const float *)src, src_pitch / 4, (float *)dst, dst_pitch / 4, t->meta, t->factor, fs, cs, w, min_val_f, y0, y1);
    }
}
