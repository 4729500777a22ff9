/*
 * simd_avx512.c -- TEST / BENCH INFRASTRUCTURE ONLY: an own-written AVX-512 implementation of the per-frame path in the
 * summation order of the reference's opt = 3 code (/root/reference/src/resize_plane_avx512.cpp:45-100 is the order to match:
 * 16 lanes across lx with fused multiply-add into ONE 512-bit partial-sum register over all kernel rows, then the fold
 * 512 -> 256 (low half + high half), 256 -> 128 (low + high), hadd of hadd; integer samples leave through cvtps_epi32 and
 * packus, i.e. saturated to the type's range, not to the clip's peak; float samples are clamped from below at 0 / -0.5 first).
 * It is NOT the reference and NOT bit-equal to opt = 0; it is bit-equal to oracle_resize_plane_simd(order = 3)
 * (tests/test_simd_order.py) and is timed beside the AVX2-order code as the second fast CPU baseline of bench.py
 * (north_star: "the reference AVX2/AVX-512 path timed on the same box's host cores"), because the reference itself cannot be
 * built on the GPU box.  Compile with -mavx512f -mavx512bw -mavx512dq -mavx512vl -mfma (this file only; the reference's flags
 * for its AVX-512 file, CMakeLists.txt:60).
 *
 * Round 6: as in simd_avx2.c the pixel loop is one plain function per sample type over a range of rows
 * (loop invariants in const locals, `restrict` destination row, the row's tail mask computed once per call), not the body of an
 * OpenMP-outlined loop with a per-pixel branch on the sample size.  Compile with -O3.
 *
 * Unlike the reference (avx512:51, :71: a full 16-sample load per group, reading past the window and, in the plane's last rows,
 * past the plane) the loads here stay inside the window's rows: a row's last group is fetched under a lane mask that covers
 * exactly the samples the window has; the masked-off lanes are zero and meet the zero padding of the coefficient row.
 */
#include <immintrin.h>
#include <stddef.h>
#include <stdint.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "jinc_oracle.h"

static inline __mmask16 first_lanes(int n) { return n >= 16 ? (__mmask16)0xFFFF : (__mmask16)((1u << n) - 1u); }

/* 512 -> 256 -> 128 -> one value in every lane (avx512:60-62) */
static inline __m128 fold(__m512 r)
{
    const __m256 q = _mm256_add_ps(_mm512_castps512_ps256(r), _mm256_castpd_ps(_mm512_extractf64x4_pd(_mm512_castps_pd(r), 1)));
    const __m128 h = _mm_add_ps(_mm256_castps256_ps128(q), _mm256_extractf128_ps(q, 1));
    return _mm_hadd_ps(_mm_hadd_ps(h, h), _mm_hadd_ps(h, h));
}

int oracle_avx512_available(void)
{
    return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512dq") &&
           __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("fma");
}

/* Rows [y0, y1) of one plane, pitches in samples.  A kernel row is `whole_taps` taps in whole groups of 16 and one last group
 * under `tail` (the lanes the window has; masked-off lanes are zero and meet the zero padding of the coefficient row).  Two
 * neighbouring pixels are computed side by side, each with its own accumulator and order (see simd_avx2.c). */
#define ROWS_FN(NAME, T, LOAD_FULL, LOAD_PART, OUT)                                                                                     \
    static __attribute__((noinline)) void NAME(const T *restrict src, ptrdiff_t pitch, T *restrict dst, ptrdiff_t dpitch,                \
                                               const oracle_meta *restrict meta, const float *restrict factor, int fs, int cs, int w,   \
                                               __m512 min_val, int y0, int y1)                                                          \
    {                                                                                                                                   \
        const int whole_taps = fs & ~15, rem = fs & 15;                                                                                 \
        const __mmask16 tail = rem ? first_lanes(rem) : (__mmask16)0; /* last group of a row of more than 16 taps */                    \
        const __mmask16 only = first_lanes(fs);                       /* the only group of a row of at most 16 taps */                  \
        (void)min_val;                                                                                                                  \
        for (int y = y0; y < y1; ++y) {                                                                                                 \
            const oracle_meta *restrict m = meta + (int64_t)y * w;                                                                      \
            T *restrict drow = dst + y * dpitch;                                                                                        \
            int x = 0;                                                                                                                  \
            for (; x + 1 < w; x += 2) {                                                                                                 \
                /* coefficient rows: 64-byte aligned, zero padded to coeff_stride (ref JincResize.cpp:290, :476) */                     \
                const float *c0 = factor + m[x].coeff_meta, *c1 = factor + m[x + 1].coeff_meta;                                         \
                const T *s0 = src + m[x].start_y * pitch + m[x].start_x, *s1 = src + m[x + 1].start_y * pitch + m[x + 1].start_x;       \
                __m512 a0 = _mm512_setzero_ps(), a1 = _mm512_setzero_ps();                                                              \
                if (fs <= 16) { /* one group per kernel row: taps 1 ... 7 at >= 1x */                                                   \
                    for (int ly = 0; ly < fs; ++ly, c0 += cs, c1 += cs, s0 += pitch, s1 += pitch) {                                     \
                        a0 = _mm512_fmadd_ps(LOAD_PART(only, s0), _mm512_load_ps(c0), a0);                                              \
                        a1 = _mm512_fmadd_ps(LOAD_PART(only, s1), _mm512_load_ps(c1), a1);                                              \
                    }                                                                                                                   \
                } else {                                                                                                                \
                    for (int ly = 0; ly < fs; ++ly, c0 += cs, c1 += cs, s0 += pitch, s1 += pitch) {                                     \
                        int lx = 0;                                                                                                     \
                        for (; lx < whole_taps; lx += 16) {                                                                             \
                            a0 = _mm512_fmadd_ps(LOAD_FULL(s0 + lx), _mm512_load_ps(c0 + lx), a0);                                      \
                            a1 = _mm512_fmadd_ps(LOAD_FULL(s1 + lx), _mm512_load_ps(c1 + lx), a1);                                      \
                        }                                                                                                               \
                        if (tail) {                                                                                                     \
                            a0 = _mm512_fmadd_ps(LOAD_PART(tail, s0 + lx), _mm512_load_ps(c0 + lx), a0);                                \
                            a1 = _mm512_fmadd_ps(LOAD_PART(tail, s1 + lx), _mm512_load_ps(c1 + lx), a1);                                \
                        }                                                                                                               \
                    }                                                                                                                   \
                }                                                                                                                       \
                drow[x] = OUT(a0);                                                                                                      \
                drow[x + 1] = OUT(a1);                                                                                                  \
            }                                                                                                                           \
            for (; x < w; ++x) { /* an odd last pixel */                                                                                \
                const float *cp = factor + m[x].coeff_meta;                                                                             \
                const T *sp = src + m[x].start_y * pitch + m[x].start_x;                                                                \
                __m512 acc = _mm512_setzero_ps();                                                                                       \
                for (int ly = 0; ly < fs; ++ly, cp += cs, sp += pitch) {                                                                \
                    int lx = 0;                                                                                                         \
                    for (; lx < whole_taps; lx += 16)                                                                                   \
                        acc = _mm512_fmadd_ps(LOAD_FULL(sp + lx), _mm512_load_ps(cp + lx), acc);                                        \
                    if (tail)                                                                                                           \
                        acc = _mm512_fmadd_ps(LOAD_PART(tail, sp + lx), _mm512_load_ps(cp + lx), acc);                                  \
                }                                                                                                                       \
                drow[x] = OUT(acc);                                                                                                     \
            }                                                                                                                           \
        }                                                                                                                               \
    }

#define FULL_U8(p) _mm512_cvtepi32_ps(_mm512_cvtepu8_epi32(_mm_loadu_si128((const __m128i *)(p))))
#define PART_U8(k, p) _mm512_cvtepi32_ps(_mm512_cvtepu8_epi32(_mm_maskz_loadu_epi8((k), (p))))
#define FULL_U16(p) _mm512_cvtepi32_ps(_mm512_cvtepu16_epi32(_mm256_loadu_si256((const __m256i *)(p))))
#define PART_U16(k, p) _mm512_cvtepi32_ps(_mm512_cvtepu16_epi32(_mm256_maskz_loadu_epi16((k), (p))))
/* the lower clamp first, as the reference does (avx512:91); surplus lanes become max(0, min_val) = 0 (min_val <= 0) */
#define FULL_F32(p) _mm512_max_ps(_mm512_loadu_ps(p), min_val)
#define PART_F32(k, p) _mm512_max_ps(_mm512_maskz_loadu_ps((k), (p)), min_val)

static inline uint8_t out_u8(__m512 acc)
{
    const __m128i zero = _mm_setzero_si128();
    return (uint8_t)_mm_cvtsi128_si32(_mm_packus_epi16(_mm_packus_epi32(_mm_cvtps_epi32(fold(acc)), zero), zero));
}
static inline uint16_t out_u16(__m512 acc) { return (uint16_t)_mm_cvtsi128_si32(_mm_packus_epi32(_mm_cvtps_epi32(fold(acc)), _mm_setzero_si128())); }
static inline float out_f32(__m512 acc) { return _mm_cvtss_f32(fold(acc)); }

ROWS_FN(rows_u8, uint8_t, FULL_U8, PART_U8, out_u8)
ROWS_FN(rows_u16, uint16_t, FULL_U16, PART_U16, out_u16)
ROWS_FN(rows_f32, float, FULL_F32, PART_F32, out_f32)

void oracle_resize_plane_avx512(const void *src, int src_pitch, size_t src_bytes, void *dst, int dst_pitch, const oracle_table *t,
                                int sample_bytes, float min_val_f, int threads)
{
    const int fs = t->filter_size, cs = t->coeff_stride, w = t->dst_width, h = t->dst_height;
    const __m512 min_val = _mm512_set1_ps(min_val_f);
    (void)src_bytes; /* every row's last group is masked to the window and whole groups lie inside it: the plane's end never matters */
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
            rows_u8((const uint8_t *)src, src_pitch, (uint8_t *)dst, dst_pitch, t->meta, t->factor, fs, cs, w, min_val, y0, y1);
        else if (sample_bytes == 2)
            rows_u16((const uint16_t *)src, src_pitch / 2, (uint16_t *)dst, dst_pitch / 2, t->meta, t->factor, fs, cs, w, min_val, y0, y1);
        else
            rows_f32((const float *)src, src_pitch / 4, (float *)dst, dst_pitch / 4, t->meta, t->factor, fs, cs, w, min_val, y0, y1);
    }
}
