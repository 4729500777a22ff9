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
 * Unlike the reference (avx512:51, :71: a full 16-sample load per group, reading past the window and, in the plane's last rows,
 * past the plane) the loads here stay inside the window's rows: a row's last group is fetched under a lane mask that covers
 * exactly the samples the window has; the masked-off lanes are zero and meet the zero padding of the coefficient row.
 */
#include <immintrin.h>
#include <stdint.h>

#include "jinc_oracle.h"

static inline __mmask16 first_lanes(int n) { return n >= 16 ? (__mmask16)0xFFFF : (__mmask16)((1u << n) - 1u); }

static inline __m512 load16_u8(const uint8_t *p, int n)
{
    return _mm512_cvtepi32_ps(_mm512_cvtepu8_epi32(_mm_maskz_loadu_epi8(first_lanes(n), p)));
}

static inline __m512 load16_u16(const uint16_t *p, int n)
{
    return _mm512_cvtepi32_ps(_mm512_cvtepu16_epi32(_mm256_maskz_loadu_epi16(first_lanes(n), p)));
}

static inline __m512 load16_f32(const float *p, int n, __m512 min_val)
{
    /* the lower clamp first, as the reference does (avx512:91); surplus lanes become max(0, min_val) = 0 (min_val <= 0) */
    return _mm512_max_ps(_mm512_maskz_loadu_ps(first_lanes(n), p), min_val);
}

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

void oracle_resize_plane_avx512(const void *src, int src_pitch, size_t src_bytes, void *dst, int dst_pitch, const oracle_table *t,
                                int sample_bytes, float min_val_f, int threads)
{
    const int fs = t->filter_size, cs = t->coeff_stride, w = t->dst_width, h = t->dst_height;
    const __m512 min_val = _mm512_set1_ps(min_val_f);
    int y;
    (void)src_bytes; /* every load is masked to the window: the plane's end never matters */
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
            const float *cp = t->factor + m->coeff_meta; /* rows of coeff_stride floats, zero padded (ref JincResize.cpp:290,:476) */
            __m512 acc = _mm512_setzero_ps();
            int ly, lx;
            if (sample_bytes == 1) {
                const uint8_t *sp = (const uint8_t *)src + m->start_y * (int64_t)src_pitch + m->start_x;
                for (ly = 0; ly < fs; ++ly, cp += cs, sp += src_pitch)
                    for (lx = 0; lx < fs; lx += 16)
                        acc = _mm512_fmadd_ps(load16_u8(sp + lx, fs - lx), _mm512_loadu_ps(cp + lx), acc);
                ((uint8_t *)drow)[x] = (uint8_t)_mm_cvtsi128_si32(
                    _mm_packus_epi16(_mm_packus_epi32(_mm_cvtps_epi32(fold(acc)), _mm_setzero_si128()), _mm_setzero_si128()));
            } else if (sample_bytes == 2) {
                const uint16_t *sp = (const uint16_t *)src + m->start_y * (int64_t)(src_pitch / 2) + m->start_x;
                for (ly = 0; ly < fs; ++ly, cp += cs, sp += src_pitch / 2)
                    for (lx = 0; lx < fs; lx += 16)
                        acc = _mm512_fmadd_ps(load16_u16(sp + lx, fs - lx), _mm512_loadu_ps(cp + lx), acc);
                ((uint16_t *)drow)[x] = (uint16_t)_mm_cvtsi128_si32(_mm_packus_epi32(_mm_cvtps_epi32(fold(acc)), _mm_setzero_si128()));
            } else {
                const float *sp = (const float *)src + m->start_y * (int64_t)(src_pitch / 4) + m->start_x;
                for (ly = 0; ly < fs; ++ly, cp += cs, sp += src_pitch / 4)
                    for (lx = 0; lx < fs; lx += 16)
                        acc = _mm512_fmadd_ps(load16_f32(sp + lx, fs - lx, min_val), _mm512_loadu_ps(cp + lx), acc);
                ((float *)drow)[x] = _mm_cvtss_f32(fold(acc));
            }
        }
    }
}
