/*
 * jinc_oracle.c -- TEST INFRASTRUCTURE ONLY (see jinc_oracle.h).
 *
 * Plain-C restatement of the reference's opt=0 path.  Every function cites the reference lines
 * it follows ("ref:" = /root/reference/src/JincResize.cpp).  Compile with
 *   gcc -O2 -ffp-contract=off   (no -march / -mfma / -ffast-math: the reference TU has none,
 *                                CMakeLists.txt:57-61 puts ISA flags on the SIMD files only)
 *
 * Third-party arithmetic outside /root/reference: std::cyl_bessel_j (libstdc++ of GCC 11.4) is
 * called by the reference for LUT arguments x^2 >= 17.99 (tap >= 5; ref :231-244).  It is reached
 * through oracle_cyl_bessel_j1() in bessel_shim.cpp -- the same library routine the reference
 * binds.  Answers to reference-derived known answers in tests/golden/kat.json: SURVEY.md 8(c)'s (taps 3, 4, 8) and the
 * round-5 judge's (LUT hash of every tap 1..16; 22 outputs incl. taps 5, 6, 7, 12, 16) -- see jinc_oracle.h.  The
 * reference cannot be built in this image (no avisynth_c.h), so there is no oracle/_ref: PARITY UNPINNED by the tier's
 * rule beyond those known answers.
 */
#define _GNU_SOURCE /* sincos() */
#include "jinc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

double oracle_cyl_bessel_j1(double x); /* bessel_shim.cpp */

/* ---- constants (data): ref :49-82 Taylor coefficients of 2*J1(pi x)/(pi x) in x^2 ---- */
static const double k_taylor[31] = {
    1.0, -1.23370055013616982735431137, 0.507339015802096027273126733,
    -0.104317403816764804365258186, 0.0128696438477519721233840271,
    -0.00105848577966854543020422691, 6.21835470803998638484476598e-05,
    -2.73985272294670461142756204e-06, 9.38932725442064547796003405e-08,
    -2.57413737759717407304931036e-09, 5.77402672521402031756429343e-11,
    -1.07930605263598241754572977e-12, 1.70710316782347356046974552e-14,
    -2.31434518382749184406648762e-16, 2.71924659665997312120515390e-18,
    -2.79561335187943028518083529e-20, 2.53599244866299622352138464e-22,
    -2.04487273140961494085786452e-24, 1.47529860450204338866792475e-26,
    -9.57935105257523453155043307e-29, 5.62764317309979254140393917e-31,
    -3.00555258814860366342363867e-33, 1.46559362903641161989338221e-35,
    -6.55110024064596600335624426e-38, 2.69403199029404093412381643e-40,
    -1.02265499954159964097119923e-42, 3.59444454568084324694180635e-45,
    -1.17313973900539982313119019e-47, 3.56478606255557746426034301e-50,
    -1.01100655781438313239513538e-52, 2.68232117541264485328658605e-55};

/* ref :84-102 zeros of the jinc function = EWA radius per tap count */
static const double k_zeros[16] = {
    1.2196698912665045, 2.2331305943815286, 3.2383154841662362, 4.2410628637960699,
    5.2427643768701817, 6.2439216898644877, 7.2447598687199570, 8.2453949139520427,
    9.2458926849494673, 10.246293348754916, 11.246622794877883, 12.246898461138105,
    13.247132522181061, 14.247333735806849, 15.247508563037300, 16.247661874700962};

double oracle_jinc_zero(int tap) { return k_zeros[tap - 1]; }

/* ref :110-140 */
static double eval_rational(const double *num, const double *den, double z, int count)
{
    double s1, s2;
    int i;
    if (z <= 1.0) {
        s1 = num[count - 1];
        s2 = den[count - 1];
        for (i = count - 2; i >= 0; --i) {
            s1 *= z;
            s2 *= z;
            s1 += num[i];
            s2 += den[i];
        }
    } else {
        z = 1.0 / z; /* ref :127 writes 1.0f / z: the float literal is promoted, same value */
        s1 = num[0];
        s2 = den[0];
        for (i = 1; i < count; ++i) {
            s1 *= z;
            s2 *= z;
            s1 += num[i];
            s2 += den[i];
        }
    }
    return s1 / s2;
}

/* ref :148-198 (large-argument J1 via rational approximations) */
static double jinc_sqr_asymptotic(double x2)
{
    static const double PC[7] = {-4.4357578167941278571e+06, -9.9422465050776411957e+06,
                                 -6.6033732483649391093e+06, -1.5235293511811373833e+06,
                                 -1.0982405543459346727e+05, -1.6116166443246101165e+03, 0.0};
    static const double QC[7] = {-4.4357578167941278568e+06, -9.9341243899345856590e+06,
                                 -6.5853394797230870728e+06, -1.5118095066341608816e+06,
                                 -1.0726385991103820119e+05, -1.4550094401904961825e+03, 1.0};
    static const double PS[7] = {3.3220913409857223519e+04, 8.5145160675335701966e+04,
                                 6.6178836581270835179e+04, 1.8494262873223866797e+04,
                                 1.7063754290207680021e+03, 3.5265133846636032186e+01, 0.0};
    static const double QS[7] = {7.0871281941028743574e+05, 1.8194580422439972989e+06,
                                 1.4194606696037208929e+06, 4.0029443582266975117e+05,
                                 3.7890229745772202641e+04, 8.6383677696049909675e+02, 1.0};
    const double y2 = M_PI * M_PI * x2;
    const double xp = sqrt(y2);
    const double y2p = 64.0 / y2;
    /* GCC -O2/-O3 (the reference's CMake Release build) merges sin(xp) and cos(xp) into one sincos() call, whose
       glibc result differs from sin()/cos() by one ulp at some arguments; call it explicitly, and through a volatile
       pointer because GCC -O0 lowers the sincos builtin back into sin() + cos(), so that the oracle does not depend on
       its own optimisation level. */
    void (*volatile glibc_sincos)(double, double *, double *) = sincos;
    double sx, cx;
    glibc_sincos(xp, &sx, &cx);
    const double rc = eval_rational(PC, QC, y2p, 7);
    const double rs = eval_rational(PS, QS, y2p, 7);
    return (sqrt(xp / M_PI) * 2.0 / y2) * (rc * (sx - cx) + (8.0 / xp) * rs * (sx + cx));
}

static double horner(double x2, int terms)
{
    double res = 0.0;
    int j;
    for (j = terms; j > 0; --j)
        res = res * x2 + k_taylor[j - 1];
    return res;
}

/* ref :201-245 */
double oracle_jinc_sqr(double x2)
{
    if (x2 < 1.49)
        return horner(x2, 16);
    if (x2 < 4.97)
        return horner(x2, 21);
    if (x2 < 10.49)
        return horner(x2, 26);
    if (x2 < 17.99)
        return horner(x2, 31);
    if (x2 < 52.57) {
        const double x = M_PI * sqrt(x2);
        return 2.0 * oracle_cyl_bessel_j1(x) / x;
    }
    if (x2 < 68.07)
        return jinc_sqr_asymptotic(x2);
    {
        const double x = M_PI * sqrt(x2);
        return 2.0 * oracle_cyl_bessel_j1(x) / x;
    }
}

/* ref :247-256 */
static double sample_sqr(double x2, double blur2, double radius2)
{
    if (blur2 > 0.0)
        x2 /= blur2;
    if (x2 < radius2)
        return oracle_jinc_sqr(x2);
    return 0.0;
}

/* ref :258, :265-275 */
void oracle_lut_init(double *lut, int lut_size, double radius, double blur)
{
    const double jinc_zero_sqr = 1.48759464366204680005356;
    const double radius2 = radius * radius;
    const double blur2 = blur * blur;
    int i;
    for (i = 0; i < lut_size; ++i) {
        const double t2 = i / (lut_size - 1.0);
        lut[i] = sample_sqr(radius2 * t2, blur2, radius2) * sample_sqr(jinc_zero_sqr * t2, 1.0, radius2);
    }
}

/* ref :277-282 */
static float lut_factor(const double *lut, int lut_size, int index)
{
    if (index >= lut_size)
        return 0.f;
    return (float)lut[index];
}

/* avs/minmax.h semantics used by the reference (JincResize.h:9) */
static float clampf(float n, float lo, float hi)
{
    n = n > hi ? hi : n;
    return n < lo ? lo : n;
}
static double mind(double a, double b) { return a < b ? a : b; }
static float maxf(float a, float b) { return a > b ? a : b; }
static int maxi(int a, int b) { return a > b ? a : b; }

/* ref :336-533 */
int oracle_table_generate(const double *lut, const oracle_table_params *p, oracle_table *out)
{
    const double round_magic = 6755399441055744.0; /* ref :284 */
    const int quantize_x = p->quantize_x, quantize_y = p->quantize_y;
    const int samples = p->samples;
    const int src_width = p->src_width, src_height = p->src_height;
    const int dst_width = p->dst_width, dst_height = p->dst_height;
    const double radius = p->radius;

    /* ref :349-356 */
    const double filter_step_x = mind((double)dst_width / p->crop_width, 1.0);
    const double filter_step_y = mind((double)dst_height / p->crop_height, 1.0);
    const float filter_support_x = (float)(radius / filter_step_x);
    const float filter_support_y = (float)(radius / filter_step_y);
    const float filter_support = maxf(filter_support_x, filter_support_y);
    const int filter_size = maxi((int)ceil(filter_support_x * 2.0), (int)ceil(filter_support_y * 2.0));

    /* ref :358-364 */
    const float start_x = (float)(p->crop_left + (p->crop_width / dst_width - 1.0) / 2.0);
    const float x_step = (float)(p->crop_width / dst_width);
    const float y_step = (float)(p->crop_height / dst_height);
    float xpos = start_x;
    float ypos = (float)(p->crop_top + (p->crop_height - dst_height) / (dst_height * (int64_t)2));

    const double radius2 = radius * radius;
    const int coeff_stride = (filter_size + 15) & ~15; /* ref :290 */
    const int coeff_per_pixel = coeff_stride * filter_size; /* ref :380 */

    int *factor_map;
    float *factor = NULL;
    size_t capacity = 0, size = 0;
    int top = 0;
    int x, y, lx, ly;

    memset(out, 0, sizeof(*out));
    out->filter_size = filter_size;
    out->coeff_stride = coeff_stride;
    out->dst_width = dst_width;
    out->dst_height = dst_height;
    out->meta = (oracle_meta *)calloc((size_t)dst_width * dst_height, sizeof(oracle_meta));
    factor_map = (int *)calloc((size_t)quantize_x * quantize_y, sizeof(int));
    if (!out->meta || !factor_map) {
        free(out->meta);
        free(factor_map);
        return -1;
    }

    for (y = 0; y < dst_height; ++y) {
        for (x = 0; x < dst_width; ++x) {
            int is_border = 0;
            oracle_meta *meta = &out->meta[(size_t)y * dst_width + x];

            /* ref :392-421 */
            int window_end_x = (int)(xpos + filter_support);
            int window_end_y = (int)(ypos + filter_support);
            int window_begin_x, window_begin_y;
            if (window_end_x >= src_width) {
                window_end_x = src_width - 1;
                is_border = 1;
            }
            if (window_end_y >= src_height) {
                window_end_y = src_height - 1;
                is_border = 1;
            }
            window_begin_x = window_end_x - filter_size + 1;
            window_begin_y = window_end_y - filter_size + 1;
            if (window_begin_x < 0) {
                window_begin_x = 0;
                is_border = 1;
            }
            if (window_begin_y < 0) {
                window_begin_y = 0;
                is_border = 1;
            }
            meta->start_x = window_begin_x;
            meta->start_y = window_begin_y;

            {
                /* ref :424-429 */
                const int qx_int = (int)(xpos * quantize_x);
                const int qy_int = (int)(ypos * quantize_y);
                const int qx_val = qx_int % quantize_x;
                const int qy_val = qy_int % quantize_y;
                const float qxpos = (float)qx_int / quantize_x;
                const float qypos = (float)qy_int / quantize_y;

                if (!is_border && factor_map[qy_val * quantize_x + qx_val] != 0) {
                    meta->coeff_meta = factor_map[qy_val * quantize_x + qx_val] - 1; /* ref :434 */
                } else {
                    float divider = 0.f;
                    int window_x, window_y, ptr;
                    const float px = is_border ? xpos : qxpos;
                    const float py = is_border ? ypos : qypos;

                    if (!is_border) { /* ref :446-451 */
                        window_begin_x = (int)(qxpos + filter_support) - filter_size + 1;
                        window_begin_y = (int)(qypos + filter_support) - filter_size + 1;
                    }
                    window_x = window_begin_x;
                    window_y = window_begin_y;

                    if (size + (size_t)coeff_per_pixel > capacity) {
                        size_t ncap = capacity ? capacity + capacity / 2 : (size_t)1 << 20;
                        float *nf;
                        if (ncap < size + (size_t)coeff_per_pixel)
                            ncap = size + (size_t)coeff_per_pixel;
                        nf = (float *)realloc(factor, ncap * sizeof(float));
                        if (!nf) {
                            free(factor);
                            free(factor_map);
                            free(out->meta);
                            memset(out, 0, sizeof(*out));
                            return -1;
                        }
                        factor = nf;
                        capacity = ncap;
                    }
                    memset(factor + size, 0, (size_t)coeff_per_pixel * sizeof(float)); /* ref :476 */
                    size += (size_t)coeff_per_pixel;

                    ptr = top;
                    for (ly = 0; ly < filter_size; ++ly) { /* ref :480-502 */
                        for (lx = 0; lx < filter_size; ++lx) {
                            const double dx = (clampf(px, 0.f, (float)(src_width - 1)) - window_x) * filter_step_x;
                            const double dy = (clampf(py, 0.f, (float)(src_height - 1)) - window_y) * filter_step_y;
                            const int index = (int)llround((samples - 1) * (dx * dx + dy * dy) / radius2 + round_magic);
                            const float f = lut_factor(lut, samples, index);
                            factor[ptr + lx] = f;
                            divider += f;
                            ++window_x;
                        }
                        ptr += coeff_stride;
                        window_x = window_begin_x;
                        ++window_y;
                    }

                    ptr = top; /* ref :505-514 */
                    for (ly = 0; ly < filter_size; ++ly) {
                        for (lx = 0; lx < filter_size; ++lx)
                            factor[ptr + lx] /= divider;
                        ptr += coeff_stride;
                    }

                    if (!is_border) { /* ref :517-518 */
                        factor_map[qy_val * quantize_x + qx_val] = top + 1;
                        out->cached_phases++;
                    }
                    meta->coeff_meta = top;
                    top += coeff_per_pixel;
                }
            }
            xpos += x_step; /* ref :524 */
        }
        ypos += y_step; /* ref :527-528 */
        xpos = start_x;
    }

    free(factor_map);
    /* ref :6-21, :466-470: the reference keeps `factor` 64-byte aligned (its SIMD paths load coefficient rows with aligned
     * loads); the growth above goes through realloc, so the finished array is moved once into aligned storage. */
    {
        void *aligned = NULL;
        if (posix_memalign(&aligned, 64, (size_t)(top > 0 ? top : 1) * sizeof(float)) != 0) {
            free(factor);
            free(out->meta);
            memset(out, 0, sizeof(*out));
            return -1;
        }
        if (top > 0)
            memcpy(aligned, factor, (size_t)top * sizeof(float));
        free(factor);
        factor = (float *)aligned;
    }
    out->factor = factor;
    out->factor_count = top;
    return 0;
}

void oracle_table_free(oracle_table *t)
{
    free(t->factor);
    free(t->meta);
    memset(t, 0, sizeof(*t));
}

/* ref :560-586, one template instantiation per sample type */
#define DEFINE_ROW(NAME, T, IS_INT)                                                                  \
    static void NAME(const T *srcp, int src_stride, T *dstp, const oracle_table *t, int y, float peak) \
    {                                                                                                \
        const int fs = t->filter_size, cs = t->coeff_stride, w = t->dst_width;                       \
        int x, lx, ly;                                                                               \
        for (x = 0; x < w; ++x) {                                                                    \
            const oracle_meta *m = t->meta + (int64_t)y * w + x;                                     \
            const T *sp = srcp + m->start_y * (int64_t)src_stride + m->start_x;                      \
            const float *cp = t->factor + m->coeff_meta;                                             \
            float result = 0.f;                                                                      \
            for (ly = 0; ly < fs; ++ly) {                                                            \
                for (lx = 0; lx < fs; ++lx)                                                          \
                    result += sp[lx] * cp[lx];                                                       \
                cp += cs;                                                                            \
                sp += src_stride;                                                                    \
            }                                                                                        \
            if (IS_INT)                                                                              \
                dstp[x] = (T)lrintf(clampf(result, 0.f, peak));                                      \
            else                                                                                     \
                dstp[x] = (T)result;                                                                 \
        }                                                                                            \
    }

DEFINE_ROW(row_u8, uint8_t, 1)
DEFINE_ROW(row_u16, uint16_t, 1)
DEFINE_ROW(row_f32, float, 0)

/* ref :536-601 (one plane; the plane loop and table choice of :542-558 are done by the caller) */
void oracle_resize_plane(const void *src, int src_pitch, void *dst, int dst_pitch,
                         const oracle_table *t, int sample_bytes, float peak, int threads)
{
    const int h = t->dst_height;
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
        if (sample_bytes == 1)
            row_u8((const uint8_t *)src, src_pitch, (uint8_t *)drow, t, y, peak);
        else if (sample_bytes == 2)
            row_u16((const uint16_t *)src, src_pitch / 2, (uint16_t *)drow, t, y, peak);
        else
            row_f32((const float *)src, src_pitch / 4, (float *)drow, t, y, peak);
    }
}

/* SURVEY.md Appendix A item 4 */
uint32_t oracle_lcg_fill(void *plane, int pitch, int width, int height, int sample_bytes, int bits, uint32_t s)
{
    int x, y;
    for (y = 0; y < height; ++y) {
        char *row = (char *)plane + (int64_t)y * pitch;
        for (x = 0; x < width; ++x) {
            uint32_t r;
            s = s * 1664525u + 1013904223u;
            r = s >> 8;
            if (sample_bytes == 1)
                ((uint8_t *)row)[x] = (uint8_t)(r & 0xffu);
            else if (sample_bytes == 2)
                ((uint16_t *)row)[x] = (uint16_t)(r & ((1u << bits) - 1u));
            else
                ((float *)row)[x] = (float)(r & 0xffffffu) / 16777215.0f;
        }
    }
    return s;
}

uint64_t oracle_fnv1a64(const void *data, size_t n, uint64_t h)
{
    const unsigned char *p = (const unsigned char *)data;
    size_t i;
    if (h == 0)
        h = 0xcbf29ce484222325ull;
    for (i = 0; i < n; ++i) {
        h ^= p[i];
        h *= 0x100000001b3ull;
    }
    return h;
}
