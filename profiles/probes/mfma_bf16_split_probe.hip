// mfma_bf16_split_probe.hip -- measurement tool (not part of the product).  Follow-up of mfma_product_probe:
// the f32-input MFMAs run on the VALU's own fp32 lanes (their time ADDS to the adds' time, profiles/round6/
// mfma_product_probe.log), so they cannot take the multiply off the VALU.  The bf16 MFMAs run on the matrix pipe proper.
// A fp32 coefficient is the exact sum of three bf16 pieces (8 + 8 + 8 significand bits, truncation split), an 8-bit
// sample IS a bf16 value, a 16-bit sample is two, a fp32 sample three; every piece product is exact in fp32 (16 bits), so
//     D[i][j] = sum_k piece_k(c_i) * piece_k'(s_j)          (K = 3 / 6 / 9 of the instruction's 4 or 16)
// equals fl(c_i * s_j) IF the instruction sums its K products exactly and rounds once.  That is a property of the
// hardware the ISA manual does not state: Part 2 measures it (bit for bit against v_mul_f32).
// Part 1: the rate of (bf16 MFMA + N adds of its D registers), as in mfma_product_probe.
//
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form mfma_bf16_split_probe.hip -o mfma_bf16_split_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#pragma clang fp contract(off)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f32v __attribute__((ext_vector_type(32)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);     \
            return 1;                                                                              \
        }                                                                                          \
    } while (0)

// SHAPE: 1 = v_mfma_f32_32x32x16_bf16 (16 D), 2 = v_mfma_f32_32x32x4_2b_bf16 (32 D), 3 = v_mfma_f32_16x16x4_4b_bf16 (16 D)
template <int SHAPE> struct Shape;
template <> struct Shape<1> { typedef f16v D; static constexpr int N = 16; typedef u4 Op; };
template <> struct Shape<2> { typedef f32v D; static constexpr int N = 32; typedef u2 Op; };
template <> struct Shape<3> { typedef f16v D; static constexpr int N = 16; typedef u2 Op; };

template <int SHAPE>
__device__ __forceinline__ typename Shape<SHAPE>::D mfma(typename Shape<SHAPE>::Op a, typename Shape<SHAPE>::Op b) {
    typename Shape<SHAPE>::D zero = {};
    if constexpr (SHAPE == 1) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, a), __builtin_bit_cast(b8, b), zero, 0, 0, 0);
    else if constexpr (SHAPE == 2) return __builtin_amdgcn_mfma_f32_32x32x4bf16_1k(__builtin_bit_cast(s4, a), __builtin_bit_cast(s4, b), zero, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x4bf16_1k(__builtin_bit_cast(s4, a), __builtin_bit_cast(s4, b), zero, 0, 0, 0);
}

// ------------------------------------------------------------------ Part 1: rate
__device__ unsigned long long g_cycles[4096];
__device__ unsigned long long g_real[4096];  // s_memrealtime ticks (100 MHz) of the same interval: shader clock = cycles / real x 100 MHz

template <int SHAPE, int ADD, int NADD>
__global__ __launch_bounds__(256) void rate_kernel(float* out, const uint32_t* in, int iters) {
    extern __shared__ float lds_pad[];
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    typedef typename Shape<SHAPE>::Op Op;
    typedef typename Shape<SHAPE>::D D;
    Op a, b;
    for (int k = 0; k < int(sizeof(Op) / 4); ++k) {
        a[k] = in[(threadIdx.x & 63) * 4 + k];
        b[k] = in[256 + (threadIdx.x & 63) * 4 + k];
    }
    float acc[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) acc[k] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        asm volatile("" : "+v"(a), "+v"(b));
        D d = mfma<SHAPE>(a, b);
        if constexpr (ADD == 1) {
#pragma unroll
            for (int k = 0; k < NADD / 2; ++k) {
                f2 s = f2{acc[2 * k], acc[2 * k + 1]};
                s = s + f2{d[2 * k], d[2 * k + 1]};
                acc[2 * k] = s.x;
                acc[2 * k + 1] = s.y;
            }
        } else if constexpr (ADD == 2) {
#pragma unroll
            for (int k = 0; k < NADD; ++k) acc[k] = acc[k] + d[k];
        } else {
            acc[0] = acc[0] + d[0];
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) s += acc[k];
    out[tid] = s;
    if (threadIdx.x == 0 && blockIdx.x < 4096) {
        g_cycles[blockIdx.x] = t1 - t0;
        g_real[blockIdx.x] = r1 - r0;
    }
}

template <int SHAPE, int ADD, int NADD>
int run_rate(const char* name, int cus, float* out, const uint32_t* in, hipEvent_t e0, hipEvent_t e1, int wps, int iters) {
    auto kern = rate_kernel<SHAPE, ADD, NADD>;
    hipFuncAttributes attr;
    CHECK(hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(kern)));
    const size_t lds = std::min<size_t>(64 * 1024, (160 * 1024 / wps) - 512);
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    int max_blocks = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&max_blocks, kern, 256, lds));
    if (max_blocks < wps) {
        std::printf("%-44s waves/SIMD %d : skipped (%d VGPRs allow %d)\n", name, wps, attr.numRegs, max_blocks);
        return 0;
    }
    const int blocks = cus * wps;
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, out, in, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    std::vector<unsigned long long> cyc(std::min(blocks, 4096));
    CHECK(hipMemcpyFromSymbol(cyc.data(), HIP_SYMBOL(g_cycles), cyc.size() * sizeof(unsigned long long)));
    std::vector<unsigned long long> real(cyc.size());
    CHECK(hipMemcpyFromSymbol(real.data(), HIP_SYMBOL(g_real), real.size() * sizeof(unsigned long long)));
    std::sort(cyc.begin(), cyc.end());
    std::sort(real.begin(), real.end());
    const double med = double(cyc[cyc.size() / 2]);
    const double mhz = med / double(real[real.size() / 2]) * 100.0;
    constexpr int terms_per_lane = ADD != 0 ? NADD : Shape<SHAPE>::N;
    const double terms = double(terms_per_lane) * 64.0 * 4.0 * blocks * iters;
    // s_memtime ticks per step and SIMD: every wave of a SIMD spans the whole launch, so a SIMD's step takes (ticks / iters) / wps
    std::printf("%-44s waves/SIMD %d : %8.3f ms  %7.2f Tops/s eq  shader clock %5.0f MHz  clk/step/SIMD %6.1f  (%d VGPRs%s)\n", name, wps, best,
                2.0 * terms / best * 1e-9, mhz, med / iters / wps, attr.numRegs, ADD == 0 ? "; products only" : "");
    return 0;
}

// ------------------------------------------------------------------ Part 2: exactness of the split product
// truncation split of a fp32 value into three bf16 pieces (each the top 16 bits of a float, sign included): c = p0 + p1 + p2 exactly
__host__ __device__ inline void split3(float c, uint32_t p[3]) {
    float r = c;
    for (int k = 0; k < 3; ++k) {
        uint32_t u = __builtin_bit_cast(uint32_t, r) & 0xffff0000u;
        p[k] = u >> 16;
        r = r - __builtin_bit_cast(float, u);  // exact: removes the leading 8 significand bits
    }
}

// KIND 0: u8 sample (one piece), 1: u16 sample (two pieces: (s >> 8) * 256 and s & 255), 2: fp32 sample (three pieces).
// Lane l brings coefficient cv[l] (row i of its block) and sample sv[l] (column j of its block); K slots are laid out
// k = 3 * (sample piece) + (coefficient piece).  SHAPE 1 only (K = 16: lanes 0..31 carry k 0..7, lanes 32..63 k 8..15);
// SHAPEs 2 / 3 (K = 4) for KIND 0 only.
template <int SHAPE, int KIND>
__global__ __launch_bounds__(64) void exact_kernel(const float* cv, const float* sv, uint32_t* got, uint32_t* want) {
    const int l = threadIdx.x;
    const size_t base = size_t(blockIdx.x) * 64;
    typedef typename Shape<SHAPE>::Op Op;
    constexpr int N = Shape<SHAPE>::N;
    // SHAPE 1 has 32 rows / columns per MFMA: rows and columns come from lanes 0..31; the upper lanes carry k 8..15 of the same row / column
    const int src_lane = SHAPE == 1 ? (l & 31) : l;
    const float c = cv[base + src_lane], s = sv[base + src_lane];
    uint32_t cp[3], sp[3] = {0, 0, 0};
    split3(c, cp);
    int ns;
    if constexpr (KIND == 0) { sp[0] = __builtin_bit_cast(uint32_t, s) >> 16; ns = 1; }
    else if constexpr (KIND == 1) {
        const uint32_t si = static_cast<uint32_t>(s);
        sp[0] = __builtin_bit_cast(uint32_t, float(si & 0xff00u)) >> 16;
        sp[1] = __builtin_bit_cast(uint32_t, float(si & 0xffu)) >> 16;
        ns = 2;
    } else { split3(s, sp); ns = 3; }
    uint32_t ka[16], kb[16];
    for (int k = 0; k < 16; ++k) { ka[k] = 0; kb[k] = 0; }
    for (int q = 0; q < ns; ++q)
        for (int k = 0; k < 3; ++k) { ka[3 * q + k] = cp[k]; kb[3 * q + k] = sp[q]; }
    Op a, b;
    if constexpr (SHAPE == 1) {
        const int k0 = 8 * (l >> 5);
        for (int w = 0; w < 4; ++w) {
            a[w] = ka[k0 + 2 * w] | (ka[k0 + 2 * w + 1] << 16);
            b[w] = kb[k0 + 2 * w] | (kb[k0 + 2 * w + 1] << 16);
        }
    } else {
        for (int w = 0; w < 2; ++w) {
            a[w] = ka[2 * w] | (ka[2 * w + 1] << 16);
            b[w] = kb[2 * w] | (kb[2 * w + 1] << 16);
        }
    }
    typename Shape<SHAPE>::D d = mfma<SHAPE>(a, b);
#pragma unroll
    for (int r = 0; r < N; ++r) {
        int la, lb;
        if constexpr (SHAPE == 1) {
            la = 8 * (r / 4) + 4 * (l / 32) + (r % 4);
            lb = l % 32;
        } else if constexpr (SHAPE == 2) {
            const int blk = r / 16;
            la = 32 * blk + 8 * ((r % 16) / 4) + 4 * (l / 32) + (r % 4);
            lb = 32 * blk + l % 32;
        } else {
            const int blk = r / 4;
            la = 16 * blk + 4 * (l / 16) + (r % 4);
            lb = 16 * blk + l % 16;
        }
        const float cx = __shfl(c, la, 64), sx = __shfl(s, lb, 64);
        float p;
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p) : "v"(cx), "v"(sx));
        const float dr = d[r];
        got[(base + l) * N + r] = __builtin_bit_cast(uint32_t, dr);
        want[(base + l) * N + r] = __builtin_bit_cast(uint32_t, p);
    }
}

template <int SHAPE, int KIND>
int run_exact(const char* name, const std::vector<float>& c, const std::vector<float>& s) {
    constexpr int N = Shape<SHAPE>::N;
    const size_t n = c.size();
    float *dc, *ds;
    uint32_t *dg, *dw;
    CHECK(hipMalloc(&dc, n * 4));
    CHECK(hipMalloc(&ds, n * 4));
    CHECK(hipMalloc(&dg, n * N * 4));
    CHECK(hipMalloc(&dw, n * N * 4));
    CHECK(hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(ds, s.data(), n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((exact_kernel<SHAPE, KIND>), dim3(static_cast<unsigned>(n / 64)), dim3(64), 0, 0, dc, ds, dg, dw);
    CHECK(hipDeviceSynchronize());
    std::vector<uint32_t> g(n * N), w(n * N);
    CHECK(hipMemcpy(g.data(), dg, n * N * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(w.data(), dw, n * N * 4, hipMemcpyDeviceToHost));
    CHECK(hipFree(dc)); CHECK(hipFree(ds)); CHECK(hipFree(dg)); CHECK(hipFree(dw));
    size_t total = 0, equal = 0, negzero = 0, off1 = 0, other = 0;
    int shown = 0;
    for (size_t k = 0; k < n * N; ++k) {
        ++total;
        if (g[k] == w[k]) { ++equal; continue; }
        if (w[k] == 0x80000000u && g[k] == 0u) { ++negzero; continue; }
        const int64_t dist = int64_t(g[k] & 0x7fffffffu) - int64_t(w[k] & 0x7fffffffu);
        if ((g[k] >> 31) == (w[k] >> 31) && (dist == 1 || dist == -1)) ++off1; else ++other;
        if (shown < 4) { std::printf("    %s differs: v_mul=%08x  mfma=%08x\n", name, w[k], g[k]); ++shown; }
    }
    std::printf("exact %-46s: %10zu products, equal %10zu, -0 -> +0 %7zu, one ulp off %9zu, OTHER %zu\n", name, total, equal, negzero, off1, other);
    return 0;
}

int main(int argc, char** argv) {
    const char* what = argc > 1 ? argv[1] : "all";
    CHECK(hipSetDevice(0));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    std::printf("device %s, %d CUs\n", prop.name, cus);
    if (!std::strcmp(what, "exact") || !std::strcmp(what, "all")) {
        std::mt19937 rng(4242);
        const size_t n = 1 << 20;
        std::vector<float> coef(n), coef_any(n), s8(n), s16(n), sf(n), sany(n);
        std::uniform_real_distribution<float> cd(-0.35f, 1.1f), u01(0.f, 1.f);
        for (size_t k = 0; k < n; ++k) {
            float c = cd(rng);
            switch (rng() % 6) { case 0: c *= 1e-3f; break; case 1: c *= 1e-6f; break; case 2: c *= 3e-9f; break; default: break; }
            coef[k] = c;
            uint32_t u = (rng() & 0x807fffffu) | ((rng() % 200 + 20) << 23);  // any normal float of a middling exponent
            std::memcpy(&coef_any[k], &u, 4);
            s8[k] = float(rng() & 0xff);
            s16[k] = float(rng() & 0xffff);
            sf[k] = u01(rng);
            u = (rng() & 0x807fffffu) | ((rng() % 60 + 90) << 23);
            std::memcpy(&sany[k], &u, 4);
        }
        if (run_exact<1, 0>("32x32x16_bf16   u8 x coefficient (K=3)", coef, s8)) return 1;
        if (run_exact<2, 0>("32x32x4_2b_bf16 u8 x coefficient (K=3)", coef, s8)) return 1;
        if (run_exact<3, 0>("16x16x4_4b_bf16 u8 x coefficient (K=3)", coef, s8)) return 1;
        if (run_exact<1, 0>("32x32x16_bf16   u8 x any normal float (K=3)", coef_any, s8)) return 1;
        if (run_exact<2, 0>("32x32x4_2b_bf16 u8 x any normal float (K=3)", coef_any, s8)) return 1;
        if (run_exact<1, 1>("32x32x16_bf16   u16 x coefficient (K=6)", coef, s16)) return 1;
        if (run_exact<1, 1>("32x32x16_bf16   u16 x any normal float (K=6)", coef_any, s16)) return 1;
        if (run_exact<1, 2>("32x32x16_bf16   float[0,1] x coefficient (K=9)", coef, sf)) return 1;
        if (run_exact<1, 2>("32x32x16_bf16   normal float x normal float (K=9)", coef_any, sany)) return 1;
        std::fflush(stdout);
    }
    if (!std::strcmp(what, "rate") || !std::strcmp(what, "all")) {
        float* out = nullptr;
        uint32_t* in = nullptr;
        CHECK(hipMalloc(&out, sizeof(float) * 256 * cus * 8));
        CHECK(hipMalloc(&in, 4 * 512));
        std::vector<uint32_t> hin(512);
        std::mt19937 rng(7);
        for (auto& x : hin) { x = (0x3f00u | (rng() & 0xff)) | ((0x3e80u | (rng() & 0x7f)) << 16); }
        CHECK(hipMemcpy(in, hin.data(), 4 * 512, hipMemcpyHostToDevice));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        const int iters = 40000;
        for (int wps : {1, 2, 3, 4, 6, 8}) {
#define RUN(S, A, N, NAME) if (run_rate<S, A, N>(NAME, cus, out, in, e0, e1, wps, iters)) return 1
            RUN(1, 0, 0, "bf16 32x32x16 alone");
            RUN(2, 0, 0, "bf16 32x32x4_2b alone");
            RUN(3, 0, 0, "bf16 16x16x4_4b alone");
            RUN(1, 1, 16, "bf16 32x32x16 + 8 pk_add");
            RUN(1, 1, 12, "bf16 32x32x16 + 6 pk_add (12 of 16 live)");
            RUN(1, 2, 16, "bf16 32x32x16 + 16 v_add");
            RUN(1, 2, 12, "bf16 32x32x16 + 12 v_add");
            RUN(2, 1, 32, "bf16 32x32x4_2b + 16 pk_add");
            RUN(2, 1, 24, "bf16 32x32x4_2b + 12 pk_add (24 of 32 live)");
            RUN(2, 2, 32, "bf16 32x32x4_2b + 32 v_add");
            RUN(2, 2, 24, "bf16 32x32x4_2b + 24 v_add");
            RUN(3, 1, 16, "bf16 16x16x4_4b + 8 pk_add");
            RUN(3, 2, 16, "bf16 16x16x4_4b + 16 v_add");
#undef RUN
            std::fflush(stdout);
        }
    }
    return 0;
}
