// mfma_product_probe.hip -- measurement tool (not part of the product).  VERDICT r5 "Next 1":
// can the matrix pipe serve as an exact PRODUCT generator beside an un-fused VALU add chain?
//
//   fma(s, c, +0) = fl(s * c): one rounding, the bits of v_mul_f32 (a -0 product becomes +0, which the next add of a
//   chain that started at +0 cannot see).  K = 1 MFMAs (v_mfma_f32_{32x32x1_2b,16x16x1_4b,4x4x1_16b}_f32, C = 0) deliver
//   outer products of one sample with 32 / 16 / 4 coefficients into the lane that owns the chains; the adds stay on the
//   VALU (v_pk_add_f32 or v_add_f32), in (ly, lx) order.  The parity rule forbids a fused ACCUMULATE, not this.
//
// Part 1 (rate): per step one MFMA + N adds of its D registers into per-lane accumulators, chip filled at 1 / 2 / 4 / 8
// waves per SIMD (as registers allow), beside the pair the kernels run today (v_pk_mul_f32 + v_pk_add_f32).  Reported:
// chain terms per second (a chain term = one product added to one accumulator of one lane), as "Tops/s equivalent"
// = 2 x terms/s, the unit of bench.py's instruction-pair probe.
// Part 2 (exactness): MFMA product against v_mul_f32, bit for bit, over: every u8 / a spread of u16 samples x a table of
// 1024 coefficient-like floats; random bit patterns; denormal inputs and outputs; +-0, inf, NaN.  Also checks the D
// register layout assumed by any kernel built on this (which lane receives which product).
//
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form mfma_product_probe.hip -o mfma_product_probe
//        (-amdgpu-mfma-vgpr-form: D in VGPRs; without it the compiler parks D in AGPRs and copies every register out)
// run:   ./mfma_product_probe [rate|exact|all]
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
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f32v __attribute__((ext_vector_type(32)));

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);     \
            return 1;                                                                              \
        }                                                                                          \
    } while (0)

template <int SHAPE>
struct DVec;
template <>
struct DVec<32> { typedef f32v type; static constexpr int N = 32; };
template <>
struct DVec<16> { typedef f16v type; static constexpr int N = 16; };
template <>
struct DVec<4> { typedef f4 type; static constexpr int N = 4; };

template <int SHAPE>
__device__ __forceinline__ typename DVec<SHAPE>::type mfma_product(float a, float b) {
    typename DVec<SHAPE>::type zero = {};
    if constexpr (SHAPE == 32) return __builtin_amdgcn_mfma_f32_32x32x1f32(a, b, zero, 0, 0, 0);
    else if constexpr (SHAPE == 16) return __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, zero, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, zero, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------------------------------
// Part 1: rate.  ADD: 0 none (MFMA alone), 1 v_pk_add_f32, 2 v_add_f32.  NADD = D registers added per step (<= N).
// DBUF = 1: two MFMAs in flight per wave (the adds of step k run beside the MFMA of step k + 1).
// SHAPE = 0: today's pair (v_pk_mul_f32 + v_pk_add_f32), NADD chain terms per lane per step.
// ---------------------------------------------------------------------------------------------------------------------
__device__ unsigned long long g_cycles[4096];
__device__ unsigned long long g_real[4096];  // s_memrealtime ticks (100 MHz) of the same interval: shader clock = cycles / real x 100 MHz

template <int SHAPE, int ADD, int NADD, int DBUF>
__global__ __launch_bounds__(256) void rate_kernel(float* out, const float* in, int iters) {
    extern __shared__ float lds_pad[];  // only to cap the occupancy
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    float a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)];
    float acc[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) acc[k] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if constexpr (SHAPE == 0) {
        f2 w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = f2{a + k, b - k};
        const f2 cc = {1e-7f, 2e-7f};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < NADD / 2; ++k) {
                f2 t, s = f2{acc[2 * k], acc[2 * k + 1]};
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(w[k & 3]), "v"(cc));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(s) : "v"(t));
                acc[2 * k] = s.x;
                acc[2 * k + 1] = s.y;
            }
        }
    } else {
        typedef typename DVec<SHAPE>::type D;
        constexpr int N = DVec<SHAPE>::N;
        auto add_in = [&](const D& d) {
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
                // MFMA alone: keep the result alive with one add
                acc[0] = acc[0] + d[0];
            }
        };
        static_assert(NADD <= N, "");
        if constexpr (DBUF) {
            D d0 = mfma_product<SHAPE>(a, b);
            for (int i = 0; i < iters; i += 2) {
                asm volatile("" : "+v"(a), "+v"(b));
                D d1 = mfma_product<SHAPE>(a, b);
                add_in(d0);
                asm volatile("" : "+v"(a), "+v"(b));
                d0 = mfma_product<SHAPE>(a, b);
                add_in(d1);
            }
            add_in(d0);
        } else {
            for (int i = 0; i < iters; ++i) {
                asm volatile("" : "+v"(a), "+v"(b));
                D d = mfma_product<SHAPE>(a, b);
                add_in(d);
            }
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

struct RateResult {
    double ms, terms_per_s, cyc_per_step;
};

template <int SHAPE, int ADD, int NADD, int DBUF>
int run_rate(const char* name, int cus, float* out, const float* in, hipEvent_t e0, hipEvent_t e1, int wps, int iters) {
    auto kern = rate_kernel<SHAPE, ADD, NADD, DBUF>;
    hipFuncAttributes attr;
    CHECK(hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(kern)));
    // occupancy cap through LDS: a CU has 160 KB; wps workgroups of 4 waves per CU
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
    constexpr int terms_per_lane = (SHAPE == 0 || ADD != 0) ? NADD : (SHAPE == 32 ? 32 : SHAPE == 16 ? 16 : 4);
    const double terms = double(terms_per_lane) * 64.0 * 4.0 * blocks * iters;
    const double steps_per_simd = double(iters) * wps;
    // s_memtime ticks per step and SIMD: every wave of a SIMD spans the whole launch, so a SIMD's step takes (ticks / iters) / wps
    std::printf("%-44s waves/SIMD %d : %8.3f ms  %7.2f Tops/s eq  shader clock %5.0f MHz  clk/step/SIMD %6.1f  (%d VGPRs%s)\n", name, wps, best,
                2.0 * terms / best * 1e-9, mhz, med / steps_per_simd, attr.numRegs, ADD == 0 && SHAPE != 0 ? "; products only" : "");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Part 2: exactness + layout.  Each lane brings one a (coefficient) and one b (sample).  The kernel writes every D
// register of every lane, and beside it v_mul_f32 of the operands that the ISA's layout says meet there:
//   32x32x1_2b : D[r] of lane l = a[lane 32*blk + i] * b[lane 32*blk + j],  blk = r / 16, j = l % 32,
//                i = 8 * ((r % 16) / 4) + 4 * (l / 32) + (r % 4)
//   16x16x1_4b : blk = r / 4, j = l % 16, i = 4 * (l / 16) + (r % 4); a, b from lane 16 * blk + {i, j}
//   4x4x1_16b  : blk = l / 4, j = l % 4, i = r; a, b from lane 4 * blk + {i, j}
// ---------------------------------------------------------------------------------------------------------------------
template <int SHAPE>
__global__ __launch_bounds__(64) void exact_kernel(const float* av, const float* bv, uint32_t* got, uint32_t* want) {
    const int l = threadIdx.x;
    const size_t base = size_t(blockIdx.x) * 64;
    const float a = av[base + l], b = bv[base + l];
    typename DVec<SHAPE>::type d = mfma_product<SHAPE>(a, b);
    constexpr int N = DVec<SHAPE>::N;
#pragma unroll
    for (int r = 0; r < N; ++r) {
        int la, lb;
        if constexpr (SHAPE == 32) {
            const int blk = r / 16, j = l % 32, i = 8 * ((r % 16) / 4) + 4 * (l / 32) + (r % 4);
            la = 32 * blk + i;
            lb = 32 * blk + j;
        } else if constexpr (SHAPE == 16) {
            const int blk = r / 4, j = l % 16, i = 4 * (l / 16) + (r % 4);
            la = 16 * blk + i;
            lb = 16 * blk + j;
        } else {
            const int blk = l / 4, j = l % 4, i = r;
            la = 4 * blk + i;
            lb = 4 * blk + j;
        }
        const float ax = __shfl(a, la, 64), bx = __shfl(b, lb, 64);
        float p;
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p) : "v"(ax), "v"(bx));
        const float dr = d[r];  // (bit-casting the vector element in place reads element 0 for every r with this clang)
        got[(base + l) * N + r] = __builtin_bit_cast(uint32_t, dr);
        want[(base + l) * N + r] = __builtin_bit_cast(uint32_t, p);
    }
}

struct ExactStats {
    size_t total = 0, equal = 0, neg_zero_to_pos = 0, nan_both = 0, denorm_out_diff = 0, denorm_in_diff = 0, other = 0;
};

static bool is_denorm(uint32_t u) { return (u & 0x7f800000u) == 0 && (u & 0x007fffffu) != 0; }
static bool is_nan(uint32_t u) { return (u & 0x7f800000u) == 0x7f800000u && (u & 0x007fffffu) != 0; }

template <int SHAPE>
int run_exact(const char* name, const std::vector<float>& a, const std::vector<float>& b, ExactStats& st, bool print) {
    constexpr int N = DVec<SHAPE>::N;
    const size_t n = a.size();  // multiple of 64
    float *da, *db;
    uint32_t *dg, *dw;
    CHECK(hipMalloc(&da, n * 4));
    CHECK(hipMalloc(&db, n * 4));
    CHECK(hipMalloc(&dg, n * N * 4));
    CHECK(hipMalloc(&dw, n * N * 4));
    CHECK(hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(exact_kernel<SHAPE>, dim3(static_cast<unsigned>(n / 64)), dim3(64), 0, 0, da, db, dg, dw);
    CHECK(hipDeviceSynchronize());
    std::vector<uint32_t> g(n * N), w(n * N);
    CHECK(hipMemcpy(g.data(), dg, n * N * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(w.data(), dw, n * N * 4, hipMemcpyDeviceToHost));
    CHECK(hipFree(da));
    CHECK(hipFree(db));
    CHECK(hipFree(dg));
    CHECK(hipFree(dw));
    int shown = 0;
    for (size_t k = 0; k < n * N; ++k) {
        ++st.total;
        if (g[k] == w[k]) { ++st.equal; continue; }
        if (w[k] == 0x80000000u && g[k] == 0u) { ++st.neg_zero_to_pos; continue; }
        if (is_nan(g[k]) && is_nan(w[k])) { ++st.nan_both; continue; }
        // classify: recompute which operands met here
        const size_t lane_idx = k / N;
        const int r = int(k % N), l = int(lane_idx % 64);
        const size_t base = lane_idx - l;
        int la, lb;
        if (SHAPE == 32) { const int blk = r / 16, j = l % 32, i = 8 * ((r % 16) / 4) + 4 * (l / 32) + (r % 4); la = 32 * blk + i; lb = 32 * blk + j; }
        else if (SHAPE == 16) { const int blk = r / 4, j = l % 16, i = 4 * (l / 16) + (r % 4); la = 16 * blk + i; lb = 16 * blk + j; }
        else { const int blk = l / 4, j = l % 4, i = r; la = 4 * blk + i; lb = 4 * blk + j; }
        uint32_t ua, ub;
        std::memcpy(&ua, &a[base + la], 4);
        std::memcpy(&ub, &b[base + lb], 4);
        if (is_denorm(ua) || is_denorm(ub)) ++st.denorm_in_diff;
        else if (is_denorm(w[k]) || (is_denorm(g[k]))) ++st.denorm_out_diff;
        else ++st.other;
        if (print && shown < 6) {
            std::printf("    %s differs: a=%08x b=%08x  v_mul=%08x  mfma=%08x\n", name, ua, ub, w[k], g[k]);
            ++shown;
        }
    }
    return 0;
}

static void report(const char* shape, const char* set, const ExactStats& s) {
    std::printf("exact %-10s %-34s: %10zu products, equal %10zu, -0 -> +0 %7zu, NaN both (payload differs) %7zu, denormal-input diffs %7zu, denormal-output diffs %7zu, OTHER %zu\n",
                shape, set, s.total, s.equal, s.neg_zero_to_pos, s.nan_both, s.denorm_in_diff, s.denorm_out_diff, s.other);
}

template <int SHAPE>
int exact_sets(const char* shape) {
    std::mt19937 rng(12345);
    auto pad64 = [](std::vector<float>& v, float fill) { while (v.size() % 64) v.push_back(fill); };
    // (1) integer samples x coefficient-like floats: every u8 and 1024 u16 values against 1024 values in (-0.3, 1.1) of jinc-table magnitudes
    {
        std::vector<float> coef(1024);
        for (int i = 0; i < 1024; ++i) {
            const double t = i / 1023.0;
            double v = std::cos(7.1 * t) * std::exp(-3.0 * t) * (i % 7 == 3 ? 1e-6 : 1.0) / (1.0 + (i % 5));
            coef[i] = static_cast<float>(v);
        }
        std::vector<float> a, b;
        for (int s = 0; s < 256; ++s)
            for (int i = 0; i < 1024; ++i) { a.push_back(coef[i]); b.push_back(float(s)); }
        for (int s = 0; s < 1024; ++s)
            for (int i = 0; i < 1024; ++i) { a.push_back(coef[(i * 7 + s) & 1023]); b.push_back(float((s * 64 + 63) & 0xffff)); }
        // shuffle within the vectors so that every lane position sees every kind of operand (the layout pairs lane groups)
        std::vector<size_t> perm(a.size());
        for (size_t k = 0; k < perm.size(); ++k) perm[k] = k;
        std::shuffle(perm.begin(), perm.end(), rng);
        std::vector<float> a2(a.size()), b2(b.size());
        for (size_t k = 0; k < perm.size(); ++k) { a2[k] = a[perm[k]]; b2[k] = b[(perm[k] * 2654435761ull) % b.size()]; }
        pad64(a2, 1.f); pad64(b2, 1.f);
        ExactStats st;
        if (run_exact<SHAPE>(shape, a2, b2, st, true)) return 1;
        report(shape, "u8/u16 samples x coefficients", st);
    }
    // (2) random finite floats in [0,1] x coefficients (float planes, ordinary data)
    {
        std::vector<float> a(1 << 20), b(1 << 20);
        std::uniform_real_distribution<float> u01(0.f, 1.f), coefd(-0.3f, 1.1f);
        for (auto& x : a) x = coefd(rng) * (rng() % 9 == 0 ? 1e-5f : 1.f);
        for (auto& x : b) x = u01(rng);
        ExactStats st;
        if (run_exact<SHAPE>(shape, a, b, st, true)) return 1;
        report(shape, "floats in [0,1] x coefficients", st);
    }
    // (3) random bit patterns (all exponents, NaN, inf, denormals)
    {
        std::vector<float> a(1 << 20), b(1 << 20);
        for (auto& x : a) { uint32_t u = rng(); std::memcpy(&x, &u, 4); }
        for (auto& x : b) { uint32_t u = rng(); std::memcpy(&x, &u, 4); }
        ExactStats st;
        if (run_exact<SHAPE>(shape, a, b, st, true)) return 1;
        report(shape, "random bit patterns", st);
    }
    // (4) denormal inputs and denormal results
    {
        std::vector<float> a(1 << 18), b(1 << 18);
        for (size_t k = 0; k < a.size(); ++k) {
            uint32_t ua, ub;
            switch (k % 4) {
                case 0: ua = rng() & 0x807fffffu; ub = 0x3f000000u | (rng() & 0x00ffffffu); break;               // denormal x ~[0.5, 2)
                case 1: ua = (rng() & 0x007fffffu) | ((rng() % 40 + 1) << 23); ub = (rng() & 0x007fffffu) | ((rng() % 60 + 50) << 23); break;  // tiny x small: results around / below 2^-126
                case 2: ua = rng() & 0x807fffffu; ub = rng() & 0x807fffffu; break;                                // denormal x denormal
                default: ua = (rng() & 0x807fffffu) | (20u << 23); ub = (rng() & 0x007fffffu) | (100u << 23); break;  // result denormal-ish
            }
            std::memcpy(&a[k], &ua, 4);
            std::memcpy(&b[k], &ub, 4);
        }
        ExactStats st;
        if (run_exact<SHAPE>(shape, a, b, st, true)) return 1;
        report(shape, "denormal inputs / results", st);
    }
    // (5) specials: +-0, +-inf, NaN, 1, -1, max, min-normal in all pairs (spread over lanes)
    {
        const uint32_t sp[] = {0x00000000u, 0x80000000u, 0x7f800000u, 0xff800000u, 0x7fc00000u, 0xffc00001u, 0x7f800001u, 0x3f800000u,
                               0xbf800000u, 0x7f7fffffu, 0x00800000u, 0x80800000u, 0x00000001u, 0x807fffffu, 0x3eaaaaabu, 0x43000000u};
        std::vector<float> a, b;
        for (int rep = 0; rep < 64; ++rep)
            for (uint32_t x : sp)
                for (uint32_t y : sp) { float fx, fy; std::memcpy(&fx, &x, 4); std::memcpy(&fy, &y, 4); a.push_back(fx); b.push_back(fy); }
        std::vector<size_t> perm(a.size());
        for (size_t k = 0; k < perm.size(); ++k) perm[k] = k;
        std::shuffle(perm.begin(), perm.end(), rng);
        std::vector<float> a2(a.size()), b2(b.size());
        for (size_t k = 0; k < perm.size(); ++k) { a2[k] = a[perm[k]]; b2[k] = b[perm[(k * 7 + 3) % perm.size()]]; }
        pad64(a2, 1.f); pad64(b2, 1.f);
        ExactStats st;
        if (run_exact<SHAPE>(shape, a2, b2, st, true)) return 1;
        report(shape, "specials (+-0, inf, NaN, extremes)", st);
    }
    return 0;
}

int main(int argc, char** argv) {
    const char* what = argc > 1 ? argv[1] : "all";
    CHECK(hipSetDevice(0));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    std::printf("device %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
    if (!std::strcmp(what, "exact") || !std::strcmp(what, "all")) {
        if (exact_sets<32>("32x32x1_2b")) return 1;
        if (exact_sets<16>("16x16x1_4b")) return 1;
        if (exact_sets<4>("4x4x1_16b")) return 1;
        std::fflush(stdout);
    }
    if (!std::strcmp(what, "rate") || !std::strcmp(what, "all")) {
        float *out = nullptr, *in = nullptr;
        CHECK(hipMalloc(&out, sizeof(float) * 256 * cus * 8));
        CHECK(hipMalloc(&in, sizeof(float) * 128));
        std::vector<float> hin(128);
        std::mt19937 rng(7);
        std::uniform_real_distribution<float> u(0.01f, 1.f);
        for (auto& x : hin) x = u(rng);
        CHECK(hipMemcpy(in, hin.data(), 128 * 4, hipMemcpyHostToDevice));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        const int iters = 40000;
        for (int wps : {1, 2, 4, 6, 8}) {
#define RUN(S, A, N, DB, NAME) if (run_rate<S, A, N, DB>(NAME, cus, out, in, e0, e1, wps, iters)) return 1
            RUN(0, 1, 16, 0, "pair today: pk_mul + pk_add, 16 terms/step");
            RUN(0, 1, 32, 0, "pair today: pk_mul + pk_add, 32 terms/step");
            RUN(32, 0, 0, 0, "mfma 32x32x1_2b alone");
            RUN(16, 0, 0, 0, "mfma 16x16x1_4b alone");
            RUN(4, 0, 0, 0, "mfma 4x4x1_16b alone");
            RUN(32, 1, 32, 0, "32x32x1_2b + 16 pk_add");
            RUN(32, 1, 24, 0, "32x32x1_2b + 12 pk_add (24 of 32 live)");
            RUN(32, 1, 32, 1, "32x32x1_2b + 16 pk_add, 2 in flight");
            RUN(32, 1, 24, 1, "32x32x1_2b + 12 pk_add, 2 in flight");
            RUN(32, 2, 32, 0, "32x32x1_2b + 32 v_add");
            RUN(32, 2, 24, 0, "32x32x1_2b + 24 v_add");
            RUN(32, 2, 24, 1, "32x32x1_2b + 24 v_add, 2 in flight");
            RUN(16, 1, 16, 0, "16x16x1_4b + 8 pk_add");
            RUN(16, 1, 12, 0, "16x16x1_4b + 6 pk_add (12 of 16 live)");
            RUN(16, 1, 16, 1, "16x16x1_4b + 8 pk_add, 2 in flight");
            RUN(16, 1, 12, 1, "16x16x1_4b + 6 pk_add, 2 in flight");
            RUN(16, 2, 16, 0, "16x16x1_4b + 16 v_add");
            RUN(16, 2, 16, 1, "16x16x1_4b + 16 v_add, 2 in flight");
            RUN(4, 1, 4, 0, "4x4x1_16b + 2 pk_add");
            RUN(4, 1, 4, 1, "4x4x1_16b + 2 pk_add, 2 in flight");
            RUN(4, 2, 4, 1, "4x4x1_16b + 4 v_add, 2 in flight");
#undef RUN
            std::fflush(stdout);
        }
    }
    return 0;
}
