// valu_probe.hip -- measurement tool (not part of the product): what is the un-fused fp32 VALU
// ceiling on this MI355X?  The EWA accumulation must stay `v_mul_f32` + `v_add_f32` per tap
// (bit-exactness vs the reference's opt=0 path), so the binding roof of the hot kernel is the rate
// of those two instructions, and the question is whether the packed forms (v_pk_mul_f32 /
// v_pk_add_f32, two IEEE results per lane per instruction) raise it.
//
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize valu_probe.hip -o valu_probe
// run:   ./valu_probe            (prints one line per mode x occupancy)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#pragma clang fp contract(off)

typedef float f2 __attribute__((ext_vector_type(2)));

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            return 1;                                                             \
        }                                                                         \
    } while (0)

// MODE 0: 8 independent scalar chains  a = a + a*c        (v_mul_f32 + v_add_f32)
// MODE 1: 4 independent packed chains  A = A + A*C        (v_pk_mul_f32 + v_pk_add_f32)
// MODE 2: 8 independent fma chains     a = fma(a,c,a)     (v_fma/v_fmac; reference point only)
// MODE 3: 1 scalar chain, products independent of the chain: a = a + w_k*c_k (the real kernel's shape)
// per-wave clock stamps: s_memtime = shader clock ticks, s_memrealtime = 100 MHz constant clock
__device__ unsigned long long g_stamps[4];

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, float c, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long t0 = 0, r0 = 0;
    if (tid == 0) {
        t0 = __builtin_amdgcn_s_memtime();
        r0 = __builtin_amdgcn_s_memrealtime();
    }
    float seed = 1.0f + 1e-3f * (threadIdx.x & 63);
    if constexpr (MODE == 0) {
        float a[8];
        for (int k = 0; k < 8; ++k) a[k] = seed + k;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = a[k] + a[k] * c;
        }
        float s = 0;
        for (int k = 0; k < 8; ++k) s += a[k];
        out[tid] = s;
    } else if constexpr (MODE == 1) {
        f2 a[4];
        const f2 cc = {c, c};
        for (int k = 0; k < 4; ++k) a[k] = f2{seed + k, seed - k};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] = a[k] + a[k] * cc;
        }
        float s = 0;
        for (int k = 0; k < 4; ++k) s += a[k].x + a[k].y;
        out[tid] = s;
    } else if constexpr (MODE == 2) {
        float a[8];
        for (int k = 0; k < 8; ++k) a[k] = seed + k;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = __builtin_fmaf(a[k], c, a[k]);
        }
        float s = 0;
        for (int k = 0; k < 8; ++k) s += a[k];
        out[tid] = s;
    } else {
        float w[8];
        for (int k = 0; k < 8; ++k) w[k] = seed * (k + 1);
        float a = 0.f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float t;
                // keep the product inside the loop (w is loop-invariant otherwise)
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t) : "s"(c), "v"(w[k]));
                a = a + t;
            }
        }
        out[tid] = a;
    }
    if (tid == 0) {
        g_stamps[0] = t0;
        g_stamps[1] = r0;
        g_stamps[2] = __builtin_amdgcn_s_memtime();
        g_stamps[3] = __builtin_amdgcn_s_memrealtime();
    }
}

int main() {
    int dev = 0;
    CHECK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    std::printf("device %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
    float* out = nullptr;
    CHECK(hipMalloc(&out, sizeof(float) * 256 * cus * 8 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int iters = 20000;
    const char* names[4] = {"mul+add x8 chains", "pk_mul+pk_add x4 chains", "fma x8 chains", "mul(indep)+add 1 chain"};
    for (int mode = 0; mode < 4; ++mode) {
        for (int wps : {1, 2, 4, 8}) {  // waves per SIMD
            const int blocks = cus * wps;  // 256 threads = 4 waves = one wave per SIMD per block
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                CHECK(hipEventRecord(e0));
                switch (mode) {
                    case 0: hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, out, 1e-7f, iters); break;
                    case 1: hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, out, 1e-7f, iters); break;
                    case 2: hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(256), 0, 0, out, 1e-7f, iters); break;
                    default: hipLaunchKernelGGL(probe<3>, dim3(blocks), dim3(256), 0, 0, out, 1e-7f, iters); break;
                }
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            // per lane per iteration: 8 multiplies + 8 adds = 16 IEEE ops (fma mode: 8 fma = 16 flop)
            const double ops = 16.0 * iters * 256.0 * blocks;
            unsigned long long st[4];
            CHECK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof(st)));
            const double mhz = double(st[2] - st[0]) / double(st[3] - st[1]) * 100.0;
            const double cyc_per_instr = double(st[2] - st[0]) / (double(iters) * (mode == 0 || mode == 3 ? 16 : 8) * wps);
            std::printf("%-26s waves/SIMD %d : %8.3f ms  %7.2f Tops/s  shader clock %6.0f MHz  %.2f clk per wave-instr per SIMD\n",
                        names[mode], wps, best, ops / best * 1e-9, mhz, cyc_per_instr);
        }
    }
    return 0;
}
