// residency_probe.hip -- measurement tool (not part of the product).
// Question: how many waves are really co-resident per SIMD in a VALU-only kernel, and what does one
// wave64 VALU instruction cost per SIMD when the SIMD is saturated?  Every wave stamps s_memtime at its
// start and end; the host sweeps the intervals.
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize residency_probe.hip -o residency_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#pragma clang fp contract(off)

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// one add chain fed by independent products (the EWA kernel's shape), `UNROLL` taps per iteration
__global__ void chain(float* out, unsigned long long* stamps, float c, int iters) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float w[8];
    const float seed = 1.0f + 1e-3f * (threadIdx.x & 63);
    for (int k = 0; k < 8; ++k) w[k] = seed * (k + 1);
    float a = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float t;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t) : "s"(c), "v"(w[k]));
            a = a + t;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) {
        const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        stamps[2 * wave] = t0;
        stamps[2 * wave + 1] = t1;
    }
}

int main() {
    CHECK(hipSetDevice(0));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, simds = cus * 4;
    const int iters = 20000;
    float* out;
    unsigned long long* stamps;
    const size_t max_waves = size_t(cus) * 32 * 4;
    CHECK(hipMalloc(&out, max_waves * 64 * sizeof(float)));
    CHECK(hipMalloc(&stamps, max_waves * 2 * sizeof(unsigned long long)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::printf("%-10s %-10s %8s %9s %12s %12s %14s\n", "block", "waves/SIMD", "ms", "Tops/s", "avg resident", "max resident", "clk/instr/SIMD");
    for (int block : {64, 256, 512, 1024}) {
        for (int wps : {1, 2, 3, 4, 6, 8, 16}) {
            const long long waves = (long long)simds * wps;
            const int blocks = int(waves * 64 / block);
            if (size_t(waves) > max_waves) continue;
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(chain, dim3(blocks), dim3(block), 0, 0, out, stamps, 1e-7f, iters);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                best = std::min(best, ms);
            }
            std::vector<unsigned long long> st(waves * 2);
            CHECK(hipMemcpy(st.data(), stamps, st.size() * sizeof(st[0]), hipMemcpyDeviceToHost));
            std::vector<std::pair<unsigned long long, int>> ev;
            unsigned long long lo = ~0ull, hi = 0, life = 0;
            for (long long w = 0; w < waves; ++w) {
                ev.push_back({st[2 * w], +1});
                ev.push_back({st[2 * w + 1], -1});
                lo = std::min(lo, st[2 * w]);
                hi = std::max(hi, st[2 * w + 1]);
                life += st[2 * w + 1] - st[2 * w];
            }
            std::sort(ev.begin(), ev.end());
            long long cur = 0, mx = 0;
            for (auto& e : ev) { cur += e.second; mx = std::max(mx, cur); }
            const double span = double(hi - lo);
            const double ops = 16.0 * iters * 64.0 * waves;
            const double instr_per_simd = 16.0 * iters * wps;
            std::printf("%-10d %-10d %8.3f %9.2f %12.2f %12.2f %14.2f\n", block, wps, best, ops / best * 1e-9,
                        double(life) / span / simds, double(mx) / simds, span / instr_per_simd);
        }
    }
    return 0;
}
