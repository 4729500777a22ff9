// kernel_probe.hip -- measurement hooks, not part of the hot path: a shader-clock sampler that runs BESIDE the kernels
// being timed (bench.py: roofline.shader_clock_ghz).  One lane per workgroup, eight workgroups (consecutive workgroups go to
// consecutive XCDs), each stamps s_memtime (shader clock ticks) and s_memrealtime (constant 100 MHz) when it starts and
// when the host raises the stop flag (device memory, set by a memset on another stream); clock = d(memtime) / d(memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS item 6).
// The samplers sleep between looks at the flag: a few dozen scalar instructions per microsecond on 8 of 1024 SIMDs.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace jinc {
namespace {

__global__ __launch_bounds__(64) void clock_sampler_kernel(const volatile int* stop, unsigned long long* out, unsigned long long max_real_ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long c1 = c0, r1 = r0;
    // exit: the host's flag, or max_real_ticks of wall time (a host that died must not leave waves spinning)
    while (r1 - r0 < max_real_ticks) {
        __builtin_amdgcn_s_sleep(127);
        __builtin_amdgcn_s_sleep(127);
        c1 = __builtin_amdgcn_s_memtime();
        r1 = __builtin_amdgcn_s_memrealtime();
        if (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;  // device memory, served by the L2
    }
    out[2 * blockIdx.x] = c1 - c0;
    out[2 * blockIdx.x + 1] = r1 - r0;
}

// What plain v_mul_f32 + v_add_f32 sustain on this part: the hot kernels' instruction pair (coefficient in an SGPR, sample
// in a VGPR, one accumulation chain per lane, products independent of the chain), nothing else in the loop.  8 taps per
// pass, `iters` passes; the caller fills the chip with `waves_per_simd` waves per SIMD and times the launch.
__global__ __launch_bounds__(256) void valu_pair_probe_kernel(float* out, float c, int iters) {
    float w[8];
    const float seed = 1.0f + 1e-3f * (threadIdx.x & 63);
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = seed * (k + 1);
    float a = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float t;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t) : "s"(c), "v"(w[k]));  // stays in the loop: w is loop-invariant
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(t));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}

}  // namespace

int launch_valu_pair_probe(float* out, int blocks, int iters, void* stream) {
    hipLaunchKernelGGL(valu_pair_probe_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, static_cast<hipStream_t>(stream), out, 1e-7f, iters);
    return hipGetLastError();
}

int launch_clock_sampler(const int* stop_flag, unsigned long long* out, int samplers, double max_seconds, void* stream) {
    hipLaunchKernelGGL(clock_sampler_kernel, dim3(static_cast<unsigned>(samplers)), dim3(64), 0, static_cast<hipStream_t>(stream), stop_flag, out,
                       static_cast<unsigned long long>(max_seconds * 1e8));
    return hipGetLastError();
}

}  // namespace jinc
