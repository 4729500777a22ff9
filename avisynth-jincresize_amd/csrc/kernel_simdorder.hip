// kernel_simdorder.hip -- ewa_simd_order_kernel: compatibility modes that reproduce the SUMMATION ORDER of the
// reference's SIMD paths for users who diff against opt = 1 / 2 / 3 output:
//   order 1  /root/reference/src/resize_plane_sse41.cpp:41-90    4 lane-partial sums, multiply then add
//   order 2  /root/reference/src/resize_plane_avx2.cpp:45-98     8 lane-partial sums, fused multiply-add
//   order 3  /root/reference/src/resize_plane_avx512.cpp:45-103  16 lane-partial sums, fused multiply-add
// then the horizontal sum 512 -> 256 -> 128 -> (h0 + h1) + (h2 + h3), cvtps_epi32 + packus saturation for integer
// planes (to the TYPE's range, not the clip's peak) and the lower clamp of float source samples.  Selected only by the
// private switch jinc_filter_set_simd_order(); the public `opt` argument keeps meaning opt = 0 results.
// One GPU lane owns one output sample and all W partial sums of it, so the result is the reference's bit for bit
// (checked against a scalar CPU restatement in the test suite that reproduces the 3 / 7 / 7 pixel differences the reference's opt 1 / 2 / 3
// show against opt = 0 on 640x360 -> 1280x720).  This is the ONLY translation unit that contains fused multiply-adds
// (explicit __builtin_fmaf; the file is still compiled with -ffp-contract=off).  Not a fast path: no LDS staging (the common filter sizes are unrolled, with a kernel
// row's coefficients as 16-byte loads).
#include "device_common.hpp"

#pragma clang fp contract(off)

namespace jinc {
namespace {

__device__ __forceinline__ int32_t cvtps_epi32(float v) {  // x86: NaN and out-of-range -> "integer indefinite" INT_MIN
    return (v >= -2147483648.0f && v < 2147483648.0f) ? static_cast<int32_t>(__builtin_rintf(v)) : INT32_MIN;
}
__device__ __forceinline__ uint32_t packus_epi32(int32_t v) { return v < 0 ? 0u : (v > 65535 ? 65535u : static_cast<uint32_t>(v)); }
__device__ __forceinline__ uint32_t packus_epi16(uint32_t w) {  // the 16-bit pattern read as SIGNED, saturated to 0..255
    const int16_t s = static_cast<int16_t>(w);
    return s < 0 ? 0u : (s > 255 ? 255u : static_cast<uint32_t>(s));
}

// FS: the filter size at compile time (7, 9, 13, 17: taps 3, 4, 6, 8 at >= 1x -- rows unrolled, a kernel row's coefficients as
// 16-byte loads), or 0: any size at run time.
template <typename T, int W, bool FUSED, int FS>
__global__ __launch_bounds__(256) void ewa_simd_order_kernel(const DevicePlan p, const PlaneIO io, const float min_val) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= p.dst_w || y >= p.dst_h) return;
    const size_t frame = blockIdx.z;
    const int fs = FS ? FS : p.fs, fsp = padded_row(fs);
    const int rc = p.row_class[y], cc = p.col_class[x];
    int set;
    if (rc < 0)
        set = p.brow_set[static_cast<size_t>(~rc) * p.dst_w + x];
    else if (cc < 0)
        set = p.bcol_set[static_cast<size_t>(~cc) * p.dst_h + y];
    else
        set = p.interior_set[rc * p.n_col_classes + cc];
    const float* c = p.coeffs + static_cast<size_t>(set) * fs * fsp;
    const char* srow = static_cast<const char*>(io.src) + frame * io.src_frame_stride +
                       static_cast<size_t>(p.row_start[y]) * io.src_pitch + static_cast<size_t>(p.col_start[x]) * sizeof(T);
    float part[W];
#pragma unroll
    for (int l = 0; l < W; ++l) part[l] = 0.f;
    auto tap = [&](float& sum, float v, float cv) {
        if constexpr (std::is_same_v<T, float>) v = v > min_val ? v : min_val;  // _mm_max_ps(src, min_val)
        if constexpr (FUSED)
            sum = __builtin_fmaf(v, cv, sum);
        else
            sum = sum + v * cv;
    };
    if constexpr (FS != 0) {
        constexpr int FSP = padded_row(FS);
        for (int ly = 0; ly < FS; ++ly) {
            const T* s = reinterpret_cast<const T*>(srow);
            float cr[FSP];  // the kernel row's coefficients (rows are padded to multiples of 4 floats, 16-byte aligned)
#pragma unroll
            for (int q = 0; q < FSP / 4; ++q) {
                const float4 v4 = *reinterpret_cast<const float4*>(c + 4 * q);
                cr[4 * q] = v4.x, cr[4 * q + 1] = v4.y, cr[4 * q + 2] = v4.z, cr[4 * q + 3] = v4.w;
            }
            T sv[FS];
#pragma unroll
            for (int lx = 0; lx < FS; ++lx) sv[lx] = s[lx];
#pragma unroll
            for (int lx = 0; lx < FS; ++lx) tap(part[lx % W], to_float(sv[lx]), cr[lx]);  // (lanes past the window: zero padding, no-ops)
            c += FSP;
            srow += io.src_pitch;
        }
    } else {
        for (int ly = 0; ly < fs; ++ly) {
            const T* s = reinterpret_cast<const T*>(srow);
            for (int lx0 = 0; lx0 < fs; lx0 += W) {
#pragma unroll
                for (int l = 0; l < W; ++l) {
                    const int lx = lx0 + l;
                    if (lx < fs) tap(part[l], to_float(s[lx]), c[lx]);  // (lanes past the window multiply by the row's zero padding: no-ops)
                }
            }
            c += fsp;
            srow += io.src_pitch;
        }
    }
    float q[8], h[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if constexpr (W == 16)
            q[i] = part[i] + part[i + 8];
        else
            q[i] = part[i < W ? i : 0];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if constexpr (W >= 8)
            h[i] = q[i] + q[i + 4];
        else
            h[i] = q[i];
    }
    const float r = (h[0] + h[1]) + (h[2] + h[3]);
    T* d = reinterpret_cast<T*>(static_cast<char*>(io.dst) + frame * io.dst_frame_stride + static_cast<size_t>(y) * io.dst_pitch) + x;
    if constexpr (std::is_same_v<T, float>)
        *d = r;
    else if constexpr (std::is_same_v<T, uint16_t>)
        *d = static_cast<uint16_t>(packus_epi32(cvtps_epi32(r)));
    else
        *d = static_cast<uint8_t>(packus_epi16(packus_epi32(cvtps_epi32(r))));
}

template <typename T>
int launch_so_t(const DevicePlan& p, const PlaneIO& io, int order, float min_val, hipStream_t s) {
    const dim3 grid((p.dst_w + 63) / 64, (p.dst_h + 3) / 4, io.nframes), block(256);
#define JINC_SO_LAUNCH(FS)                                                                                                       \
    switch (order) {                                                                                                             \
        case 1: hipLaunchKernelGGL((ewa_simd_order_kernel<T, 4, false, FS>), grid, block, 0, s, p, io, min_val); break;          \
        case 2: hipLaunchKernelGGL((ewa_simd_order_kernel<T, 8, true, FS>), grid, block, 0, s, p, io, min_val); break;           \
        default: hipLaunchKernelGGL((ewa_simd_order_kernel<T, 16, true, FS>), grid, block, 0, s, p, io, min_val); break;         \
    }
    switch (p.fs) {
        case 7: JINC_SO_LAUNCH(7) break;
        case 9: JINC_SO_LAUNCH(9) break;
        case 13: JINC_SO_LAUNCH(13) break;
        case 17: JINC_SO_LAUNCH(17) break;
        default: JINC_SO_LAUNCH(0) break;
    }
#undef JINC_SO_LAUNCH
    return static_cast<int>(hipGetLastError());
}

}  // namespace

int launch_simd_order(const DevicePlan& plan, const PlaneIO& io, int order, float min_val, void* stream) {
    if (io.nframes <= 0 || plan.dst_w <= 0 || plan.dst_h <= 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (io.sample_bytes) {
        case 1: return launch_so_t<uint8_t>(plan, io, order, min_val, s);
        case 2: return launch_so_t<uint16_t>(plan, io, order, min_val, s);
        default: return launch_so_t<float>(plan, io, order, min_val, s);
    }
}

}  // namespace jinc
