// device_common.hpp -- helpers shared by the gfx950 kernels of libjincresize_hip.so (included by every
// kernel_*.hip translation unit; everything lives in an anonymous namespace, so each TU has its own copy).
//
// The result of the resampling path is DEFINED by the reference's opt=0 code: per output sample a strictly
// sequential fp32 chain   r = 0; for ly: for lx: r = fl(r + fl(float(src) * coeff))
// (/root/reference/src/JincResize.cpp:570-579) followed by clamp + round-half-even for integer planes
// (ref :581-582).  Therefore, in every kernel:
//   * one lane owns one output sample's whole chain; cross-lane operations move data only;
//   * multiply and add stay un-fused: the kernel files are compiled with -ffp-contract=off AND carry the
//     pragma below; tests/test_build.py greps the ISA for v_fma / v_fmac / v_mad / v_pk_fma / v_mfma;
//   * fp32 denormals are kept (gfx9 default; no -fgpu-flush-denormals-to-zero).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

#pragma clang fp contract(off)

namespace jinc {
namespace {

#define JINC_CONSTANT __attribute__((address_space(4)))

template <typename T>
__device__ __forceinline__ float to_float(T v) {
    return static_cast<float>(v);
}

// ref :581-582 -- clamp(result, 0, peak) then lrintf (round-half-even).  For every non-NaN input
// v_med3_f32(r, 0, peak) equals the reference's "upper bound first, then lower" clamp; a NaN (only
// reachable through non-finite coefficients) ends as 0 on both sides.
__device__ __forceinline__ uint32_t round_sample(float r, float peak) {
    return static_cast<uint32_t>(__builtin_rintf(__builtin_amdgcn_fmed3f(r, 0.f, peak)));
}

// 8-bit planes (peak is always 255): v_cvt_pk_u8_f32 rounds to nearest even (MODE.fp_round default) and
// saturates to [0, 255] in one instruction -- the same value as clamp + lrintf for every input, NaN -> 0.
// tests/test_gpu_parity.py::test_integer_conversion_ties checks ties, bounds and specials on the device.
__device__ __forceinline__ uint32_t round_sample_u8(float r) { return __builtin_amdgcn_cvt_pk_u8_f32(r, 0u, 0u); }

// Two 9 ... 16-bit samples as one dword, (a) in the low half: clamp as round_sample, then round-half-even by adding 2^23 -- the
// sum of a value in [0, 65535] and 8388608.0f has an ulp of 1, so its low mantissa bits ARE the sample rounded to nearest, ties to
// even (2^23 is even: the tie rule of the sum is the tie rule of the sample) -- and v_perm_b32 picks the two low halves: five
// instructions per pair against seven (v_med3, v_rndne, v_cvt_u32 each and a v_lshl_or).  The same value as round_sample for every
// input (NaN: v_med3 returns 0 first); tests/test_gpu_parity.py::test_integer_conversion_ties runs both forms over ties and bounds.
__device__ __forceinline__ uint32_t round_pair_u16(float a, float b, float peak) {
    const float ca = __builtin_amdgcn_fmed3f(a, 0.f, peak) + 8388608.0f, cb = __builtin_amdgcn_fmed3f(b, 0.f, peak) + 8388608.0f;
    return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, cb), __builtin_bit_cast(uint32_t, ca), 0x05040100u);
}

template <typename T>
__device__ __forceinline__ T convert_sample(float r, float peak) {
    if constexpr (std::is_same_v<T, float>)
        return r;
    else if constexpr (std::is_same_v<T, uint8_t>)
        return static_cast<uint8_t>(round_sample_u8(r));
    else
        return static_cast<T>(round_sample(r, peak));
}

template <typename T>
__device__ __forceinline__ void store_sample(T* p, float r, float peak) {
    *p = convert_sample<T>(r, peak);
}

// Store through a buffer resource: per-lane byte offset in a VGPR that never changes, the row offset in
// an SGPR -- no address arithmetic on the VALU (the binding unit of these kernels) per stored sample.
using BufferRsrc = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ BufferRsrc make_rsrc(void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, bytes, 0x00020000);
}
template <typename T>
__device__ __forceinline__ void store_sample_buf(BufferRsrc rsrc, uint32_t voffset, uint32_t soffset, float r, float peak) {
    if constexpr (std::is_same_v<T, float>)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, r), rsrc, voffset, soffset, 0);
    else if constexpr (std::is_same_v<T, uint8_t>)
        __builtin_amdgcn_raw_buffer_store_b8(static_cast<uint8_t>(round_sample_u8(r)), rsrc, voffset, soffset, 0);
    else
        __builtin_amdgcn_raw_buffer_store_b16(static_cast<uint16_t>(round_sample(r, peak)), rsrc, voffset, soffset, 0);
}

// Coefficient rows are padded to a multiple of 4 floats on the device (16-byte aligned rows).
__host__ __device__ constexpr int padded_row(int fs) { return (fs + 3) & ~3; }

// XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (each with its own L2) in
// linear-id order, so neighbouring linear ids land on different XCDs and every XCD would fetch its own copy
// of the halo rows/columns shared by adjacent tiles.  Re-map the linear id so that each XCD walks a
// contiguous run of tiles of the frame (bijective for any tile count; placement affects speed only).
__device__ __forceinline__ void swizzled_tile(int& tx, int& ty) {
    const int n = gridDim.x * gridDim.y;
    const int lin = blockIdx.y * gridDim.x + blockIdx.x;
    constexpr int kXcds = 8;
    const int q = n / kXcds, rem = n % kXcds;
    const int xcd = lin % kXcds, idx = lin / kXcds;
    // XCD k owns q (+1 if k < rem) consecutive tiles starting at k*q + min(k, rem)
    const int tile_id = xcd * q + (xcd < rem ? xcd : rem) + idx;
    ty = tile_id / gridDim.x;
    tx = tile_id - ty * gridDim.x;
}

}  // namespace
}  // namespace jinc
