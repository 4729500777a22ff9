// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for the EWA-Jinc gather-MAC.
//
// Replaces resize_plane_{c,sse41,avx2,avx512}<T,thr,subsampled> and their row fan-out
// (/root/reference/src/JincResize.cpp:536-601, resize_plane_*.cpp).  The result is DEFINED by the
// reference's opt=0 path: per output sample a strictly sequential fp32 chain
//     r = 0; for ly: for lx: r = fl(r + fl(float(src) * coeff))          (ref :570-579)
// followed by clamp + round-half-even for integer planes (ref :581-582).  Therefore:
//   * one lane owns one output sample's whole chain; cross-lane operations move data only;
//   * multiply and add stay un-fused: this file is compiled with -ffp-contract=off AND carries the
//     pragma below; the build greps the ISA for v_fma/v_fmac/v_mad/v_pk_fma (tests/test_build.py);
//   * fp32 denormals are kept (gfx9 default; no -fgpu-flush-denormals-to-zero).
//
// Two kernels:
//   ewa_gather_kernel   -- any plan: one lane per output pixel, per-lane window origin and
//                          coefficient-set pointer (plan arrays are L2-resident).
//   ewa_periodic_kernel -- phase-periodic interior (integer upscales): a wave owns ONE phase, so its
//                          fs*fs coefficients are wave-uniform and live in SGPRs (s_load through a
//                          constant-address-space pointer; v_mul_f32 takes the SGPR operand directly);
//                          lanes are consecutive source-aligned columns; the source tile is converted
//                          to fp32 once and staged in LDS; each lane keeps an fs x fs register window
//                          that slides down the tile (one new LDS row per output row, rotation done by
//                          compile-time unrolling, no register moves).
#include <hip/hip_runtime.h>

#include <cstdint>
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

// ------------------------------------------------------------------------------------------------
// Generic gather kernel
// ------------------------------------------------------------------------------------------------
struct GatherArgs {
    DevicePlan plan;
    PlaneIO io;
    RectList rects;
    int block_begin[5];
    int blocks_a[4];  // blocks along the lane axis of each rectangle
    int lane_axis[4]; // 0: lanes run along x (wide rectangles), 1: along y (border columns)
    int stride[4];    // lane stride P: lane l of an item handles coordinate origin + P*l + residue
    int lines[4];     // lines (rows for lane_axis 0, columns for lane_axis 1) per block
};

constexpr int kGatherLdsFloats = 6144;  // 24 KB source tile per block

// Coefficient rows are padded to a multiple of 4 floats on the device (16-byte aligned rows).
__host__ __device__ constexpr int padded_row(int fs) { return (fs + 3) & ~3; }

// Any plan.  A block covers 64*P coordinates along the lane axis x 4..32 lines; its source footprint is
// staged once in LDS as fp32.  Work items = (line, residue): the 64 lanes of an item are P apart, P
// being the plan's dominant phase period, so that (nearly) all lanes of an item share one coefficient
// set.  The item then runs a waterfall over the distinct sets actually present: the set of the first
// pending lane is made wave-uniform (readlane), its coefficients are fetched with scalar loads into
// SGPRs and every lane that uses this set runs its sequential chain.  For exactly periodic plans
// that is one pass; ratios whose phases drift (1.5x, 3x: the reference accumulates positions in
// float) add a pass per deviation.  There is no per-lane coefficient traffic.
template <typename T, int FS>
__global__ __launch_bounds__(256) void ewa_gather_kernel(const GatherArgs a) {
    __shared__ float tile[kGatherLdsFloats];
    const DevicePlan& p = a.plan;
    const int b = blockIdx.x;
    int r = 0;
    while (r + 1 < a.rects.n && b >= a.block_begin[r + 1]) ++r;
    const int local = b - a.block_begin[r];
    const int axis = a.lane_axis[r];
    const int P = a.stride[r];
    const int ba = local % a.blocks_a[r];  // block index along the lane axis
    const int bl = local / a.blocks_a[r];  // block index along the line axis
    const int rx0 = a.rects.x0[r], ry0 = a.rects.y0[r];
    const int rx1 = rx0 + a.rects.w[r], ry1 = ry0 + a.rects.h[r];
    // block extent in output pixels (inclusive last pixel), all wave-uniform
    const int nlines = a.lines[r];
    const int bx0 = axis == 0 ? rx0 + ba * 64 * P : rx0 + bl * nlines;
    const int by0 = axis == 0 ? ry0 + bl * nlines : ry0 + ba * 64 * P;
    const int bx1 = min(bx0 + (axis == 0 ? 64 * P : nlines), rx1) - 1;
    const int by1 = min(by0 + (axis == 0 ? nlines : 64 * P), ry1) - 1;

    const int fs = FS ? FS : p.fs;
    const int fsp = FS ? padded_row(FS) : padded_row(p.fs);
    const size_t frame = blockIdx.y;
    const char* sframe = static_cast<const char*>(a.io.src) + frame * a.io.src_frame_stride;
    char* dframe = static_cast<char*>(a.io.dst) + frame * a.io.dst_frame_stride;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // Source footprint of the block (window origins are non-decreasing in x and in y).
    const int tx0 = p.col_start[bx0], ty0 = p.row_start[by0];
    const int tw = p.col_start[bx1] + fs - tx0;
    const int th = p.row_start[by1] + fs - ty0;
    const int pitch = tw | 1;
    const bool staged = pitch * th <= kGatherLdsFloats;  // wave-uniform; huge down-scales read global memory
    if (staged) {
        for (int rr = wave; rr < th; rr += 4) {
            const T* srow = reinterpret_cast<const T*>(sframe + static_cast<size_t>(ty0 + rr) * a.io.src_pitch) + tx0;
            for (int c = lane; c < tw; c += 64) tile[rr * pitch + c] = to_float(srow[c]);
        }
    }
    __syncthreads();

    const int nitems = nlines * P;
    for (int item = wave; item < nitems; item += 4) {
        const int line = item / P;
        const int res = item - line * P;
        const int x = axis == 0 ? bx0 + P * lane + res : bx0 + line;
        const int y = axis == 0 ? by0 + line : by0 + P * lane + res;
        const bool active = x <= bx1 && y <= by1;
        if (!__builtin_amdgcn_readfirstlane(__ballot(active) != 0)) continue;

        // Inactive lanes (block overhang) look up the block's first pixel so every address stays in range.
        const int qx = active ? x : bx0, qy = active ? y : by0;
        const int sx = p.col_start[qx];
        const int sy = p.row_start[qy];
        const int rc = p.row_class[qy];
        const int cc = p.col_class[qx];
        int set;
        if (rc < 0)
            set = p.brow_set[static_cast<size_t>(~rc) * p.dst_w + qx];
        else if (cc < 0)
            set = p.bcol_set[static_cast<size_t>(~cc) * p.dst_h + qy];
        else
            set = p.interior_set[rc * p.n_col_classes + cc];

        float acc = 0.f;
        if constexpr (FS != 0) {
            if (staged) {
                const float* s = tile + (sy - ty0) * pitch + (sx - tx0);
                constexpr bool kWindowInRegs = FS <= 9;  // small windows: read LDS once, reuse across passes
                float w[kWindowInRegs ? FS * FS : 1];
                if constexpr (kWindowInRegs) {
#pragma unroll
                    for (int ly = 0; ly < FS; ++ly)
#pragma unroll
                        for (int lx = 0; lx < FS; ++lx) w[ly * FS + lx] = s[ly * pitch + lx];
                }
                unsigned long long todo = __ballot(active);
                while (todo) {
                    const int leader = __ffsll(static_cast<long long>(todo)) - 1;
                    const int u = __builtin_amdgcn_readlane(set, leader);
                    const bool mine = active && set == u;
                    const JINC_CONSTANT float* cs =
                        (const JINC_CONSTANT float*)(p.coeffs + static_cast<size_t>(u) * (FS * padded_row(FS)));
                    if (mine) {
                        if constexpr (kWindowInRegs) {
#pragma unroll
                            for (int ly = 0; ly < FS; ++ly)
#pragma unroll
                                for (int lx = 0; lx < FS; ++lx)
                                    acc = acc + w[ly * FS + lx] * cs[ly * padded_row(FS) + lx];
                        } else {
                            const float* sr = s;
                            for (int ly = 0; ly < FS; ++ly) {
                                float c[FS];
#pragma unroll
                                for (int lx = 0; lx < FS; ++lx) c[lx] = cs[ly * padded_row(FS) + lx];
#pragma unroll
                                for (int lx = 0; lx < FS; ++lx) acc = acc + sr[lx] * c[lx];
                                sr += pitch;
                            }
                        }
                    }
                    todo &= ~__ballot(mine);
                }
            }
        }
        if (FS == 0 || !staged) {
            // Fallback: run-time filter size (even sizes of down-scales, fs > 17) or a source footprint
            // larger than the LDS tile: per-lane loads through L1/L2.
            const float* c = p.coeffs + static_cast<size_t>(set) * fs * fsp;
            if (staged) {
                const float* s = tile + (sy - ty0) * pitch + (sx - tx0);
                for (int ly = 0; ly < fs; ++ly) {
                    for (int lx = 0; lx < fs; ++lx) acc = acc + s[lx] * c[lx];
                    s += pitch;
                    c += fsp;
                }
            } else {
                const char* srow = sframe + static_cast<size_t>(sy) * a.io.src_pitch + static_cast<size_t>(sx) * sizeof(T);
                for (int ly = 0; ly < fs; ++ly) {
                    const T* s = reinterpret_cast<const T*>(srow);
                    for (int lx = 0; lx < fs; ++lx) acc = acc + to_float(s[lx]) * c[lx];
                    srow += a.io.src_pitch;
                    c += fsp;
                }
            }
        }
        if (active) {
            T* d = reinterpret_cast<T*>(dframe + static_cast<size_t>(y) * a.io.dst_pitch) + x;
            store_sample<T>(d, acc, a.io.peak);
        }
    }
}

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

// ------------------------------------------------------------------------------------------------
// Periodic interior kernel
// ------------------------------------------------------------------------------------------------
constexpr int kTileCols = 64;  // source-aligned columns per tile = lanes of a wave

// RG: row groups of FS rows per tile.  A/B on MI355X (C2): 8 groups 2.3 % faster than 4, 16 groups 10 % slower.
template <int FS, int RG = (FS <= 7 ? 8 : 6)>
struct PeriodicCfg {
    static constexpr int kRowGroups = RG;
    static constexpr int kTileRows = FS * kRowGroups;     // period-rows per tile (multiple of FS)
    static constexpr int kLdsCols = kTileCols + FS;       // 64 + (FS-1) halo + 1 phase spread
    static constexpr int kLdsPitch = kLdsCols + 1;        // odd pitch not needed for row reads; keeps staging writes spread
    static constexpr int kLdsRows = kTileRows + FS;       // TJ + (FS-1) halo + 1 phase spread
};

template <typename T, int FS, int RG>
__global__ __launch_bounds__(256) void ewa_periodic_kernel(const PeriodicArgs a, const PlaneIO io) {
    using Cfg = PeriodicCfg<FS, RG>;
    __shared__ float tile[Cfg::kLdsRows * Cfg::kLdsPitch];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int tile_x, tile_y;
    swizzled_tile(tile_x, tile_y);
    const int i0 = tile_x * kTileCols;
    const int j0 = tile_y * Cfg::kTileRows;
    const size_t frame = blockIdx.z;

    // ---- stage the source tile as fp32 (each source sample converted once) ----
    {
        const int gx0 = a.min_sx + i0;
        const int gy0 = a.min_sy + j0;
        const char* sbase = static_cast<const char*>(io.src) + frame * io.src_frame_stride;
        for (int r = wave; r < Cfg::kLdsRows; r += 4) {
            int gy = gy0 + r;
            gy = gy < a.src_h ? gy : a.src_h - 1;
            const T* srow = reinterpret_cast<const T*>(sbase + static_cast<size_t>(gy) * io.src_pitch);
#pragma unroll
            for (int c = lane; c < Cfg::kLdsCols; c += 64) {
                int gx = gx0 + c;
                gx = gx < a.src_w ? gx : a.src_w - 1;
                tile[r * Cfg::kLdsPitch + c] = to_float(srow[gx]);
            }
        }
    }
    __syncthreads();

    const int nphase = a.px * a.py;
    for (int ph = wave; ph < nphase; ph += 4) {
        const int q = ph / a.px;
        const int p = ph - q * a.px;

        // wave-uniform coefficients -> SGPRs
        const JINC_CONSTANT float* cs =
            (const JINC_CONSTANT float*)(a.coeffs + static_cast<size_t>(a.set[ph]) * (FS * padded_row(FS)));
        float cf[FS * FS];
#pragma unroll
        for (int k = 0; k < FS * FS; ++k) cf[k] = cs[(k / FS) * padded_row(FS) + (k % FS)];

        const float* base = tile + (a.start_y[q] - a.min_sy) * Cfg::kLdsPitch + (a.start_x[p] - a.min_sx) + lane;

        const unsigned x = a.ix0 + a.px * (i0 + lane) + p;  // per-lane output column
        if ((i0 + lane) >= a.ni) continue;                   // whole phase loop under one exec mask
        const BufferRsrc drsrc = make_rsrc(static_cast<char*>(io.dst) + frame * io.dst_frame_stride,
                                           static_cast<uint32_t>(io.dst_pitch) * a.dst_h);  // wave-uniform
        const uint32_t xoff = x * static_cast<uint32_t>(sizeof(T));

        float win[FS][FS];
#pragma unroll
        for (int r = 0; r < FS - 1; ++r)
#pragma unroll
            for (int lx = 0; lx < FS; ++lx) win[r][lx] = base[r * Cfg::kLdsPitch + lx];

        for (int g = 0; g < Cfg::kRowGroups; ++g) {
            if (j0 + g * FS >= a.nj) break;  // wave-uniform: the bottom tiles usually need fewer groups
            const float* gbase = base + (g * FS) * Cfg::kLdsPitch;
#pragma unroll
            for (int u = 0; u < FS; ++u) {
                // newest window row: tile row g*FS + u + FS-1 -> slot (u + FS-1) % FS
#pragma unroll
                for (int lx = 0; lx < FS; ++lx)
                    win[(u + FS - 1) % FS][lx] = gbase[(u + FS - 1) * Cfg::kLdsPitch + lx];

                float acc = 0.f;
#pragma unroll
                for (int ly = 0; ly < FS; ++ly)
#pragma unroll
                    for (int lx = 0; lx < FS; ++lx) acc = acc + win[(u + ly) % FS][lx] * cf[ly * FS + lx];

                const int j = j0 + g * FS + u;  // wave-uniform
                if (j < a.nj) {
                    const int y = a.iy0 + a.py * j + q;
                    store_sample_buf<T>(drsrc, xoff, static_cast<uint32_t>(y) * io.dst_pitch, acc, io.peak);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Quasi-periodic interior kernel (ratios whose phases drift: 1.5x, 3x, ...)
// ------------------------------------------------------------------------------------------------
// The reference accumulates output positions in float, so for ratios such as 3/2 or 3 the quantised
// phase of a column/row is only NEARLY periodic: along 1920 columns a residue class changes its
// coefficient class about ten times.  The window ORIGINS, however, stay exactly affine per residue.
// This kernel keeps everything of ewa_periodic_kernel that depends on the origins only -- source tile
// staged once as fp32 in LDS, lanes = consecutive periods, fs x fs register window sliding down the
// tile with compile-time rotation (SY new source rows per output row) -- and looks the coefficient set
// up per (row, lane): row class from an LDS copy (wave-uniform), lane class loaded once per phase, set id
// from an LDS copy of the class-pair table.  The set of the previous row stays in SGPRs; a waterfall
// over the distinct sets of the wave reloads them only at the rare change points, so the common case is
// again 2 VALU instructions per tap with wave-uniform SGPR coefficients.
template <typename T, int FS, int SX, int SY>
__global__ __launch_bounds__(256) void ewa_quasi_kernel(const QuasiArgs a, const PlaneIO io) {
    extern __shared__ __attribute__((aligned(16))) float q_smem[];
    float* tile = q_smem;
    int* l_iset = reinterpret_cast<int*>(tile + a.lds_rows * a.lds_pitch);
    int* l_rc = l_iset + a.n_row_classes * a.n_col_classes;

    const int nthreads = blockDim.x;
    const int nwaves = nthreads >> 6;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile_rows = a.rg * FS;  // output rows (per row phase) of a tile
    int tile_x, tile_y;
    swizzled_tile(tile_x, tile_y);
    const int i0 = tile_x * 64;
    const int j0 = tile_y * tile_rows;
    const size_t frame = blockIdx.z;
    const int pitch = a.lds_pitch;

    {
        const int gx0 = a.min_sx + SX * i0;
        const int gy0 = a.min_sy + SY * j0;
        const char* sbase = static_cast<const char*>(io.src) + frame * io.src_frame_stride;
        for (int r = wave; r < a.lds_rows; r += nwaves) {
            int gy = gy0 + r;
            gy = gy < a.src_h ? gy : a.src_h - 1;
            const T* srow = reinterpret_cast<const T*>(sbase + static_cast<size_t>(gy) * io.src_pitch);
            for (int c = lane; c < a.lds_cols; c += 64) {
                int gx = gx0 + c;
                gx = gx < a.src_w ? gx : a.src_w - 1;
                tile[r * pitch + c] = to_float(srow[gx]);
            }
        }
        const int ntab = a.n_row_classes * a.n_col_classes;
        for (int k = threadIdx.x; k < ntab; k += nthreads) l_iset[k] = a.interior_set[k];
        // classes of the tile's output rows, in output order: entry py*jj + q <-> row iy0 + py*(j0+jj) + q
        const int nrows = a.py * tile_rows;
        for (int k = threadIdx.x; k < nrows; k += nthreads) {
            const int jj = k / a.py;
            l_rc[k] = (j0 + jj) < a.nj ? a.row_class[a.iy0 + a.py * j0 + k] : 0;
        }
    }
    __syncthreads();

    const BufferRsrc drsrc = make_rsrc(static_cast<char*>(io.dst) + frame * io.dst_frame_stride,
                                       static_cast<uint32_t>(io.dst_pitch) * a.dst_h);
    const int nphase = a.px * a.py;
    for (int ph = wave; ph < nphase; ph += nwaves) {
        const int q = ph / a.px;
        const int p = ph - q * a.px;
        const int ia = i0 + lane;
        const bool valid = ia < a.ni;
        const unsigned long long valid_mask = __ballot(valid);
        if (valid_mask == 0) continue;
        const unsigned x = a.ix0 + a.px * (valid ? ia : i0) + p;
        const int cc = a.col_class[x];  // one global load per phase and lane
        const uint32_t xoff = x * static_cast<uint32_t>(sizeof(T));
        const float* base = tile + (a.start_y[q] - a.min_sy) * pitch + (a.start_x[p] - a.min_sx) + SX * lane;

        float win[FS][FS];
#pragma unroll
        for (int r = 0; r < FS - SY; ++r)
#pragma unroll
            for (int lx = 0; lx < FS; ++lx) win[r][lx] = base[r * pitch + lx];

        float cf[FS * FS];
#pragma unroll
        for (int k = 0; k < FS * FS; ++k) cf[k] = 0.f;
        int cur_set = -1;

        for (int g = 0; g < a.rg; ++g) {
            if (j0 + g * FS >= a.nj) break;  // wave-uniform
            const float* gbase = base + (SY * g * FS) * pitch;
#pragma unroll
            for (int t = 0; t < FS; ++t) {
                // window of output row t covers relative source rows SY*t .. SY*t+FS-1; the SY newest arrive now
#pragma unroll
                for (int k = 0; k < SY; ++k) {
                    const int rel = SY * t + FS - SY + k;
#pragma unroll
                    for (int lx = 0; lx < FS; ++lx) win[rel % FS][lx] = gbase[rel * pitch + lx];
                }
                const int jj = g * FS + t;
                if (j0 + jj < a.nj) {  // wave-uniform
                    const int rc = __builtin_amdgcn_readfirstlane(l_rc[a.py * jj + q]);
                    const int set = l_iset[rc * a.n_col_classes + cc];
                    float acc = 0.f;
                    unsigned long long todo = valid_mask;
                    while (todo) {
                        const int leader = __ffsll(static_cast<long long>(todo)) - 1;
                        const int u = __builtin_amdgcn_readlane(set, leader);
                        if (u != cur_set) {  // wave-uniform: only at the class change points of the drift
                            const JINC_CONSTANT float* cs =
                                (const JINC_CONSTANT float*)(a.coeffs + static_cast<size_t>(u) * (FS * padded_row(FS)));
#pragma unroll
                            for (int k = 0; k < FS * FS; ++k) cf[k] = cs[(k / FS) * padded_row(FS) + (k % FS)];
                            cur_set = u;
                        }
                        const bool mine = valid && set == u;
                        if (mine) {
#pragma unroll
                            for (int ly = 0; ly < FS; ++ly)
#pragma unroll
                                for (int lx = 0; lx < FS; ++lx)
                                    acc = acc + win[(SY * t + ly) % FS][lx] * cf[ly * FS + lx];
                        }
                        todo &= ~__ballot(mine);
                    }
                    if (valid) {
                        const int y = a.iy0 + a.py * (j0 + jj) + q;
                        store_sample_buf<T>(drsrc, xoff, static_cast<uint32_t>(y) * io.dst_pitch, acc, io.peak);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Periodic interior kernel, packed-math form (experimental A/B variant)
// ------------------------------------------------------------------------------------------------
// Same algorithm as ewa_periodic_kernel, but a lane owns TWO source-aligned columns 64 apart and keeps
// both register windows as 2-vectors, so every tap is one v_pk_mul_f32 + one v_pk_add_f32 (two
// independent IEEE products / sums per instruction; nothing is fused or reassociated: each half is
// exactly the scalar chain).  The coefficient stays in an SGPR and is broadcast to both halves by op_sel.
typedef float f32x2 __attribute__((ext_vector_type(2)));

// acc-independent product of a 2-vector with ONE coefficient taken from the low (HI = false) or high half of
// an aligned SGPR pair, broadcast to both halves by op_sel -- written as asm because the compiler otherwise
// materialises every (c, c) splat as its own SGPR pair (98 SGPRs for fs = 7 -> spills).  Register-only VALU.
// One kernel row (7 taps) of the packed chain as a single asm statement: acc += w[lx] * c[lx], lx = 0..6, each tap
// v_pk_mul_f32 (coefficient = low or high half of an aligned SGPR pair, broadcast to both halves by op_sel)
// followed by v_pk_add_f32 -- un-fused, in order.  One statement per row keeps the compiler's per-statement
// boundary pad (one s_nop) at 1 per 14 instructions; register-only VALU, interlocked by hardware.
__device__ __forceinline__ void pk_row7(f32x2& acc, const f32x2 (&w)[7], f32x2 p01, f32x2 p23, f32x2 p45, f32x2 p6x) {
    f32x2 t;
    asm("v_pk_mul_f32 %1, %2, %9 op_sel_hi:[1,0]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %3, %9 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %4, %10 op_sel_hi:[1,0]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %5, %10 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %6, %11 op_sel_hi:[1,0]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %7, %11 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %8, %12 op_sel_hi:[1,0]\n\t"
        "v_pk_add_f32 %0, %0, %1"
        : "+v"(acc), "=&v"(t)
        : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "s"(p01), "s"(p23), "s"(p45), "s"(p6x));
}

template <int FS, int RG>
struct PeriodicPkCfg {
    static constexpr int kRowGroups = RG;
    static constexpr int kTileRows = FS * RG;
    static constexpr int kTileCols = 128;
    static constexpr int kSrcCols = kTileCols + FS;  // source columns staged per tile row
    // LDS row = pairs: pair k = (source column k, source column 64 + k), k = 0 .. 64+FS-1, so that a lane's two
    // windows (columns lane+lx and 64+lane+lx) arrive as one aligned ds_read_b64 -> one VGPR pair.
    static constexpr int kPairsPerRow = 64 + FS;
    static constexpr int kLdsPitch = 2 * kPairsPerRow + 2;  // floats; even (8-byte aligned rows)
    static constexpr int kLdsRows = kTileRows + FS;
};

template <typename T, int FS, int RG>
__global__ __launch_bounds__(256) void ewa_periodic_pk_kernel(const PeriodicArgs a, const PlaneIO io) {
    using Cfg = PeriodicPkCfg<FS, RG>;
    __shared__ __attribute__((aligned(16))) float tile[Cfg::kLdsRows * Cfg::kLdsPitch];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int tile_x, tile_y;
    swizzled_tile(tile_x, tile_y);
    const int i0 = tile_x * Cfg::kTileCols;
    const int j0 = tile_y * Cfg::kTileRows;
    const size_t frame = blockIdx.z;
    {
        const int gx0 = a.min_sx + i0;
        const int gy0 = a.min_sy + j0;
        const char* sbase = static_cast<const char*>(io.src) + frame * io.src_frame_stride;
        for (int r = wave; r < Cfg::kLdsRows; r += 4) {
            int gy = gy0 + r;
            gy = gy < a.src_h ? gy : a.src_h - 1;
            const T* srow = reinterpret_cast<const T*>(sbase + static_cast<size_t>(gy) * io.src_pitch);
            float* trow = tile + r * Cfg::kLdsPitch;
#pragma unroll
            for (int c = lane; c < Cfg::kSrcCols; c += 64) {
                int gx = gx0 + c;
                gx = gx < a.src_w ? gx : a.src_w - 1;
                const float v = to_float(srow[gx]);
                if (c < Cfg::kPairsPerRow) trow[2 * c] = v;        // first element of pair c
                if (c >= 64) trow[2 * (c - 64) + 1] = v;           // second element of pair c - 64
            }
        }
    }
    __syncthreads();

    const int nphase = a.px * a.py;
    for (int ph = wave; ph < nphase; ph += 4) {
        const int q = ph / a.px;
        const int p = ph - q * a.px;
        const JINC_CONSTANT float* cs =
            (const JINC_CONSTANT float*)(a.coeffs + static_cast<size_t>(a.set[ph]) * (FS * padded_row(FS)));
        // coefficient rows as aligned 64-bit SGPR pairs (rows are padded to a multiple of 4 floats)
        constexpr int kPairs = padded_row(FS) / 2;
        const JINC_CONSTANT f32x2* cs2 = (const JINC_CONSTANT f32x2*)cs;
        f32x2 cp[FS * kPairs];
#pragma unroll
        for (int k = 0; k < FS * kPairs; ++k) cp[k] = cs2[k];

        const f32x2* base = reinterpret_cast<const f32x2*>(tile + (a.start_y[q] - a.min_sy) * Cfg::kLdsPitch) +
                            (a.start_x[p] - a.min_sx) + lane;
        constexpr int kPitch2 = Cfg::kLdsPitch / 2;  // row pitch in pairs
        const int ia = i0 + lane;
        if (ia >= a.ni) continue;
        const bool b_ok = ia + 64 < a.ni;
        const BufferRsrc drsrc = make_rsrc(static_cast<char*>(io.dst) + frame * io.dst_frame_stride,
                                           static_cast<uint32_t>(io.dst_pitch) * a.dst_h);
        const uint32_t xoff_a = (a.ix0 + a.px * ia + p) * static_cast<uint32_t>(sizeof(T));
        const uint32_t xoff_b = xoff_a + 64u * a.px * static_cast<uint32_t>(sizeof(T));

        f32x2 win[FS][FS];
#pragma unroll
        for (int r = 0; r < FS - 1; ++r)
#pragma unroll
            for (int lx = 0; lx < FS; ++lx) win[r][lx] = base[r * kPitch2 + lx];

        for (int g = 0; g < Cfg::kRowGroups; ++g) {
            if (j0 + g * FS >= a.nj) break;
            const f32x2* gbase = base + (g * FS) * kPitch2;
#pragma unroll
            for (int u = 0; u < FS; ++u) {
#pragma unroll
                for (int lx = 0; lx < FS; ++lx) win[(u + FS - 1) % FS][lx] = gbase[(u + FS - 1) * kPitch2 + lx];
                f32x2 acc = {0.f, 0.f};
                static_assert(FS == 7, "packed variant is written for fs = 7");
#pragma unroll
                for (int ly = 0; ly < FS; ++ly)
                    pk_row7(acc, win[(u + ly) % FS], cp[ly * kPairs], cp[ly * kPairs + 1], cp[ly * kPairs + 2], cp[ly * kPairs + 3]);
                const int j = j0 + g * FS + u;
                if (j < a.nj) {
                    const uint32_t soff = static_cast<uint32_t>(a.iy0 + a.py * j + q) * io.dst_pitch;
                    store_sample_buf<T>(drsrc, xoff_a, soff, acc.x, io.peak);
                    if (b_ok) store_sample_buf<T>(drsrc, xoff_b, soff, acc.y, io.peak);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Periodic interior kernel, row-streamed form (any filter size, used for fs > 9)
// ------------------------------------------------------------------------------------------------
// Same phase-uniform idea as ewa_periodic_kernel (one phase per wave => coefficients in SGPRs), but
// sized for footprints whose fs x fs window does not fit the register file (fs = 17: 289 values):
//   * a lane owns K = 4 consecutive source-aligned columns and R = 4 consecutive period-rows
//     (16 independent accumulation chains), so one LDS row segment of fs+K-1 samples feeds K*fs taps;
//   * the loop runs ly-major: the fs coefficients of kernel row ly sit in SGPRs and are reused by
//     all 16 pixels; each pixel still sees its taps in (ly, lx) raster order, so every chain is the
//     reference's sequential chain;
//   * the LDS tile is stored as K column-planes (column c -> plane c % K, index c / K) so that the
//     64 lanes of a wave, which are K columns apart, read consecutive LDS words (no bank conflicts).
template <int FS, int KC = 4>
struct RowsCfg {
    static constexpr int K = KC;       // columns per lane (4, or 3 when that wastes fewer overhanging columns)
    static constexpr int R = 4;        // rows per chunk (accumulators per lane = R * K)
    static constexpr int kChunks = 4;  // chunks per tile
    static constexpr int kTileRows = R * kChunks;
    static constexpr int kTileCols = 64 * K;
    static constexpr int kCols = kTileCols + FS;  // + (FS-1) halo + 1 phase spread
    static constexpr int kPlaneMin = 64 + (FS + K - 1) / K + 1;
    static constexpr int kPlane = ((kPlaneMin - 8 + 31) / 32) * 32 + 8;  // == 8 (mod 32): the K planes start on distinct banks
    static constexpr int kRows = kTileRows + FS;  // + (FS-1) halo + 1 phase spread
    // LDS layout [plane][row][index]: every ds_read of a lane stays within 255 dwords of one of K
    // per-plane base registers (ds_read2_b32 immediate range), so the inner loop has no address VALU.
    static constexpr int kPlaneStride = kRows * kPlane;  // kRows is odd -> plane bases fall on distinct banks
    static constexpr int kWaves = 8;
    static constexpr int kThreads = 64 * kWaves;
};

template <typename T, int FS, int KC, int OFF>
__device__ __forceinline__ void rows_item(const float* __restrict__ tile, unsigned base_off, const JINC_CONSTANT float* cs, BufferRsrc drsrc,
                                          int dst_pitch, float peak, int y0, int ystep, int rows_valid, unsigned x0,
                                          unsigned xstep, int cols_valid) {
    using Cfg = RowsCfg<FS, KC>;
    constexpr int K = Cfg::K, R = Cfg::R;
    float acc[R][K];
#pragma unroll
    for (int jj = 0; jj < R; ++jj)
#pragma unroll
        for (int k = 0; k < K; ++k) acc[jj][k] = 0.f;

    // One LDS word offset per plane, kept in its own VGPR (the empty asm stops the compiler from
    // re-deriving plane bases as "base + large constant" with a VALU add in front of every read).
    unsigned plane_off[K];
#pragma unroll
    for (int m = 0; m < K; ++m) {
        plane_off[m] = base_off + m * Cfg::kPlaneStride;
        asm volatile("" : "+v"(plane_off[m]));
    }

    for (int ly = 0; ly < FS; ++ly) {
        float c[FS];
#pragma unroll
        for (int lx = 0; lx < FS; ++lx) c[lx] = cs[ly * padded_row(FS) + lx];
#pragma unroll
        for (int jj = 0; jj < R; ++jj) {
            float seg[FS + K - 1];
#pragma unroll
            for (int u = 0; u < FS + K - 1; ++u)
                seg[u] = tile[plane_off[(u + OFF) % K] + (jj * Cfg::kPlane + (u + OFF) / K)];
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int lx = 0; lx < FS; ++lx) acc[jj][k] = acc[jj][k] + seg[k + lx] * c[lx];
        }
#pragma unroll
        for (int m = 0; m < K; ++m) plane_off[m] += Cfg::kPlane;  // next kernel row
    }
#pragma unroll
    for (int jj = 0; jj < R; ++jj) {
        if (jj < rows_valid) {  // wave-uniform
            const uint32_t soff = static_cast<uint32_t>(y0 + jj * ystep) * dst_pitch;
#pragma unroll
            for (int k = 0; k < K; ++k)
                if (k < cols_valid)
                    store_sample_buf<T>(drsrc, (x0 + k * xstep) * static_cast<uint32_t>(sizeof(T)), soff, acc[jj][k], peak);
        }
    }
}

template <typename T, int FS, int KC>
__global__ __launch_bounds__(512) void ewa_periodic_rows_kernel(const PeriodicArgs a, const PlaneIO io) {
    using Cfg = RowsCfg<FS, KC>;
    constexpr int K = Cfg::K, R = Cfg::R;
    __shared__ float tile[K * Cfg::kPlaneStride];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int tile_x, tile_y;
    swizzled_tile(tile_x, tile_y);
    const int i0 = tile_x * Cfg::kTileCols;
    const int j0 = tile_y * Cfg::kTileRows;
    const size_t frame = blockIdx.z;

    {
        const int gx0 = a.min_sx + i0;
        const int gy0 = a.min_sy + j0;
        const char* sbase = static_cast<const char*>(io.src) + frame * io.src_frame_stride;
        for (int r = wave; r < Cfg::kRows; r += Cfg::kWaves) {
            int gy = gy0 + r;
            gy = gy < a.src_h ? gy : a.src_h - 1;
            const T* srow = reinterpret_cast<const T*>(sbase + static_cast<size_t>(gy) * io.src_pitch);
#pragma unroll
            for (int c = lane; c < Cfg::kCols; c += 64) {
                int gx = gx0 + c;
                gx = gx < a.src_w ? gx : a.src_w - 1;
                tile[(c % K) * Cfg::kPlaneStride + r * Cfg::kPlane + c / K] = to_float(srow[gx]);
            }
        }
    }
    __syncthreads();

    const BufferRsrc dframe = make_rsrc(static_cast<char*>(io.dst) + frame * io.dst_frame_stride,
                                        static_cast<uint32_t>(io.dst_pitch) * a.dst_h);
    const int nphase = a.px * a.py;
    const int nitems = nphase * Cfg::kChunks;
    const int cols_valid = a.ni - (i0 + K * lane);  // per lane: how many of its K columns exist
    if (cols_valid <= 0) return;
    for (int item = wave; item < nitems; item += Cfg::kWaves) {
        const int ch = item / nphase;
        const int ph = item - ch * nphase;
        const int q = ph / a.px;
        const int p = ph - q * a.px;
        const int j = j0 + ch * R;  // first period-row of the chunk
        const int rows_valid = a.nj - j;
        if (rows_valid <= 0) continue;
        const JINC_CONSTANT float* cs =
            (const JINC_CONSTANT float*)(a.coeffs + static_cast<size_t>(a.set[ph]) * (FS * padded_row(FS)));
        const unsigned base = ((a.start_y[q] - a.min_sy) + ch * R) * Cfg::kPlane + lane;
        const int y0 = a.iy0 + a.py * j + q;
        const unsigned x0 = a.ix0 + a.px * (i0 + K * lane) + p;
        if (a.start_x[p] - a.min_sx)
            rows_item<T, FS, KC, 1>(tile, base, cs, dframe, io.dst_pitch, io.peak, y0, a.py, rows_valid, x0, a.px, cols_valid);
        else
            rows_item<T, FS, KC, 0>(tile, base, cs, dframe, io.dst_pitch, io.peak, y0, a.py, rows_valid, x0, a.px, cols_valid);
    }
}

template <typename T, int FS>
int launch_gather_t(const GatherArgs& ga, int total_blocks, hipStream_t stream) {
    dim3 grid(total_blocks, ga.io.nframes, 1), block(256, 1, 1);
    hipLaunchKernelGGL((ewa_gather_kernel<T, FS>), grid, block, 0, stream, ga);
    return static_cast<int>(hipGetLastError());
}

template <typename T>
int launch_gather_fs(const GatherArgs& ga, int total_blocks, hipStream_t stream) {
    switch (ga.plan.fs) {
        case 3: return launch_gather_t<T, 3>(ga, total_blocks, stream);
        case 4: return launch_gather_t<T, 4>(ga, total_blocks, stream);
        case 5: return launch_gather_t<T, 5>(ga, total_blocks, stream);
        case 6: return launch_gather_t<T, 6>(ga, total_blocks, stream);
        case 7: return launch_gather_t<T, 7>(ga, total_blocks, stream);
        case 8: return launch_gather_t<T, 8>(ga, total_blocks, stream);
        case 9: return launch_gather_t<T, 9>(ga, total_blocks, stream);
        case 10: return launch_gather_t<T, 10>(ga, total_blocks, stream);
        case 11: return launch_gather_t<T, 11>(ga, total_blocks, stream);
        case 12: return launch_gather_t<T, 12>(ga, total_blocks, stream);
        case 13: return launch_gather_t<T, 13>(ga, total_blocks, stream);
        case 14: return launch_gather_t<T, 14>(ga, total_blocks, stream);
        case 15: return launch_gather_t<T, 15>(ga, total_blocks, stream);
        case 16: return launch_gather_t<T, 16>(ga, total_blocks, stream);
        case 17: return launch_gather_t<T, 17>(ga, total_blocks, stream);
        default: return launch_gather_t<T, 0>(ga, total_blocks, stream);
    }
}

template <typename T, int FS, int RG = PeriodicCfg<FS>::kRowGroups>
int launch_periodic_t(const PeriodicArgs& pa, const PlaneIO& io, hipStream_t stream) {
    using Cfg = PeriodicCfg<FS, RG>;
    dim3 grid((pa.ni + kTileCols - 1) / kTileCols, (pa.nj + Cfg::kTileRows - 1) / Cfg::kTileRows, io.nframes);
    hipLaunchKernelGGL((ewa_periodic_kernel<T, FS, RG>), grid, dim3(256, 1, 1), 0, stream, pa, io);
    return static_cast<int>(hipGetLastError());
}

template <typename T, int FS, int RG>
int launch_periodic_pk_t(const PeriodicArgs& pa, const PlaneIO& io, hipStream_t stream) {
    using Cfg = PeriodicPkCfg<FS, RG>;
    dim3 grid((pa.ni + Cfg::kTileCols - 1) / Cfg::kTileCols, (pa.nj + Cfg::kTileRows - 1) / Cfg::kTileRows, io.nframes);
    hipLaunchKernelGGL((ewa_periodic_pk_kernel<T, FS, RG>), grid, dim3(256, 1, 1), 0, stream, pa, io);
    return static_cast<int>(hipGetLastError());
}

template <typename T, int FS, int KC>
int launch_rows_k(const PeriodicArgs& pa, const PlaneIO& io, hipStream_t stream) {
    using Cfg = RowsCfg<FS, KC>;
    dim3 grid((pa.ni + Cfg::kTileCols - 1) / Cfg::kTileCols, (pa.nj + Cfg::kTileRows - 1) / Cfg::kTileRows, io.nframes);
    hipLaunchKernelGGL((ewa_periodic_rows_kernel<T, FS, KC>), grid, dim3(Cfg::kThreads, 1, 1), 0, stream, pa, io);
    return static_cast<int>(hipGetLastError());
}

// Tile width 256 (K = 4) or 192 (K = 3) columns: whichever leaves fewer overhanging (idle) lanes in the last tile
// column; ties go to K = 4 (fewer LDS reads per tap).
template <typename T, int FS>
int launch_rows_t(const PeriodicArgs& pa, const PlaneIO& io, hipStream_t stream) {
    const long long cover4 = (pa.ni + 255) / 256 * 256LL, cover3 = (pa.ni + 191) / 192 * 192LL;
    if (cover3 * 100 < cover4 * 97) return launch_rows_k<T, FS, 3>(pa, io, stream);
    return launch_rows_k<T, FS, 4>(pa, io, stream);
}

template <typename T>
int launch_periodic_fs(const PeriodicArgs& pa, int fs, const PlaneIO& io, hipStream_t stream, int variant) {
    if (variant == 3 && fs == 7) return launch_periodic_pk_t<T, 7, 4>(pa, io, stream);
    if (variant == 4 && fs == 7) return launch_periodic_pk_t<T, 7, 8>(pa, io, stream);
    if (variant == 2 && fs == 7) return launch_periodic_t<T, 7, 4>(pa, io, stream);
    if (variant == 2 && fs == 9) return launch_periodic_t<T, 9, 3>(pa, io, stream);
    if (variant == 1) {
        if (fs == 7) return launch_rows_t<T, 7>(pa, io, stream);
        if (fs == 9) return launch_rows_t<T, 9>(pa, io, stream);
    }
    switch (fs) {
        case 3: return launch_rows_t<T, 3>(pa, io, stream);
        case 5: return launch_rows_t<T, 5>(pa, io, stream);
        case 7: return launch_periodic_t<T, 7>(pa, io, stream);
        case 9: return launch_periodic_t<T, 9>(pa, io, stream);
        case 11: return launch_rows_t<T, 11>(pa, io, stream);
        case 13: return launch_rows_t<T, 13>(pa, io, stream);
        case 15: return launch_rows_t<T, 15>(pa, io, stream);
        case 17: return launch_rows_t<T, 17>(pa, io, stream);
        default: return static_cast<int>(hipErrorInvalidValue);
    }
}

}  // namespace

namespace {
// Runs exactly the conversion + store code of the resampling kernels on caller-supplied sums.
template <typename T>
__global__ void convert_kernel(const float* in, T* out, int n, float peak) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const BufferRsrc rsrc = make_rsrc(out, static_cast<uint32_t>(n) * sizeof(T));
    if (i & 1)
        store_sample_buf<T>(rsrc, static_cast<uint32_t>(i) * sizeof(T), 0u, in[i], peak);  // periodic kernels' path
    else
        store_sample<T>(out + i, in[i], peak);                                             // gather kernel's path
}
}  // namespace

int launch_debug_convert(const float* in, void* out, int n, int sample_bytes, float peak, void* stream) {
    if (n <= 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((n + 255) / 256), block(256);
    switch (sample_bytes) {
        case 1: hipLaunchKernelGGL(convert_kernel<uint8_t>, grid, block, 0, s, in, static_cast<uint8_t*>(out), n, peak); break;
        case 2: hipLaunchKernelGGL(convert_kernel<uint16_t>, grid, block, 0, s, in, static_cast<uint16_t*>(out), n, peak); break;
        default: hipLaunchKernelGGL(convert_kernel<float>, grid, block, 0, s, in, static_cast<float*>(out), n, peak); break;
    }
    return static_cast<int>(hipGetLastError());
}

int launch_gather(const DevicePlan& plan, const PlaneIO& io, const RectList& rects, void* stream) {
    GatherArgs ga;
    ga.plan = plan;
    ga.io = io;
    ga.rects = rects;
    int total = 0;
    for (int r = 0; r < 4; ++r) {
        ga.block_begin[r] = total;
        ga.blocks_a[r] = 1;
        ga.lane_axis[r] = 0;
        ga.stride[r] = 1;
        ga.lines[r] = 4;
        if (r < rects.n && rects.w[r] > 0 && rects.h[r] > 0) {
            // Narrow rectangles (border columns) put the lanes along y so that a wave is not mostly idle.
            const int axis = rects.w[r] < 32 && rects.h[r] > rects.w[r] ? 1 : 0;
            const int P = axis == 0 ? plan.gather_period_x : plan.gather_period_y;
            const int along = axis == 0 ? rects.w[r] : rects.h[r];
            const int across = axis == 0 ? rects.h[r] : rects.w[r];
            // More lines per block amortise the block's fixed cost (bounds, staging, barrier) and its halo;
            // thin rectangles (the border frame) get as many lines as they have.
            int nl = 4;
            while (nl < 32 && nl < across) nl *= 2;
            // ... but keep enough blocks in flight for the chip (256 CUs): thin border rectangles of a small
            // batch would otherwise collapse into a few hundred long-running blocks
            while (nl > 4 && static_cast<long long>((along + 64 * P - 1) / (64 * P)) * ((across + nl - 1) / nl) * io.nframes < 1024)
                nl /= 2;
            {   // ... as long as the block's source footprint still fits the LDS tile (down-scales widen it)
                const double rx = static_cast<double>(plan.src_w) / plan.dst_w, ry = static_cast<double>(plan.src_h) / plan.dst_h;
                const double r_along = axis == 0 ? rx : ry, r_across = axis == 0 ? ry : rx;
                const double w_along = 64.0 * P * r_along + plan.fs + 2;
                while (nl > 1 && w_along * (nl * r_across + plan.fs + 2) > 0.9 * kGatherLdsFloats) nl /= 2;
            }
            ga.lane_axis[r] = axis;
            ga.stride[r] = P;
            ga.lines[r] = nl;
            ga.blocks_a[r] = (along + 64 * P - 1) / (64 * P);
            total += ga.blocks_a[r] * ((across + nl - 1) / nl);
        }
    }
    ga.block_begin[4] = total;
    if (total == 0 || io.nframes <= 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (io.sample_bytes) {
        case 1: return launch_gather_fs<uint8_t>(ga, total, s);
        case 2: return launch_gather_fs<uint16_t>(ga, total, s);
        default: return launch_gather_fs<float>(ga, total, s);
    }
}

bool quasi_supported(int fs, int px, int py, int sx, int sy, int n_col_classes, int n_row_classes) {
    if (fs != 3 && fs != 5 && fs != 7 && fs != 9) return false;           // register window
    if (px < 1 || py < 1 || px > 8 || py > 8) return false;
    if (sx < 1 || sy < 1 || sx > 3 || sy > 3) return false;
    return static_cast<long long>(n_col_classes) * n_row_classes <= 4096;   // class-pair table held in LDS
}

bool quasi_configure(QuasiArgs& a, int fs, int spread_x, int spread_y) {
    const int nphase = a.px * a.py;
    a.nwaves = nphase % 4 == 0 ? 4 : (nphase % 3 == 0 ? 3 : (nphase % 2 == 0 ? 2 : (nphase >= 4 ? 4 : nphase)));
    a.lds_cols = a.sx * 64 + fs + spread_x;
    a.lds_pitch = a.lds_cols | 1;
    for (int rg = 8; rg >= 1; --rg) {
        const int rows = a.sy * rg * fs + fs + spread_y;
        const size_t bytes = sizeof(float) * static_cast<size_t>(rows) * a.lds_pitch +
                             sizeof(int) * (static_cast<size_t>(a.n_col_classes) * a.n_row_classes + a.py * rg * fs);
        if (bytes <= 40 * 1024) {
            a.rg = rg;
            a.lds_rows = rows;
            return true;
        }
    }
    return false;
}

namespace {
template <typename T, int FS, int SX, int SY>
int launch_quasi_t(const QuasiArgs& qa, const PlaneIO& io, hipStream_t stream) {
    const size_t lds = sizeof(float) * static_cast<size_t>(qa.lds_rows) * qa.lds_pitch +
                       sizeof(int) * (static_cast<size_t>(qa.n_col_classes) * qa.n_row_classes + qa.py * qa.rg * FS);
    const int tile_rows = qa.rg * FS;
    dim3 grid((qa.ni + 63) / 64, (qa.nj + tile_rows - 1) / tile_rows, io.nframes);
    hipLaunchKernelGGL((ewa_quasi_kernel<T, FS, SX, SY>), grid, dim3(64 * qa.nwaves, 1, 1), lds, stream, qa, io);
    return static_cast<int>(hipGetLastError());
}
template <typename T, int FS, int SX>
int launch_quasi_sy(const QuasiArgs& qa, const PlaneIO& io, hipStream_t stream) {
    switch (qa.sy) {
        case 1: return launch_quasi_t<T, FS, SX, 1>(qa, io, stream);
        case 2: return launch_quasi_t<T, FS, SX, 2>(qa, io, stream);
        default: return launch_quasi_t<T, FS, SX, 3>(qa, io, stream);
    }
}
template <typename T, int FS>
int launch_quasi_sx(const QuasiArgs& qa, const PlaneIO& io, hipStream_t stream) {
    switch (qa.sx) {
        case 1: return launch_quasi_sy<T, FS, 1>(qa, io, stream);
        case 2: return launch_quasi_sy<T, FS, 2>(qa, io, stream);
        default: return launch_quasi_sy<T, FS, 3>(qa, io, stream);
    }
}
template <typename T>
int launch_quasi_fs(const QuasiArgs& qa, int fs, const PlaneIO& io, hipStream_t stream) {
    switch (fs) {
        case 3: return launch_quasi_sx<T, 3>(qa, io, stream);
        case 5: return launch_quasi_sx<T, 5>(qa, io, stream);
        case 7: return launch_quasi_sx<T, 7>(qa, io, stream);
        case 9: return launch_quasi_sx<T, 9>(qa, io, stream);
        default: return static_cast<int>(hipErrorInvalidValue);
    }
}
}  // namespace

int launch_quasi(const QuasiArgs& args, int fs, const PlaneIO& io, void* stream) {
    if (args.ni <= 0 || args.nj <= 0 || io.nframes <= 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (io.sample_bytes) {
        case 1: return launch_quasi_fs<uint8_t>(args, fs, io, s);
        case 2: return launch_quasi_fs<uint16_t>(args, fs, io, s);
        default: return launch_quasi_fs<float>(args, fs, io, s);
    }
}

bool periodic_supported(int fs, int px, int py, int sx, int sy) {
    if (sx != 1 || sy != 1) return false;
    if (px < 1 || py < 1 || px > 8 || py > 8) return false;
    return fs >= 3 && fs <= 17 && (fs & 1);  // taps 1..8 at >= 1x scale
}

int launch_periodic(const PeriodicArgs& args, int fs, const PlaneIO& io, void* stream, int variant) {
    if (args.ni <= 0 || args.nj <= 0 || io.nframes <= 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (io.sample_bytes) {
        case 1: return launch_periodic_fs<uint8_t>(args, fs, io, s, variant);
        case 2: return launch_periodic_fs<uint16_t>(args, fs, io, s, variant);
        default: return launch_periodic_fs<float>(args, fs, io, s, variant);
    }
}

}  // namespace jinc
