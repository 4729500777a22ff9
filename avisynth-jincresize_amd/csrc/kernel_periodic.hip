// kernel_periodic.hip -- interior kernels of exactly phase-periodic plans (integer up-scales):
//   ewa_periodic_kernel       fs 7 / 9: one phase per wave, coefficients in SGPRs, source tile in LDS as fp32,
//                             fs x fs register window sliding down the tile (compile-time rotation)
//   ewa_periodic_pk_kernel    the same with two columns per lane on v_pk_mul_f32 / v_pk_add_f32 (A/B variant)
//   ewa_periodic_rows_kernel  any odd fs <= 17: 4 x 4 chains per lane, ly-major loop, column-plane LDS layout
// See device_common.hpp for the parity rules.
#include "device_common.hpp"
#include "knobs.h"

#pragma clang fp contract(off)

namespace jinc {
namespace {

#include "kernel_periodic_common.inc"

// ------------------------------------------------------------------------------------------------
// Periodic interior kernel
// ------------------------------------------------------------------------------------------------
constexpr int kTileCols = 64;  // source-aligned columns per tile = lanes of a wave

// RG: row groups of FS rows per tile.  A/B on MI355X: fs 7 (C2) 8 groups = 4 groups (was +2.3 % before the staging loads
// were issued together), 16 groups 10 % slower; fs 9 (C4, 88 VGPRs = 5 waves/SIMD) 9 groups +3.9 % over 6 -- 27 KB of
// LDS per block still allows the 5 blocks per CU the registers allow.
template <int FS, int RG = (FS <= 7 ? 8 : 9)>
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
    if (skips_frame(a, frame)) return;  // float planes: the other launch's frame

    // The coefficients of the wave's first phase do not depend on the tile: request them (scalar loads) BEFORE the
    // staging loads and the barrier, so that their latency overlaps the staging instead of following it.
    const int nphase = a.px * a.py;
    float cf[FS * FS];
    if (wave < nphase) {
        const JINC_CONSTANT float* cs0 =
            (const JINC_CONSTANT float*)(a.coeffs + static_cast<size_t>(a.set[wave]) * (FS * padded_row(FS)));
#pragma unroll
        for (int k = 0; k < FS * FS; ++k) cf[k] = cs0[(k / FS) * padded_row(FS) + (k % FS)];
    }

    // ---- stage the source tile as fp32 (each source sample converted once) ----
    {
        const int gx0 = a.min_sx + i0;
        const int gy0 = a.min_sy + j0;
        const char* sbase = static_cast<const char*>(io.src) + frame * io.src_frame_stride;
        // Fully unrolled with all loads in front of the LDS writes: one memory latency per tile instead of one per
        // row (a rolled loop is load -> wait -> write per iteration, 16 round trips during which the block's four
        // waves compute nothing).
        constexpr int kRowsPerWave = (Cfg::kLdsRows + 3) / 4;
        constexpr int kColsPerLane = (Cfg::kLdsCols + 63) / 64;
        T staged[kRowsPerWave][kColsPerLane];
        NonFinite<T> nonfinite(a, frame);
#pragma unroll
        for (int i = 0; i < kRowsPerWave; ++i) {
            int gy = gy0 + wave + 4 * i;
            gy = gy < a.src_h ? gy : a.src_h - 1;
            const T* srow = reinterpret_cast<const T*>(sbase + static_cast<size_t>(gy) * io.src_pitch);
#pragma unroll
            for (int k = 0; k < kColsPerLane; ++k) {
                int gx = gx0 + lane + 64 * k;
                gx = gx < a.src_w ? gx : a.src_w - 1;
                staged[i][k] = srow[gx];
            }
        }
#pragma unroll
        for (int i = 0; i < kRowsPerWave; ++i) {
            const int r = wave + 4 * i;
#pragma unroll
            for (int k = 0; k < kColsPerLane; ++k) {
                const int c = lane + 64 * k;
                if (r < Cfg::kLdsRows && c < Cfg::kLdsCols) tile[r * Cfg::kLdsPitch + c] = nonfinite.take(staged[i][k]);
            }
        }
    }
    __syncthreads();

    for (int ph = wave; ph < nphase; ph += 4) {
        const int q = ph / a.px;
        const int p = ph - q * a.px;

        // wave-uniform coefficients -> SGPRs (the first phase's are already on their way)
        if (ph != wave) {
            const JINC_CONSTANT float* cs =
                (const JINC_CONSTANT float*)(a.coeffs + static_cast<size_t>(a.set[ph]) * (FS * padded_row(FS)));
#pragma unroll
            for (int k = 0; k < FS * FS; ++k) cf[k] = cs[(k / FS) * padded_row(FS) + (k % FS)];
        }

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
// Periodic interior kernel, packed-math form (experimental A/B variant)
// ------------------------------------------------------------------------------------------------
// Same algorithm as ewa_periodic_kernel, but a lane owns TWO source-aligned columns 64 apart and keeps
// both register windows as 2-vectors, so every tap is one v_pk_mul_f32 + one v_pk_add_f32 (two
// independent IEEE products / sums per instruction; nothing is fused or reassociated: each half is
// exactly the scalar chain).  The coefficient stays in an SGPR and is broadcast to both halves by op_sel.

// Two horizontally adjacent samples of a lane through the buffer resource: one 2-sample store.
template <typename T>
__device__ __forceinline__ void store_pair_buf(BufferRsrc rsrc, uint32_t voffset, uint32_t soffset, f32x2 r, float peak) {
    if constexpr (std::is_same_v<T, float>) {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, r), rsrc, voffset, soffset, 0);
    } else if constexpr (std::is_same_v<T, uint8_t>) {
        uint32_t w = __builtin_amdgcn_cvt_pk_u8_f32(r.x, 0u, 0u);
        w = __builtin_amdgcn_cvt_pk_u8_f32(r.y, 1u, w);
        __builtin_amdgcn_raw_buffer_store_b16(static_cast<uint16_t>(w), rsrc, voffset, soffset, 0);
    } else {
        __builtin_amdgcn_raw_buffer_store_b32(round_pair_u16(r.x, r.y, peak), rsrc, voffset, soffset, 0);
    }
}

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
    if (skips_frame(a, frame)) return;  // float planes: the other launch's frame
    {
        const int gx0 = a.min_sx + i0;
        const int gy0 = a.min_sy + j0;
        const char* sbase = static_cast<const char*>(io.src) + frame * io.src_frame_stride;
        // all loads in front of the LDS writes (see ewa_periodic_kernel)
        constexpr int kRowsPerWave = (Cfg::kLdsRows + 3) / 4;
        constexpr int kColsPerLane = (Cfg::kSrcCols + 63) / 64;
        T staged[kRowsPerWave][kColsPerLane];
        NonFinite<T> nonfinite(a, frame);
#pragma unroll
        for (int i = 0; i < kRowsPerWave; ++i) {
            int gy = gy0 + wave + 4 * i;
            gy = gy < a.src_h ? gy : a.src_h - 1;
            const T* srow = reinterpret_cast<const T*>(sbase + static_cast<size_t>(gy) * io.src_pitch);
#pragma unroll
            for (int k = 0; k < kColsPerLane; ++k) {
                int gx = gx0 + lane + 64 * k;
                gx = gx < a.src_w ? gx : a.src_w - 1;
                staged[i][k] = srow[gx];
            }
        }
#pragma unroll
        for (int i = 0; i < kRowsPerWave; ++i) {
            const int r = wave + 4 * i;
            float* trow = tile + r * Cfg::kLdsPitch;
#pragma unroll
            for (int k = 0; k < kColsPerLane; ++k) {
                const int c = lane + 64 * k;
                if (r < Cfg::kLdsRows && c < Cfg::kSrcCols) {
                    const float v = nonfinite.take(staged[i][k]);
                    if (c < Cfg::kPairsPerRow) trow[2 * c] = v;        // first element of pair c
                    if (c >= 64) trow[2 * (c - 64) + 1] = v;           // second element of pair c - 64
                }
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
// Periodic interior kernel, quad form: 2x up-scales whose two phases per axis share their window origin
// ------------------------------------------------------------------------------------------------
// For a 2x up-scale the output pixels (2i, 2i+1) x (2j, 2j+1) of a period read the SAME fs x fs source window (plan:
// start_x[0] == start_x[1], start_y[0] == start_y[1]) with four different coefficient sets.  A lane owns one period
// column and keeps ONE register window for all four chains; the chains of the two horizontally adjacent pixels travel
// as the two halves of a register pair, so every tap of both is one v_pk_mul_f32 (sample broadcast to both halves by
// op_sel, coefficient pair = (phase 0, phase 1) from an aligned SGPR pair) and one v_pk_add_f32 -- two exact IEEE
// products / sums per instruction, each half the reference's sequential chain bit for bit, nothing fused or reassociated.
// Against ewa_periodic_kernel: half the VALU instructions per output sample on the faster packed pair (70 against 58
// Tops/s in the probe), a quarter of the LDS reads and stores, the same 8 waves per SIMD (one window per lane) -- but the
// 4 x fs x fs coefficients no longer fit the SGPR file: the pairs of one kernel row (2 x 16 dwords) are re-read from the
// scalar cache per kernel row and output row pair (always hits: 4 sets).
// Coefficient layout (host: attach_quad): quad[ly][q][8 pairs][p], the pair lx = (set(p=0,q), set(p=1,q))[ly][lx].
template <int FS>
struct QuadTaps;  // packed taps of one kernel row, window row held as pairs of a flat register array

// 7 taps starting at an EVEN flat index: pairs w0..w3 hold taps (0,1) (2,3) (4,5) (6,-)
__device__ __forceinline__ void quad_row7_even(f32x2& acc, f32x2 w0, f32x2 w1, f32x2 w2, f32x2 w3, f32x2 c0, f32x2 c1, f32x2 c2, f32x2 c3,
                                               f32x2 c4, f32x2 c5, f32x2 c6) {
    f32x2 t;
    asm("v_pk_mul_f32 %1, %2, %6 op_sel_hi:[0,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %2, %7 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %3, %8 op_sel_hi:[0,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %3, %9 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %4, %10 op_sel_hi:[0,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %4, %11 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %5, %12 op_sel_hi:[0,1]\n\t"
        "v_pk_add_f32 %0, %0, %1"
        : "+v"(acc), "=&v"(t)
        : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "s"(c0), "s"(c1), "s"(c2), "s"(c3), "s"(c4), "s"(c5), "s"(c6));
}
// 7 taps starting at an ODD flat index: pairs w0..w3 hold taps (-,0) (1,2) (3,4) (5,6)
__device__ __forceinline__ void quad_row7_odd(f32x2& acc, f32x2 w0, f32x2 w1, f32x2 w2, f32x2 w3, f32x2 c0, f32x2 c1, f32x2 c2, f32x2 c3,
                                              f32x2 c4, f32x2 c5, f32x2 c6) {
    f32x2 t;
    asm("v_pk_mul_f32 %1, %2, %6 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %3, %7 op_sel_hi:[0,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %3, %8 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %4, %9 op_sel_hi:[0,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %4, %10 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %5, %11 op_sel_hi:[0,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %5, %12 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_add_f32 %0, %0, %1"
        : "+v"(acc), "=&v"(t)
        : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "s"(c0), "s"(c1), "s"(c2), "s"(c3), "s"(c4), "s"(c5), "s"(c6));
}

// fs 9: nine taps starting at an EVEN flat index: pairs w0..w4 hold taps (0,1) (2,3) (4,5) (6,7) (8,-)
__device__ __forceinline__ void quad_row9_even(f32x2& acc, f32x2 w0, f32x2 w1, f32x2 w2, f32x2 w3, f32x2 w4, f32x2 c0, f32x2 c1, f32x2 c2,
                                               f32x2 c3, f32x2 c4, f32x2 c5, f32x2 c6, f32x2 c7, f32x2 c8) {
    f32x2 t;
    asm("v_pk_mul_f32 %1, %2, %7 op_sel_hi:[0,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %2, %8 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %3, %9 op_sel_hi:[0,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %3, %10 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %4, %11 op_sel_hi:[0,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %4, %12 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %5, %13 op_sel_hi:[0,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %5, %14 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %6, %15 op_sel_hi:[0,1]\n\t"
        "v_pk_add_f32 %0, %0, %1"
        : "+v"(acc), "=&v"(t)
        : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(w4), "s"(c0), "s"(c1), "s"(c2), "s"(c3), "s"(c4), "s"(c5), "s"(c6), "s"(c7), "s"(c8));
}
// ... at an ODD flat index: pairs w0..w4 hold taps (-,0) (1,2) (3,4) (5,6) (7,8)
__device__ __forceinline__ void quad_row9_odd(f32x2& acc, f32x2 w0, f32x2 w1, f32x2 w2, f32x2 w3, f32x2 w4, f32x2 c0, f32x2 c1, f32x2 c2,
                                              f32x2 c3, f32x2 c4, f32x2 c5, f32x2 c6, f32x2 c7, f32x2 c8) {
    f32x2 t;
    asm("v_pk_mul_f32 %1, %2, %7 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %3, %8 op_sel_hi:[0,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %3, %9 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %4, %10 op_sel_hi:[0,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %4, %11 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %5, %12 op_sel_hi:[0,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %5, %13 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %6, %14 op_sel_hi:[0,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %6, %15 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_add_f32 %0, %0, %1"
        : "+v"(acc), "=&v"(t)
        : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(w4), "s"(c0), "s"(c1), "s"(c2), "s"(c3), "s"(c4), "s"(c5), "s"(c6), "s"(c7), "s"(c8));
}
template <int SLOT>
__device__ __forceinline__ void quad_row9(f32x2& acc, const f32x2 (&w)[41], const f32x2 (&c)[10]) {
    constexpr int K0 = 9 * SLOT, P = K0 / 2;
    if constexpr (K0 % 2 == 0)
        quad_row9_even(acc, w[P], w[P + 1], w[P + 2], w[P + 3], w[P + 4], c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8]);
    else
        quad_row9_odd(acc, w[P], w[P + 1], w[P + 2], w[P + 3], w[P + 4], c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8]);
}
template <int SLOT>
__device__ __forceinline__ void quad_load_row9(f32x2 (&w)[41], const float* p) {
#pragma unroll
    for (int lx = 0; lx < 9; ++lx) {
        constexpr int K0 = 9 * SLOT;
        const float v = p[lx];
        if ((K0 + lx) % 2 == 0) w[(K0 + lx) / 2].x = v;
        else w[(K0 + lx) / 2].y = v;
    }
}
// fs 9: the 2 x 10 coefficient pairs of a kernel row do not fit twice next to everything else (2 x 40 SGPRs), so the two
// phase rows q alternate through two sets of 20 SGPRs: the pairs of (ly, q = 1) are requested before the taps of (ly, q = 0)
// are issued, those of (ly + 1, q = 0) before the taps of (ly, q = 1).  Layout: quad[ly][q][10 pairs][p].
__device__ __forceinline__ void quad_fetch10(f32x2 (&c)[10], const JINC_CONSTANT f32x2* quad, int row) {
#pragma unroll
    for (int k = 0; k < 10; ++k) c[k] = quad[row * 10 + k];
}
__device__ __forceinline__ void quad_arrived10(f32x2 (&c)[10]) {
#pragma unroll
    for (int k = 0; k < 10; ++k) asm volatile("" : "+s"(c[k]));
}
template <int U>
__device__ __forceinline__ void quad_pixel9(f32x2& acc0, f32x2& acc1, const f32x2 (&w)[41], const JINC_CONSTANT f32x2* quad) {
    f32x2 ca[10], cb[10];
    quad_fetch10(ca, quad, 0);
#define JINC_QUAD9_STEP(LY)                                                   \
    quad_fetch10(cb, quad, 2 * LY + 1);                                        \
    __builtin_amdgcn_sched_barrier(0);                                         \
    quad_arrived10(ca);                                                        \
    quad_row9<(U + LY) % 9>(acc0, w, ca);                                      \
    __builtin_amdgcn_sched_barrier(0);                                         \
    if constexpr (LY < 8) quad_fetch10(ca, quad, 2 * LY + 2);                  \
    __builtin_amdgcn_sched_barrier(0);                                         \
    quad_arrived10(cb);                                                        \
    quad_row9<(U + LY) % 9>(acc1, w, cb);                                      \
    __builtin_amdgcn_sched_barrier(0);
    JINC_QUAD9_STEP(0) JINC_QUAD9_STEP(1) JINC_QUAD9_STEP(2) JINC_QUAD9_STEP(3) JINC_QUAD9_STEP(4) JINC_QUAD9_STEP(5)
    JINC_QUAD9_STEP(6) JINC_QUAD9_STEP(7) JINC_QUAD9_STEP(8)
#undef JINC_QUAD9_STEP
}

// Window slot `slot` (7 samples at flat indices 7 * slot ..) times the seven coefficient pairs c[0..6] onto acc.
template <int SLOT>
__device__ __forceinline__ void quad_row7(f32x2& acc, const f32x2 (&w)[25], const f32x2 (&c)[8]) {
    constexpr int K0 = 7 * SLOT, P = K0 / 2;
    if constexpr (K0 % 2 == 0)
        quad_row7_even(acc, w[P], w[P + 1], w[P + 2], w[P + 3], c[0], c[1], c[2], c[3], c[4], c[5], c[6]);
    else
        quad_row7_odd(acc, w[P], w[P + 1], w[P + 2], w[P + 3], c[0], c[1], c[2], c[3], c[4], c[5], c[6]);
}

template <int SLOT>
__device__ __forceinline__ void quad_load_row7(f32x2 (&w)[25], const float* p) {
#pragma unroll
    for (int lx = 0; lx < 7; ++lx) {
        constexpr int K0 = 7 * SLOT;
        const float v = p[lx];
        if ((K0 + lx) % 2 == 0) w[(K0 + lx) / 2].x = v;
        else w[(K0 + lx) / 2].y = v;
    }
}

// The 16 coefficient pairs of kernel row LY (q = 0: pairs 0..7, q = 1: pairs 8..15): two s_load_dwordx16.
__device__ __forceinline__ void quad_fetch(f32x2 (&c)[16], const JINC_CONSTANT f32x2* quad, int ly) {
#pragma unroll
    for (int k = 0; k < 16; ++k) c[k] = quad[ly * 16 + k];
}
__device__ __forceinline__ void quad_arrived(f32x2 (&c)[16]) {
#pragma unroll
    for (int k = 0; k < 16; ++k) asm volatile("" : "+s"(c[k]));
}

// One output row pair: window slots (U + ly) % 7 hold kernel rows ly = 0..6; acc0 = (q = 0: phases p = 0, 1), acc1 = (q = 1).
// The coefficient pairs of kernel row ly + 1 are requested before the taps of row ly are issued (two sets of 32 SGPRs
// taken alternately); `quad` must not be loop-invariant in the caller, or the compiler hoists all 224 loads out of the row
// loop and spills them (measured: 264 spilled SGPRs).
template <int U>
__device__ __forceinline__ void quad_pixel7(f32x2& acc0, f32x2& acc1, const f32x2 (&w)[25], const JINC_CONSTANT f32x2* quad) {
    f32x2 ca[16], cb[16];
    quad_fetch(ca, quad, 0);
#define JINC_QUAD_STEP(LY, CUR, NEXT)                                         \
    if constexpr (LY < 6) quad_fetch(NEXT, quad, LY + 1);                      \
    __builtin_amdgcn_sched_barrier(0);                                         \
    quad_arrived(CUR);                                                         \
    quad_row7<(U + LY) % 7>(acc0, w, reinterpret_cast<const f32x2(&)[8]>(CUR[0])); \
    quad_row7<(U + LY) % 7>(acc1, w, reinterpret_cast<const f32x2(&)[8]>(CUR[8])); \
    __builtin_amdgcn_sched_barrier(0);
    JINC_QUAD_STEP(0, ca, cb)
    JINC_QUAD_STEP(1, cb, ca)
    JINC_QUAD_STEP(2, ca, cb)
    JINC_QUAD_STEP(3, cb, ca)
    JINC_QUAD_STEP(4, ca, cb)
    JINC_QUAD_STEP(5, cb, ca)
    JINC_QUAD_STEP(6, ca, cb)
#undef JINC_QUAD_STEP
}

template <typename T, int RG>
__global__ __launch_bounds__(256, 6) void ewa_periodic_quad_kernel(const PeriodicArgs a, const PlaneIO io) {
    constexpr int FS = 7;
    using Cfg = PeriodicCfg<FS, RG>;
    static_assert(RG % 4 == 0, "the four waves of a workgroup take RG / 4 row groups each");
    __shared__ float tile[Cfg::kLdsRows * Cfg::kLdsPitch];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int tile_x, tile_y;
    swizzled_tile(tile_x, tile_y);
    const int i0 = tile_x * kTileCols;
    const int j0 = tile_y * Cfg::kTileRows;
    const size_t frame = blockIdx.z;
    if (skips_frame(a, frame)) return;  // float planes: the other launch's frame
    {   // stage the source tile as fp32, all loads in front of the LDS writes (see ewa_periodic_kernel)
        const int gx0 = a.min_sx + i0;
        const int gy0 = a.min_sy + j0;
        const char* sbase = static_cast<const char*>(io.src) + frame * io.src_frame_stride;
        constexpr int kRowsPerWave = (Cfg::kLdsRows + 3) / 4;
        constexpr int kColsPerLane = (Cfg::kLdsCols + 63) / 64;
        T staged[kRowsPerWave][kColsPerLane];
        NonFinite<T> nonfinite(a, frame);
#pragma unroll
        for (int i = 0; i < kRowsPerWave; ++i) {
            int gy = gy0 + wave + 4 * i;
            gy = gy < a.src_h ? gy : a.src_h - 1;
            const T* srow = reinterpret_cast<const T*>(sbase + static_cast<size_t>(gy) * io.src_pitch);
#pragma unroll
            for (int k = 0; k < kColsPerLane; ++k) {
                int gx = gx0 + lane + 64 * k;
                gx = gx < a.src_w ? gx : a.src_w - 1;
                staged[i][k] = srow[gx];
            }
        }
#pragma unroll
        for (int i = 0; i < kRowsPerWave; ++i) {
            const int r = wave + 4 * i;
#pragma unroll
            for (int k = 0; k < kColsPerLane; ++k) {
                const int c = lane + 64 * k;
                if (r < Cfg::kLdsRows && c < Cfg::kLdsCols) tile[r * Cfg::kLdsPitch + c] = nonfinite.take(staged[i][k]);
            }
        }
    }
    __syncthreads();
    if ((i0 + lane) >= a.ni) return;  // no barrier below

    const JINC_CONSTANT f32x2* quad = (const JINC_CONSTANT f32x2*)(a.quad);
    // both phases of an axis share the window origin (host: quad != nullptr only then)
    const float* base = tile + (a.start_y[0] - a.min_sy) * Cfg::kLdsPitch + (a.start_x[0] - a.min_sx) + lane;
    const BufferRsrc drsrc = make_rsrc(static_cast<char*>(io.dst) + frame * io.dst_frame_stride,
                                       static_cast<uint32_t>(io.dst_pitch) * a.dst_h);  // wave-uniform
    const uint32_t xoff = static_cast<uint32_t>(a.ix0 + 2 * (i0 + lane)) * static_cast<uint32_t>(sizeof(T));

    constexpr int kGroupsPerWave = RG / 4;
    const int g_first = wave * kGroupsPerWave;
    if (j0 + g_first * FS >= a.nj) return;  // wave-uniform: bottom tiles
    f32x2 win[25];
    {
        const float* wb = base + (g_first * FS) * Cfg::kLdsPitch;
        quad_load_row7<0>(win, wb + 0 * Cfg::kLdsPitch);
        quad_load_row7<1>(win, wb + 1 * Cfg::kLdsPitch);
        quad_load_row7<2>(win, wb + 2 * Cfg::kLdsPitch);
        quad_load_row7<3>(win, wb + 3 * Cfg::kLdsPitch);
        quad_load_row7<4>(win, wb + 4 * Cfg::kLdsPitch);
        quad_load_row7<5>(win, wb + 5 * Cfg::kLdsPitch);
    }
    for (int g = g_first; g < g_first + kGroupsPerWave; ++g) {
        if (j0 + g * FS >= a.nj) break;  // wave-uniform
        const float* gbase = base + (g * FS) * Cfg::kLdsPitch;
#define JINC_QUAD_ROW(U)                                                                                          \
    {                                                                                                             \
        quad_load_row7<(U + FS - 1) % FS>(win, gbase + (U + FS - 1) * Cfg::kLdsPitch);                             \
        f32x2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};                                                                \
        uint32_t zero;                                                                                             \
        asm volatile("s_mov_b32 %0, 0" : "=s"(zero)); /* opaque: keeps the coefficient loads inside the row loop */ \
        quad_pixel7<U>(acc0, acc1, win, quad + zero);                                                              \
        const int j = j0 + g * FS + U;                                                                             \
        if (j < a.nj) {                                                                                            \
            const uint32_t so = static_cast<uint32_t>(a.iy0 + 2 * j) * io.dst_pitch;                               \
            store_pair_buf<T>(drsrc, xoff, so, acc0, io.peak);                                                     \
            store_pair_buf<T>(drsrc, xoff, so + static_cast<uint32_t>(io.dst_pitch), acc1, io.peak);               \
        }                                                                                                          \
    }
        JINC_QUAD_ROW(0) JINC_QUAD_ROW(1) JINC_QUAD_ROW(2) JINC_QUAD_ROW(3) JINC_QUAD_ROW(4) JINC_QUAD_ROW(5) JINC_QUAD_ROW(6)
#undef JINC_QUAD_ROW
    }
}

// fs 9 (tap 4 at 2x: C4): the same kernel with a 9 x 9 window (41 register pairs: 5 waves per SIMD, as the window kernel of
// fs 9) and the coefficient pairs of the two phase rows taken alternately (quad_pixel9).
template <typename T, int RG>
__global__ __launch_bounds__(256, 5) void ewa_periodic_quad9_kernel(const PeriodicArgs a, const PlaneIO io) {
    constexpr int FS = 9;
    using Cfg = PeriodicCfg<FS, RG>;
    static_assert(RG % 4 == 0, "the four waves of a workgroup take RG / 4 row groups each");
    __shared__ float tile[Cfg::kLdsRows * Cfg::kLdsPitch];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int tile_x, tile_y;
    swizzled_tile(tile_x, tile_y);
    const int i0 = tile_x * kTileCols;
    const int j0 = tile_y * Cfg::kTileRows;
    const size_t frame = blockIdx.z;
    if (skips_frame(a, frame)) return;  // float planes: the other launch's frame
    {   // stage the source tile as fp32, all loads in front of the LDS writes (see ewa_periodic_kernel)
        const int gx0 = a.min_sx + i0;
        const int gy0 = a.min_sy + j0;
        const char* sbase = static_cast<const char*>(io.src) + frame * io.src_frame_stride;
        constexpr int kRowsPerWave = (Cfg::kLdsRows + 3) / 4;
        constexpr int kColsPerLane = (Cfg::kLdsCols + 63) / 64;
        T staged[kRowsPerWave][kColsPerLane];
        NonFinite<T> nonfinite(a, frame);
#pragma unroll
        for (int i = 0; i < kRowsPerWave; ++i) {
            int gy = gy0 + wave + 4 * i;
            gy = gy < a.src_h ? gy : a.src_h - 1;
            const T* srow = reinterpret_cast<const T*>(sbase + static_cast<size_t>(gy) * io.src_pitch);
#pragma unroll
            for (int k = 0; k < kColsPerLane; ++k) {
                int gx = gx0 + lane + 64 * k;
                gx = gx < a.src_w ? gx : a.src_w - 1;
                staged[i][k] = srow[gx];
            }
        }
#pragma unroll
        for (int i = 0; i < kRowsPerWave; ++i) {
            const int r = wave + 4 * i;
#pragma unroll
            for (int k = 0; k < kColsPerLane; ++k) {
                const int c = lane + 64 * k;
                if (r < Cfg::kLdsRows && c < Cfg::kLdsCols) tile[r * Cfg::kLdsPitch + c] = nonfinite.take(staged[i][k]);
            }
        }
    }
    __syncthreads();
    if ((i0 + lane) >= a.ni) return;  // no barrier below

    const JINC_CONSTANT f32x2* quad = (const JINC_CONSTANT f32x2*)(a.quad);
    // both phases of an axis share the window origin (host: quad != nullptr only then)
    const float* base = tile + (a.start_y[0] - a.min_sy) * Cfg::kLdsPitch + (a.start_x[0] - a.min_sx) + lane;
    const BufferRsrc drsrc = make_rsrc(static_cast<char*>(io.dst) + frame * io.dst_frame_stride,
                                       static_cast<uint32_t>(io.dst_pitch) * a.dst_h);  // wave-uniform
    const uint32_t xoff = static_cast<uint32_t>(a.ix0 + 2 * (i0 + lane)) * static_cast<uint32_t>(sizeof(T));

    constexpr int kGroupsPerWave = RG / 4;
    const int g_first = wave * kGroupsPerWave;
    if (j0 + g_first * FS >= a.nj) return;  // wave-uniform: bottom tiles
    f32x2 win[41];
    {
        const float* wb = base + (g_first * FS) * Cfg::kLdsPitch;
        quad_load_row9<0>(win, wb + 0 * Cfg::kLdsPitch);
        quad_load_row9<1>(win, wb + 1 * Cfg::kLdsPitch);
        quad_load_row9<2>(win, wb + 2 * Cfg::kLdsPitch);
        quad_load_row9<3>(win, wb + 3 * Cfg::kLdsPitch);
        quad_load_row9<4>(win, wb + 4 * Cfg::kLdsPitch);
        quad_load_row9<5>(win, wb + 5 * Cfg::kLdsPitch);
        quad_load_row9<6>(win, wb + 6 * Cfg::kLdsPitch);
        quad_load_row9<7>(win, wb + 7 * Cfg::kLdsPitch);
    }
    for (int g = g_first; g < g_first + kGroupsPerWave; ++g) {
        if (j0 + g * FS >= a.nj) break;  // wave-uniform
        const float* gbase = base + (g * FS) * Cfg::kLdsPitch;
#define JINC_QUAD_ROW(U)                                                                                          \
    {                                                                                                             \
        quad_load_row9<(U + FS - 1) % FS>(win, gbase + (U + FS - 1) * Cfg::kLdsPitch);                             \
        f32x2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};                                                                \
        uint32_t zero;                                                                                             \
        asm volatile("s_mov_b32 %0, 0" : "=s"(zero)); /* opaque: keeps the coefficient loads inside the row loop */ \
        quad_pixel9<U>(acc0, acc1, win, quad + zero);                                                              \
        const int j = j0 + g * FS + U;                                                                             \
        if (j < a.nj) {                                                                                            \
            const uint32_t so = static_cast<uint32_t>(a.iy0 + 2 * j) * io.dst_pitch;                               \
            store_pair_buf<T>(drsrc, xoff, so, acc0, io.peak);                                                     \
            store_pair_buf<T>(drsrc, xoff, so + static_cast<uint32_t>(io.dst_pitch), acc1, io.peak);               \
        }                                                                                                          \
    }
        JINC_QUAD_ROW(0) JINC_QUAD_ROW(1) JINC_QUAD_ROW(2) JINC_QUAD_ROW(3) JINC_QUAD_ROW(4) JINC_QUAD_ROW(5) JINC_QUAD_ROW(6) JINC_QUAD_ROW(7) JINC_QUAD_ROW(8)
#undef JINC_QUAD_ROW
    }
}

// ------------------------------------------------------------------------------------------------
// Periodic interior kernel, quad form on a trimmed 8 x 8 support (tap 4 at 2x: Jinc64Resize, C4)
// ------------------------------------------------------------------------------------------------
// The fs-7 quad form on eight taps per kernel row: every row starts on a register pair (no odd / even variants), the eight
// coefficient pairs of a kernel row and q are exactly one s_load_dwordx16.  Against the window kernel on the same support:
// half the VALU instructions, a quarter of the LDS reads, and -- what counts on float planes -- the two horizontally adjacent
// samples of a lane leave as ONE store, so a wave's row is one contiguous run instead of every other sample of two.
__device__ __forceinline__ void quad_row8(f32x2& acc, f32x2 w0, f32x2 w1, f32x2 w2, f32x2 w3, f32x2 c0, f32x2 c1, f32x2 c2, f32x2 c3, f32x2 c4,
                                          f32x2 c5, f32x2 c6, f32x2 c7) {
    f32x2 t;
    asm("v_pk_mul_f32 %1, %2, %6 op_sel_hi:[0,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %2, %7 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %3, %8 op_sel_hi:[0,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %3, %9 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %4, %10 op_sel_hi:[0,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %4, %11 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %5, %12 op_sel_hi:[0,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %5, %13 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_add_f32 %0, %0, %1"
        : "+v"(acc), "=&v"(t)
        : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "s"(c0), "s"(c1), "s"(c2), "s"(c3), "s"(c4), "s"(c5), "s"(c6), "s"(c7));
}

// Kernel rows of the 8 x 8 support without their first and last t taps (t = 1: taps 1 .. 6, t = 2: taps 2 .. 5): rows in which those
// taps carry zero coefficients for both phases p of a q -- the disc's chords near the box's top and bottom (compile-time
// pattern TR8 of the quad kernels below; generated, like the full rows, as one asm statement each).
__device__ __forceinline__ void quad_row8_t1(f32x2& acc, f32x2 w0, f32x2 w1, f32x2 w2, f32x2 w3, f32x2 c1, f32x2 c2, f32x2 c3, f32x2 c4, f32x2 c5, f32x2 c6) {
    f32x2 t;
    asm("v_pk_mul_f32 %1, %2, %6 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %3, %7 op_sel_hi:[0,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %3, %8 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %4, %9 op_sel_hi:[0,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %4, %10 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %5, %11 op_sel_hi:[0,1]\n\t"
        "v_pk_add_f32 %0, %0, %1"
        : "+v"(acc), "=&v"(t)
        : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "s"(c1), "s"(c2), "s"(c3), "s"(c4), "s"(c5), "s"(c6));
}
__device__ __forceinline__ void quad_row8_t2(f32x2& acc, f32x2 w1, f32x2 w2, f32x2 c2, f32x2 c3, f32x2 c4, f32x2 c5) {
    f32x2 t;
    asm("v_pk_mul_f32 %1, %2, %4 op_sel_hi:[0,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %2, %5 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %3, %6 op_sel_hi:[0,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %3, %7 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_add_f32 %0, %0, %1"
        : "+v"(acc), "=&v"(t)
        : "v"(w1), "v"(w2), "s"(c2), "s"(c3), "s"(c4), "s"(c5));
}
__device__ __forceinline__ void quad2_row8_t1(f32x2& acc_a, f32x2& acc_b, f32x2 w0, f32x2 w1, f32x2 w2, f32x2 w3, f32x2 c1, f32x2 c2, f32x2 c3, f32x2 c4, f32x2 c5, f32x2 c6) {
    f32x2 ta, tb;
    asm("v_pk_mul_f32 %2, %4, %8 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_mul_f32 %3, %5, %8 op_sel_hi:[0,1]\n\t"
        "v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3\n\t"
        "v_pk_mul_f32 %2, %5, %9 op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %3, %5, %9 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3\n\t"
        "v_pk_mul_f32 %2, %5, %10 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_mul_f32 %3, %6, %10 op_sel_hi:[0,1]\n\t"
        "v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3\n\t"
        "v_pk_mul_f32 %2, %6, %11 op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %3, %6, %11 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3\n\t"
        "v_pk_mul_f32 %2, %6, %12 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_mul_f32 %3, %7, %12 op_sel_hi:[0,1]\n\t"
        "v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3\n\t"
        "v_pk_mul_f32 %2, %7, %13 op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %3, %7, %13 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3"
        : "+v"(acc_a), "+v"(acc_b), "=&v"(ta), "=&v"(tb)
        : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "s"(c1), "s"(c2), "s"(c3), "s"(c4), "s"(c5), "s"(c6));
}
__device__ __forceinline__ void quad2_row8_t2(f32x2& acc_a, f32x2& acc_b, f32x2 w1, f32x2 w2, f32x2 w3, f32x2 c2, f32x2 c3, f32x2 c4, f32x2 c5) {
    f32x2 ta, tb;
    asm("v_pk_mul_f32 %2, %4, %7 op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %3, %4, %7 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3\n\t"
        "v_pk_mul_f32 %2, %4, %8 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_mul_f32 %3, %5, %8 op_sel_hi:[0,1]\n\t"
        "v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3\n\t"
        "v_pk_mul_f32 %2, %5, %9 op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %3, %5, %9 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3\n\t"
        "v_pk_mul_f32 %2, %5, %10 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_mul_f32 %3, %6, %10 op_sel_hi:[0,1]\n\t"
        "v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3"
        : "+v"(acc_a), "+v"(acc_b), "=&v"(ta), "=&v"(tb)
        : "v"(w1), "v"(w2), "v"(w3), "s"(c2), "s"(c3), "s"(c4), "s"(c5));
}

// TR8: compile-time pattern of such rows, two bits per (kernel row ly, q) at bit 2 * (2 * ly + q).  Instantiated for "none" and
// for the 2x up-scale with tap 4 (blur 1 and 0.98): q = 0 rows 0 / 6 / 7 leave out 1 / 1 / 2 taps per side, q = 1 rows 0 / 1 / 7
// leave out 2 / 1 / 1 -- 56 instead of 64 taps per sample.  The launcher takes the pattern when the plan's rows allow at least it.
constexpr uint32_t kQuad8TrimTap4 = kQuad8TrimTap4Value;
constexpr int quad8_trim_of(uint32_t tr8, int ly, int q) { return static_cast<int>((tr8 >> (2 * (2 * ly + q))) & 3u); }

template <int SLOT>
__device__ __forceinline__ void quad_load_row8(f32x2 (&w)[32], const float* p) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        w[4 * SLOT + m].x = p[2 * m];
        w[4 * SLOT + m].y = p[2 * m + 1];
    }
}

template <int U, uint32_t TR8>
__device__ __forceinline__ void quad_pixel8(f32x2& acc0, f32x2& acc1, const f32x2 (&w)[32], const JINC_CONSTANT f32x2* quad) {
    f32x2 ca[16], cb[16];
    quad_fetch(ca, quad, 0);
#define JINC_QUAD8_STEP(LY, CUR, NEXT)                                                                                              \
    if constexpr (LY < 7) quad_fetch(NEXT, quad, LY + 1);                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                                               \
    quad_arrived(CUR);                                                                                                               \
    {                                                                                                                                \
        constexpr int S = 4 * ((U + LY) % 8);                                                                                        \
        if constexpr (quad8_trim_of(TR8, LY, 0) == 2)                                                                                \
            quad_row8_t2(acc0, w[S + 1], w[S + 2], CUR[2], CUR[3], CUR[4], CUR[5]);                                                  \
        else if constexpr (quad8_trim_of(TR8, LY, 0) == 1)                                                                           \
            quad_row8_t1(acc0, w[S], w[S + 1], w[S + 2], w[S + 3], CUR[1], CUR[2], CUR[3], CUR[4], CUR[5], CUR[6]);                  \
        else                                                                                                                         \
            quad_row8(acc0, w[S], w[S + 1], w[S + 2], w[S + 3], CUR[0], CUR[1], CUR[2], CUR[3], CUR[4], CUR[5], CUR[6], CUR[7]);     \
        if constexpr (quad8_trim_of(TR8, LY, 1) == 2)                                                                                \
            quad_row8_t2(acc1, w[S + 1], w[S + 2], CUR[10], CUR[11], CUR[12], CUR[13]);                                              \
        else if constexpr (quad8_trim_of(TR8, LY, 1) == 1)                                                                           \
            quad_row8_t1(acc1, w[S], w[S + 1], w[S + 2], w[S + 3], CUR[9], CUR[10], CUR[11], CUR[12], CUR[13], CUR[14]);             \
        else                                                                                                                         \
            quad_row8(acc1, w[S], w[S + 1], w[S + 2], w[S + 3], CUR[8], CUR[9], CUR[10], CUR[11], CUR[12], CUR[13], CUR[14], CUR[15]); \
    }                                                                                                                                \
    __builtin_amdgcn_sched_barrier(0);
    JINC_QUAD8_STEP(0, ca, cb)
    JINC_QUAD8_STEP(1, cb, ca)
    JINC_QUAD8_STEP(2, ca, cb)
    JINC_QUAD8_STEP(3, cb, ca)
    JINC_QUAD8_STEP(4, ca, cb)
    JINC_QUAD8_STEP(5, cb, ca)
    JINC_QUAD8_STEP(6, ca, cb)
    JINC_QUAD8_STEP(7, cb, ca)
#undef JINC_QUAD8_STEP
}

template <typename T, int RG, uint32_t TR8>
__global__ __launch_bounds__(256, 6) void ewa_periodic_quad8_kernel(const PeriodicArgs a, const PlaneIO io) {
    constexpr int FS = 8;
    using Cfg = PeriodicCfg<FS, RG>;
    static_assert(RG % 4 == 0, "the four waves of a workgroup take RG / 4 row groups each");
    __shared__ float tile[Cfg::kLdsRows * Cfg::kLdsPitch];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int tile_x, tile_y;
    swizzled_tile(tile_x, tile_y);
    const int i0 = tile_x * kTileCols;
    const int j0 = tile_y * Cfg::kTileRows;
    const size_t frame = blockIdx.z;
    if (skips_frame(a, frame)) return;  // float planes: the other launch's frame
    {   // stage the source tile as fp32, all loads in front of the LDS writes (see ewa_periodic_kernel)
        const int gx0 = a.min_sx + i0;
        const int gy0 = a.min_sy + j0;
        const char* sbase = static_cast<const char*>(io.src) + frame * io.src_frame_stride;
        constexpr int kRowsPerWave = (Cfg::kLdsRows + 3) / 4;
        constexpr int kColsPerLane = (Cfg::kLdsCols + 63) / 64;
        T staged[kRowsPerWave][kColsPerLane];
        NonFinite<T> nonfinite(a, frame);
#pragma unroll
        for (int i = 0; i < kRowsPerWave; ++i) {
            int gy = gy0 + wave + 4 * i;
            gy = gy < a.src_h ? gy : a.src_h - 1;
            const T* srow = reinterpret_cast<const T*>(sbase + static_cast<size_t>(gy) * io.src_pitch);
#pragma unroll
            for (int k = 0; k < kColsPerLane; ++k) {
                int gx = gx0 + lane + 64 * k;
                gx = gx < a.src_w ? gx : a.src_w - 1;
                staged[i][k] = srow[gx];
            }
        }
#pragma unroll
        for (int i = 0; i < kRowsPerWave; ++i) {
            const int r = wave + 4 * i;
#pragma unroll
            for (int k = 0; k < kColsPerLane; ++k) {
                const int c = lane + 64 * k;
                if (r < Cfg::kLdsRows && c < Cfg::kLdsCols) tile[r * Cfg::kLdsPitch + c] = nonfinite.take(staged[i][k]);
            }
        }
    }
    __syncthreads();
    if ((i0 + lane) >= a.ni) return;  // no barrier below

    const JINC_CONSTANT f32x2* quad = (const JINC_CONSTANT f32x2*)(a.quad);
    const float* base = tile + (a.start_y[0] - a.min_sy) * Cfg::kLdsPitch + (a.start_x[0] - a.min_sx) + lane;
    const BufferRsrc drsrc = make_rsrc(static_cast<char*>(io.dst) + frame * io.dst_frame_stride,
                                       static_cast<uint32_t>(io.dst_pitch) * a.dst_h);  // wave-uniform
    const uint32_t xoff = static_cast<uint32_t>(a.ix0 + 2 * (i0 + lane)) * static_cast<uint32_t>(sizeof(T));

    constexpr int kGroupsPerWave = RG / 4;
    const int g_first = wave * kGroupsPerWave;
    if (j0 + g_first * FS >= a.nj) return;  // wave-uniform: bottom tiles
    f32x2 win[32];
    {
        const float* wb = base + (g_first * FS) * Cfg::kLdsPitch;
        quad_load_row8<0>(win, wb + 0 * Cfg::kLdsPitch);
        quad_load_row8<1>(win, wb + 1 * Cfg::kLdsPitch);
        quad_load_row8<2>(win, wb + 2 * Cfg::kLdsPitch);
        quad_load_row8<3>(win, wb + 3 * Cfg::kLdsPitch);
        quad_load_row8<4>(win, wb + 4 * Cfg::kLdsPitch);
        quad_load_row8<5>(win, wb + 5 * Cfg::kLdsPitch);
        quad_load_row8<6>(win, wb + 6 * Cfg::kLdsPitch);
    }
    for (int g = g_first; g < g_first + kGroupsPerWave; ++g) {
        if (j0 + g * FS >= a.nj) break;  // wave-uniform
        const float* gbase = base + (g * FS) * Cfg::kLdsPitch;
#define JINC_QUAD8_ROW(U)                                                                                          \
    {                                                                                                              \
        quad_load_row8<(U + FS - 1) % FS>(win, gbase + (U + FS - 1) * Cfg::kLdsPitch);                              \
        f32x2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};                                                                 \
        uint32_t zero;                                                                                              \
        asm volatile("s_mov_b32 %0, 0" : "=s"(zero)); /* opaque: keeps the coefficient loads inside the row loop */ \
        quad_pixel8<U, TR8>(acc0, acc1, win, quad + zero);                                                          \
        const int j = j0 + g * FS + U;                                                                              \
        if (j < a.nj) {                                                                                             \
            const uint32_t so = static_cast<uint32_t>(a.iy0 + 2 * j) * io.dst_pitch;                                \
            store_pair_buf<T>(drsrc, xoff, so, acc0, io.peak);                                                      \
            store_pair_buf<T>(drsrc, xoff, so + static_cast<uint32_t>(io.dst_pitch), acc1, io.peak);                \
        }                                                                                                           \
    }
        JINC_QUAD8_ROW(0) JINC_QUAD8_ROW(1) JINC_QUAD8_ROW(2) JINC_QUAD8_ROW(3) JINC_QUAD8_ROW(4) JINC_QUAD8_ROW(5) JINC_QUAD8_ROW(6) JINC_QUAD8_ROW(7)
#undef JINC_QUAD8_ROW
    }
}

// ------------------------------------------------------------------------------------------------
// Periodic interior kernel, quad form on a trimmed 6 x 6 support, two periods per lane
// ------------------------------------------------------------------------------------------------
// Integer planes whose coefficient sets carry exact zeros along the window's edge run on the trimmed support (host:
// device_plan.cpp, trim_periodic): the 2x up-scale with tap 3 has filter size 7, but the first kernel row and column of all
// four phase sets are 0.0f (the EWA disc of radius 3.24 spans six samples at these phases), and a tap whose coefficient is
// zero adds +0 to a chain that is never -0 -- leaving it out is exact for finite samples, which integer samples are.
// 36 taps per sample instead of 49.
// The lane mapping changes with it: a lane owns TWO horizontally adjacent periods -- 4 x 2 output samples from ONE
// 6 x 7 register window (the periods' windows are one source column apart).  Against ewa_periodic_quad_kernel, per output
// sample: half the coefficient fetches (one SGPR pair feeds both periods: the scalar cache delivered two s_load_dwordx16
// per 784 issue cycles and wave there, now two per 1152), window rows as four aligned ds_read_b64 per eight samples
// instead of seven ds_read_b32 per four, four independent chains per lane interleaved two by two, and the four 8-bit
// samples of a row leave as one dword store.  Every chain still meets its taps in (ly, lx) order, multiply and add un-fused.
// Coefficient layout: as the fs-7 quad form, quad[ly][q][8 pairs][p] with pairs 6 and 7 unused.
template <int RG>
struct Quad2Cfg {
    static constexpr int FS = 6;
    static constexpr int kPeriodsPerLane = 2;
    static constexpr int kTileCols = 64 * kPeriodsPerLane;  // periods per tile row
    static constexpr int kTileRows = FS * RG;               // period-rows per tile
    static constexpr int kLdsCols = kTileCols + FS;         // lane 63 reads columns 126 .. 133
    static constexpr int kLdsPitch = 136;                   // even: every lane's row segment is 8-byte aligned
    static constexpr int kLdsRows = kTileRows + FS - 1;
};

// One kernel row of one q for both periods of a lane: window pairs w0..w3 = source columns 0..7 of the row (period A
// reads columns 0..5, period B columns 1..6), coefficient pairs c0..c5 = (p = 0, p = 1) of taps 0..5.
__device__ __forceinline__ void quad2_row6(f32x2& acc_a, f32x2& acc_b, f32x2 w0, f32x2 w1, f32x2 w2, f32x2 w3, f32x2 c0, f32x2 c1, f32x2 c2,
                                           f32x2 c3, f32x2 c4, f32x2 c5) {
    f32x2 ta, tb;
#define JINC_LO(T, W, C) "v_pk_mul_f32 " T ", " W ", " C " op_sel_hi:[0,1]\n\t"                 /* sample = low half of the pair */
#define JINC_HI(T, W, C) "v_pk_mul_f32 " T ", " W ", " C " op_sel:[1,0] op_sel_hi:[1,1]\n\t"   /* sample = high half */
#define JINC_ADD2 "v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3\n\t"
    asm(JINC_LO("%2", "%4", "%8") JINC_HI("%3", "%4", "%8") JINC_ADD2      // tap 0: A column 0, B column 1
        JINC_HI("%2", "%4", "%9") JINC_LO("%3", "%5", "%9") JINC_ADD2      // tap 1: A 1, B 2
        JINC_LO("%2", "%5", "%10") JINC_HI("%3", "%5", "%10") JINC_ADD2    // tap 2: A 2, B 3
        JINC_HI("%2", "%5", "%11") JINC_LO("%3", "%6", "%11") JINC_ADD2    // tap 3: A 3, B 4
        JINC_LO("%2", "%6", "%12") JINC_HI("%3", "%6", "%12") JINC_ADD2    // tap 4: A 4, B 5
        JINC_HI("%2", "%6", "%13") JINC_LO("%3", "%7", "%13")              // tap 5: A 5, B 6
        "v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3"
        : "+v"(acc_a), "+v"(acc_b), "=&v"(ta), "=&v"(tb)
        : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "s"(c0), "s"(c1), "s"(c2), "s"(c3), "s"(c4), "s"(c5));
#undef JINC_LO
#undef JINC_HI
#undef JINC_ADD2
}

// Seven taps per kernel row for both periods of a lane (the 6-row x 7-column support: chroma planes sited as MPEG-2 at 2x): period A
// reads columns 0 .. 6, period B columns 1 .. 7 of the same four register pairs.
__device__ __forceinline__ void quad2_row7(f32x2& acc_a, f32x2& acc_b, f32x2 w0, f32x2 w1, f32x2 w2, f32x2 w3, f32x2 c0, f32x2 c1, f32x2 c2, f32x2 c3, f32x2 c4, f32x2 c5, f32x2 c6) {
    f32x2 ta, tb;
    asm("v_pk_mul_f32 %2, %4, %8 op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %3, %4, %8 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3\n\t"
        "v_pk_mul_f32 %2, %4, %9 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_mul_f32 %3, %5, %9 op_sel_hi:[0,1]\n\t"
        "v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3\n\t"
        "v_pk_mul_f32 %2, %5, %10 op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %3, %5, %10 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3\n\t"
        "v_pk_mul_f32 %2, %5, %11 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_mul_f32 %3, %6, %11 op_sel_hi:[0,1]\n\t"
        "v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3\n\t"
        "v_pk_mul_f32 %2, %6, %12 op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %3, %6, %12 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3\n\t"
        "v_pk_mul_f32 %2, %6, %13 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_mul_f32 %3, %7, %13 op_sel_hi:[0,1]\n\t"
        "v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3\n\t"
        "v_pk_mul_f32 %2, %7, %14 op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %3, %7, %14 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3"
        : "+v"(acc_a), "+v"(acc_b), "=&v"(ta), "=&v"(tb)
        : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "s"(c0), "s"(c1), "s"(c2), "s"(c3), "s"(c4), "s"(c5), "s"(c6));
}

// The same kernel row where tap 0 and tap 5 carry zero coefficients for BOTH phases p of this q (the disc's chord in the box's
// first / last row: host, PeriodicArgs::quad_inner): taps 1 .. 4 only.
__device__ __forceinline__ void quad2_row6_inner(f32x2& acc_a, f32x2& acc_b, f32x2 w0, f32x2 w1, f32x2 w2, f32x2 c1, f32x2 c2, f32x2 c3, f32x2 c4) {
    f32x2 ta, tb;
#define JINC_LO(T, W, C) "v_pk_mul_f32 " T ", " W ", " C " op_sel_hi:[0,1]\n\t"
#define JINC_HI(T, W, C) "v_pk_mul_f32 " T ", " W ", " C " op_sel:[1,0] op_sel_hi:[1,1]\n\t"
#define JINC_ADD2 "v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3\n\t"
    asm(JINC_HI("%2", "%4", "%7") JINC_LO("%3", "%5", "%7") JINC_ADD2      // tap 1: A column 1, B column 2
        JINC_LO("%2", "%5", "%8") JINC_HI("%3", "%5", "%8") JINC_ADD2      // tap 2: A 2, B 3
        JINC_HI("%2", "%5", "%9") JINC_LO("%3", "%6", "%9") JINC_ADD2      // tap 3: A 3, B 4
        JINC_LO("%2", "%6", "%10") JINC_HI("%3", "%6", "%10")              // tap 4: A 4, B 5
        "v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3"
        : "+v"(acc_a), "+v"(acc_b), "=&v"(ta), "=&v"(tb)
        : "v"(w0), "v"(w1), "v"(w2), "s"(c1), "s"(c2), "s"(c3), "s"(c4));
#undef JINC_LO
#undef JINC_HI
#undef JINC_ADD2
}

// quad2_row7 on a span of its taps: the chord of a (kernel row, q) of the 6-row x 7-column support -- form 1 = taps 1 .. 6, 2 = taps
// 1 .. 5, 3 = taps 2 .. 5 (form 0: all seven, quad2_row7) -- for rows whose other taps carry zero coefficients for BOTH phases p.
// Chroma planes sited as MPEG-2 at 2x with tap 3: 36 of the 42 taps per sample (PeriodicArgs::kQuadSpan7Mpeg2).
#define JINC_Q7_EVEN(W, C) /* tap t even: period A = the low half of pair t / 2, B its high half */                         \
    "v_pk_mul_f32 %2, " W ", " C " op_sel_hi:[0,1]\n\tv_pk_mul_f32 %3, " W ", " C " op_sel:[1,0] op_sel_hi:[1,1]\n\t" \
    "v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3\n\t"
#define JINC_Q7_ODD(W, WN, C) /* tap t odd: A = the high half of pair t / 2, B the low half of the next pair */             \
    "v_pk_mul_f32 %2, " W ", " C " op_sel:[1,0] op_sel_hi:[1,1]\n\tv_pk_mul_f32 %3, " WN ", " C " op_sel_hi:[0,1]\n\t" \
    "v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3\n\t"
template <int FORM>
__device__ __forceinline__ void quad2_row7_span(f32x2& acc_a, f32x2& acc_b, f32x2 w0, f32x2 w1, f32x2 w2, f32x2 w3, f32x2 c0, f32x2 c1, f32x2 c2, f32x2 c3,
                                                f32x2 c4, f32x2 c5, f32x2 c6) {
    static_assert(FORM >= 0 && FORM <= 3, "span forms of the seven-tap row");
    f32x2 ta, tb;
    if constexpr (FORM == 0) {
        quad2_row7(acc_a, acc_b, w0, w1, w2, w3, c0, c1, c2, c3, c4, c5, c6);
    } else if constexpr (FORM == 1) {  // taps 1 .. 6
        asm(JINC_Q7_ODD("%4", "%5", "%8") JINC_Q7_EVEN("%5", "%9") JINC_Q7_ODD("%5", "%6", "%10") JINC_Q7_EVEN("%6", "%11")
            JINC_Q7_ODD("%6", "%7", "%12") JINC_Q7_EVEN("%7", "%13") ""
            : "+v"(acc_a), "+v"(acc_b), "=&v"(ta), "=&v"(tb)
            : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "s"(c1), "s"(c2), "s"(c3), "s"(c4), "s"(c5), "s"(c6));
    } else if constexpr (FORM == 2) {  // taps 1 .. 5
        asm(JINC_Q7_ODD("%4", "%5", "%8") JINC_Q7_EVEN("%5", "%9") JINC_Q7_ODD("%5", "%6", "%10") JINC_Q7_EVEN("%6", "%11")
            JINC_Q7_ODD("%6", "%7", "%12") ""
            : "+v"(acc_a), "+v"(acc_b), "=&v"(ta), "=&v"(tb)
            : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "s"(c1), "s"(c2), "s"(c3), "s"(c4), "s"(c5));
    } else {                           // taps 2 .. 5
        asm(JINC_Q7_EVEN("%4", "%7") JINC_Q7_ODD("%4", "%5", "%8") JINC_Q7_EVEN("%5", "%9") JINC_Q7_ODD("%5", "%6", "%10") ""
            : "+v"(acc_a), "+v"(acc_b), "=&v"(ta), "=&v"(tb)
            : "v"(w1), "v"(w2), "v"(w3), "s"(c2), "s"(c3), "s"(c4), "s"(c5));
    }
}
#undef JINC_Q7_EVEN
#undef JINC_Q7_ODD

// Window slot SLOT <- eight source columns of one tile row (four aligned ds_read_b64).
template <int SLOT>
__device__ __forceinline__ void quad2_load_row(f32x2 (&w)[24], const float* p) {
    const f32x2* p2 = reinterpret_cast<const f32x2*>(p);
#pragma unroll
    for (int m = 0; m < 4; ++m) w[4 * SLOT + m] = p2[m];
}

// One output row pair of both periods: acc[0] / acc[1] = period A / B at q = 0, acc[2] / acc[3] at q = 1.  Coefficient pairs of
// kernel row ly + 1 are requested before the taps of row ly are issued, as in quad_pixel7.
template <int U, uint32_t INNER, int NT>
__device__ __forceinline__ void quad2_pixel6(f32x2 (&acc)[4], const f32x2 (&w)[24], const JINC_CONSTANT f32x2* quad) {
    f32x2 ca[16], cb[16];
    quad_fetch(ca, quad, 0);
#define JINC_QUAD2_STEP(LY, CUR, NEXT)                                                                                        \
    if constexpr (LY < 5) quad_fetch(NEXT, quad, LY + 1);                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                                         \
    quad_arrived(CUR);                                                                                                         \
    {                                                                                                                          \
        constexpr int S = 4 * ((U + LY) % 6);                                                                                  \
        /* INNER: a compile-time mask (a run-time test here, however uniform, cost 8 % of the kernel: round4/quad_inner_ab.log) */ \
        if constexpr (NT == 7) { /* 6 rows x 7 columns: seven taps per kernel row */                                            \
            /* (NT == 7: INNER holds the span form of every (kernel row, q), two bits each) */                              \
            quad2_row7_span<(INNER >> (2 * (2 * LY))) & 3u>(acc[0], acc[1], w[S], w[S + 1], w[S + 2], w[S + 3], CUR[0], CUR[1], CUR[2], CUR[3], CUR[4], CUR[5], CUR[6]); \
            quad2_row7_span<(INNER >> (2 * (2 * LY + 1))) & 3u>(acc[2], acc[3], w[S], w[S + 1], w[S + 2], w[S + 3], CUR[8], CUR[9], CUR[10], CUR[11], CUR[12], CUR[13], CUR[14]); \
        } else {                                                                                                               \
        if constexpr ((INNER >> (2 * LY)) & 1u)                                                                                \
            quad2_row6_inner(acc[0], acc[1], w[S], w[S + 1], w[S + 2], CUR[1], CUR[2], CUR[3], CUR[4]);                        \
        else                                                                                                                   \
            quad2_row6(acc[0], acc[1], w[S], w[S + 1], w[S + 2], w[S + 3], CUR[0], CUR[1], CUR[2], CUR[3], CUR[4], CUR[5]);    \
        if constexpr ((INNER >> (2 * LY + 1)) & 1u)                                                                            \
            quad2_row6_inner(acc[2], acc[3], w[S], w[S + 1], w[S + 2], CUR[9], CUR[10], CUR[11], CUR[12]);                     \
        else                                                                                                                   \
            quad2_row6(acc[2], acc[3], w[S], w[S + 1], w[S + 2], w[S + 3], CUR[8], CUR[9], CUR[10], CUR[11], CUR[12], CUR[13]); \
        }                                                                                                                      \
    }                                                                                                                          \
    __builtin_amdgcn_sched_barrier(0);
    JINC_QUAD2_STEP(0, ca, cb)
    JINC_QUAD2_STEP(1, cb, ca)
    JINC_QUAD2_STEP(2, ca, cb)
    JINC_QUAD2_STEP(3, cb, ca)
    JINC_QUAD2_STEP(4, ca, cb)
    JINC_QUAD2_STEP(5, cb, ca)
#undef JINC_QUAD2_STEP
}

// The four (or, for the lane that holds the plane's last odd period, two) samples of one output row of a lane: one store.
template <typename T>
__device__ __forceinline__ void store_quad_buf(BufferRsrc rsrc, uint32_t voffset, uint32_t soffset, f32x2 a, f32x2 b, float peak, bool b_ok) {
    if constexpr (std::is_same_v<T, float>) {
        // two 8-byte stores: the samples start at a 4-byte boundary (odd interior origin), and a 16-byte store that is not
        // 16-byte aligned does not write its four dwords where they belong (measured: dwords 1 and 3 took the values of 0 and 2)
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, a), rsrc, voffset, soffset, 0);
        if (b_ok) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, b), rsrc, voffset + 8u, soffset, 0);
    } else if constexpr (std::is_same_v<T, uint8_t>) {
        uint32_t w = __builtin_amdgcn_cvt_pk_u8_f32(a.x, 0u, 0u);
        w = __builtin_amdgcn_cvt_pk_u8_f32(a.y, 1u, w);
        w = __builtin_amdgcn_cvt_pk_u8_f32(b.x, 2u, w);
        w = __builtin_amdgcn_cvt_pk_u8_f32(b.y, 3u, w);
        if (b_ok) __builtin_amdgcn_raw_buffer_store_b32(w, rsrc, voffset, soffset, 0);
        else __builtin_amdgcn_raw_buffer_store_b16(static_cast<uint16_t>(w), rsrc, voffset, soffset, 0);
    } else {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        const uint32_t lo = round_pair_u16(a.x, a.y, peak);
        if (b_ok) {
            const u32x2 v = {lo, round_pair_u16(b.x, b.y, peak)};
            __builtin_amdgcn_raw_buffer_store_b64(v, rsrc, voffset, soffset, 0);
        } else {
            __builtin_amdgcn_raw_buffer_store_b32(lo, rsrc, voffset, soffset, 0);
        }
    }
}

// The plane's border columns beside this tile's rows (PeriodicArgs::EdgeColumns; integer planes, first / last tile column only) for the
// two-periods-per-lane quad forms (Cfg: Quad2Cfg / Quad2x8Cfg; NR kernel rows = the interior's trimmed rows, NC = the plan's filter size).
// Lane = period-row of the tile, one NR-row x NC-column register window per lane for every column of the side (they share their
// window origin) and both row phases (they share theirs); wave w takes row phase w & 1 of one half of the side's columns, four
// columns at a time: their samples are adjacent in the lane's output row and leave as one store.  Every sample is the reference's
// chain: taps in (ly, lx) order, multiply and add un-fused (ref /root/reference/src/JincResize.cpp:570-579); the kernel rows left
// out carry zero coefficients (checked on the host).
template <typename T>
__device__ __forceinline__ bool edge_tile_of(const PeriodicArgs& a, int tile_x) {
    if constexpr (std::is_same_v<T, float>) return false;  // (float planes: the border kernels, whatever the samples)
    return a.edge.coeffs != nullptr && ((a.edge.n[0] > 0 && tile_x == a.edge.tile_x[0]) || (a.edge.n[1] > 0 && tile_x == a.edge.tile_x[1]));
}

// An edge window that starts one source column in front of the tile (the left border's, when the interior's support was trimmed by a
// column): that column goes into the word in front of each tile row -- the previous row's last spare word, or for row 0 the word in
// front of the tile (the kernels allocate two).  Call between the staging writes and the barrier, edge tiles only.
template <typename T, typename Cfg>
__device__ __forceinline__ void stage_edge_column(const PeriodicArgs& a, const PlaneIO& io, float* tile, int tile_x, const char* sbase, int gx0, int gy0,
                                                  int wave, int lane) {
    static_assert(Cfg::kLdsPitch >= Cfg::kLdsCols + 1 && Cfg::kLdsRows <= 64, "a spare word per tile row, one tile row per lane");
    const bool in_front = (a.edge.n[0] > 0 && tile_x == a.edge.tile_x[0] && a.edge.lds_col[0] < 0) || (a.edge.n[1] > 0 && tile_x == a.edge.tile_x[1] && a.edge.lds_col[1] < 0);
    if (in_front && wave == 0 && lane < Cfg::kLdsRows && gx0 > 0) {
        int gy = gy0 + lane;
        gy = gy < a.src_h ? gy : a.src_h - 1;
        tile[lane * Cfg::kLdsPitch - 1] = to_float(reinterpret_cast<const T*>(sbase + static_cast<size_t>(gy) * io.src_pitch)[gx0 - 1]);
    }
}

template <typename T, typename Cfg, int NR, int NC>
__device__ __forceinline__ void quad2_edge_columns(const PeriodicArgs& a, const PlaneIO& io, const float* tile, int tile_x, int j0, int wave, int lane,
                                                   BufferRsrc drsrc) {
    constexpr int NCP = (NC + 3) & ~3;  // floats per coefficient row
    for (int s = 0; s < 2; ++s) {
        const int n = a.edge.n[s];
        if (n <= 0 || tile_x != a.edge.tile_x[s]) continue;  // workgroup-uniform
        const int q = wave & 1, half = (n + 1) >> 1;
        const int k_begin = (wave >> 1) * half, k_end = k_begin + half < n ? k_begin + half : n;
        if (k_begin >= k_end) continue;  // wave-uniform
        const bool live = lane < Cfg::kTileRows && j0 + lane < a.nj;
        // The window: lane l reads the NC samples of tile row l, rows l + 1 .. l + NR - 1 come over from the lanes above by whole-wave
        // shifts (read from LDS, the lanes' rows lie a pitch apart: four banks for 64 lanes, a 16-way conflict per read.  Measured
        // level with this form all the same -- round5/edge_cols_ab.log -- the chains are what the edge tiles pay for).
        static_assert(Cfg::kLdsRows <= 64, "one tile row per lane");
        const float* wp = tile + (lane < Cfg::kLdsRows ? lane : 0) * Cfg::kLdsPitch + a.edge.lds_col[s];
        float w[NR][NC];
#pragma unroll
        for (int lx = 0; lx < NC; ++lx) w[0][lx] = wp[lx];
#pragma unroll
        for (int ly = 1; ly < NR; ++ly)
#pragma unroll
            for (int lx = 0; lx < NC; ++lx)  // wave_shl:1 -- lane l takes lane l + 1's value
                w[ly][lx] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, w[ly - 1][lx]), 0x130, 0xf, 0xf, false));
        const uint32_t row_off = static_cast<uint32_t>(a.iy0 + 2 * (j0 + lane) + q) * static_cast<uint32_t>(io.dst_pitch);
        for (int k0 = k_begin; k0 < k_end; k0 += 4) {  // wave-uniform
            const int nk = k_end - k0 < 4 ? k_end - k0 : 4;
            float r[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                r[k] = 0.f;
                if (k < nk) {  // wave-uniform
                    const JINC_CONSTANT float* cs = (const JINC_CONSTANT float*)(a.edge.coeffs) + ((s * PeriodicArgs::EdgeColumns::kMaxPerSide + k0 + k) * 2 + q) * (NR * NCP);
                    float acc = 0.f;
#pragma unroll
                    for (int ly = 0; ly < NR; ++ly)
#pragma unroll
                        for (int lx = 0; lx < NC; ++lx) acc = acc + w[ly][lx] * cs[ly * NCP + lx];
                    r[k] = acc;
                }
            }
            if (!live) continue;
            const uint32_t voff = row_off + static_cast<uint32_t>(a.edge.x0[s] + k0) * static_cast<uint32_t>(sizeof(T));
            if (nk == 4) {
                if constexpr (std::is_same_v<T, uint8_t>) {
                    uint32_t v = __builtin_amdgcn_cvt_pk_u8_f32(r[0], 0u, 0u);
                    v = __builtin_amdgcn_cvt_pk_u8_f32(r[1], 1u, v);
                    v = __builtin_amdgcn_cvt_pk_u8_f32(r[2], 2u, v);
                    v = __builtin_amdgcn_cvt_pk_u8_f32(r[3], 3u, v);
                    __builtin_amdgcn_raw_buffer_store_b32(v, drsrc, voff, 0, 0);
                } else {
                    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                    const u32x2 v = {round_pair_u16(r[0], r[1], io.peak), round_pair_u16(r[2], r[3], io.peak)};
                    __builtin_amdgcn_raw_buffer_store_b64(v, drsrc, voff, 0, 0);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    if (k < nk) store_sample_buf<T>(drsrc, voff + static_cast<uint32_t>(k * sizeof(T)), 0, r[k], io.peak);
            }
        }
    }
}

// INNER: bit 2 * ly + q set = taps 0 and 5 of kernel row ly are zero for both phases p of q and are not executed (the disc's chord
// in the box's edge rows).  Instantiated for no such rows and for the pattern of the 2x up-scale with tap 3 at blur 1 (q = 0: the
// last kernel row, q = 1: the first): the launcher takes the instantiation whose mask is a subset of the plan's.
constexpr uint32_t kQuad2InnerTap3 = PeriodicArgs::kQuadInnerTap3;
// NT: taps per kernel row -- 6, or 7 for the 6-row x 7-column support (chroma planes sited as MPEG-2 at 2x: the disc spans six
// source rows but, shifted by an eighth of a sample, seven columns; PeriodicArgs::quad_taps).
template <typename T, int RG, uint32_t INNER, int NT = 6>
__global__ __launch_bounds__(256, 6) void ewa_periodic_quad2_kernel(const PeriodicArgs a, const PlaneIO io) {
    using Cfg = Quad2Cfg<RG>;
    constexpr int FS = Cfg::FS;
    static_assert(RG % 4 == 0, "the four waves of a workgroup take RG / 4 row groups each");
    // (two words in front of the tile: row 0's "column -1", see the edge columns below; the tile stays 8-byte aligned)
    __shared__ __attribute__((aligned(16))) float tile_words[2 + Cfg::kLdsRows * Cfg::kLdsPitch];
    float* const tile = tile_words + 2;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int tile_x, tile_y;
    swizzled_tile(tile_x, tile_y);
    const int i0 = tile_x * Cfg::kTileCols;
    const int j0 = tile_y * Cfg::kTileRows;
    const size_t frame = blockIdx.z;
    if (skips_frame(a, frame)) return;  // float planes: the other launch's frame
    // integer planes: this tile column also computes the plane's border columns of its rows (PeriodicArgs::EdgeColumns)
    const bool edge_tile = edge_tile_of<T>(a, tile_x);
    {   // stage the source tile as fp32, all loads in front of the LDS writes (see ewa_periodic_kernel)
        const int gx0 = a.min_sx + i0;
        const int gy0 = a.min_sy + j0;
        const char* sbase = static_cast<const char*>(io.src) + frame * io.src_frame_stride;
        constexpr int kRowsPerWave = (Cfg::kLdsRows + 3) / 4;
        constexpr int kColsPerLane = (Cfg::kLdsCols + 63) / 64;
        T staged[kRowsPerWave][kColsPerLane];
        NonFinite<T> nonfinite(a, frame);
#pragma unroll
        for (int i = 0; i < kRowsPerWave; ++i) {
            int gy = gy0 + wave + 4 * i;
            gy = gy < a.src_h ? gy : a.src_h - 1;
            const T* srow = reinterpret_cast<const T*>(sbase + static_cast<size_t>(gy) * io.src_pitch);
#pragma unroll
            for (int k = 0; k < kColsPerLane; ++k) {
                int gx = gx0 + lane + 64 * k;
                gx = gx < a.src_w ? gx : a.src_w - 1;
                staged[i][k] = srow[gx];
            }
        }
#pragma unroll
        for (int i = 0; i < kRowsPerWave; ++i) {
            const int r = wave + 4 * i;
#pragma unroll
            for (int k = 0; k < kColsPerLane; ++k) {
                const int c = lane + 64 * k;
                // (the pitch's two spare words stay unwritten here: the second is the next row's column -1)
                if (r < Cfg::kLdsRows && c < Cfg::kLdsCols) tile[r * Cfg::kLdsPitch + c] = nonfinite.take(staged[i][k]);
            }
        }
        if constexpr (!std::is_same_v<T, float>) {
            if (edge_tile) stage_edge_column<T, Cfg>(a, io, tile, tile_x, sbase, gx0, gy0, wave, lane);
        }
    }
    __syncthreads();
    const BufferRsrc drsrc = make_rsrc(static_cast<char*>(io.dst) + frame * io.dst_frame_stride,
                                       static_cast<uint32_t>(io.dst_pitch) * a.dst_h);  // wave-uniform
    if constexpr (!std::is_same_v<T, float>) {
        if (edge_tile) quad2_edge_columns<T, Cfg, 6, 7>(a, io, tile, tile_x, j0, wave, lane, drsrc);
    }
    const int ia = i0 + 2 * lane;  // the lane's first period
    if (ia >= a.ni) return;        // no barrier below
    const bool b_ok = ia + 1 < a.ni;

    const JINC_CONSTANT f32x2* quad = (const JINC_CONSTANT f32x2*)(a.quad);
    // both phases of an axis share the window origin, which is also the tile's (host: quad != nullptr only then)
    const float* base = tile + 2 * lane;
    const uint32_t xoff = static_cast<uint32_t>(a.ix0 + 2 * ia) * static_cast<uint32_t>(sizeof(T));

    constexpr int kGroupsPerWave = RG / 4;
    const int g_first = wave * kGroupsPerWave;
    if (j0 + g_first * FS >= a.nj) return;  // wave-uniform: bottom tiles
    f32x2 win[24];
    {
        const float* wb = base + (g_first * FS) * Cfg::kLdsPitch;
        quad2_load_row<0>(win, wb + 0 * Cfg::kLdsPitch);
        quad2_load_row<1>(win, wb + 1 * Cfg::kLdsPitch);
        quad2_load_row<2>(win, wb + 2 * Cfg::kLdsPitch);
        quad2_load_row<3>(win, wb + 3 * Cfg::kLdsPitch);
        quad2_load_row<4>(win, wb + 4 * Cfg::kLdsPitch);
    }
    for (int g = g_first; g < g_first + kGroupsPerWave; ++g) {
        if (j0 + g * FS >= a.nj) break;  // wave-uniform
        const float* gbase = base + (g * FS) * Cfg::kLdsPitch;
#define JINC_QUAD2_ROW(U)                                                                                          \
    {                                                                                                              \
        quad2_load_row<(U + FS - 1) % FS>(win, gbase + (U + FS - 1) * Cfg::kLdsPitch);                              \
        f32x2 acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};                                            \
        uint32_t zero;                                                                                              \
        asm volatile("s_mov_b32 %0, 0" : "=s"(zero)); /* opaque: keeps the coefficient loads inside the row loop */ \
        quad2_pixel6<U, INNER, NT>(acc, win, quad + zero);                                                          \
        const int j = j0 + g * FS + U;                                                                              \
        if (j < a.nj) {                                                                                             \
            const uint32_t so = static_cast<uint32_t>(a.iy0 + 2 * j) * io.dst_pitch;                                \
            store_quad_buf<T>(drsrc, xoff, so, acc[0], acc[1], io.peak, b_ok);                                      \
            store_quad_buf<T>(drsrc, xoff, so + static_cast<uint32_t>(io.dst_pitch), acc[2], acc[3], io.peak, b_ok); \
        }                                                                                                           \
    }
        JINC_QUAD2_ROW(0) JINC_QUAD2_ROW(1) JINC_QUAD2_ROW(2) JINC_QUAD2_ROW(3) JINC_QUAD2_ROW(4) JINC_QUAD2_ROW(5)
#undef JINC_QUAD2_ROW
    }
}

// ------------------------------------------------------------------------------------------------
// Two periods per lane on the trimmed 8 x 8 support (ewa_periodic_quad2x8_kernel): ewa_periodic_quad2_kernel's mapping for tap 4
// ------------------------------------------------------------------------------------------------
// A lane owns two adjacent periods = 4 x 2 output samples from one 8-row x 9-column register window (rows kept ten wide: five
// register pairs, five aligned ds_read_b64 per row), four chains interleaved two by two, coefficient pairs shared by both
// periods.  100 registers: five waves per SIMD.
template <int RG>
struct Quad2x8Cfg {
    static constexpr int FS = 8;
    static constexpr int kTileCols = 128;
    static constexpr int kTileRows = FS * RG;
    static constexpr int kLdsCols = kTileCols + FS;   // lane 63 reads columns 126 .. 135
    static constexpr int kLdsPitch = 138;             // even
    static constexpr int kLdsRows = kTileRows + FS - 1;
};

__device__ __forceinline__ void quad2_row8(f32x2& acc_a, f32x2& acc_b, f32x2 w0, f32x2 w1, f32x2 w2, f32x2 w3, f32x2 w4, f32x2 c0, f32x2 c1,
                                           f32x2 c2, f32x2 c3, f32x2 c4, f32x2 c5, f32x2 c6, f32x2 c7) {
    f32x2 ta, tb;
#define JINC_LO(T, W, C) "v_pk_mul_f32 " T ", " W ", " C " op_sel_hi:[0,1]\n\t"
#define JINC_HI(T, W, C) "v_pk_mul_f32 " T ", " W ", " C " op_sel:[1,0] op_sel_hi:[1,1]\n\t"
#define JINC_ADD2 "v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3\n\t"
    asm(JINC_LO("%2", "%4", "%9") JINC_HI("%3", "%4", "%9") JINC_ADD2       // tap 0: A column 0, B column 1
        JINC_HI("%2", "%4", "%10") JINC_LO("%3", "%5", "%10") JINC_ADD2     // tap 1: A 1, B 2
        JINC_LO("%2", "%5", "%11") JINC_HI("%3", "%5", "%11") JINC_ADD2     // tap 2
        JINC_HI("%2", "%5", "%12") JINC_LO("%3", "%6", "%12") JINC_ADD2     // tap 3
        JINC_LO("%2", "%6", "%13") JINC_HI("%3", "%6", "%13") JINC_ADD2     // tap 4
        JINC_HI("%2", "%6", "%14") JINC_LO("%3", "%7", "%14") JINC_ADD2     // tap 5
        JINC_LO("%2", "%7", "%15") JINC_HI("%3", "%7", "%15") JINC_ADD2     // tap 6
        JINC_HI("%2", "%7", "%16") JINC_LO("%3", "%8", "%16")               // tap 7: A 7, B 8
        "v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3"
        : "+v"(acc_a), "+v"(acc_b), "=&v"(ta), "=&v"(tb)
        : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(w4), "s"(c0), "s"(c1), "s"(c2), "s"(c3), "s"(c4), "s"(c5), "s"(c6), "s"(c7));
#undef JINC_LO
#undef JINC_HI
#undef JINC_ADD2
}

template <int SLOT>
__device__ __forceinline__ void quad2x8_load_row(f32x2 (&w)[40], const float* p) {
    const f32x2* p2 = reinterpret_cast<const f32x2*>(p);
#pragma unroll
    for (int m = 0; m < 5; ++m) w[5 * SLOT + m] = p2[m];
}

template <int U, uint32_t TR8>
__device__ __forceinline__ void quad2_pixel8(f32x2 (&acc)[4], const f32x2 (&w)[40], const JINC_CONSTANT f32x2* quad) {
    f32x2 ca[16], cb[16];
    quad_fetch(ca, quad, 0);
#define JINC_QUAD2X8_STEP(LY, CUR, NEXT)                                                                                                  \
    if constexpr (LY < 7) quad_fetch(NEXT, quad, LY + 1);                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                                                     \
    quad_arrived(CUR);                                                                                                                     \
    {                                                                                                                                      \
        constexpr int S = 5 * ((U + LY) % 8);                                                                                              \
        if constexpr (quad8_trim_of(TR8, LY, 0) == 2)                                                                                      \
            quad2_row8_t2(acc[0], acc[1], w[S + 1], w[S + 2], w[S + 3], CUR[2], CUR[3], CUR[4], CUR[5]);                                   \
        else if constexpr (quad8_trim_of(TR8, LY, 0) == 1)                                                                                 \
            quad2_row8_t1(acc[0], acc[1], w[S], w[S + 1], w[S + 2], w[S + 3], CUR[1], CUR[2], CUR[3], CUR[4], CUR[5], CUR[6]);             \
        else                                                                                                                               \
            quad2_row8(acc[0], acc[1], w[S], w[S + 1], w[S + 2], w[S + 3], w[S + 4], CUR[0], CUR[1], CUR[2], CUR[3], CUR[4], CUR[5],       \
                       CUR[6], CUR[7]);                                                                                                    \
        if constexpr (quad8_trim_of(TR8, LY, 1) == 2)                                                                                      \
            quad2_row8_t2(acc[2], acc[3], w[S + 1], w[S + 2], w[S + 3], CUR[10], CUR[11], CUR[12], CUR[13]);                               \
        else if constexpr (quad8_trim_of(TR8, LY, 1) == 1)                                                                                 \
            quad2_row8_t1(acc[2], acc[3], w[S], w[S + 1], w[S + 2], w[S + 3], CUR[9], CUR[10], CUR[11], CUR[12], CUR[13], CUR[14]);        \
        else                                                                                                                               \
            quad2_row8(acc[2], acc[3], w[S], w[S + 1], w[S + 2], w[S + 3], w[S + 4], CUR[8], CUR[9], CUR[10], CUR[11], CUR[12], CUR[13],   \
                       CUR[14], CUR[15]);                                                                                                  \
    }                                                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);
    JINC_QUAD2X8_STEP(0, ca, cb)
    JINC_QUAD2X8_STEP(1, cb, ca)
    JINC_QUAD2X8_STEP(2, ca, cb)
    JINC_QUAD2X8_STEP(3, cb, ca)
    JINC_QUAD2X8_STEP(4, ca, cb)
    JINC_QUAD2X8_STEP(5, cb, ca)
    JINC_QUAD2X8_STEP(6, ca, cb)
    JINC_QUAD2X8_STEP(7, cb, ca)
#undef JINC_QUAD2X8_STEP
}

// Nine taps per kernel row for both periods of a lane (the 8-row x 9-column support: chroma planes sited as MPEG-2 at 2x with tap 4):
// period A reads columns 0 .. 8, period B columns 1 .. 9 of the row's five register pairs; FORM = the span of the row's taps that
// is executed (0: all nine, 1: taps 1 .. 8, 2: 1 .. 7, 3: 2 .. 7, 4: 2 .. 6 -- the others carry zero coefficients for both phases p).
#define JINC_Q9_EVEN(W, C) /* tap t even: period A = the low half of pair t / 2, B its high half */                         \
    "v_pk_mul_f32 %2, " W ", " C " op_sel_hi:[0,1]\n\tv_pk_mul_f32 %3, " W ", " C " op_sel:[1,0] op_sel_hi:[1,1]\n\t" \
    "v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3\n\t"
#define JINC_Q9_ODD(W, WN, C) /* tap t odd: A = the high half of pair t / 2, B the low half of the next pair */             \
    "v_pk_mul_f32 %2, " W ", " C " op_sel:[1,0] op_sel_hi:[1,1]\n\tv_pk_mul_f32 %3, " WN ", " C " op_sel_hi:[0,1]\n\t" \
    "v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3\n\t"
#define JINC_Q9_T0 JINC_Q9_EVEN("%4", "%9")
#define JINC_Q9_T1 JINC_Q9_ODD("%4", "%5", "%10")
#define JINC_Q9_T2 JINC_Q9_EVEN("%5", "%11")
#define JINC_Q9_T3 JINC_Q9_ODD("%5", "%6", "%12")
#define JINC_Q9_T4 JINC_Q9_EVEN("%6", "%13")
#define JINC_Q9_T5 JINC_Q9_ODD("%6", "%7", "%14")
#define JINC_Q9_T6 JINC_Q9_EVEN("%7", "%15")
#define JINC_Q9_T7 JINC_Q9_ODD("%7", "%8", "%16")
#define JINC_Q9_T8 JINC_Q9_EVEN("%8", "%17")
#define JINC_Q9_OPERANDS                                                                                                          \
    : "+v"(acc_a), "+v"(acc_b), "=&v"(ta), "=&v"(tb)                                                                               \
    : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "s"(c[0]), "s"(c[1]), "s"(c[2]), "s"(c[3]), "s"(c[4]), "s"(c[5]), "s"(c[6]), \
      "s"(c[7]), "s"(c[8])
template <int FORM>
__device__ __forceinline__ void quad2_row9_span(f32x2& acc_a, f32x2& acc_b, const f32x2 (&w)[5], const f32x2 (&c)[10]) {
    static_assert(FORM >= 0 && FORM <= 4, "span forms of the nine-tap row");
    f32x2 ta, tb;
    if constexpr (FORM == 0)
        asm(JINC_Q9_T0 JINC_Q9_T1 JINC_Q9_T2 JINC_Q9_T3 JINC_Q9_T4 JINC_Q9_T5 JINC_Q9_T6 JINC_Q9_T7 JINC_Q9_T8 "" JINC_Q9_OPERANDS);
    else if constexpr (FORM == 1)
        asm(JINC_Q9_T1 JINC_Q9_T2 JINC_Q9_T3 JINC_Q9_T4 JINC_Q9_T5 JINC_Q9_T6 JINC_Q9_T7 JINC_Q9_T8 "" JINC_Q9_OPERANDS);
    else if constexpr (FORM == 2)
        asm(JINC_Q9_T1 JINC_Q9_T2 JINC_Q9_T3 JINC_Q9_T4 JINC_Q9_T5 JINC_Q9_T6 JINC_Q9_T7 "" JINC_Q9_OPERANDS);
    else if constexpr (FORM == 3)
        asm(JINC_Q9_T2 JINC_Q9_T3 JINC_Q9_T4 JINC_Q9_T5 JINC_Q9_T6 JINC_Q9_T7 "" JINC_Q9_OPERANDS);
    else
        asm(JINC_Q9_T2 JINC_Q9_T3 JINC_Q9_T4 JINC_Q9_T5 JINC_Q9_T6 "" JINC_Q9_OPERANDS);
}
#undef JINC_Q9_EVEN
#undef JINC_Q9_ODD
#undef JINC_Q9_T0
#undef JINC_Q9_T1
#undef JINC_Q9_T2
#undef JINC_Q9_T3
#undef JINC_Q9_T4
#undef JINC_Q9_T5
#undef JINC_Q9_T6
#undef JINC_Q9_T7
#undef JINC_Q9_T8
#undef JINC_Q9_OPERANDS

// One output row pair of both periods on the 8 x 9 support.  The ten coefficient pairs of a (kernel row, q) are 20 SGPRs: the two
// row phases alternate through two sets (as quad_pixel9): the pairs of (ly, q = 1) are requested before the taps of (ly, q = 0) are
// issued, those of (ly + 1, q = 0) before the taps of (ly, q = 1).  Layout: quad[ly][q][10 pairs][p] (device_plan.cpp attach_quad, NX = 9).
template <int U, uint64_t SPAN9>
__device__ __forceinline__ void quad2_pixel9(f32x2 (&acc)[4], const f32x2 (&w)[40], const JINC_CONSTANT f32x2* quad) {
    f32x2 ca[10], cb[10];
    quad_fetch10(ca, quad, 0);
#define JINC_QUAD2X9_STEP(LY)                                                                                      \
    quad_fetch10(cb, quad, 2 * LY + 1);                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    quad_arrived10(ca);                                                                                            \
    {                                                                                                              \
        constexpr int S = 5 * ((U + LY) % 8);                                                                      \
        const f32x2 row[5] = {w[S], w[S + 1], w[S + 2], w[S + 3], w[S + 4]};                                       \
        quad2_row9_span<static_cast<int>((SPAN9 >> (3 * (2 * LY))) & 7u)>(acc[0], acc[1], row, ca);                 \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        if constexpr (LY < 7) quad_fetch10(ca, quad, 2 * LY + 2);                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        quad_arrived10(cb);                                                                                        \
        quad2_row9_span<static_cast<int>((SPAN9 >> (3 * (2 * LY + 1))) & 7u)>(acc[2], acc[3], row, cb);             \
    }                                                                                                              \
    __builtin_amdgcn_sched_barrier(0);
    JINC_QUAD2X9_STEP(0) JINC_QUAD2X9_STEP(1) JINC_QUAD2X9_STEP(2) JINC_QUAD2X9_STEP(3) JINC_QUAD2X9_STEP(4) JINC_QUAD2X9_STEP(5)
    JINC_QUAD2X9_STEP(6) JINC_QUAD2X9_STEP(7)
#undef JINC_QUAD2X9_STEP
}

// NT: taps per kernel row -- 8, or 9 for the 8-row x 9-column support (then SPAN9 = the span form of every (kernel row, q), TR8 unused).
template <typename T, int RG, uint32_t TR8, int NT = 8, uint64_t SPAN9 = 0>
__global__ __launch_bounds__(256, 5) void ewa_periodic_quad2x8_kernel(const PeriodicArgs a, const PlaneIO io) {
    using Cfg = Quad2x8Cfg<RG>;
    constexpr int FS = Cfg::FS;
    static_assert(RG % 4 == 0, "the four waves of a workgroup take RG / 4 row groups each");
    // (two words in front of the tile: row 0's "column -1" of the edge columns, stage_edge_column; the tile stays 8-byte aligned)
    __shared__ __attribute__((aligned(16))) float tile_words[2 + Cfg::kLdsRows * Cfg::kLdsPitch];
    float* const tile = tile_words + 2;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int tile_x, tile_y;
    swizzled_tile(tile_x, tile_y);
    const int i0 = tile_x * Cfg::kTileCols;
    const int j0 = tile_y * Cfg::kTileRows;
    const size_t frame = blockIdx.z;
    if (skips_frame(a, frame)) return;  // float planes: the other launch's frame
    const bool edge_tile = edge_tile_of<T>(a, tile_x);  // integer planes: this tile column computes the plane's border columns of its rows
    {   // stage the source tile as fp32, in two halves of the rows (all loads of a half in front of its LDS writes): the whole
        // tile at once would hold more staged registers than the compute phase has
        const int gx0 = a.min_sx + i0;
        const int gy0 = a.min_sy + j0;
        const char* sbase = static_cast<const char*>(io.src) + frame * io.src_frame_stride;
        constexpr int kRowsPerWave = (Cfg::kLdsRows + 3) / 4;
        constexpr int kHalf = (kRowsPerWave + 1) / 2;
        constexpr int kColsPerLane = (Cfg::kLdsPitch + 63) / 64;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            T staged[kHalf][kColsPerLane];
            NonFinite<T> nonfinite(a, frame);
#pragma unroll
            for (int i = 0; i < kHalf; ++i) {
                int gy = gy0 + wave + 4 * (h * kHalf + i);
                gy = gy < a.src_h ? gy : a.src_h - 1;
                const T* srow = reinterpret_cast<const T*>(sbase + static_cast<size_t>(gy) * io.src_pitch);
#pragma unroll
                for (int k = 0; k < kColsPerLane; ++k) {
                    int gx = gx0 + lane + 64 * k;
                    gx = gx < a.src_w ? gx : a.src_w - 1;
                    staged[i][k] = srow[gx];
                }
            }
#pragma unroll
            for (int i = 0; i < kHalf; ++i) {
                const int r = wave + 4 * (h * kHalf + i);
#pragma unroll
                for (int k = 0; k < kColsPerLane; ++k) {
                    const int c = lane + 64 * k;
                    // (the pitch's two spare words stay unwritten here: the second is the next row's column -1)
                    if (r < Cfg::kLdsRows && c < Cfg::kLdsCols) tile[r * Cfg::kLdsPitch + c] = nonfinite.take(staged[i][k]);
                }
            }
        }
        if constexpr (!std::is_same_v<T, float>) {
            if (edge_tile) stage_edge_column<T, Cfg>(a, io, tile, tile_x, sbase, gx0, gy0, wave, lane);
        }
    }
    __syncthreads();
    const BufferRsrc drsrc = make_rsrc(static_cast<char*>(io.dst) + frame * io.dst_frame_stride,
                                       static_cast<uint32_t>(io.dst_pitch) * a.dst_h);  // wave-uniform
    if constexpr (!std::is_same_v<T, float>) {
        if (edge_tile) quad2_edge_columns<T, Cfg, 8, 9>(a, io, tile, tile_x, j0, wave, lane, drsrc);
    }
    const int ia = i0 + 2 * lane;
    if (ia >= a.ni) return;  // no barrier below
    const bool b_ok = ia + 1 < a.ni;

    const JINC_CONSTANT f32x2* quad = (const JINC_CONSTANT f32x2*)(a.quad);
    const float* base = tile + 2 * lane;
    const uint32_t xoff = static_cast<uint32_t>(a.ix0 + 2 * ia) * static_cast<uint32_t>(sizeof(T));

    constexpr int kGroupsPerWave = RG / 4;
    const int g_first = wave * kGroupsPerWave;
    if (j0 + g_first * FS >= a.nj) return;  // wave-uniform: bottom tiles
    f32x2 win[40];
    {
        const float* wb = base + (g_first * FS) * Cfg::kLdsPitch;
        quad2x8_load_row<0>(win, wb + 0 * Cfg::kLdsPitch);
        quad2x8_load_row<1>(win, wb + 1 * Cfg::kLdsPitch);
        quad2x8_load_row<2>(win, wb + 2 * Cfg::kLdsPitch);
        quad2x8_load_row<3>(win, wb + 3 * Cfg::kLdsPitch);
        quad2x8_load_row<4>(win, wb + 4 * Cfg::kLdsPitch);
        quad2x8_load_row<5>(win, wb + 5 * Cfg::kLdsPitch);
        quad2x8_load_row<6>(win, wb + 6 * Cfg::kLdsPitch);
    }
    for (int g = g_first; g < g_first + kGroupsPerWave; ++g) {
        if (j0 + g * FS >= a.nj) break;  // wave-uniform
        const float* gbase = base + (g * FS) * Cfg::kLdsPitch;
#define JINC_QUAD2X8_ROW(U)                                                                                        \
    {                                                                                                              \
        quad2x8_load_row<(U + FS - 1) % FS>(win, gbase + (U + FS - 1) * Cfg::kLdsPitch);                            \
        f32x2 acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};                                            \
        uint32_t zero;                                                                                              \
        asm volatile("s_mov_b32 %0, 0" : "=s"(zero)); /* opaque: keeps the coefficient loads inside the row loop */ \
        if constexpr (NT == 9) quad2_pixel9<U, SPAN9>(acc, win, quad + zero);                                       \
        else quad2_pixel8<U, TR8>(acc, win, quad + zero);                                                           \
        const int j = j0 + g * FS + U;                                                                              \
        if (j < a.nj) {                                                                                             \
            const uint32_t so = static_cast<uint32_t>(a.iy0 + 2 * j) * io.dst_pitch;                                \
            store_quad_buf<T>(drsrc, xoff, so, acc[0], acc[1], io.peak, b_ok);                                      \
            store_quad_buf<T>(drsrc, xoff, so + static_cast<uint32_t>(io.dst_pitch), acc[2], acc[3], io.peak, b_ok); \
        }                                                                                                           \
    }
        JINC_QUAD2X8_ROW(0) JINC_QUAD2X8_ROW(1) JINC_QUAD2X8_ROW(2) JINC_QUAD2X8_ROW(3) JINC_QUAD2X8_ROW(4) JINC_QUAD2X8_ROW(5) JINC_QUAD2X8_ROW(6)
        JINC_QUAD2X8_ROW(7)
#undef JINC_QUAD2X8_ROW
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

// One kernel row of a rows item on the taps lx = TR .. FS-1-TR: the taps in front of and behind that span carry the
// coefficient 0.0f in this kernel row of this phase (the EWA disc's chord; host: trim_periodic, row_trim) and are left out,
// which is exact for the integer planes the trimmed support is built for.  TR = 0: every tap.
template <int FS, int KC, int OFF, int TR>
__device__ __forceinline__ void rows_kernel_row(float (&acc)[4][KC], const float* __restrict__ tile, const unsigned (&plane_off)[KC],
                                                const JINC_CONSTANT float* crow) {
    using Cfg = RowsCfg<FS, KC>;
    constexpr int K = Cfg::K, R = Cfg::R, N = FS - 2 * TR;
    static_assert(R == 4, "four chain rows per lane");
    float c[N];
#pragma unroll
    for (int lx = 0; lx < N; ++lx) c[lx] = crow[TR + lx];
#pragma unroll
    for (int jj = 0; jj < R; ++jj) {
        float seg[N + K - 1];
#pragma unroll
        for (int u = 0; u < N + K - 1; ++u)
            seg[u] = tile[plane_off[(u + TR + OFF) % K] + (jj * Cfg::kPlane + (u + TR + OFF) / K)];
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int lx = 0; lx < N; ++lx) acc[jj][k] = acc[jj][k] + seg[k + lx] * c[lx];
    }
}

template <typename T, int FS, int KC, int OFF>
__device__ __forceinline__ void rows_item(const float* __restrict__ tile, unsigned base_off, const JINC_CONSTANT float* cs,
                                          const JINC_CONSTANT int32_t* row_trim, int nrows, BufferRsrc drsrc,
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

    for (int ly = 0; ly < nrows; ++ly) {  // (FS, or fewer on a support that is wider than tall: PeriodicArgs::rows_ny)
        const JINC_CONSTANT float* crow = cs + ly * padded_row(FS);
        // taps this kernel row leaves out on either side (wave-uniform: the phase's, from the plan); spans are offered in
        // steps of one tap up to five, a row whose zero flanks are wider takes the widest
        const int tr = row_trim ? row_trim[ly] : 0;
        if constexpr (FS >= 12) {
            switch (tr) {
                case 0: rows_kernel_row<FS, KC, OFF, 0>(acc, tile, plane_off, crow); break;
                case 1: rows_kernel_row<FS, KC, OFF, 1>(acc, tile, plane_off, crow); break;
                case 2: rows_kernel_row<FS, KC, OFF, 2>(acc, tile, plane_off, crow); break;
                case 3: rows_kernel_row<FS, KC, OFF, 3>(acc, tile, plane_off, crow); break;
                case 4: rows_kernel_row<FS, KC, OFF, 4>(acc, tile, plane_off, crow); break;
                default: rows_kernel_row<FS, KC, OFF, 5>(acc, tile, plane_off, crow); break;
            }
        } else if constexpr (FS >= 6) {
            switch (tr) {
                case 0: rows_kernel_row<FS, KC, OFF, 0>(acc, tile, plane_off, crow); break;
                case 1: rows_kernel_row<FS, KC, OFF, 1>(acc, tile, plane_off, crow); break;
                default: rows_kernel_row<FS, KC, OFF, 2>(acc, tile, plane_off, crow); break;
            }
        } else {
            rows_kernel_row<FS, KC, OFF, 0>(acc, tile, plane_off, crow);
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
__global__ __launch_bounds__(512, 8) void ewa_periodic_rows_kernel(const PeriodicArgs a, const PlaneIO io) {
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
    if (skips_frame(a, frame)) return;  // float planes: the other launch's frame

    {
        const int gx0 = a.min_sx + i0;
        const int gy0 = a.min_sy + j0;
        const char* sbase = static_cast<const char*>(io.src) + frame * io.src_frame_stride;
        // all loads in front of the LDS writes: one memory latency per tile instead of one per row (see ewa_periodic_kernel)
        constexpr int kRowsPerWave = (Cfg::kRows + Cfg::kWaves - 1) / Cfg::kWaves;
        constexpr int kColsPerLane = (Cfg::kCols + 63) / 64;
        T staged[kRowsPerWave][kColsPerLane];
        NonFinite<T> nonfinite(a, frame);
#pragma unroll
        for (int i = 0; i < kRowsPerWave; ++i) {
            int gy = gy0 + wave + Cfg::kWaves * i;
            gy = gy < a.src_h ? gy : a.src_h - 1;
            const T* srow = reinterpret_cast<const T*>(sbase + static_cast<size_t>(gy) * io.src_pitch);
#pragma unroll
            for (int k = 0; k < kColsPerLane; ++k) {
                int gx = gx0 + lane + 64 * k;
                gx = gx < a.src_w ? gx : a.src_w - 1;
                staged[i][k] = srow[gx];
            }
        }
#pragma unroll
        for (int i = 0; i < kRowsPerWave; ++i) {
            const int r = wave + Cfg::kWaves * i;
#pragma unroll
            for (int k = 0; k < kColsPerLane; ++k) {
                const int c = lane + 64 * k;
                if (r < Cfg::kRows && c < Cfg::kCols)
                    tile[(c % K) * Cfg::kPlaneStride + r * Cfg::kPlane + c / K] = nonfinite.take(staged[i][k]);
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
        const int nrows = a.rows_ny ? a.rows_ny : FS;
        const JINC_CONSTANT float* cs =
            (const JINC_CONSTANT float*)(a.coeffs + static_cast<size_t>(a.set[ph]) * (static_cast<size_t>(nrows) * padded_row(FS)));
        const unsigned base = ((a.start_y[q] - a.min_sy) + ch * R) * Cfg::kPlane + lane;
        const int y0 = a.iy0 + a.py * j + q;
        const unsigned x0 = a.ix0 + a.px * (i0 + K * lane) + p;
        const JINC_CONSTANT int32_t* row_trim = a.row_trim ? (const JINC_CONSTANT int32_t*)(a.row_trim) + ph * 32 : nullptr;
        if (a.start_x[p] - a.min_sx)
            rows_item<T, FS, KC, 1>(tile, base, cs, row_trim, nrows, dframe, io.dst_pitch, io.peak, y0, a.py, rows_valid, x0, a.px, cols_valid);
        else
            rows_item<T, FS, KC, 0>(tile, base, cs, row_trim, nrows, dframe, io.dst_pitch, io.peak, y0, a.py, rows_valid, x0, a.px, cols_valid);
    }
}

template <typename T, int FS, int RG = PeriodicCfg<FS>::kRowGroups>
int launch_periodic_t(const PeriodicArgs& pa, const PlaneIO& io, hipStream_t stream) {
    using Cfg = PeriodicCfg<FS, RG>;
    dim3 grid((pa.ni + kTileCols - 1) / kTileCols, (pa.nj + Cfg::kTileRows - 1) / Cfg::kTileRows, io.nframes);
    hipLaunchKernelGGL((ewa_periodic_kernel<T, FS, RG>), grid, dim3(256, 1, 1), 0, stream, pa, io);
    knobs::note_instance("ewa_periodic_kernel", "%s, %d, %d", knobs::type_name<T>(), FS, RG);
    return static_cast<int>(hipGetLastError());
}

template <typename T, int FS, int RG>
int launch_periodic_pk_t(const PeriodicArgs& pa, const PlaneIO& io, hipStream_t stream) {
    using Cfg = PeriodicPkCfg<FS, RG>;
    dim3 grid((pa.ni + Cfg::kTileCols - 1) / Cfg::kTileCols, (pa.nj + Cfg::kTileRows - 1) / Cfg::kTileRows, io.nframes);
    hipLaunchKernelGGL((ewa_periodic_pk_kernel<T, FS, RG>), grid, dim3(256, 1, 1), 0, stream, pa, io);
    knobs::note_instance("ewa_periodic_pk_kernel", "%s, %d, %d", knobs::type_name<T>(), FS, RG);
    return static_cast<int>(hipGetLastError());
}

template <typename T, int RG>
int launch_periodic_quad_t(const PeriodicArgs& pa, const PlaneIO& io, hipStream_t stream) {
    using Cfg = PeriodicCfg<7, RG>;
    dim3 grid((pa.ni + kTileCols - 1) / kTileCols, (pa.nj + Cfg::kTileRows - 1) / Cfg::kTileRows, io.nframes);
    hipLaunchKernelGGL((ewa_periodic_quad_kernel<T, RG>), grid, dim3(256, 1, 1), 0, stream, pa, io);
    knobs::note_instance("ewa_periodic_quad_kernel", "%s, %d", knobs::type_name<T>(), RG);
    return static_cast<int>(hipGetLastError());
}

template <typename T, int RG>
int launch_periodic_quad9_t(const PeriodicArgs& pa, const PlaneIO& io, hipStream_t stream) {
    using Cfg = PeriodicCfg<9, RG>;
    dim3 grid((pa.ni + kTileCols - 1) / kTileCols, (pa.nj + Cfg::kTileRows - 1) / Cfg::kTileRows, io.nframes);
    hipLaunchKernelGGL((ewa_periodic_quad9_kernel<T, RG>), grid, dim3(256, 1, 1), 0, stream, pa, io);
    knobs::note_instance("ewa_periodic_quad9_kernel", "%s, %d", knobs::type_name<T>(), RG);
    return static_cast<int>(hipGetLastError());
}

template <typename T, int RG>
int launch_periodic_quad8_t(const PeriodicArgs& pa, const PlaneIO& io, hipStream_t stream) {
    using Cfg = PeriodicCfg<8, RG>;
    dim3 grid((pa.ni + kTileCols - 1) / kTileCols, (pa.nj + Cfg::kTileRows - 1) / Cfg::kTileRows, io.nframes);
    const bool pattern = quad8_pattern_fits(pa.quad_trim8, kQuad8TrimTap4);
    if (pattern)
        hipLaunchKernelGGL((ewa_periodic_quad8_kernel<T, RG, kQuad8TrimTap4>), grid, dim3(256, 1, 1), 0, stream, pa, io);
    else
        hipLaunchKernelGGL((ewa_periodic_quad8_kernel<T, RG, 0u>), grid, dim3(256, 1, 1), 0, stream, pa, io);
    knobs::note_instance("ewa_periodic_quad8_kernel", "%s, %d, %uu", knobs::type_name<T>(), RG, pattern ? kQuad8TrimTap4 : 0u);
    return static_cast<int>(hipGetLastError());
}

template <typename T, int RG>
int launch_periodic_quad2x8_t(const PeriodicArgs& pa, const PlaneIO& io, hipStream_t stream) {
    using Cfg = Quad2x8Cfg<RG>;
    dim3 grid((pa.ni + Cfg::kTileCols - 1) / Cfg::kTileCols, (pa.nj + Cfg::kTileRows - 1) / Cfg::kTileRows, io.nframes);
    if (pa.quad_taps == 9) {  // the 8-row x 9-column support (chroma at tap 4), by the chord pattern the plan's spans fit
        const uint64_t span = quad_span9_fits(pa.quad_span7, kQuadSpan9Mpeg2) ? kQuadSpan9Mpeg2 : quad_span9_fits(pa.quad_span7, kQuadSpan9Mpeg2Swapped) ? kQuadSpan9Mpeg2Swapped : 0;
        if (span == kQuadSpan9Mpeg2)
            hipLaunchKernelGGL((ewa_periodic_quad2x8_kernel<T, RG, 0u, 9, kQuadSpan9Mpeg2>), grid, dim3(256, 1, 1), 0, stream, pa, io);
        else if (span == kQuadSpan9Mpeg2Swapped)
            hipLaunchKernelGGL((ewa_periodic_quad2x8_kernel<T, RG, 0u, 9, kQuadSpan9Mpeg2Swapped>), grid, dim3(256, 1, 1), 0, stream, pa, io);
        else
            hipLaunchKernelGGL((ewa_periodic_quad2x8_kernel<T, RG, 0u, 9, 0>), grid, dim3(256, 1, 1), 0, stream, pa, io);
        knobs::note_instance("ewa_periodic_quad2x8_kernel", "%s, %d, 0u, 9, %lluul", knobs::type_name<T>(), RG, static_cast<unsigned long long>(span));
        return static_cast<int>(hipGetLastError());
    }
    const bool pattern = quad8_pattern_fits(pa.quad_trim8, kQuad8TrimTap4);
    if (pattern)
        hipLaunchKernelGGL((ewa_periodic_quad2x8_kernel<T, RG, kQuad8TrimTap4>), grid, dim3(256, 1, 1), 0, stream, pa, io);
    else
        hipLaunchKernelGGL((ewa_periodic_quad2x8_kernel<T, RG, 0u>), grid, dim3(256, 1, 1), 0, stream, pa, io);
    knobs::note_instance("ewa_periodic_quad2x8_kernel", "%s, %d, %uu, 8, 0ul", knobs::type_name<T>(), RG, pattern ? kQuad8TrimTap4 : 0u);  // (as rocprofv3 spells it)
    return static_cast<int>(hipGetLastError());
}

template <typename T, int RG>
int launch_periodic_quad2_t(const PeriodicArgs& pa, const PlaneIO& io, hipStream_t stream) {
    using Cfg = Quad2Cfg<RG>;
    dim3 grid((pa.ni + Cfg::kTileCols - 1) / Cfg::kTileCols, (pa.nj + Cfg::kTileRows - 1) / Cfg::kTileRows, io.nframes);
    if (pa.quad_taps == 7 && quad_span7_fits(pa.quad_span7, PeriodicArgs::kQuadSpan7Mpeg2)) {  // the chords of MPEG-2 sited chroma at 2x: 36 of 42 taps
        hipLaunchKernelGGL((ewa_periodic_quad2_kernel<T, RG, PeriodicArgs::kQuadSpan7Mpeg2, 7>), grid, dim3(256, 1, 1), 0, stream, pa, io);
        knobs::note_instance("ewa_periodic_quad2_kernel", "%s, %d, %uu, 7", knobs::type_name<T>(), RG, PeriodicArgs::kQuadSpan7Mpeg2);
    } else if (pa.quad_taps == 7 && quad_span7_fits(pa.quad_span7, PeriodicArgs::kQuadSpan7Mpeg2Swapped)) {
        hipLaunchKernelGGL((ewa_periodic_quad2_kernel<T, RG, PeriodicArgs::kQuadSpan7Mpeg2Swapped, 7>), grid, dim3(256, 1, 1), 0, stream, pa, io);
        knobs::note_instance("ewa_periodic_quad2_kernel", "%s, %d, %uu, 7", knobs::type_name<T>(), RG, PeriodicArgs::kQuadSpan7Mpeg2Swapped);
    } else if (pa.quad_taps == 7) {
        hipLaunchKernelGGL((ewa_periodic_quad2_kernel<T, RG, 0u, 7>), grid, dim3(256, 1, 1), 0, stream, pa, io);
        knobs::note_instance("ewa_periodic_quad2_kernel", "%s, %d, 0u, 7", knobs::type_name<T>(), RG);
    } else if ((pa.quad_inner & kQuad2InnerTap3) == kQuad2InnerTap3) {
        hipLaunchKernelGGL((ewa_periodic_quad2_kernel<T, RG, kQuad2InnerTap3>), grid, dim3(256, 1, 1), 0, stream, pa, io);
        knobs::note_instance("ewa_periodic_quad2_kernel", "%s, %d, %uu, 6", knobs::type_name<T>(), RG, kQuad2InnerTap3);
    } else {
        hipLaunchKernelGGL((ewa_periodic_quad2_kernel<T, RG, 0u>), grid, dim3(256, 1, 1), 0, stream, pa, io);
        knobs::note_instance("ewa_periodic_quad2_kernel", "%s, %d, 0u, 6", knobs::type_name<T>(), RG);
    }
    return static_cast<int>(hipGetLastError());
}

template <typename T, int FS, int KC>
int launch_rows_k(const PeriodicArgs& pa, const PlaneIO& io, hipStream_t stream) {
    using Cfg = RowsCfg<FS, KC>;
    dim3 grid((pa.ni + Cfg::kTileCols - 1) / Cfg::kTileCols, (pa.nj + Cfg::kTileRows - 1) / Cfg::kTileRows, io.nframes);
    hipLaunchKernelGGL((ewa_periodic_rows_kernel<T, FS, KC>), grid, dim3(Cfg::kThreads, 1, 1), 0, stream, pa, io);
    knobs::note_instance("ewa_periodic_rows_kernel", "%s, %d, %d", knobs::type_name<T>(), FS, KC);
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
    if ((variant == 5 || variant == 6) && fs == 6 && pa.quad)  // trimmed support, two periods per lane: 5 = tiles of 8 row groups, 6 = of 4
        return variant == 5 ? launch_periodic_quad2_t<T, 8>(pa, io, stream) : launch_periodic_quad2_t<T, 4>(pa, io, stream);
    if (variant == 7 && fs == 8 && pa.quad) return launch_periodic_quad2x8_t<T, 4>(pa, io, stream);  // two periods per lane
    // Border columns handed over with the interior launch (PeriodicArgs::edge) are computed by the two launchers above and by
    // nobody else: a caller that dropped its own column launches on the strength of them and then lands anywhere below -- a knob
    // changed between its decision and this one, a rule edited on one side only -- must hear about it, not ship frames with
    // unwritten columns.
    if (pa.edge.coeffs) return static_cast<int>(hipErrorInvalidValue);
    if ((variant == 5 || variant == 6) && fs == 8 && pa.quad)  // trimmed 8 x 8 support: 5 = tiles of 8 row groups, 6 = of 4
        return variant == 5 ? launch_periodic_quad8_t<T, 8>(pa, io, stream) : launch_periodic_quad8_t<T, 4>(pa, io, stream);
    if ((variant == 5 || variant == 6) && fs == 7 && pa.quad)  // quad form: 5 = tiles of 8 row groups, 6 = of 4 (small calls)
        return variant == 5 ? launch_periodic_quad_t<T, 8>(pa, io, stream) : launch_periodic_quad_t<T, 4>(pa, io, stream);
    if ((variant == 5 || variant == 6) && fs == 9 && pa.quad)
        return launch_periodic_quad9_t<T, 4>(pa, io, stream);  // (8 row groups per tile: the staging's registers spill, no gain over the window kernel)
    if (variant == 3 && fs == 7) return launch_periodic_pk_t<T, 7, 4>(pa, io, stream);
    if (variant == 4 && fs == 7) return launch_periodic_pk_t<T, 7, 8>(pa, io, stream);
    if (variant == 2 && fs == 7) return launch_periodic_t<T, 7, 4>(pa, io, stream);
    if (variant == 2 && fs == 9) return launch_periodic_t<T, 9, 6>(pa, io, stream);
    if (variant == 2 && fs == 6) return launch_periodic_t<T, 6, 4>(pa, io, stream);
    if (variant == 2 && fs == 8) return launch_periodic_t<T, 8, 5>(pa, io, stream);
    if (variant == 1) {
        if (fs == 7) return launch_rows_t<T, 7>(pa, io, stream);
        if (fs == 9) return launch_rows_t<T, 9>(pa, io, stream);
        if (fs == 6) return launch_rows_t<T, 6>(pa, io, stream);
        if (fs == 8) return launch_rows_t<T, 8>(pa, io, stream);
    }
    switch (fs) {  // (even sizes: trimmed supports of integer planes, device_plan.cpp trim_periodic)
        case 3: return launch_rows_t<T, 3>(pa, io, stream);
        case 4: return launch_rows_t<T, 4>(pa, io, stream);
        case 5: return launch_rows_t<T, 5>(pa, io, stream);
        case 6: return launch_periodic_t<T, 6>(pa, io, stream);
        case 7: return launch_periodic_t<T, 7>(pa, io, stream);
        case 8: return launch_periodic_t<T, 8>(pa, io, stream);
        case 9: return launch_periodic_t<T, 9>(pa, io, stream);
        case 10: return launch_rows_t<T, 10>(pa, io, stream);
        case 11: return launch_rows_t<T, 11>(pa, io, stream);
        case 12: return launch_rows_t<T, 12>(pa, io, stream);
        case 13: return launch_rows_t<T, 13>(pa, io, stream);
        case 14: return launch_rows_t<T, 14>(pa, io, stream);
        case 15: return launch_rows_t<T, 15>(pa, io, stream);
        case 16: return launch_rows_t<T, 16>(pa, io, stream);
        case 17: return launch_rows_t<T, 17>(pa, io, stream);
        default: return static_cast<int>(hipErrorInvalidValue);
    }
}


}  // namespace

bool periodic_supported(int fs, int px, int py, int sx, int sy) {
    if (sx != 1 || sy != 1) return false;
    if (px < 1 || py < 1 || px > 8 || py > 8) return false;
    return fs >= 3 && fs <= 17 && (fs & 1);  // taps 1..8 at >= 1x scale
}

int launch_periodic(const PeriodicArgs& args, int fs, const PlaneIO& io, void* stream, int variant) {
    if (args.ni <= 0 || args.nj <= 0 || io.nframes <= 0) return 0;
    // 2x up-scales with 12 .. 17 taps per kernel row: the rows kernel's packed phase-pair form (kernel_rowpair.hip) wherever the
    // plan carries its coefficient pairs; variant 1 (kernel mode 3) and the knob ROWS_PAIR = 0 keep ewa_periodic_rows_kernel
    // Short kernel rows (6 .. 9 taps: taps 3 and 4) stay with the window and quad forms, which measure ahead there (C2 847 : 816
    // Gpix/s, C4 152 : 139; profiles/round5/rowpair_small_ab.log); the knob ROWPAIR_SMALL = 1 puts them on the pair form wherever
    // the plan carries the pairs and no A/B variant of the other kernels is asked for.
    if (args.rowpair && knobs::flag(JINC_KNOB_ROWS_PAIR, true)) {
        const int taps = args.quad_taps ? args.quad_taps : fs;  // (6 rows x 7 columns: fs = 6 rows, 7 taps per row)
        const bool auto_variant = variant == 0 || variant == 2 || variant == 5 || variant == 6 || variant == 7;
        if (args.rowpair_n == taps && !args.edge.coeffs &&  // (edge columns: the caller counts on a two-periods quad form)
            ((fs >= 10 && variant == 0) || (fs < 10 && auto_variant && knobs::geti(JINC_KNOB_ROWPAIR_SMALL, 0) == 1)))
            return launch_rowpair(args, io, stream);
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (io.sample_bytes) {
        case 1: return launch_periodic_fs<uint8_t>(args, fs, io, s, variant);
        case 2: return launch_periodic_fs<uint16_t>(args, fs, io, s, variant);
        default: return launch_periodic_fs<float>(args, fs, io, s, variant);
    }
}


}  // namespace jinc
