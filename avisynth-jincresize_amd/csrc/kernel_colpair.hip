// kernel_colpair.hip -- ewa_colpair_kernel: the border columns of exactly periodic plans at source step 1 (filter sizes 7 .. 17), interior
// rows, on packed column pairs.
//
// Every border column of a side reads the same fs source columns (the reference shifts the window back inside the image, ref
// /root/reference/src/JincResize.cpp:395-418) and, at output row iy0 + py * j + q, the source rows start_y[q] + j ...: the columns of
// a side differ in their coefficient sets only -- one per (column, row phase q), repeating with the interior's period along the
// column (device_plan.cpp plan_direct: strips_ok).  So two ADJACENT border columns are two chains over the SAME samples with
// different coefficients: the halves of a register pair, a tap of both one v_pk_mul_f32 (the sample broadcast to both halves by
// op_sel, the coefficient pair (set(x, q), set(x + 1, q))[ly][lx] from an aligned SGPR pair) and one v_pk_add_f32 -- each half the
// reference's chain bit for bit: taps in (ly, lx) order, multiply and add un-fused (ref :570-579).
//
//   * lane = period-row j of a block of 64; a sweep = (row phase q, group of four adjacent columns = two chain pairs): the lane walks
//     the fs kernel rows, per kernel row its fs samples of tile row (start_y[q] - min_sy) + lane + ly as ds_read_b128 (rows a pitch
//     of 12 / 20 words apart: eight lanes cover all banks) and the row's 2 fs coefficient pairs as scalar loads; the four samples of
//     a sweep are adjacent in the lane's output row and leave as one store;
//   * a workgroup = four waves = one block of 64 period-rows of one side; it stages the block's (64 + fs) x fs source samples as fp32
//     in LDS once, its waves take the side's sweeps in turn.
// Every tap is executed (no zero-tap elision): exact for every sample, float infinities and NaNs included.
// The kernels this replaces ran the columns of filter size 17 at a tenth of the VALU peak (ewa_colstrip_kernel: one LDS read per tap;
// C3 0.24 ms per step of 32 frames for 2.4 G operations; profiles/round5/colpair_ab.log).
#include "device_common.hpp"
#include "knobs.h"

#pragma clang fp contract(off)

namespace jinc {
namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int FS>
struct ColPairCfg {
    static constexpr int kLanes = 64;                          // period-rows per block
    static constexpr int kRows = kLanes + FS;                  // + (fs - 1) halo + 1 phase spread
    static constexpr int kQuads = (FS + 3) / 4;                // 16-byte reads per lane and kernel row
    static constexpr int kPitch = 4 * (kQuads | 1);            // words per tile row, an odd number of quads: no bank conflict between the lanes' rows
    static constexpr int kThreads = 256;
    static constexpr int kCoeffRow = 4 * FS;                   // floats per kernel row of a column group: fs x (pair 0, pair 1)
};

#define JINC_CP_LO(T, W, C) "v_pk_mul_f32 " T ", " W ", " C " op_sel_hi:[0,1]\n\t"               /* the sample = the low half of the window pair */
#define JINC_CP_HI(T, W, C) "v_pk_mul_f32 " T ", " W ", " C " op_sel:[1,0] op_sel_hi:[1,1]\n\t" /* ... the high half */
#define JINC_CP_ADD "v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3\n\t"

// Four taps (samples wa.x, wa.y, wb.x, wb.y) onto both chain pairs; c[2 t] / c[2 t + 1] = the coefficient pairs of tap t for the pairs.
__device__ __forceinline__ void colpair_taps4(f32x2& a0, f32x2& a1, f32x2 wa, f32x2 wb, f32x2 c0, f32x2 c1, f32x2 c2, f32x2 c3, f32x2 c4, f32x2 c5,
                                              f32x2 c6, f32x2 c7) {
    f32x2 t0, t1;
    asm(JINC_CP_LO("%2", "%4", "%6") JINC_CP_LO("%3", "%4", "%7") JINC_CP_ADD
        JINC_CP_HI("%2", "%4", "%8") JINC_CP_HI("%3", "%4", "%9") JINC_CP_ADD
        JINC_CP_LO("%2", "%5", "%10") JINC_CP_LO("%3", "%5", "%11") JINC_CP_ADD
        JINC_CP_HI("%2", "%5", "%12") JINC_CP_HI("%3", "%5", "%13")
        "v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3"
        : "+v"(a0), "+v"(a1), "=&v"(t0), "=&v"(t1)
        : "v"(wa), "v"(wb), "s"(c0), "s"(c1), "s"(c2), "s"(c3), "s"(c4), "s"(c5), "s"(c6), "s"(c7));
}
// Three taps (wa.x, wa.y, wb.x).
__device__ __forceinline__ void colpair_taps3(f32x2& a0, f32x2& a1, f32x2 wa, f32x2 wb, f32x2 c0, f32x2 c1, f32x2 c2, f32x2 c3, f32x2 c4, f32x2 c5) {
    f32x2 t0, t1;
    asm(JINC_CP_LO("%2", "%4", "%6") JINC_CP_LO("%3", "%4", "%7") JINC_CP_ADD
        JINC_CP_HI("%2", "%4", "%8") JINC_CP_HI("%3", "%4", "%9") JINC_CP_ADD
        JINC_CP_LO("%2", "%5", "%10") JINC_CP_LO("%3", "%5", "%11")
        "v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3"
        : "+v"(a0), "+v"(a1), "=&v"(t0), "=&v"(t1)
        : "v"(wa), "v"(wb), "s"(c0), "s"(c1), "s"(c2), "s"(c3), "s"(c4), "s"(c5));
}
// One tap (wa.x).
__device__ __forceinline__ void colpair_taps1(f32x2& a0, f32x2& a1, f32x2 wa, f32x2 c0, f32x2 c1) {
    f32x2 t0, t1;
    asm(JINC_CP_LO("%2", "%4", "%5") JINC_CP_LO("%3", "%4", "%6")
        "v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3"
        : "+v"(a0), "+v"(a1), "=&v"(t0), "=&v"(t1)
        : "v"(wa), "s"(c0), "s"(c1));
}
#undef JINC_CP_LO
#undef JINC_CP_HI
#undef JINC_CP_ADD

template <typename T, int FS>
__global__ __launch_bounds__(256) void ewa_colpair_kernel(const ColPairArgs a, const PlaneIO io) {
    using Cfg = ColPairCfg<FS>;
    static_assert(FS % 4 == 1 || FS % 4 == 3, "odd filter sizes: groups of four taps and one of one or three");
    __shared__ __attribute__((aligned(16))) float tile[Cfg::kRows * Cfg::kPitch];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int side = blockIdx.y;
    const int n = a.n[side];
    if (n <= 0) return;  // workgroup-uniform
    const int j0 = static_cast<int>(blockIdx.x) * Cfg::kLanes;
    const size_t frame = blockIdx.z;
    {   // stage word [r][c] = source(row min_sy + j0 + r, column origin + c), rows clamped to the plane like every kernel's halo (the
        // columns lie inside it: host); all loads in front of the LDS writes
        const char* sbase = static_cast<const char*>(io.src) + frame * io.src_frame_stride;
        constexpr int kElems = Cfg::kRows * FS, kPer = (kElems + Cfg::kThreads - 1) / Cfg::kThreads;
        T staged[kPer];
        const int gx0 = a.origin[side], gy0 = a.min_sy + j0;
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            const int e = min(static_cast<int>(threadIdx.x) + Cfg::kThreads * i, kElems - 1);
            const int r = e / FS, c = e - r * FS;
            int gy = gy0 + r;
            gy = gy < a.src_h ? gy : a.src_h - 1;
            staged[i] = reinterpret_cast<const T*>(sbase + static_cast<size_t>(gy) * io.src_pitch)[gx0 + c];
        }
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            const int e = static_cast<int>(threadIdx.x) + Cfg::kThreads * i;
            if (e < kElems) tile[(e / FS) * Cfg::kPitch + (e % FS)] = to_float(staged[i]);
        }
    }
    __syncthreads();
    const int j = j0 + lane;           // the lane's period-row
    const bool live = j < a.nj;        // (no barrier below; dead lanes read their tile rows all the same: inside the tile)
    const BufferRsrc drsrc = make_rsrc(static_cast<char*>(io.dst) + frame * io.dst_frame_stride, static_cast<uint32_t>(io.dst_pitch) * a.dst_h);
    const int ngroups = (n + 3) >> 2;  // column groups of four (two chain pairs); the last may hold fewer columns: zero coefficients
    const int nsweeps = a.py * ngroups;
    for (int sweep = wave; sweep < nsweeps; sweep += 4) {  // wave-uniform
        const int g = sweep / a.py, q = sweep - g * a.py;
        const float* rowbase = tile + ((a.start_y[q] - a.min_sy) + lane) * Cfg::kPitch;
        const JINC_CONSTANT f32x2* cs = (const JINC_CONSTANT f32x2*)(a.coeffs) +
                                        static_cast<size_t>((side * ColPairArgs::kMaxGroups + g) * a.py + q) * (FS * Cfg::kCoeffRow / 2);
        f32x2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
#pragma nounroll
        for (int ly = 0; ly < FS; ++ly) {  // (rolled: a kernel row's 2 fs coefficient pairs are most of the SGPR file)
            f32x2 w[2 * Cfg::kQuads];
            const f32x4* row = reinterpret_cast<const f32x4*>(rowbase + ly * Cfg::kPitch);
#pragma unroll
            for (int m = 0; m < Cfg::kQuads; ++m) {
                const f32x4 v = row[m];
                w[2 * m] = f32x2{v.x, v.y};
                w[2 * m + 1] = f32x2{v.z, v.w};
            }
            const JINC_CONSTANT f32x2* c = cs + ly * (Cfg::kCoeffRow / 2);
#pragma unroll
            for (int t = 0; t + 4 <= FS; t += 4)
                colpair_taps4(acc0, acc1, w[t / 2], w[t / 2 + 1], c[2 * t], c[2 * t + 1], c[2 * t + 2], c[2 * t + 3], c[2 * t + 4], c[2 * t + 5],
                              c[2 * t + 6], c[2 * t + 7]);
            constexpr int t = FS / 4 * 4;
            if constexpr (FS % 4 == 3)
                colpair_taps3(acc0, acc1, w[t / 2], w[t / 2 + 1], c[2 * t], c[2 * t + 1], c[2 * t + 2], c[2 * t + 3], c[2 * t + 4], c[2 * t + 5]);
            else
                colpair_taps1(acc0, acc1, w[t / 2], c[2 * t], c[2 * t + 1]);
        }
        if (!live) continue;
        const int nk = n - 4 * g < 4 ? n - 4 * g : 4;  // wave-uniform
        const uint32_t voff = static_cast<uint32_t>(a.iy0 + a.py * j + q) * static_cast<uint32_t>(io.dst_pitch) +
                              static_cast<uint32_t>(a.x0[side] + 4 * g) * static_cast<uint32_t>(sizeof(T));
        const float r[4] = {acc0.x, acc0.y, acc1.x, acc1.y};
        if (nk == 4) {
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            if constexpr (std::is_same_v<T, uint8_t>) {
                uint32_t v = __builtin_amdgcn_cvt_pk_u8_f32(r[0], 0u, 0u);
                v = __builtin_amdgcn_cvt_pk_u8_f32(r[1], 1u, v);
                v = __builtin_amdgcn_cvt_pk_u8_f32(r[2], 2u, v);
                v = __builtin_amdgcn_cvt_pk_u8_f32(r[3], 3u, v);
                __builtin_amdgcn_raw_buffer_store_b32(v, drsrc, voff, 0, 0);
            } else if constexpr (std::is_same_v<T, uint16_t>) {
                const u32x2 v = {round_pair_u16(r[0], r[1], io.peak), round_pair_u16(r[2], r[3], io.peak)};
                __builtin_amdgcn_raw_buffer_store_b64(v, drsrc, voff, 0, 0);
            } else {
                // (8-byte stores: a 16-byte store that is not 16-byte aligned does not put its dwords where they belong on this part)
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, acc0), drsrc, voff, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, acc1), drsrc, voff + 8u, 0, 0);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (k < nk) store_sample_buf<T>(drsrc, voff + static_cast<uint32_t>(k * sizeof(T)), 0, r[k], io.peak);
        }
    }
}

template <typename T, int FS>
int launch_colpair_fs(const ColPairArgs& a, const PlaneIO& io, hipStream_t stream) {
    dim3 grid(static_cast<unsigned>((a.nj + ColPairCfg<FS>::kLanes - 1) / ColPairCfg<FS>::kLanes), 2u, static_cast<unsigned>(io.nframes));
    hipLaunchKernelGGL((ewa_colpair_kernel<T, FS>), grid, dim3(ColPairCfg<FS>::kThreads), 0, stream, a, io);
    return static_cast<int>(hipGetLastError());
}

template <typename T>
int launch_colpair_t(const ColPairArgs& a, const PlaneIO& io, hipStream_t stream) {
    switch (a.fs) {
        case 7: return launch_colpair_fs<T, 7>(a, io, stream);
        case 9: return launch_colpair_fs<T, 9>(a, io, stream);
        case 11: return launch_colpair_fs<T, 11>(a, io, stream);
        case 13: return launch_colpair_fs<T, 13>(a, io, stream);
        case 15: return launch_colpair_fs<T, 15>(a, io, stream);
        case 17: return launch_colpair_fs<T, 17>(a, io, stream);
        default: return static_cast<int>(hipErrorInvalidValue);
    }
}

}  // namespace

bool colpair_supported(int fs, int py, int sy, int spread) { return fs >= 7 && fs <= 17 && fs % 2 == 1 && py >= 1 && py <= 4 && sy == 1 && spread >= 0 && spread <= 1; }

int launch_colpair(const ColPairArgs& args, const PlaneIO& io, void* stream) {
    if (args.nj <= 0 || io.nframes <= 0 || (args.n[0] <= 0 && args.n[1] <= 0)) return 0;
    if (!args.coeffs || !colpair_supported(args.fs, args.py, 1, args.spread) || args.n[0] > 4 * ColPairArgs::kMaxGroups || args.n[1] > 4 * ColPairArgs::kMaxGroups)
        return static_cast<int>(hipErrorInvalidValue);
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (io.sample_bytes) {
        case 1: return launch_colpair_t<uint8_t>(args, io, s);
        case 2: return launch_colpair_t<uint16_t>(args, io, s);
        default: return launch_colpair_t<float>(args, io, s);
    }
}

}  // namespace jinc
