// kernel_framelane_sub.hip -- ewa_framelane_sub_kernel: the sliding-window frame-lane form for groups of FEWER than 64 frames.
//
// The frame-lane kernels (kernel_framelane.hip) put the 64 lanes of a wave on the same output pixel of 64 frames; their rate is
// proportional to the lanes that carry a frame, and a host at look-ahead 32 hands over groups of 16 (1.37x, 16 frames per launch:
// 94 Gpix/s, 0.12 of the VALU peak, against 376 with all lanes filled).  Here a wave is G sub-groups of 64 / G lanes: sub-group
// g walks output row G * strip + g of the tile, its lanes are the frames.  The plan is separable in its window origins
// (col_start depends on x only), so the G rows of a wave advance their windows through the same source columns in the same
// steps: the walk by window origin, the ring phase and every loop bound stay wave-uniform, exactly as in
// ewa_framelane_win_kernel.  What differs between the sub-groups is per-lane data: the source row the window starts at (an LDS
// offset) and the coefficient set of the pixel, which cannot be a scalar operand any more.  The lanes of a sub-group need the SAME
// coefficients, though: every row of 16 lanes keeps ONE copy of the set, four floats per lane (one 16-byte vector load per lane
// and pixel), and the multiply of a tap reads its coefficient from the lane that holds it through a DPP operand (row_newbcast; by
// quads for sub-groups of 8 or 4 frames) -- see SubSet.  Each lane still owns one output sample's whole chain in (ly, lx) order
// with un-fused multiply and add.  DESIGN.md section 4.6 has the measurements of the forms tried on the way.
#include <algorithm>

#include "device_common.hpp"

#pragma clang fp contract(off)

namespace jinc {
namespace {

#include "kernel_framelane_common.inc"

// LDS bytes per source position: the group's frames side by side + 4 (the staging writes of neighbouring columns fall on
// different banks), as kFrameLanePosBytes for 64 frames.
constexpr int kSubPosBytes(int frames, size_t sample_bytes) { return static_cast<int>(frames * sample_bytes + 4); }

template <typename T, int FS, int PS>
__device__ __forceinline__ void sub_load_col(float (&col)[FS], const char* p) {
#pragma unroll
    for (int ly = 0; ly < FS; ++ly) col[ly] = to_float(*reinterpret_cast<const T*>(p + ly * PS));
}

template <int E>
__device__ __forceinline__ float f4_elem(const float4& v) {
    if constexpr (E == 0) return v.x;
    else if constexpr (E == 1) return v.y;
    else if constexpr (E == 2) return v.z;
    else return v.w;
}

// One tap of a chain: t = w * (the value lane L of the caller's lane group holds in c), and the PREVIOUS tap's product joins the
// chain behind it (acc += tprev), so that no instruction waits for the one in front of it.  ROW: lane L of the caller's row of
// 16 lanes (DPP row_newbcast), otherwise lane L of its quad (DPP quad_perm): a modifier on the multiply's first operand, no
// instruction of its own.  (As a __builtin_amdgcn_mov_dpp the compiler keeps a v_mov_b32_dpp per tap; with the add left to the
// compiler it puts an s_nop between every product and its add.)  The hardware wants two wait states between a VALU write of c
// and this read; c comes straight from a vector load here, and tests/test_build.py checks the listing for VALU writes in front
// of every DPP read.
template <bool ROW, int L>
__device__ __forceinline__ float tap_first(float c, float w) {
    float t;
    if constexpr (ROW)
        asm("v_mul_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=&v"(t) : "v"(c), "v"(w), "n"(L));
    else
        asm("v_mul_f32_dpp %0, %1, %2 quad_perm:[%3,%3,%3,%3] row_mask:0xf bank_mask:0xf" : "=&v"(t) : "v"(c), "v"(w), "n"(L));
    return t;
}
template <bool ROW, int L>
__device__ __forceinline__ float tap_next(float& acc, float tprev, float c, float w) {
    float t;
    if constexpr (ROW)
        asm("v_mul_f32_dpp %0, %2, %3 row_newbcast:%4 row_mask:0xf bank_mask:0xf\n\tv_add_f32 %1, %1, %5"
            : "=&v"(t), "+v"(acc)
            : "v"(c), "v"(w), "n"(L), "v"(tprev));
    else
        asm("v_mul_f32_dpp %0, %2, %3 quad_perm:[%4,%4,%4,%4] row_mask:0xf bank_mask:0xf\n\tv_add_f32 %1, %1, %5"
            : "=&v"(t), "+v"(acc)
            : "v"(c), "v"(w), "n"(L), "v"(tprev));
    return t;
}

// A pixel's coefficient set, spread over the lanes that share it.  The lanes of a sub-group compute the same pixel, so they
// need the same fs x fs coefficients; one copy per lane through vector loads costs 64 lanes x fs x fs x 4 bytes of vector-memory
// traffic per pixel step against 2 fs x fs arithmetic instructions, and the texture path bounds the kernel (measured: a third
// of the 64-frame form's rate per filled lane).  Instead every row of 16 lanes (sub-groups of 16 or 32 frames) holds ONE copy,
// four floats per lane -- one 16-byte load per lane and pixel for filter sizes up to 8 -- and the multiply of tap i reads its
// coefficient from lane (i % 64) / 4 of the row; sub-groups of 8 frames share by quads (16 floats per chunk).  Chunk k of
// lane l holds floats (4 * LPC) k + 4 l .. + 3 of the set (rows padded to FSP floats); the last chunk may run past the set
// (the table has slack, device_plan.cpp upload_table).
template <int FS, bool ROW>
struct SubSet {
    static constexpr int FSP = padded_row(FS);
    static constexpr int LPC = ROW ? 16 : 4;  // lanes per copy
    static constexpr int NK = (FS * FSP + 4 * LPC - 1) / (4 * LPC);
    float4 c[NK];
    __device__ __forceinline__ void load(const char* cl) {  // cl: the set's address + 16 * (lane % LPC)
#pragma unroll
        for (int k = 0; k < NK; ++k) c[k] = *reinterpret_cast<const float4*>(cl + 16 * LPC * k);
    }
};

template <int FS, int PH, bool ROW, int I = 0>
__device__ __forceinline__ void sub_taps(float& acc, float tprev, const float (&w)[FS][FS], const SubSet<FS, ROW>& set) {
    if constexpr (I < FS * FS) {
        constexpr int ly = I / FS, lx = I % FS, i = ly * SubSet<FS, ROW>::FSP + lx, per = 4 * SubSet<FS, ROW>::LPC;
        const float c = f4_elem<i % 4>(set.c[i / per]), wv = w[(PH + lx) % FS][ly];
        float t;
        if constexpr (I == 0)
            t = tap_first<ROW, (i % per) / 4>(c, wv);
        else
            t = tap_next<ROW, (i % per) / 4>(acc, tprev, c, wv);
        sub_taps<FS, PH, ROW, I + 1>(acc, t, w, set);
    } else {
        asm("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(tprev));  // the last tap's product
    }
}

// The chain of one pixel in (ly, lx) order: window w[slot][ly] at ring phase PH, coefficients in `cur`; the next pixel's set
// (cl_next: its address + the lane's 16 bytes) is requested into `nxt` first, a whole pixel ahead of its use.
template <int FS, int PH, bool ROW>
__device__ __forceinline__ float sub_mac(const float (&w)[FS][FS], const SubSet<FS, ROW>& cur, SubSet<FS, ROW>& nxt, const char* cl_next) {
    nxt.load(cl_next);
    float acc = 0.f;  // (the chain starts as 0 + the first product, as the reference's does)
    sub_taps<FS, PH, ROW>(acc, 0.f, w, cur);
    return acc;
}

// One window origin `s` of the wave's strips at ring phase I (see fl_win_step in kernel_framelane.hip): the column completing
// the windows, then every pixel with this origin.  sets_row: the lane's row of the tile's set table (LDS); set_a / set_b: the
// coefficients of the strip's even / odd pixels; set_next: the byte offset of pixel j + 1's set, read from LDS one pixel earlier still.
template <typename T, int FS, int PS, bool ROW, int I>
__device__ __forceinline__ bool sub_step(float (&w)[FS][FS], const char*& pc, int pc_step, int s, int s_last, int& j, int npix, int csv,
                                         const int* sets_row, uint32_t& set_next, SubSet<FS, ROW>& set_a, SubSet<FS, ROW>& set_b, const char* cbase, float (&res)[4],
                                         char* drow, bool lane_on, bool vec_ok, float peak) {
    constexpr int SB = static_cast<int>(sizeof(T));
    constexpr uint32_t kSetBytes = FS * padded_row(FS) * 4;
    if (s > s_last) return false;  // wave-uniform
    sub_load_col<T, FS, PS>(w[(I + FS - 1) % FS], pc);  // column s + FS - 1
    pc += pc_step;
    while (j < npix && __builtin_amdgcn_readlane(csv, j) == s) {
        const uint32_t snext = set_next;  // pixel j + 1's set (the strip's last pixel: its own, fetched and dropped)
        set_next = static_cast<uint32_t>(sets_row[min(j + 2, npix - 1)]) * kSetBytes;
        // (even pixels of the strip compute from set_a and fill set_b, odd pixels the other way round: no register copies)
        const float acc = (j & 1) ? sub_mac<FS, I, ROW>(w, set_b, set_a, cbase + snext) : sub_mac<FS, I, ROW>(w, set_a, set_b, cbase + snext);
        const int q = j & 3;
        const bool flush = q == 3 || j == npix - 1;  // wave-uniform
        if constexpr (std::is_same_v<T, uint8_t>) {
            uint32_t& pk = reinterpret_cast<uint32_t&>(res[0]);
            pk = __builtin_amdgcn_cvt_pk_u8_f32(acc, static_cast<uint32_t>(q), pk);
            if (flush && lane_on) {
                char* d = drow + static_cast<size_t>(j & ~3);
                if (vec_ok && q == 3) {
                    *reinterpret_cast<uint32_t*>(d) = pk;
                } else {
#pragma unroll
                    for (int xx = 0; xx < 4; ++xx)
                        if (xx <= q) reinterpret_cast<uint8_t*>(d)[xx] = static_cast<uint8_t>(pk >> (8 * xx));
                }
            }
        } else {
            switch (q) {
                case 0: res[0] = acc; break;
                case 1: res[1] = acc; break;
                case 2: res[2] = acc; break;
                default: res[3] = acc; break;
            }
            if (flush && lane_on) fl_store4<T>(drow + static_cast<size_t>(j & ~3) * SB, res, q + 1, vec_ok && q == 3, peak);
        }
        ++j;
    }
    return true;
}

template <typename T, int FS, int G>
__global__ __launch_bounds__(512, 4) void ewa_framelane_sub_kernel(const FrameLaneArgs a) {
    static_assert(G == 2 || G == 4 || G == 8 || G == 16, "sub-groups per wave");
    constexpr int FPG = 64 / G;  // frames per workgroup = lanes per sub-group
    extern __shared__ __attribute__((aligned(16))) char fl_smem[];
    int* cs = reinterpret_cast<int*>(fl_smem);
    int* rs = cs + kFrameLaneMaxTile;
    int* sets = rs + kFrameLaneMaxTile;
    char* tile = fl_smem + kFrameLaneTableBytes(a.ty_shift);
    constexpr int PS = kSubPosBytes(FPG, sizeof(T));
    constexpr int SB = static_cast<int>(sizeof(T));
    constexpr uint32_t kSetBytes = FS * padded_row(FS) * 4;
    const DevicePlan& p = a.plan;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    FlTile t;
    if (!fl_locate<FPG>(a, FS, t)) return;  // whole block, before any barrier
    fl_tables(a, t, cs, rs, sets);
    const int nstrips = t.by1 - t.by0 + 1, npix = t.bx1 - t.bx0 + 1;
    const int nsg = (nstrips + G - 1) / G;  // strip groups: G output rows per wave and pass
    // (No L2 prefetch of the next strip group's sets, as the 64-frame forms have it: the sets already come through the vector-memory
    // path, which is the busy one here -- 55 % of the kernel's cycles with the prefetch, whose requests doubled the traffic for an L2
    // hit rate that is 91 % anyway; without it 1.37x at 16 frames 204 -> 247 Gpix/s, 8 frames 128 -> 168; round4/fl_sub_prefetch_ab.log.)
    const int thp = t.th | 1;  // column-major positions, odd column pitch (as ewa_framelane_win_kernel)
    constexpr int UF = FPG >= 16 ? FPG / 8 : 1;  // frames a wave of an 8-wave workgroup owns in a full group
    fl_stage<T, PS, UF, 16 / UF>(a, t, tile, 1, thp, lane, wave, nwaves);
    __syncthreads();

    const int grp = lane / FPG, fl = lane % FPG;
    // (lanes without a frame or a row stay active: the strip's origins live one pixel per lane and are read with v_readlane)
    const bool frame_on = fl < t.nfg;
    char* dframe = static_cast<char*>(a.io.dst) + static_cast<size_t>(t.f0 + (frame_on ? fl : 0)) * a.io.dst_frame_stride;
    const char* lds_lane = tile + fl * SB;
    constexpr bool ROW = FPG >= 16;  // a row of 16 lanes lies inside one sub-group
    const char* cbase = reinterpret_cast<const char*>(p.coeffs) + 16 * (lane & (SubSet<FS, ROW>::LPC - 1));  // the lane's part of a set: SubSet
    const bool vec_ok = (a.vec_store_ok & 1) && ((t.bx0 & 3) == 0);
    for (int sg = wave; sg < nsg; sg += nwaves) {
        const int r = min(sg * G + grp, nstrips - 1);
        const bool lane_on = frame_on && sg * G + grp < nstrips;
        const int sy = rs[r];
        const int pj = min(lane & 31, npix - 1);
        const int csv = cs[pj];  // lane j: window origin of the strips' pixel j
        const int* sets_row = sets + r * kFrameLaneMaxTile;
        SubSet<FS, ROW> set_a, set_b;
        set_a.load(cbase + static_cast<uint32_t>(sets_row[0]) * kSetBytes);
        uint32_t set_next = static_cast<uint32_t>(sets_row[min(1, npix - 1)]) * kSetBytes;
        const char* lrow = lds_lane + (sy - t.ty0) * PS;
        char* drow = dframe + static_cast<size_t>(t.by0 + r) * a.io.dst_pitch + static_cast<size_t>(t.bx0) * SB;
        float w[FS][FS];  // w[slot][ly]; source column c of the strip lives in slot (c - s_first) % FS
        const int s_first = __builtin_amdgcn_readlane(csv, 0), s_last = __builtin_amdgcn_readlane(csv, npix - 1);
        const char* pc = lrow + (s_first - t.tx0) * thp * PS;  // next column to load
#pragma unroll
        for (int i = 0; i < FS - 1; ++i) {
            sub_load_col<T, FS, PS>(w[i], pc);
            pc += thp * PS;
        }
        int j = 0;
        float res[4] = {0.f, 0.f, 0.f, 0.f};
        static_assert(FS <= 9, "the step list below has nine entries");
        for (int s0 = s_first; s0 <= s_last; s0 += FS) {
#define JINC_SUB_STEP(I)                                                                                                          \
    if constexpr (I < FS) {                                                                                                        \
        if (!sub_step<T, FS, PS, ROW, (I < FS ? I : 0)>(w, pc, thp * PS, s0 + I, s_last, j, npix, csv, sets_row, set_next, set_a, set_b, cbase, res, \
                                                    drow, lane_on, vec_ok, a.io.peak))                                                    \
            break;                                                                                                                 \
    }
            JINC_SUB_STEP(0) JINC_SUB_STEP(1) JINC_SUB_STEP(2) JINC_SUB_STEP(3) JINC_SUB_STEP(4) JINC_SUB_STEP(5) JINC_SUB_STEP(6)
            JINC_SUB_STEP(7) JINC_SUB_STEP(8)
#undef JINC_SUB_STEP
        }
    }
}

template <typename T, int FS, int G>
int launch_sub(const FrameLaneArgs& a, hipStream_t stream) {
    constexpr int FPG = 64 / G;
    // the tile configuration is the 64-frame form's (framelane_configure); a position holds FPG frames here, and a workgroup has
    // as many waves as the tile has groups of G rows (at most 8: 128 registers per lane, two workgroups per CU)
    const int table_bytes = kFrameLaneTableBytes(a.ty_shift);
    const int positions = (a.lds_bytes - table_bytes) / kFrameLanePosBytes(sizeof(T));
    const int lds_bytes = table_bytes + positions * kSubPosBytes(FPG, sizeof(T));
    const int threads = 64 * std::min(8, std::max(1, (1 << a.ty_shift) / G));
    const int ntiles = a.block_begin[4];
    dim3 grid(static_cast<unsigned>((ntiles + 7) / 8) * 8u, static_cast<unsigned>((a.io.nframes + FPG - 1) / FPG), 1);
    dim3 block(static_cast<unsigned>(threads), 1, 1);
    hipLaunchKernelGGL((ewa_framelane_sub_kernel<T, FS, G>), grid, block, static_cast<size_t>(lds_bytes), stream, a);
    return static_cast<int>(hipGetLastError());
}

template <typename T, int FS>
int launch_sub_g(const FrameLaneArgs& a, hipStream_t stream) {
    switch (a.subgroups) {
        case 2: return launch_sub<T, FS, 2>(a, stream);
        case 4: return launch_sub<T, FS, 4>(a, stream);
        case 8:
            if constexpr (FS != 9) return launch_sub<T, FS, 8>(a, stream);
            return static_cast<int>(hipErrorInvalidValue);
        case 16:
            if constexpr (FS != 9) return launch_sub<T, FS, 16>(a, stream);
            return static_cast<int>(hipErrorInvalidValue);
        default: return static_cast<int>(hipErrorInvalidValue);
    }
}

template <typename T>
int launch_sub_fs(const FrameLaneArgs& a, hipStream_t stream) {
    switch (a.plan.fs) {
        case 5: return launch_sub_g<T, 5>(a, stream);
        case 7: return launch_sub_g<T, 7>(a, stream);
        case 8: return launch_sub_g<T, 8>(a, stream);
        case 9: return launch_sub_g<T, 9>(a, stream);
        default: return static_cast<int>(hipErrorInvalidValue);
    }
}

}  // namespace

bool framelane_sub_supported(int fs, int subgroups, int ty_shift) {
    // (filter size 9 with 8 or 16 sub-groups: the quad-shared coefficient chunks, twice over, do not fit 128 registers beside the window)
    return (fs == 5 || fs == 7 || fs == 8 || fs == 9) && (subgroups == 2 || subgroups == 4 || ((subgroups == 8 || subgroups == 16) && fs != 9)) &&
           (1 << ty_shift) >= subgroups;
}

int launch_framelane_sub(const FrameLaneArgs& args, void* stream) {
    if (args.block_begin[4] <= 0 || args.io.nframes <= 0) return 0;
    if (!framelane_sub_supported(args.plan.fs, args.subgroups, args.ty_shift)) return static_cast<int>(hipErrorInvalidValue);
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (args.io.sample_bytes) {
        case 1: return launch_sub_fs<uint8_t>(args, s);
        case 2: return launch_sub_fs<uint16_t>(args, s);
        default: return launch_sub_fs<float>(args, s);
    }
}

}  // namespace jinc
