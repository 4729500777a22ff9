// kernel_framelane_pair.hip -- ewa_framelane_pair_kernel: the sliding-window frame-lane kernel with TWO frames per lane.
//
// kernel_framelane.hip makes the 64 lanes of a wave the same output pixel of 64 frames, so that the pixel's coefficient
// set is wave-uniform (scalar loads into SGPRs) for any plan.  What holds it back on plans without phase structure is
// the per-pixel cost that does NOT scale with the frames: one 224-byte coefficient set through the scalar cache per
// pixel and wave (a miss for practically every line: 87 376 sets; measured: the scalar data caches are busy for the whole
// kernel and push back in 30 % of their cycles; with all sets made equal the same kernel runs 25 % faster), two
// v_readlane, ~32 scalar instructions.  Here a workgroup takes 128 frames and a lane owns the two adjacent frames 2l and
// 2l + 1 of the group:
//   * every per-pixel cost above is paid once per 128 frames instead of once per 64;
//   * the two chains of a lane are independent and use the same coefficient, so each tap is ONE v_pk_mul_f32 (coefficient
//     broadcast from an SGPR by op_sel) and ONE v_pk_add_f32: two exact IEEE products / sums per instruction, each half
//     the scalar chain bit for bit (nothing fused, nothing reassociated); the packed forms sustain more un-fused
//     operations per clock than v_mul_f32 / v_add_f32 (profiles/probes/valu_probe.hip: 63 vs 52 Tops/s at 4 waves/SIMD);
//   * the two frames' samples of a source position are adjacent in LDS: one ds_read per window sample of both frames.
// Price: two fs x fs windows per lane (98 registers for fs 7: 4 waves per SIMD instead of 8) and 128 frames of the
// tile's footprint in LDS (tiles of 32 x 16 pixels instead of 32 x 32: more halo).  Filter sizes 5 and 7.
#include <atomic>

#include "device_common.hpp"

#pragma clang fp contract(off)

namespace jinc {
namespace {

#include "kernel_framelane_common.inc"

typedef float f32x2 __attribute__((ext_vector_type(2)));

// The two adjacent frames' samples of one source position -> (frame 2l, frame 2l + 1) as floats.
template <typename T>
__device__ __forceinline__ f32x2 flp_load(const char* p) {
    f32x2 r;
    if constexpr (std::is_same_v<T, uint8_t>) {
        const uint32_t v = *reinterpret_cast<const uint16_t*>(p);
        r.x = static_cast<float>(v & 0xffu);  // v_cvt_f32_ubyte0
        r.y = static_cast<float>(v >> 8);     // v_cvt_f32_ubyte1
    } else if constexpr (std::is_same_v<T, uint16_t>) {
        const uint32_t v = *reinterpret_cast<const uint32_t*>(p);
        r.x = static_cast<float>(v & 0xffffu);
        r.y = static_cast<float>(v >> 16);
    } else {
        r = *reinterpret_cast<const f32x2*>(p);
    }
    return r;
}

// One kernel row of both chains: acc += w[lx] * c[lx] for lx = 0 .. FS-1 in order, each tap v_pk_mul_f32 (coefficient =
// low or high half of an aligned SGPR pair, broadcast to both halves by op_sel) followed by v_pk_add_f32.  Written as one
// asm statement per row: the compiler otherwise materialises every (c, c) splat as an SGPR pair of its own, and pads
// every asm statement with an s_nop.  Register-only VALU, interlocked by hardware.
__device__ __forceinline__ void flp_row7(f32x2& acc, f32x2 w0, f32x2 w1, f32x2 w2, f32x2 w3, f32x2 w4, f32x2 w5, f32x2 w6, f32x2 p01,
                                         f32x2 p23, f32x2 p45, f32x2 p6x) {
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
        : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(w4), "v"(w5), "v"(w6), "s"(p01), "s"(p23), "s"(p45), "s"(p6x));
}
__device__ __forceinline__ void flp_row5(f32x2& acc, f32x2 w0, f32x2 w1, f32x2 w2, f32x2 w3, f32x2 w4, f32x2 p01, f32x2 p23, f32x2 p4x) {
    f32x2 t;
    asm("v_pk_mul_f32 %1, %2, %7 op_sel_hi:[1,0]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %3, %7 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %4, %8 op_sel_hi:[1,0]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %5, %8 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_mul_f32 %1, %6, %9 op_sel_hi:[1,0]\n\t"
        "v_pk_add_f32 %0, %0, %1"
        : "+v"(acc), "=&v"(t)
        : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(w4), "s"(p01), "s"(p23), "s"(p4x));
}

// Both chains of one output pixel: taps in (ly, lx) order, window column of tap lx = ring slot (PH + lx) % FS.
template <int FS, int PH>
__device__ __forceinline__ f32x2 flp_mac(const f32x2 (&w)[FS][FS], const JINC_CONSTANT char* cset) {
    static_assert(FS == 5 || FS == 7, "row forms exist for filter sizes 5 and 7");
    constexpr int RP = padded_row(FS) / 2;  // aligned SGPR pairs per coefficient row
    const JINC_CONSTANT f32x2* cp = reinterpret_cast<const JINC_CONSTANT f32x2*>(cset);
    f32x2 c[FS * RP];
    // one contiguous run -> s_load_dwordx16 / x8 pieces; every piece is requested before the first multiply
#pragma unroll
    for (int i = 0; i < FS * RP; ++i) c[i] = cp[i];
#pragma unroll
    for (int i = 0; i < FS * RP; ++i) asm volatile("" : "+s"(c[i]));
    f32x2 acc = {0.f, 0.f};
#pragma unroll
    for (int ly = 0; ly < FS; ++ly) {
        if constexpr (FS == 7)
            flp_row7(acc, w[(PH + 0) % FS][ly], w[(PH + 1) % FS][ly], w[(PH + 2) % FS][ly], w[(PH + 3) % FS][ly], w[(PH + 4) % FS][ly],
                     w[(PH + 5) % FS][ly], w[(PH + 6) % FS][ly], c[ly * RP], c[ly * RP + 1], c[ly * RP + 2], c[ly * RP + 3]);
        else
            flp_row5(acc, w[(PH + 0) % FS][ly], w[(PH + 1) % FS][ly], w[(PH + 2) % FS][ly], w[(PH + 3) % FS][ly], w[(PH + 4) % FS][ly],
                     c[ly * RP], c[ly * RP + 1], c[ly * RP + 2]);
    }
    return acc;
}

template <typename T, int FS>
__device__ __forceinline__ void flp_load_col(f32x2 (&col)[FS], const char* p) {
    constexpr int PS = kFrameLanePairPosBytes(sizeof(T));
#pragma unroll
    for (int ly = 0; ly < FS; ++ly) col[ly] = flp_load<T>(p + ly * PS);
}

// Results of the strip's current group of four pixels, per frame of the lane's pair.
template <typename T>
struct FlpOut {
    float a[4], b[4];
};
template <>
struct FlpOut<uint8_t> {
    uint32_t a, b;  // four converted samples each, packed as they are computed
};

// One window origin `s` of a strip at ring phase I: the column completing its window, then every pixel with this origin
// (see fl_win_step in kernel_framelane.hip).  False: past the strip's last origin.
template <typename T, int FS, int I>
__device__ __forceinline__ bool flp_step(f32x2 (&w)[FS][FS], const char*& pc, int pc_step, int s, int s_last, int& j, int npix, int csv,
                                         int setv, const JINC_CONSTANT char* cbase, FlpOut<T>& res, uint32_t (&oba)[4], uint32_t (&obb)[4], char* drow, size_t fstride,
                                         bool on_a, bool on_b, bool vec_ok, int vec_strip, float peak) {
    constexpr int SB = static_cast<int>(sizeof(T));
    if (s > s_last) return false;  // wave-uniform
    flp_load_col<T, FS>(w[(I + FS - 1) % FS], pc);  // column s + FS - 1
    pc += pc_step;
    while (j < npix && __builtin_amdgcn_readlane(csv, j) == s) {
        const uint32_t soff = static_cast<uint32_t>(__builtin_amdgcn_readlane(setv, j));
        const f32x2 acc = flp_mac<FS, I>(w, cbase + soff);
        const int q = j & 3;
        const bool flush = q == 3 || j == npix - 1;  // wave-uniform
        if constexpr (std::is_same_v<T, uint8_t>) {
            // (byte q replaced, the others kept: bytes above q are stale until the group is complete and never stored before)
            res.a = __builtin_amdgcn_cvt_pk_u8_f32(acc.x, static_cast<uint32_t>(q), res.a);
            res.b = __builtin_amdgcn_cvt_pk_u8_f32(acc.y, static_cast<uint32_t>(q), res.b);
            if (flush) {  // a finished word is parked; 16 pixels (or the strip's tail) leave as one store per frame
                fl_park4(oba, j >> 2, res.a);
                fl_park4(obb, j >> 2, res.b);
                if ((j & 15) == 15 || j == npix - 1) {
                    char* d = drow + static_cast<size_t>(j & ~15);
                    if (on_a) fl_flush16_u8(oba, d, (j & 15) + 1, vec_strip);
                    if (on_b) fl_flush16_u8(obb, d + fstride, (j & 15) + 1, vec_strip);
                }
            }
        } else {
            switch (q) {
                case 0: res.a[0] = acc.x, res.b[0] = acc.y; break;
                case 1: res.a[1] = acc.x, res.b[1] = acc.y; break;
                case 2: res.a[2] = acc.x, res.b[2] = acc.y; break;
                default: res.a[3] = acc.x, res.b[3] = acc.y; break;
            }
            if (flush) {
                char* d = drow + static_cast<size_t>(j & ~3) * SB;
                if (on_a) fl_store4<T>(d, res.a, q + 1, vec_ok && q == 3, peak);
                if (on_b) fl_store4<T>(d + fstride, res.b, q + 1, vec_ok && q == 3, peak);
            }
        }
        ++j;
    }
    return true;
}

// 512 threads, at most 128 registers: two workgroups per CU = 4 waves per SIMD, each workgroup with up to 80 KB of LDS.
template <typename T, int FS>
__global__ __launch_bounds__(512, 4) void ewa_framelane_pair_kernel(const FrameLaneArgs a) {
    extern __shared__ __attribute__((aligned(16))) char fl_smem[];
    int* cs = reinterpret_cast<int*>(fl_smem);
    int* rs = cs + kFrameLaneMaxTile;
    int* sets = rs + kFrameLaneMaxTile;
    char* tile = fl_smem + kFrameLaneTableBytes(a.ty_shift);
    constexpr int PS = kFrameLanePairPosBytes(sizeof(T));
    constexpr int SB = static_cast<int>(sizeof(T));
    constexpr uint32_t kSetBytes = FS * padded_row(FS) * 4;
    const DevicePlan& p = a.plan;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    FlTile t;
    if (!fl_locate<kFrameLanePairFrames>(a, FS, t)) return;  // whole block, before any barrier
    fl_tables(a, t, cs, rs, sets);
    __syncthreads();

    // Coefficient sets of a strip (one output row of the tile, <= 32 pixels), requested one strip ahead with VECTOR loads,
    // one cache line per lane (kernel_framelane.hip): the scalar loads of the tap loop then hit L2.
    const int nstrips = t.by1 - t.by0 + 1, npix = t.bx1 - t.bx0 + 1;
    auto prefetch_strip = [&](int st) -> uint32_t {
        uint32_t keep = 0;
        if (st < nstrips) {
            const int set = sets[st * kFrameLaneMaxTile + min(lane & 31, npix - 1)];
            const char* sp = reinterpret_cast<const char*>(p.coeffs) + static_cast<size_t>(static_cast<uint32_t>(set) * kSetBytes);
            for (uint32_t off = static_cast<uint32_t>(lane >> 5) * 64u; off < kSetBytes + 60u; off += 128u)
                keep |= *reinterpret_cast<const uint32_t*>(sp + (off < kSetBytes - 4u ? off : kSetBytes - 4u));
        }
        return keep;
    };
    uint32_t pf_keep = prefetch_strip(wave);

    const int thp = t.th | 1;  // odd column pitch: the staging writes of neighbouring columns fall on different banks
    fl_stage<T, PS, 16, 1>(a, t, tile, 1, thp, lane, wave, nwaves);  // 8 waves, 128 frames: 16 frames per wave
    __syncthreads();
    asm volatile("" ::"v"(pf_keep));
    // Lanes without a frame (last group of the batch) stay active -- the strip tables live one pixel per LANE and are read
    // with v_readlane -- compute on unwritten LDS and store nothing.
    const bool on_a = 2 * lane < t.nfg, on_b = 2 * lane + 1 < t.nfg;
    const size_t fstride = a.io.dst_frame_stride;
    char* dframe = static_cast<char*>(a.io.dst) + static_cast<size_t>(t.f0 + (on_a ? 2 * lane : 0)) * fstride;
    const char* lds_lane = tile + lane * 2 * SB;
    const JINC_CONSTANT char* cbase = (const JINC_CONSTANT char*)(p.coeffs);
    const bool vec_ok = (a.vec_store_ok & 1) && ((t.bx0 & 3) == 0);
    const int vec_strip = (vec_ok ? 1 : 0) | (((a.vec_store_ok & 2) && (t.bx0 & 15) == 0) ? 2 : 0);  // 8-bit: dword / 16-byte stores
    for (int st = wave; st < nstrips; st += nwaves) {
        asm volatile("" ::"v"(pf_keep));
        pf_keep = prefetch_strip(st + nwaves);
        const int sy = __builtin_amdgcn_readfirstlane(rs[st]);
        const int pj = min(lane & 31, npix - 1);
        const int csv = cs[pj];  // lane j: window origin of the strip's pixel j
        const int setv = static_cast<int>(static_cast<uint32_t>(sets[st * kFrameLaneMaxTile + pj]) * kSetBytes);  // ... byte offset of its set
        const char* lrow = lds_lane + (sy - t.ty0) * PS;
        char* drow = dframe + static_cast<size_t>(t.by0 + st) * a.io.dst_pitch + static_cast<size_t>(t.bx0) * SB;
        // walked by WINDOW ORIGIN with the origin loop unrolled FS times: the ring phase, and with it every register
        // index of the tap loop, is a compile-time constant (kernel_framelane_win_body.inc)
        f32x2 w[FS][FS];  // w[slot][ly]; source column c of the strip lives in slot (c - s_first) % FS
        const int s_first = __builtin_amdgcn_readlane(csv, 0), s_last = __builtin_amdgcn_readlane(csv, npix - 1);
        const char* pc = lrow + (s_first - t.tx0) * thp * PS;  // next column to load
#pragma unroll
        for (int i = 0; i < FS - 1; ++i) {
            flp_load_col<T, FS>(w[i], pc);
            pc += thp * PS;
        }
        int j = 0;
        FlpOut<T> res = {};
        uint32_t oba[4] = {0u, 0u, 0u, 0u}, obb[4] = {0u, 0u, 0u, 0u};  // 8-bit: packed results of the current 16 pixels, per frame
        static_assert(FS <= 7, "the step list below has seven entries");
        for (int s0 = s_first; s0 <= s_last; s0 += FS) {
#define JINC_FLP_STEP(I)                                                                                                               \
    if constexpr (I < FS) {                                                                                                            \
        if (!flp_step<T, FS, (I < FS ? I : 0)>(w, pc, thp * PS, s0 + I, s_last, j, npix, csv, setv, cbase, res, oba, obb, drow, fstride, on_a, \
                                               on_b, vec_ok, vec_strip, a.io.peak))                                                                     \
            break;                                                                                                                     \
    }
            JINC_FLP_STEP(0) JINC_FLP_STEP(1) JINC_FLP_STEP(2) JINC_FLP_STEP(3) JINC_FLP_STEP(4) JINC_FLP_STEP(5) JINC_FLP_STEP(6)
#undef JINC_FLP_STEP
        }
    }
}

template <typename T, int FS>
int launch_flp(const FrameLaneArgs& a, hipStream_t stream) {
    // dynamic LDS beyond 64 KB needs the attribute once per kernel AND device (a process may drive several: jinc_batch_*)
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return static_cast<int>(hipGetLastError());
    if (dev < 0 || dev >= 64 || !attr_set[dev].load()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&ewa_framelane_pair_kernel<T, FS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                80 * 1024) != hipSuccess)
            return static_cast<int>(hipGetLastError());
        if (dev >= 0 && dev < 64) attr_set[dev].store(true);
    }
    const int ntiles = a.block_begin[4];
    dim3 grid(static_cast<unsigned>((ntiles + 7) / 8) * 8u,
              static_cast<unsigned>((a.io.nframes + kFrameLanePairFrames - 1) / kFrameLanePairFrames), 1);
    dim3 block(static_cast<unsigned>(a.threads), 1, 1);
    hipLaunchKernelGGL((ewa_framelane_pair_kernel<T, FS>), grid, block, static_cast<size_t>(a.lds_bytes), stream, a);
    return static_cast<int>(hipGetLastError());
}

template <typename T>
int launch_flp_fs(const FrameLaneArgs& a, hipStream_t stream) {
    switch (a.plan.fs) {
        case 5: return launch_flp<T, 5>(a, stream);
        case 7: return launch_flp<T, 7>(a, stream);
        default: return static_cast<int>(hipErrorInvalidValue);  // (host: framelane_pair_configure)
    }
}

}  // namespace

int launch_framelane_pair(const FrameLaneArgs& args, void* stream) {
    if (args.block_begin[4] <= 0 || args.io.nframes <= 0) return 0;
    if (!args.pair || args.lds_bytes > 80 * 1024 || args.threads > 512) return static_cast<int>(hipErrorInvalidValue);
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (args.io.sample_bytes) {
        case 1: return launch_flp_fs<uint8_t>(args, s);
        case 2: return launch_flp_fs<uint16_t>(args, s);
        default: return launch_flp_fs<float>(args, s);
    }
}

}  // namespace jinc
