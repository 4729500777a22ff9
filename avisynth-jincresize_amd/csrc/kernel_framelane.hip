// kernel_framelane.hip -- ewa_framelane_kernel: batches of independent frames, ANY plan (no phase structure needed).
//
// The kernels of kernel_periodic / kernel_quasi / kernel_direct put the 64 lanes of a wave on 64 pixels of ONE frame
// and need those pixels to share a coefficient set (phase structure) to keep the coefficients wave-uniform.  A ratio
// without structure (1280 -> 1754: 256 x 256 phase classes, practically one coefficient set per pixel of a row) leaves
// them nothing to share, and a lane-per-pixel kernel then streams 4 * fs * fs bytes of coefficients per output pixel
// out of L2 (kernel_gather: 8 % of the VALU peak).
//
// Frames of a clip are independent and use the same plan (GetFrame touches frame n only, ref JincResize.cpp:603-630),
// so this kernel turns the wave by 90 degrees: the 64 lanes of a wave are the SAME output pixel of 64 DIFFERENT
// frames.  Everything that depends on the pixel -- window origin, coefficient set, loop bounds, LDS offsets -- is
// then wave-uniform for every plan: coefficients arrive through scalar loads in SGPRs (v_mul_f32 v, s, v), 64 x less
// coefficient traffic than lane-per-pixel, no waterfall, no per-lane table look-ups; border pixels (private sets,
// shifted windows) are ordinary pixels.  Each lane still owns one output sample's whole chain in (ly, lx) order with
// un-fused multiply and add (device_common.hpp).
//
//   * grid = (output tiles, groups of 64 frames); tiles are dealt to the XCDs in contiguous runs (shared halos stay
//     in one L2).
//   * The block stages the tile's source footprint of its 64 frames in LDS in the source format, transposed to
//     [row][column][frame]: global loads run along the columns of one frame (coalesced), LDS reads run along the frames
//     (64 consecutive samples, conflict-free).  Position stride = 64 samples + 4 bytes, so the staging writes of
//     neighbouring columns fall on different banks.
//   * A wave takes units of 4 (x) by K (y) output pixels.  For one output column the K pixels below each other share
//     their window columns exactly, so each source row segment (fs samples) is read from LDS and converted ONCE and
//     feeds up to K chains (each with its own coefficient row, scalar-loaded); the register indices of the tap loop are
//     compile-time constants whatever the geometry -- only scalar addresses and wave-uniform branches are run-time.
//   * The 4 horizontally adjacent results of a lane are packed and stored with one 4-sample store per row (a lane
//     writes to its own frame: 64 different cache lines per store instruction, so bytes per store matter).
//   * Whole groups of 128 frames of filter sizes 5 and 7 run on kernel_framelane_pair.hip instead (two frames per lane);
//     the pieces both share -- tile location, per-tile tables, staging, packed stores -- are kernel_framelane_common.inc.
#include <atomic>

#include "device_common.hpp"

#pragma clang fp contract(off)

namespace jinc {
namespace {

#include "kernel_framelane_common.inc"

// One chunk of N <= 8 taps of one source row for the chains KLO..KHI of an output column (all of them hold this source
// row): samples from the LDS tile (read and converted once), one coefficient row per chain through scalar loads.
template <typename T, int K, int N, int KLO, int KHI>
__device__ __forceinline__ void fl_chunk(float (&acc)[K], const char* lds, const JINC_CONSTANT char* cbase, const uint32_t (&co)[K],
                                         uint32_t cofs) {
    constexpr int PS = kFrameLanePosBytes(sizeof(T));
    constexpr int NL = (N + 3) & ~3;  // coefficient rows are padded to multiples of 4 floats
    float seg[N];
#pragma unroll
    for (int j = 0; j < N; ++j) seg[j] = to_float(*reinterpret_cast<const T*>(lds + j * PS));
    float c[K][NL];
#pragma unroll
    for (int k = KLO; k <= KHI; ++k) {
        // one base pointer + a 32-bit byte offset per chain: s_load_dwordxN sdst, sbase, soffset (no 64-bit scalar
        // address arithmetic per row and chain)
        const JINC_CONSTANT float* cp = reinterpret_cast<const JINC_CONSTANT float*>(cbase + (co[k] + cofs));
#pragma unroll
        for (int j = 0; j < NL; ++j) c[k][j] = cp[j];
    }
    // every coefficient row of the chunk is requested before the first multiply (one wait per chunk)
#pragma unroll
    for (int k = KLO; k <= KHI; ++k)
#pragma unroll
        for (int j = 0; j < NL; ++j) asm volatile("" : "+s"(c[k][j]));
#pragma unroll
    for (int k = KLO; k <= KHI; ++k)
#pragma unroll
        for (int j = 0; j < N; ++j) acc[k] = acc[k] + seg[j] * c[k][j];
}

// `nrows` consecutive source rows, starting with row `rr`, of one output column, for the chains KLO..KHI -- exactly the
// chains whose windows hold all of these rows, so the loop has no per-row conditions: the scalar unit (one per CU,
// shared by the four SIMDs) only advances KHI-KLO+1 offsets and the counter per row.
template <typename T, int FS, int K, int KLO, int KHI>
__device__ __forceinline__ void fl_rows(float (&acc)[K], const char* lrow, int lrow_step, int rr, int nrows, const int (&sy)[K],
                                        const uint32_t (&cb)[K], const JINC_CONSTANT char* cbase, uint32_t row_bytes, int fs) {
    constexpr int PS = kFrameLanePosBytes(sizeof(T));
    uint32_t co[K];
#pragma unroll
    for (int k = 0; k < K; ++k) co[k] = cb[k] + static_cast<uint32_t>(rr - sy[k]) * row_bytes;  // (used for KLO..KHI only)
    for (int i = 0; i < nrows; ++i) {
        if constexpr (FS != 0) {
#pragma unroll
            for (int c0 = 0; c0 < FS; c0 += 8) {
                if (FS - c0 >= 8)
                    fl_chunk<T, K, 8, KLO, KHI>(acc, lrow + c0 * PS, cbase, co, c0 * 4);
                else
                    fl_chunk<T, K, (FS % 8 ? FS % 8 : 8), KLO, KHI>(acc, lrow + c0 * PS, cbase, co, c0 * 4);
            }
        } else {
            for (int c0 = 0; c0 < fs; c0 += 8) {
                const int n = fs - c0;  // wave-uniform
                if (n >= 8) {
                    fl_chunk<T, K, 8, KLO, KHI>(acc, lrow + c0 * PS, cbase, co, c0 * 4);
                } else {
                    switch (n) {
                        case 1: fl_chunk<T, K, 1, KLO, KHI>(acc, lrow + c0 * PS, cbase, co, c0 * 4); break;
                        case 2: fl_chunk<T, K, 2, KLO, KHI>(acc, lrow + c0 * PS, cbase, co, c0 * 4); break;
                        case 3: fl_chunk<T, K, 3, KLO, KHI>(acc, lrow + c0 * PS, cbase, co, c0 * 4); break;
                        case 4: fl_chunk<T, K, 4, KLO, KHI>(acc, lrow + c0 * PS, cbase, co, c0 * 4); break;
                        case 5: fl_chunk<T, K, 5, KLO, KHI>(acc, lrow + c0 * PS, cbase, co, c0 * 4); break;
                        case 6: fl_chunk<T, K, 6, KLO, KHI>(acc, lrow + c0 * PS, cbase, co, c0 * 4); break;
                        default: fl_chunk<T, K, 7, KLO, KHI>(acc, lrow + c0 * PS, cbase, co, c0 * 4); break;
                    }
                }
            }
        }
#pragma unroll
        for (int k = KLO; k <= KHI; ++k) co[k] += row_bytes;
        lrow += lrow_step;
    }
}

template <typename T, int FS, int K>
__device__ __forceinline__ void fl_rows_dispatch(int klo, int khi, float (&acc)[K], const char* lrow, int lrow_step, int rr, int nrows,
                                                 const int (&sy)[K], const uint32_t (&cb)[K], const JINC_CONSTANT char* cbase,
                                                 uint32_t row_bytes, int fs) {
    static_assert(K == 4, "the dispatch below lists the chain ranges of K = 4");
#define JINC_FL_ROWS(LO, HI) \
    case LO * 4 + HI: fl_rows<T, FS, K, LO, HI>(acc, lrow, lrow_step, rr, nrows, sy, cb, cbase, row_bytes, fs); break;
    switch (klo * 4 + khi) {
        JINC_FL_ROWS(0, 0) JINC_FL_ROWS(0, 1) JINC_FL_ROWS(0, 2) JINC_FL_ROWS(0, 3) JINC_FL_ROWS(1, 1) JINC_FL_ROWS(1, 2)
        JINC_FL_ROWS(1, 3) JINC_FL_ROWS(2, 2) JINC_FL_ROWS(2, 3) JINC_FL_ROWS(3, 3)
        default: break;
    }
#undef JINC_FL_ROWS
}

// ------------------------------------------------------------------------------------------------
// Sliding-window form, filter sizes up to 9: a wave walks along an output row; the fs x fs window of its 64 frames stays
// in registers as a ring over source columns (compile-time ring phase: FS code variants), so a pixel costs the new
// window columns (source step, 0.5 .. 2 per pixel) x fs LDS reads and conversions, fs scalar coefficient-row loads and
// the 2 * fs * fs multiply / add with static register indices -- no per-row conditions, about 25 scalar instructions.
// LDS positions are column-major here ((col, row) -> col * (th | 1) + row): a window column is fs consecutive positions.
// ------------------------------------------------------------------------------------------------
template <typename T, int FS, int PH>
__device__ __forceinline__ float fl_win_mac(const float (&w)[FS][FS], const JINC_CONSTANT char* cset) {
    constexpr int FSP = padded_row(FS);
    constexpr int G = 56 / FSP >= FS ? FS : (56 / FSP);  // coefficient rows per batch of scalar loads (<= 56 SGPRs)
    float acc = 0.f;
#pragma unroll
    for (int g0 = 0; g0 < FS; g0 += G) {
        constexpr int NR = G;  // rows of this batch (the last batch may be shorter)
        float c[NR * FSP];
        const JINC_CONSTANT float* cp = reinterpret_cast<const JINC_CONSTANT float*>(cset + g0 * FSP * 4);
        // one contiguous run of floats -> s_load_dwordx16 / x8 / x4 pieces, one address computation per batch
#pragma unroll
        for (int i = 0; i < NR * FSP; ++i)
            if (g0 * FSP + i < FS * FSP) c[i] = cp[i];
#pragma unroll
        for (int i = 0; i < NR * FSP; ++i)
            if (g0 * FSP + i < FS * FSP) asm volatile("" : "+s"(c[i]));
#pragma unroll
        for (int g = 0; g < NR; ++g)
            if (g0 + g < FS) {
#pragma unroll
                for (int lx = 0; lx < FS; ++lx) acc = acc + w[(PH + lx) % FS][g0 + g] * c[g * FSP + lx];
            }
    }
    return acc;
}

template <typename T, int FS>
__device__ __forceinline__ void fl_win_load_col(float (&col)[FS], const char* p) {
    constexpr int PS = kFrameLanePosBytes(sizeof(T));
#pragma unroll
    for (int ly = 0; ly < FS; ++ly) col[ly] = to_float(*reinterpret_cast<const T*>(p + ly * PS));
}

// One window origin `s` of a strip at ring phase I: the column completing its window, then every pixel with this origin.
// False: past the strip's last origin.
template <typename T, int FS, int I>
__device__ __forceinline__ bool fl_win_step(float (&w)[FS][FS], const char*& pc, int pc_step, int s, int s_last, int& j, int npix, int csv,
                                            int setv, const JINC_CONSTANT char* cbase, float (&res)[4], char* drow, bool lane_on,
                                            bool vec_ok, float peak) {
    constexpr int SB = static_cast<int>(sizeof(T));
    if (s > s_last) return false;  // wave-uniform
    fl_win_load_col<T, FS>(w[(I + FS - 1) % FS], pc);  // column s + FS - 1
    pc += pc_step;
    while (j < npix && __builtin_amdgcn_readlane(csv, j) == s) {
        const uint32_t soff = static_cast<uint32_t>(__builtin_amdgcn_readlane(setv, j));
        const float acc = fl_win_mac<T, FS, I>(w, cbase + soff);
        const int q = j & 3;
        const bool flush = q == 3 || j == npix - 1;  // wave-uniform
        if constexpr (std::is_same_v<T, uint8_t>) {
            // 8-bit: the converted sample goes straight into byte q of the packed word (byte select from an SGPR): one
            // vector instruction instead of a four-way compare / select on the result registers
            uint32_t& pk = reinterpret_cast<uint32_t&>(res[0]);
            // (byte q replaced, the others kept: bytes above q are stale until the group is complete and never stored before)
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

// (fs 8: 80 registers, so that three workgroups of a 53 KB tile share a CU: 6 waves per SIMD instead of 4)
template <typename T, int FS>
__global__ __launch_bounds__(512, (FS == 8 ? 6 : 4)) void ewa_framelane_win_kernel(const FrameLaneArgs a) {
#define JINC_FL_WIN_STAGE_FRAMES 8
#include "kernel_framelane_win_body.inc"
#undef JINC_FL_WIN_STAGE_FRAMES
}

// The same kernel as 1024-thread workgroups held to 64 VGPRs: two workgroups per CU = 8 waves per SIMD (the kernel reacts
// to waves per SIMD more than to anything else), each with a tile of up to 80 KB (32 x 32 pixels for up-scales: less halo,
// half the barriers).  fs 7 only: the larger windows do not fit 64 registers.
template <typename T, int FS>
__global__ __launch_bounds__(1024, 8) void ewa_framelane_win1k_kernel(const FrameLaneArgs a) {
#define JINC_FL_WIN_STAGE_FRAMES 4
#include "kernel_framelane_win_body.inc"
#undef JINC_FL_WIN_STAGE_FRAMES
}

// ------------------------------------------------------------------------------------------------
// Row-segment form, any filter size: units of 4 x K output pixels, K chains below each other share each source row
// segment read from LDS.
// ------------------------------------------------------------------------------------------------
template <typename T, int FS, int K>
__global__ __launch_bounds__(512) void ewa_framelane_kernel(const FrameLaneArgs a) {
    extern __shared__ __attribute__((aligned(16))) char fl_smem[];
    int* cs = reinterpret_cast<int*>(fl_smem);  // window origin of the tile's columns
    int* rs = cs + kFrameLaneMaxTile;           // ... and rows
    int* sets = rs + kFrameLaneMaxTile;         // coefficient set of every pixel of the tile
    char* tile = fl_smem + kFrameLaneTableBytes(a.ty_shift);
    constexpr int PS = kFrameLanePosBytes(sizeof(T));
    constexpr int SB = static_cast<int>(sizeof(T));

    const DevicePlan& p = a.plan;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const int fs = FS ? FS : p.fs;
    const int fsp = padded_row(fs);
    FlTile t;
    if (!fl_locate(a, fs, t)) return;  // padding ids; whole block, before any barrier
    const int bx0 = t.bx0, by0 = t.by0, bx1 = t.bx1, by1 = t.by1, tx0 = t.tx0, ty0 = t.ty0, tw = t.tw, f0 = t.f0, nfg = t.nfg;
    fl_tables(a, t, cs, rs, sets);
    __syncthreads();

    // Coefficient sets of a unit (4 x K pixels), requested with VECTOR loads one unit ahead, one cache line per lane:
    // for a plan without phase structure the sets of a tile are 4 x K x fs x fs coefficients scattered over a table of
    // tens of MB (far beyond the XCD's 4 MB L2), and the scalar loads of the tap loop would each wait for the
    // Infinity Cache; the prefetch turns that into an L2 hit.  The loaded words are only kept alive, never used.
    const uint32_t row_bytes = static_cast<uint32_t>(fsp) * 4u, set_bytes = static_cast<uint32_t>(fs) * row_bytes;
    const int ux_shift = a.tx_shift - 2;
    constexpr int KS = K == 4 ? 2 : (K == 2 ? 1 : 0);
    const int nunits = 1 << (ux_shift + a.ty_shift - KS);
    auto prefetch_unit = [&](int u2) -> uint32_t {
        uint32_t keep = 0;
        if (u2 < nunits) {
            const int ux2 = u2 & ((1 << ux_shift) - 1), uy2 = u2 >> ux_shift;
            const int s = lane & (4 * K - 1);  // pixel of the unit: column s / K, row s % K
            const int set = sets[(K * uy2 + (s % K)) * kFrameLaneMaxTile + 4 * ux2 + s / K];
            const char* sp = reinterpret_cast<const char*>(p.coeffs) + static_cast<size_t>(static_cast<uint32_t>(set) * set_bytes);
            constexpr int kGroups = 64 / (4 * K);  // lanes per pixel = cache lines requested per load instruction and set
            for (uint32_t off = static_cast<uint32_t>(lane / (4 * K)) * 64u; off < set_bytes + 60u; off += 64u * kGroups)
                keep |= *reinterpret_cast<const uint32_t*>(sp + (off < set_bytes - 4u ? off : set_bytes - 4u));
        }
        return keep;
    };
    uint32_t pf_keep = prefetch_unit(wave);  // the wave's first unit: in flight during the staging below

    fl_stage<T, PS, 8, 2>(a, t, tile, tw, 1, lane, wave, nwaves);  // row-major positions: a row segment is fs consecutive positions
    __syncthreads();
    asm volatile("" ::"v"(pf_keep));
    if (lane >= nfg) return;  // lanes without a frame (last group of the batch); no barrier below

    // ---- compute: units of 4 x K output pixels; lane = frame ----
    char* dframe = static_cast<char*>(a.io.dst) + static_cast<size_t>(f0 + lane) * a.io.dst_frame_stride;
    const char* lds_lane = tile + lane * SB;
    const JINC_CONSTANT char* cbase = (const JINC_CONSTANT char*)(p.coeffs);
    for (int u = wave; u < nunits; u += nwaves) {
        asm volatile("" ::"v"(pf_keep));           // (keeps the previous prefetch's loads alive; they completed long ago)
        pf_keep = prefetch_unit(u + nwaves);
        const int ux = u & ((1 << ux_shift) - 1), uy = u >> ux_shift;
        const int x0 = bx0 + 4 * ux, y0 = by0 + K * uy;
        if (x0 > bx1 || y0 > by1) continue;
        const int nvx = min(4, bx1 - x0 + 1);
        int sy[K];
#pragma unroll
        for (int k = 0; k < K; ++k) sy[k] = __builtin_amdgcn_readfirstlane(rs[K * uy + k]);  // rows past by1 repeat the last row's
        const int rr0 = sy[0], rr1 = sy[K - 1] + fs;
        const int kvalid = min(K, by1 - y0 + 1);

        float res[K][4];
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int xx = 0; xx < 4; ++xx) res[k][xx] = 0.f;
#pragma nounroll
        for (int xx = 0; xx < nvx; ++xx) {
            const int sx = __builtin_amdgcn_readfirstlane(cs[4 * ux + xx]);
            uint32_t cb[K];  // byte offset of each chain's coefficient set (host: table < 4 GiB)
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int set = __builtin_amdgcn_readfirstlane(sets[(K * uy + k) * kFrameLaneMaxTile + 4 * ux + xx]);
                cb[k] = static_cast<uint32_t>(set) * set_bytes;
            }
            float acc[K];
#pragma unroll
            for (int k = 0; k < K; ++k) acc[k] = 0.f;
            const char* lcol = lds_lane + (sx - tx0) * PS;
            // Source rows rr0 .. rr1-1 in segments over which the set of chains holding the row is constant: chain k
            // holds rows sy[k] .. sy[k]+fs-1 and sy is ascending, so that set is a range klo..khi.
            int rr = rr0;
            while (rr < rr1) {
                int klo = 0, khi = 0, next = rr1;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    if (k < kvalid) {
                        const int e = sy[k] + fs;
                        if (e <= rr) klo = k + 1;
                        if (sy[k] <= rr) khi = k;
                        if (e > rr) next = min(next, e);
                        if (sy[k] > rr) next = min(next, sy[k]);
                    }
                }
                if (klo <= khi)  // (no chain holds rr only if the windows of consecutive output rows do not overlap)
                    fl_rows_dispatch<T, FS, K>(klo, khi, acc, lcol + (rr - ty0) * tw * PS, tw * PS, rr, next - rr, sy, cb, cbase,
                                               row_bytes, fs);
                rr = next;
            }
            switch (xx) {  // wave-uniform; keeps the register indices of res[][] static without unrolling the column loop
                case 0:
#pragma unroll
                    for (int k = 0; k < K; ++k) res[k][0] = acc[k];
                    break;
                case 1:
#pragma unroll
                    for (int k = 0; k < K; ++k) res[k][1] = acc[k];
                    break;
                case 2:
#pragma unroll
                    for (int k = 0; k < K; ++k) res[k][2] = acc[k];
                    break;
                default:
#pragma unroll
                    for (int k = 0; k < K; ++k) res[k][3] = acc[k];
                    break;
            }
        }

        // ---- store: one 4-sample store per row where the address allows it ----
        const bool vec = nvx == 4 && (a.vec_store_ok & 1) && ((x0 & 3) == 0);  // wave-uniform
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (k < kvalid)
                fl_store4<T>(dframe + static_cast<size_t>(y0 + k) * a.io.dst_pitch + static_cast<size_t>(x0) * SB, res[k], nvx, vec, a.io.peak);
    }
}

template <typename T, int FS, int K>
int launch_fl_t(const FrameLaneArgs& a, hipStream_t stream) {
    const int ntiles = a.block_begin[4];
    dim3 grid(static_cast<unsigned>((ntiles + 7) / 8) * 8u, static_cast<unsigned>((a.io.nframes + 63) / 64), 1);
    dim3 block(static_cast<unsigned>(a.threads), 1, 1);
    hipLaunchKernelGGL((ewa_framelane_kernel<T, FS, K>), grid, block, static_cast<size_t>(a.lds_bytes), stream, a);
    return static_cast<int>(hipGetLastError());
}

template <typename T, int FS>
int launch_fl_win(const FrameLaneArgs& a, hipStream_t stream) {
    if (a.lds_bytes > 64 * 1024) {  // dynamic LDS beyond 64 KB needs the attribute once per kernel AND device
        static std::atomic<bool> attr_set[64];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return static_cast<int>(hipGetLastError());
        if (dev < 0 || dev >= 64 || !attr_set[dev].load()) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&ewa_framelane_win_kernel<T, FS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    80 * 1024) != hipSuccess)
                return static_cast<int>(hipGetLastError());
            if (dev >= 0 && dev < 64) attr_set[dev].store(true);
        }
    }
    const int ntiles = a.block_begin[4];
    dim3 grid(static_cast<unsigned>((ntiles + 7) / 8) * 8u, static_cast<unsigned>((a.io.nframes + 63) / 64), 1);
    dim3 block(static_cast<unsigned>(a.threads), 1, 1);
    hipLaunchKernelGGL((ewa_framelane_win_kernel<T, FS>), grid, block, static_cast<size_t>(a.lds_bytes), stream, a);
    return static_cast<int>(hipGetLastError());
}

template <typename T, int FS>
int launch_fl_win1k(const FrameLaneArgs& a, hipStream_t stream) {
    // dynamic LDS beyond 64 KB needs the attribute once per kernel AND device (a process may drive several: jinc_batch_*)
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return static_cast<int>(hipGetLastError());
    if (dev < 0 || dev >= 64 || !attr_set[dev].load()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&ewa_framelane_win1k_kernel<T, FS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                80 * 1024) != hipSuccess)
            return static_cast<int>(hipGetLastError());
        if (dev >= 0 && dev < 64) attr_set[dev].store(true);
    }
    const int ntiles = a.block_begin[4];
    dim3 grid(static_cast<unsigned>((ntiles + 7) / 8) * 8u, static_cast<unsigned>((a.io.nframes + 63) / 64), 1), block(1024, 1, 1);
    hipLaunchKernelGGL((ewa_framelane_win1k_kernel<T, FS>), grid, block, static_cast<size_t>(a.lds_bytes), stream, a);
    return static_cast<int>(hipGetLastError());
}

template <typename T>
int launch_fl_fs(const FrameLaneArgs& a, hipStream_t stream) {
    if (a.threads == 1024 && a.plan.fs == 7) return launch_fl_win1k<T, 7>(a, stream);
    if (a.variant != 1) {  // sliding-window form for the small filter sizes (variant 1: A/B, the row-segment form)
        switch (a.plan.fs) {
            case 5: return launch_fl_win<T, 5>(a, stream);
            case 7: return launch_fl_win<T, 7>(a, stream);
            case 8: return launch_fl_win<T, 8>(a, stream);
            case 9: return launch_fl_win<T, 9>(a, stream);
            default: break;
        }
    }
    switch (a.plan.fs) {
        case 7: return launch_fl_t<T, 7, 4>(a, stream);
        case 9: return launch_fl_t<T, 9, 4>(a, stream);
        case 10: return launch_fl_t<T, 10, 4>(a, stream);
        case 13: return launch_fl_t<T, 13, 4>(a, stream);
        case 17: return launch_fl_t<T, 17, 4>(a, stream);
        default: return launch_fl_t<T, 0, 4>(a, stream);
    }
}

}  // namespace

int launch_framelane(const FrameLaneArgs& args, void* stream) {
    if (args.block_begin[4] <= 0 || args.io.nframes <= 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (args.io.sample_bytes) {
        case 1: return launch_fl_fs<uint8_t>(args, s);
        case 2: return launch_fl_fs<uint16_t>(args, s);
        default: return launch_fl_fs<float>(args, s);
    }
}

}  // namespace jinc
