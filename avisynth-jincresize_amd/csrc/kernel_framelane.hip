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
#include "device_common.hpp"

#pragma clang fp contract(off)

namespace jinc {
namespace {

constexpr int kFlTableInts = 2 * kFrameLaneMaxTile + kFrameLaneMaxTile * kFrameLaneMaxTile;

template <typename T>
struct FlPack;  // four horizontally adjacent samples of one frame
template <>
struct FlPack<uint8_t> {
    using type = uint32_t;
};
template <>
struct FlPack<uint16_t> {
    using type = uint2;
};
template <>
struct FlPack<float> {
    using type = float4;
};

template <typename T>
__device__ __forceinline__ typename FlPack<T>::type fl_pack(const float (&r)[4], float peak) {
    if constexpr (std::is_same_v<T, uint8_t>) {
        uint32_t w = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) w = __builtin_amdgcn_cvt_pk_u8_f32(r[i], static_cast<uint32_t>(i), w);
        return w;
    } else if constexpr (std::is_same_v<T, uint16_t>) {
        uint2 w;
        w.x = round_sample(r[0], peak) | (round_sample(r[1], peak) << 16);
        w.y = round_sample(r[2], peak) | (round_sample(r[3], peak) << 16);
        return w;
    } else {
        return make_float4(r[0], r[1], r[2], r[3]);
    }
}

// One chunk of N <= 8 taps of one source row for the K chains of an output column: samples from the LDS tile (converted
// once), K coefficient rows through scalar loads, then the chains whose window holds this source row.
template <typename T, int K, int N>
__device__ __forceinline__ void fl_chunk(float (&acc)[K], const char* lds, const JINC_CONSTANT float* const (&cp)[K],
                                         const bool (&on)[K]) {
    constexpr int PS = kFrameLanePosBytes(sizeof(T));
    constexpr int NL = (N + 3) & ~3;  // coefficient rows are padded to multiples of 4 floats
    float seg[N];
#pragma unroll
    for (int j = 0; j < N; ++j) seg[j] = to_float(*reinterpret_cast<const T*>(lds + j * PS));
    float c[K][NL];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int j = 0; j < NL; ++j) c[k][j] = cp[k][j];
    // All K coefficient rows are requested together, in front of the wave-uniform branches (left alone the compiler sinks
    // each row's scalar loads into its branch: load -> wait -> 14 VALU, K times per source row).
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int j = 0; j < NL; ++j) asm volatile("" : "+s"(c[k][j]));
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (on[k]) {
#pragma unroll
            for (int j = 0; j < N; ++j) acc[k] = acc[k] + seg[j] * c[k][j];
        }
    }
}

template <typename T, int FS, int K>
__global__ __launch_bounds__(512) void ewa_framelane_kernel(const FrameLaneArgs a) {
    extern __shared__ __attribute__((aligned(16))) char fl_smem[];
    int* cs = reinterpret_cast<int*>(fl_smem);  // window origin of the tile's columns
    int* rs = cs + kFrameLaneMaxTile;           // ... and rows
    int* sets = rs + kFrameLaneMaxTile;         // coefficient set of every pixel of the tile
    char* tile = fl_smem + kFlTableInts * 4;
    constexpr int PS = kFrameLanePosBytes(sizeof(T));
    constexpr int SB = static_cast<int>(sizeof(T));

    const DevicePlan& p = a.plan;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;

    // ---- block -> tile.  Workgroups are dealt round-robin over the 8 XCDs by linear id; XCD k walks the contiguous
    // run of tiles [k*q + min(k, rem), ...) so that the halos shared by neighbouring tiles stay in one L2.
    const int ntiles = a.block_begin[4];
    int tid;
    {
        const int q = ntiles / 8, rem = ntiles % 8;
        const int xcd = blockIdx.x % 8, idx = blockIdx.x / 8;
        if (idx >= q + (xcd < rem ? 1 : 0)) return;  // padding ids; whole block, before any barrier
        tid = xcd * q + (xcd < rem ? xcd : rem) + idx;
    }
    int r = 0;
    while (r + 1 < a.rects.n && tid >= a.block_begin[r + 1]) ++r;
    const int local = tid - a.block_begin[r];
    const int tcx = local % a.tiles_x[r], tcy = local / a.tiles_x[r];
    const int rx1 = a.rects.x0[r] + a.rects.w[r], ry1 = a.rects.y0[r] + a.rects.h[r];
    const int bx0 = a.rects.x0[r] + (tcx << a.tx_shift), by0 = a.rects.y0[r] + (tcy << a.ty_shift);
    const int bx1 = min(bx0 + (1 << a.tx_shift), rx1) - 1, by1 = min(by0 + (1 << a.ty_shift), ry1) - 1;  // inclusive

    const int fs = FS ? FS : p.fs;
    const int fsp = padded_row(fs);
    const int f0 = blockIdx.y * 64;
    const int nfg = min(64, a.io.nframes - f0);  // frames of this group

    // ---- per-tile tables: window origins and set ids (looked up once per pixel for 64 frames) ----
    for (int t = threadIdx.x; t < (1 << a.tx_shift); t += blockDim.x) cs[t] = p.col_start[min(bx0 + t, bx1)];
    for (int t = threadIdx.x; t < (1 << a.ty_shift); t += blockDim.x) rs[t] = p.row_start[min(by0 + t, by1)];
    for (int i = threadIdx.x; i < (1 << (a.tx_shift + a.ty_shift)); i += blockDim.x) {
        const int ix = i & ((1 << a.tx_shift) - 1), iy = i >> a.tx_shift;
        const int qx = min(bx0 + ix, bx1), qy = min(by0 + iy, by1);
        const int rc = p.row_class[qy], cc = p.col_class[qx];
        int set;
        if (rc < 0)
            set = p.brow_set[static_cast<size_t>(~rc) * p.dst_w + qx];
        else if (cc < 0)
            set = p.bcol_set[static_cast<size_t>(~cc) * p.dst_h + qy];
        else
            set = p.interior_set[rc * p.n_col_classes + cc];
        sets[iy * kFrameLaneMaxTile + ix] = set;
    }

    // ---- stage the source footprint of the group's frames: [row][column][frame], source format ----
    const int tx0 = p.col_start[bx0], ty0 = p.row_start[by0];
    const int tw = p.col_start[bx1] + fs - tx0;  // <= 64 (host: framelane_configure)
    const int th = p.row_start[by1] + fs - ty0;
    {
        const int sh = tw <= 1 ? 0 : 32 - __builtin_clz(static_cast<unsigned>(tw - 1));  // lanes per row = 1 << sh >= tw
        const int lc = lane & ((1 << sh) - 1), lr = lane >> sh, rps = 64 >> sh;
        const int nsteps = (th + rps - 1) >> (6 - sh);
        const int colc = min(lc, tw - 1);
        constexpr int U = 16;  // loads in flight per lane
        for (int fi = wave; fi < nfg; fi += nwaves) {
            const char* sframe = static_cast<const char*>(a.io.src) + static_cast<size_t>(f0 + fi) * a.io.src_frame_stride +
                                 static_cast<size_t>(ty0) * a.io.src_pitch + static_cast<size_t>(tx0 + colc) * SB;
            char* lds_f = tile + fi * SB + lc * PS;
            for (int s0 = 0; s0 < nsteps; s0 += U) {
                T v[U];
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    const int row = min((s0 + j) * rps + lr, th - 1);
                    v[j] = *reinterpret_cast<const T*>(sframe + static_cast<size_t>(row) * a.io.src_pitch);
                }
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    const int row = (s0 + j) * rps + lr;
                    if (row < th && lc < tw) *reinterpret_cast<T*>(lds_f + row * tw * PS) = v[j];
                }
            }
        }
    }
    __syncthreads();
    if (lane >= nfg) return;  // lanes without a frame (last group of the batch); no barrier below

    // ---- compute: units of 4 x K output pixels; lane = frame ----
    char* dframe = static_cast<char*>(a.io.dst) + static_cast<size_t>(f0 + lane) * a.io.dst_frame_stride;
    const char* lds_lane = tile + lane * SB;
    const size_t set_floats = static_cast<size_t>(fs) * fsp;
    const int ux_shift = a.tx_shift - 2;
    constexpr int KS = K == 4 ? 2 : (K == 2 ? 1 : 0);
    const int nunits = 1 << (ux_shift + a.ty_shift - KS);
    for (int u = wave; u < nunits; u += nwaves) {
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
        for (int xx = 0; xx < 4; ++xx) {
#pragma unroll
            for (int k = 0; k < K; ++k) res[k][xx] = 0.f;
            if (xx < nvx) {
                const int sx = __builtin_amdgcn_readfirstlane(cs[4 * ux + xx]);
                const JINC_CONSTANT float* cb[K];
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const int set = __builtin_amdgcn_readfirstlane(sets[(K * uy + k) * kFrameLaneMaxTile + 4 * ux + xx]);
                    cb[k] = (const JINC_CONSTANT float*)(p.coeffs + static_cast<size_t>(set) * set_floats);
                }
                float acc[K];
#pragma unroll
                for (int k = 0; k < K; ++k) acc[k] = 0.f;
                const char* lrow = lds_lane + ((rr0 - ty0) * tw + (sx - tx0)) * PS;
                for (int rr = rr0; rr < rr1; ++rr) {
                    bool on[K];
                    const JINC_CONSTANT float* cp[K];
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const int ly = rr - sy[k];
                        on[k] = static_cast<unsigned>(ly) < static_cast<unsigned>(fs) && k < kvalid;
                        cp[k] = cb[k] + min(max(ly, 0), fs - 1) * fsp;
                    }
                    if constexpr (FS != 0) {
#pragma unroll
                        for (int c0 = 0; c0 < FS; c0 += 8) {
                            constexpr int kFull = 8;
                            if (FS - c0 >= kFull) {
                                fl_chunk<T, K, 8>(acc, lrow + c0 * PS, cp, on);
                            } else {
                                fl_chunk<T, K, (FS % 8 ? FS % 8 : 8)>(acc, lrow + c0 * PS, cp, on);
                            }
#pragma unroll
                            for (int k = 0; k < K; ++k) cp[k] += 8;
                        }
                    } else {
                        for (int c0 = 0; c0 < fs; c0 += 8) {
                            const int n = fs - c0;  // wave-uniform
                            if (n >= 8) {
                                fl_chunk<T, K, 8>(acc, lrow + c0 * PS, cp, on);
                            } else {
                                switch (n) {
                                    case 1: fl_chunk<T, K, 1>(acc, lrow + c0 * PS, cp, on); break;
                                    case 2: fl_chunk<T, K, 2>(acc, lrow + c0 * PS, cp, on); break;
                                    case 3: fl_chunk<T, K, 3>(acc, lrow + c0 * PS, cp, on); break;
                                    case 4: fl_chunk<T, K, 4>(acc, lrow + c0 * PS, cp, on); break;
                                    case 5: fl_chunk<T, K, 5>(acc, lrow + c0 * PS, cp, on); break;
                                    case 6: fl_chunk<T, K, 6>(acc, lrow + c0 * PS, cp, on); break;
                                    default: fl_chunk<T, K, 7>(acc, lrow + c0 * PS, cp, on); break;
                                }
                            }
#pragma unroll
                            for (int k = 0; k < K; ++k) cp[k] += 8;
                        }
                    }
                    lrow += tw * PS;
                }
#pragma unroll
                for (int k = 0; k < K; ++k) res[k][xx] = acc[k];
            }
        }

        // ---- store: one 4-sample store per row where the address allows it ----
        const bool vec = nvx == 4 && a.vec_store_ok && ((x0 & 3) == 0);  // wave-uniform
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (k < kvalid) {
                char* d = dframe + static_cast<size_t>(y0 + k) * a.io.dst_pitch + static_cast<size_t>(x0) * SB;
                if (vec) {
                    *reinterpret_cast<typename FlPack<T>::type*>(d) = fl_pack<T>(res[k], a.io.peak);
                } else {
#pragma unroll
                    for (int xx = 0; xx < 4; ++xx)
                        if (xx < nvx) store_sample<T>(reinterpret_cast<T*>(d) + xx, res[k][xx], a.io.peak);
                }
            }
        }
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

template <typename T>
int launch_fl_fs(const FrameLaneArgs& a, hipStream_t stream) {
    switch (a.plan.fs) {
        case 7: return launch_fl_t<T, 7, 4>(a, stream);
        case 8: return launch_fl_t<T, 8, 4>(a, stream);
        case 9: return launch_fl_t<T, 9, 4>(a, stream);
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
