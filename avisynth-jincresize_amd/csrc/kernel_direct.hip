// kernel_direct.hip -- ewa_direct_kernel: exactly phase-periodic plans of ANY filter size and source step 1..4,
// without an LDS tile.  Two roles (template parameter MODE):
//   kDirectInterior  the interior of plans the register/LDS kernels of kernel_periodic.hip do not cover: integer and
//                    rational down-scales (1/2: fs 13 step 2, 1/3: fs 20 step 3, 2/3: fs 10 step 3 period 2, ...)
//                    and up-scales with taps 9..16 (fs 19..33);
//   kDirectRowStrip  the top/bottom border rows of EVERY exactly periodic plan (interior column range): a border row
//                    is an interior row with its own window origin and its own coefficient set per column phase.
// The left/right border columns are kernel_colstrip.hip's (lanes have to run along y there -- without LDS staging that
// is one cache line per lane and fetch, measured 3x slower); the four corners stay with the gather kernel.
// See device_common.hpp for the parity rules.
//
// Why no LDS tile: a down-scale by s reads s*s source samples per output sample, so an fp32 LDS tile for the
// 16 chains x 64 lanes of one wave costs 36 KB at s = 3 before the (fs-1)-wide halo -- two waves per CU.  Instead
// every lane reads the row segment its K adjacent output columns share straight from memory in the SOURCE format
// (u8: 1 byte per sample) with 16-byte buffer loads, converts it in registers, and the L1/L2 caches supply the
// vertical reuse (fs / step times per source row).  At full VALU rate that is <= 13 B/clk/CU of cache traffic for u8.
// Buffer loads are bounds-checked by the hardware against the plane's size and only naturally aligned dwords are
// fetched, so the whole-segment fetches (up to a block of taps past the window, and whole 4-column segments of
// partially valid lanes) can never leave the aligned dwords that hold the plane.
#include "device_common.hpp"

#pragma clang fp contract(off)

namespace jinc {
namespace {

enum : int { kDirectInterior = 0, kDirectRowStrip = 1 };

// Chains per lane.  K adjacent output columns of one phase share a row segment; R output rows of one phase share
// the coefficients (a strip row owns a coefficient set: R = 1).
// B: taps per step of the lx loop -- a step's segment (B + SX*(K-1) samples) is what a lane holds in registers.
// D: steps whose segments are fetched together before any of them is computed.  The interior has enough waves in
// flight to hide the fetch latency by switching; the strip launches do not.
template <typename T, int MODE>
struct DirectShape {
    static constexpr int K = 4;
    static constexpr int R = MODE == kDirectInterior ? 4 : 1;
    static constexpr int B = MODE == kDirectInterior ? 16 : 8;
    static constexpr int D = MODE == kDirectInterior ? 1 : 4;
};

// Row segment of a lane in the source format: RW naturally aligned dwords from byte offset voffset (+ soffset), both
// multiples of 4, fetched as 16/8/4-byte pieces.  The lane's first sample sits SH = 0..3 bytes into the first dword;
// SH is the same for every lane of a wave (the lane stride SX*4 samples is a multiple of 4 bytes) and a template
// parameter of everything below, so the samples are picked out of the aligned dwords by the conversion itself
// (v_cvt_f32_ubyteN / SDWA word select) -- no funnel shifts.  Nothing outside the aligned dwords that hold the
// plane's samples is ever touched: the buffer resource bounds the rest.
// The wave-uniform row offset travels in the instruction's soffset.  LLVM documents soffset of raw.buffer.load as
// "excluded from bounds checking"; on gfx950 the hardware does include it -- a fetch is out of range when
// voffset + imm + soffset reaches num_records, also for soffset alone beyond num_records (measured:
// profiles/probes/soffset_probe.hip, profiles/round2/soffset_probe.log).  Because everything here rests on that, the same
// probe runs on the device when a filter is created (buffer_range_check_covers_soffset) and the direct kernel is not
// used where it fails; tests/test_gpu_parity.py::test_buffer_range_check_premise pins it too.  (Moving the row offset
// into a per-row descriptor instead costs 20 % of this issue-bound kernel's speed: D12 60 -> 48 % of the VALU peak.)
template <int RW>
__device__ __forceinline__ void load_raw(BufferRsrc rsrc, uint32_t voffset, uint32_t soffset, uint32_t (&raw)[RW]) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    int w = 0;
#pragma unroll
    for (; w + 4 <= RW; w += 4) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voffset + 4 * w, soffset, 0);
        raw[w] = v.x, raw[w + 1] = v.y, raw[w + 2] = v.z, raw[w + 3] = v.w;
    }
    if constexpr (RW % 4 >= 2) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voffset + 4 * w, soffset, 0);
        raw[w] = v.x, raw[w + 1] = v.y;
        w += 2;
    }
    if constexpr (RW % 2 == 1) raw[w] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, voffset + 4 * w, soffset, 0);
}

// dwords that hold NS samples starting SH bytes into the first one
template <typename T>
__host__ __device__ constexpr int segment_words(int ns, int sh) {
    return (ns * static_cast<int>(sizeof(T)) + sh + 3) / 4;
}

template <typename T, int NS, int RW, int SH>
__device__ __forceinline__ void convert_segment(const uint32_t (&raw)[RW], float (&seg)[NS]) {
    static_assert(NS * static_cast<int>(sizeof(T)) + SH <= 4 * RW, "segment larger than its raw words");
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        constexpr int SB = static_cast<int>(sizeof(T));
        const int b = i * SB + SH;  // byte position of sample i in the raw words (compile-time after unrolling)
        if constexpr (sizeof(T) == 1)
            seg[i] = static_cast<float>((raw[b / 4] >> (8 * (b % 4))) & 0xffu);  // v_cvt_f32_ubyteN
        else if constexpr (sizeof(T) == 2)
            seg[i] = static_cast<float>((raw[b / 4] >> (8 * (b % 4))) & 0xffffu);  // SDWA word select
        else
            seg[i] = __builtin_bit_cast(float, raw[b / 4]);
    }
}

// NT taps (lx = lx0 .. lx0+NT-1) of kernel row ly for all R x K chains of the lane, from segments already in
// registers.  Every chain still meets its taps in (ly, lx) raster order: steps run in lx order inside ly order.
template <typename T, int SX, int NT, int R, int K, int RW, int NC, int SH>
__device__ __forceinline__ void mac_rows(float (&acc)[R][K], const uint32_t (&raw)[R][RW], const float (&cf)[NC]) {
    constexpr int NS = NT + SX * (K - 1);
#pragma unroll
    for (int jj = 0; jj < R; ++jj) {
        // The tail variants (NT = 1 .. B-1) share their conversions and products with the full step; left alone, the
        // compiler hoists all of them above the variant ladder (hundreds of live VGPRs).  Passing the raw words
        // through an empty asm makes them private to this variant.
        uint32_t rw[RW];
#pragma unroll
        for (int w = 0; w < RW; ++w) {
            rw[w] = raw[jj][w];
            asm volatile("" : "+v"(rw[w]));
        }
        float seg[NS];
        convert_segment<T, NS, RW, SH>(rw, seg);
        // tap-major over the K chains of the row: neighbouring instructions belong to different chains, so a
        // wave does not wait for its own previous add (each chain still sees its taps in lx order)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int k = 0; k < K; ++k) acc[jj][k] = acc[jj][k] + seg[SX * k + t] * cf[t];
    }
}

// remaining >= B: a full step; otherwise the tail of the kernel row with a compile-time tap count
template <typename T, int SX, int NT, int R, int K, int RW, int NC, int SH>
struct MacSelect {
    static __device__ __forceinline__ void run(int remaining, float (&acc)[R][K], const uint32_t (&raw)[R][RW],
                                               const float (&cf)[NC]) {
        if (remaining >= NT)
            mac_rows<T, SX, NT, R, K, RW, NC, SH>(acc, raw, cf);
        else
            MacSelect<T, SX, NT - 1, R, K, RW, NC, SH>::run(remaining, acc, raw, cf);
    }
};
template <typename T, int SX, int R, int K, int RW, int NC, int SH>
struct MacSelect<T, SX, 0, R, K, RW, NC, SH> {
    static __device__ __forceinline__ void run(int, float (&)[R][K], const uint32_t (&)[R][RW], const float (&)[NC]) {}
};

// Interior form (one step at a time): every variant fetches exactly the dwords ITS tap count needs -- the tail of a
// kernel row, or a kernel row shorter than a full step (fs 13 with 16-tap steps), would otherwise fetch a full
// step's segment: one buffer_load more per row than necessary.
template <typename T, int SX, int NT, int R, int K, int NC, int SH>
struct FetchMacSelect {
    static __device__ __forceinline__ void run(int remaining, float (&acc)[R][K], BufferRsrc rsrc, const uint32_t (&voff)[R],
                                               uint32_t srow, uint32_t srow_step, const float (&cf)[NC]) {
        if (remaining >= NT) {
            constexpr int RWN = segment_words<T>(NT + SX * (K - 1), SH);
            uint32_t raw[R][RWN];
#pragma unroll
            for (int jj = 0; jj < R; ++jj) load_raw<RWN>(rsrc, voff[jj], srow + static_cast<uint32_t>(jj) * srow_step, raw[jj]);
            mac_rows<T, SX, NT, R, K, RWN, NC, SH>(acc, raw, cf);
        } else {
            FetchMacSelect<T, SX, NT - 1, R, K, NC, SH>::run(remaining, acc, rsrc, voff, srow, srow_step, cf);
        }
    }
};
template <typename T, int SX, int R, int K, int NC, int SH>
struct FetchMacSelect<T, SX, 0, R, K, NC, SH> {
    static __device__ __forceinline__ void run(int, float (&)[R][K], BufferRsrc, const uint32_t (&)[R], uint32_t, uint32_t,
                                               const float (&)[NC]) {}
};

// The whole tap loop of an item for one value of SH: steps in (ly, lx) raster order, step s covers kernel row
// s / nsx, taps (s % nsx) * B ...
template <typename T, int SX, int MODE, int SH, int R, int K>
__device__ __forceinline__ void direct_steps(float (&acc)[R][K], BufferRsrc srsrc, const uint32_t (&voff)[R], uint32_t soff0,
                                             uint32_t srow_step, uint32_t pitch, const JINC_CONSTANT float* cs, int coeff_row,
                                             int fs) {
    using Shape = DirectShape<T, MODE>;
    static_assert(R == Shape::R && K == Shape::K, "chain shape of the mode");
    constexpr int B = Shape::B, D = Shape::D;
    constexpr int RW = segment_words<T>(B + SX * (K - 1), SH);  // dwords of a full step's segment
    constexpr uint32_t SB = static_cast<uint32_t>(sizeof(T));
    const int nsx = (fs + B - 1) / B;
    const int nsteps = fs * nsx;
    int ly = 0, lxi = 0;  // coordinates of the next step to fetch
#pragma unroll 1
    for (int s0 = 0; s0 < nsteps; s0 += D) {
        if constexpr (D == 1) {
            const int lx = lxi * B;
            const uint32_t srow = soff0 + static_cast<uint32_t>(ly) * pitch + static_cast<uint32_t>(lx) * SB;
            const JINC_CONSTANT float* c = cs + static_cast<size_t>(ly) * coeff_row + lx;
            float cf1[B];
#pragma unroll
            for (int t = 0; t < B; ++t) cf1[t] = c[t];  // wave-uniform -> SGPRs (the allocation has slack past the last row)
            FetchMacSelect<T, SX, B, R, K, B, SH>::run(fs - lx, acc, srsrc, voff, srow, srow_step, cf1);
            if (++lxi == nsx) lxi = 0, ++ly;
        } else {
            uint32_t raw[D][R][RW];
            float cf[D][B];
            int rem[D];
#pragma unroll
            for (int d = 0; d < D; ++d) {
                if (s0 + d < nsteps) {  // wave-uniform
                    const int lx = lxi * B;
                    const uint32_t srow = soff0 + static_cast<uint32_t>(ly) * pitch + static_cast<uint32_t>(lx) * SB;
#pragma unroll
                    for (int jj = 0; jj < R; ++jj) load_raw<RW>(srsrc, voff[jj], srow + static_cast<uint32_t>(jj) * srow_step, raw[d][jj]);
                    const JINC_CONSTANT float* c = cs + static_cast<size_t>(ly) * coeff_row + lx;
#pragma unroll
                    for (int t = 0; t < B; ++t) cf[d][t] = c[t];  // wave-uniform -> SGPRs
                    rem[d] = fs - lx;
                    if (++lxi == nsx) lxi = 0, ++ly;
                }
            }
#pragma unroll
            for (int d = 0; d < D; ++d)
                if (s0 + d < nsteps) MacSelect<T, SX, B, R, K, RW, B, SH>::run(rem[d], acc, raw[d], cf[d]);
        }
    }
}

__device__ __forceinline__ int plan_int(const int32_t* base, size_t index) {  // wave-uniform table lookup -> s_load
    return ((const JINC_CONSTANT int32_t*)base)[index];
}

// One wave = one item.
//   interior : phase (p, q) x 256 period-columns (4 per lane) x 4 period-rows;
//   row strip: output row y x column phase p x 256 period-columns (4 per lane, 1 row);
// The four waves of a workgroup take consecutive items (phases of one row chunk / strip line first), so they share
// source rows in the L1.  A kernel row is walked in steps of B taps plus one shorter tail step.
template <typename T, int SX, int MODE>
__global__ __launch_bounds__(256) void ewa_direct_kernel(const DirectArgs a, const PlaneIO io) {
    using Shape = DirectShape<T, MODE>;
    constexpr int K = Shape::K, R = Shape::R;
    constexpr uint32_t SB = static_cast<uint32_t>(sizeof(T));
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int tile_a, tile_b;
    swizzled_tile(tile_a, tile_b);  // tile_a: along the lane axis; tile_b: groups of 4 items
    const int item = tile_b * 4 + wave;
    const size_t frame = blockIdx.z;
    const uint32_t pitch = static_cast<uint32_t>(io.src_pitch);
    const int fs = a.fs;

    // ---- which chains does this lane own? ----
    int set;                 // wave-uniform coefficient set
    uint32_t voff[R];        // per-lane byte offset of the segments of its R rows (without ly / lx)
    uint32_t soff0;          // wave-uniform byte offset of kernel row 0, tap 0
    uint32_t srow_step;      // wave-uniform byte distance between the lane's R rows (interior only)
    bool row_ok[R];          // chain rows that exist
    int cols_valid = K;      // chain columns that exist
    uint32_t dst_voff[R];    // per-lane byte offset of the first output sample of each row
    uint32_t dst_soff[R];    // wave-uniform part
    uint32_t dst_xstep = 0;  // bytes between the K output columns
    bool lane_ok;            // the lane owns at least one chain (checked after the wave-uniform set-up below: a
                             // divergent exit in front of it would push every uniform value into VGPRs)
    if constexpr (MODE == kDirectInterior) {
        const int nphase = a.px * a.py;
        const int ch = item / nphase;
        const int ph = item - ch * nphase;
        const int q = ph / a.px;
        const int p = ph - q * a.px;
        const int j = ch * R;  // first period-row of the chunk
        if (j >= a.nj) return;  // wave-uniform
        const int i_lane = tile_a * (64 * K) + K * lane;
        cols_valid = a.ni - i_lane;
        lane_ok = cols_valid > 0;
        set = a.set[ph];
        soff0 = static_cast<uint32_t>(a.start_y[q] + a.sy * j) * pitch;
        srow_step = static_cast<uint32_t>(a.sy) * pitch;
#pragma unroll
        for (int jj = 0; jj < R; ++jj) {
            voff[jj] = static_cast<uint32_t>(a.start_x[p] + SX * i_lane) * SB;
            row_ok[jj] = j + jj < a.nj;
            dst_voff[jj] = static_cast<uint32_t>(a.ix0 + a.px * i_lane + p) * SB;
            dst_soff[jj] = static_cast<uint32_t>(a.iy0 + a.py * (j + jj) + q) * static_cast<uint32_t>(io.dst_pitch);
        }
        dst_xstep = static_cast<uint32_t>(a.px) * SB;
    } else if constexpr (MODE == kDirectRowStrip) {
        const int line = item / a.px;
        const int p = item - line * a.px;
        if (line >= a.line_n[0] + a.line_n[1]) return;
        const int y = line < a.line_n[0] ? a.line0[0] + line : a.line0[1] + (line - a.line_n[0]);
        const int i_lane = tile_a * (64 * K) + K * lane;
        cols_valid = a.ni - i_lane;
        lane_ok = cols_valid > 0;
        const int xr = a.ix0 + p;  // representative column of the phase
        const int rc = plan_int(a.plan.row_class, y);
        set = rc < 0 ? plan_int(a.plan.brow_set, static_cast<size_t>(~rc) * a.plan.dst_w + xr)
                     : plan_int(a.plan.interior_set, static_cast<size_t>(rc) * a.plan.n_col_classes + plan_int(a.plan.col_class, xr));
        soff0 = static_cast<uint32_t>(plan_int(a.plan.row_start, y)) * pitch;
        srow_step = 0;
        voff[0] = static_cast<uint32_t>(a.start_x[p] + SX * i_lane) * SB;
        row_ok[0] = true;
        dst_voff[0] = static_cast<uint32_t>(a.ix0 + a.px * i_lane + p) * SB;
        dst_soff[0] = static_cast<uint32_t>(y) * static_cast<uint32_t>(io.dst_pitch);
        dst_xstep = static_cast<uint32_t>(a.px) * SB;
    }

    // Split every segment address into an aligned part and the byte shift SH of its first sample.  The host guarantees
    // that the pitch and the frame stride are multiples of 4; the plane's own misalignment (base & 3) joins the lane
    // offset.  The lane stride (SX * 4 samples) is a multiple of 4 bytes, so SH is the same for all lanes: one
    // wave-uniform switch selects the instantiation of the tap loop that has SH as a compile-time constant.
    const uintptr_t plane = reinterpret_cast<uintptr_t>(io.src) + frame * io.src_frame_stride;
    const uint32_t mis = static_cast<uint32_t>(plane & 3u) + (soff0 & 3u);
    soff0 &= ~3u;
    const uint32_t sh = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>((voff[0] + mis) & 3u)));
#pragma unroll
    for (int jj = 0; jj < R; ++jj) voff[jj] = (voff[jj] + mis) & ~3u;
    const BufferRsrc srsrc = make_rsrc(reinterpret_cast<char*>(plane & ~static_cast<uintptr_t>(3)), a.src_bytes);
    if (!lane_ok) return;
    const JINC_CONSTANT float* cs =
        (const JINC_CONSTANT float*)(a.coeffs + static_cast<size_t>(set) * (static_cast<size_t>(fs) * a.coeff_row));

    float acc[R][K];
#pragma unroll
    for (int jj = 0; jj < R; ++jj)
#pragma unroll
        for (int k = 0; k < K; ++k) acc[jj][k] = 0.f;

    if constexpr (sizeof(T) == 4) {
        direct_steps<T, SX, MODE, 0>(acc, srsrc, voff, soff0, srow_step, pitch, cs, a.coeff_row, fs);
    } else if constexpr (sizeof(T) == 2) {
        if (sh == 0)
            direct_steps<T, SX, MODE, 0>(acc, srsrc, voff, soff0, srow_step, pitch, cs, a.coeff_row, fs);
        else
            direct_steps<T, SX, MODE, 2>(acc, srsrc, voff, soff0, srow_step, pitch, cs, a.coeff_row, fs);
    } else {
        switch (sh) {
            case 0: direct_steps<T, SX, MODE, 0>(acc, srsrc, voff, soff0, srow_step, pitch, cs, a.coeff_row, fs); break;
            case 1: direct_steps<T, SX, MODE, 1>(acc, srsrc, voff, soff0, srow_step, pitch, cs, a.coeff_row, fs); break;
            case 2: direct_steps<T, SX, MODE, 2>(acc, srsrc, voff, soff0, srow_step, pitch, cs, a.coeff_row, fs); break;
            default: direct_steps<T, SX, MODE, 3>(acc, srsrc, voff, soff0, srow_step, pitch, cs, a.coeff_row, fs); break;
        }
    }

    const BufferRsrc drsrc = make_rsrc(static_cast<char*>(io.dst) + frame * io.dst_frame_stride,
                                       static_cast<uint32_t>(io.dst_pitch) * a.dst_h);
#pragma unroll
    for (int jj = 0; jj < R; ++jj) {
        if (row_ok[jj]) {  // wave-uniform
#pragma unroll
            for (int k = 0; k < K; ++k)
                if (k < cols_valid) store_sample_buf<T>(drsrc, dst_voff[jj] + k * dst_xstep, dst_soff[jj], acc[jj][k], io.peak);
        }
    }
}

template <typename T, int SX, int MODE>
int launch_direct_t(const DirectArgs& da, const PlaneIO& io, hipStream_t stream) {
    constexpr int K = DirectShape<T, MODE>::K, R = DirectShape<T, MODE>::R;
    int tiles, items;
    if (MODE == kDirectInterior) {
        tiles = (da.ni + 64 * K - 1) / (64 * K);
        items = (da.nj + R - 1) / R * da.px * da.py;
    } else {
        tiles = (da.ni + 64 * K - 1) / (64 * K);
        items = (da.line_n[0] + da.line_n[1]) * da.px;
    }
    if (tiles <= 0 || items <= 0) return 0;
    dim3 grid(tiles, (items + 3) / 4, io.nframes);
    hipLaunchKernelGGL((ewa_direct_kernel<T, SX, MODE>), grid, dim3(256, 1, 1), 0, stream, da, io);
    return static_cast<int>(hipGetLastError());
}

template <typename T, int MODE>
int launch_direct_sx(const DirectArgs& da, const PlaneIO& io, hipStream_t stream) {
    switch (da.sx) {
        case 1: return launch_direct_t<T, 1, MODE>(da, io, stream);
        case 2: return launch_direct_t<T, 2, MODE>(da, io, stream);
        case 3: return launch_direct_t<T, 3, MODE>(da, io, stream);
        case 4: return launch_direct_t<T, 4, MODE>(da, io, stream);
        default: return static_cast<int>(hipErrorInvalidValue);
    }
}

template <int MODE>
int launch_direct_mode(const DirectArgs& args, const PlaneIO& io, void* stream) {
    if (io.nframes <= 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (io.sample_bytes) {
        case 1: return launch_direct_sx<uint8_t, MODE>(args, io, s);
        case 2: return launch_direct_sx<uint16_t, MODE>(args, io, s);
        default: return launch_direct_sx<float, MODE>(args, io, s);
    }
}

}  // namespace

namespace {
// See load_raw: lane l reads dword l of a 2N-byte buffer through a descriptor of N bytes with soffset = N - 128 and with
// soffset = N + 256; the range check must zero lanes 32.. of the first read and all of the second.
__global__ void soffset_probe_kernel(const uint32_t* buf, uint32_t nbytes, uint32_t* out) {
    const BufferRsrc r = make_rsrc(const_cast<uint32_t*>(buf), nbytes);
    const uint32_t l = threadIdx.x;
    out[l] = __builtin_amdgcn_raw_buffer_load_b32(r, 4 * l, nbytes - 128, 0);
    out[64 + l] = __builtin_amdgcn_raw_buffer_load_b32(r, 4 * l, nbytes + 256, 0);
}
}  // namespace

int launch_soffset_probe(const uint32_t* buf, uint32_t nbytes, uint32_t* out, void* stream) {
    hipLaunchKernelGGL(soffset_probe_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), buf, nbytes, out);
    return static_cast<int>(hipGetLastError());
}

bool direct_supported(int fs, int px, int py, int sx, int sy) {
    if (fs < 1 || sx < 1 || sx > 4 || sy < 1 || sy > 4) return false;
    return px >= 1 && py >= 1 && px <= 16 && py <= 16 && px * py <= 256;
}

int launch_direct(const DirectArgs& args, const PlaneIO& io, void* stream) {
    if (args.ni <= 0 || args.nj <= 0) return 0;
    return launch_direct_mode<kDirectInterior>(args, io, stream);
}

int launch_direct_row_strips(const DirectArgs& args, const PlaneIO& io, void* stream) {
    if (args.ni <= 0) return 0;
    return launch_direct_mode<kDirectRowStrip>(args, io, stream);
}

}  // namespace jinc
