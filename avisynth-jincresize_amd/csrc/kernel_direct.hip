// kernel_direct.hip -- interior kernel of exactly phase-periodic plans of ANY filter size and source step:
// integer and rational down-scales (1/2: fs 13 step 2, 1/3: fs 20 step 3, 2/3: fs 10 step 3 period 2, ...) and the
// up-scales whose footprint exceeds the register/LDS kernels of kernel_periodic.hip (taps 9..16: fs 19..33).
// See device_common.hpp for the parity rules.
//
// Why no LDS tile here: a down-scale by s reads s*s source samples per output sample, so an fp32 LDS tile for the
// 16 chains x 64 lanes of one wave costs 36 KB at s = 3 before the (fs-1)-wide halo -- two waves per CU.  Instead
// every lane reads the row segment its K adjacent output columns share straight from memory in the SOURCE format
// (u8: 1 byte per sample) with 16-byte loads, converts it in registers, and the L1/L2 caches supply the vertical
// reuse (fs / step times per source row).  At full VALU rate that is <= 13 B/clk/CU of cache traffic for u8.
#include "device_common.hpp"

#pragma clang fp contract(off)

namespace jinc {
namespace {

constexpr int kDirectK = 4;  // adjacent output columns (of one phase) per lane: they share one row segment
constexpr int kDirectR = 4;  // output rows (of one phase) per lane => 16 independent chains per lane

// Taps per block of the lx loop.  A block's segment (B + SX*(K-1) samples) is what a lane holds in registers.
template <typename T>
struct DirectBlock {
    static constexpr int B = sizeof(T) == 1 ? 16 : 8;
};

// Row segment of a lane in the source format: RW dwords fetched as 16/8/4-byte pieces from an address that is
// aligned to the sample size only.  Fewer than 4 bytes past the last sample of the segment are touched.
template <int RW>
__device__ __forceinline__ void load_raw(const char* p, uint32_t (&raw)[RW]) {
    int w = 0;
#pragma unroll
    for (; w + 4 <= RW; w += 4) __builtin_memcpy(&raw[w], p + 4 * w, 16);
    if constexpr (RW % 4 >= 2) {
        __builtin_memcpy(&raw[w], p + 4 * w, 8);
        w += 2;
    }
    if constexpr (RW % 2 == 1) __builtin_memcpy(&raw[w], p + 4 * w, 4);
}

template <typename T, int NS, int RW>
__device__ __forceinline__ void convert_segment(const uint32_t (&raw)[RW], float (&seg)[NS]) {
    static_assert(NS * static_cast<int>(sizeof(T)) <= 4 * RW, "segment larger than its raw words");
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        if constexpr (sizeof(T) == 1)
            seg[i] = static_cast<float>((raw[i / 4] >> (8 * (i % 4))) & 0xffu);  // v_cvt_f32_ubyteN
        else if constexpr (sizeof(T) == 2)
            seg[i] = static_cast<float>((raw[i / 2] >> (16 * (i % 2))) & 0xffffu);
        else
            seg[i] = __builtin_bit_cast(float, raw[i]);
    }
}

template <typename T, int SX>
struct DirectGeom {
    static constexpr int B = DirectBlock<T>::B;
    static constexpr int NSB = B + SX * (kDirectK - 1);                       // samples of a full block's segment
    static constexpr int RW = (NSB * static_cast<int>(sizeof(T)) + 3) / 4;   // ... in dwords
};

// NT taps (lx = lx0 .. lx0+NT-1) of kernel row ly for all R x K chains of the lane, from segments already in
// registers.  Every chain still meets its taps in (ly, lx) raster order: steps run in lx order inside ly order.
template <typename T, int SX, int NT, int RW, int NC>
__device__ __forceinline__ void mac_rows(float (&acc)[kDirectR][kDirectK], const uint32_t (&raw)[kDirectR][RW],
                                         const float (&cf)[NC]) {
    constexpr int NS = NT + SX * (kDirectK - 1);
#pragma unroll
    for (int jj = 0; jj < kDirectR; ++jj) {
        float seg[NS];
        convert_segment<T, NS, RW>(raw[jj], seg);
#pragma unroll
        for (int k = 0; k < kDirectK; ++k)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[jj][k] = acc[jj][k] + seg[SX * k + t] * cf[t];
    }
}

// remaining >= B: a full block; otherwise the tail of the kernel row with a compile-time tap count
template <typename T, int SX, int NT, int RW, int NC>
struct MacSelect {
    static __device__ __forceinline__ void run(int remaining, float (&acc)[kDirectR][kDirectK],
                                               const uint32_t (&raw)[kDirectR][RW], const float (&cf)[NC]) {
        if (remaining >= NT)
            mac_rows<T, SX, NT, RW, NC>(acc, raw, cf);
        else
            MacSelect<T, SX, NT - 1, RW, NC>::run(remaining, acc, raw, cf);
    }
};
template <typename T, int SX, int RW, int NC>
struct MacSelect<T, SX, 0, RW, NC> {
    static __device__ __forceinline__ void run(int, float (&)[kDirectR][kDirectK], const uint32_t (&)[kDirectR][RW],
                                               const float (&)[NC]) {}
};

// One wave = one item: phase (p, q) x 64*K period-columns x R period-rows.  The four waves of a workgroup take
// consecutive items (phases of one row chunk first, then the next chunk), so they share source rows in the L1.
// A kernel row is walked in steps of B taps (+ one shorter tail step); PIPE: the segments and coefficients of
// step n+1 are fetched into a second register set before step n is computed (the waves of a SIMD are too few --
// about 100 VGPRs each -- to hide the fetch latency by switching alone).
template <typename T, int SX, bool PIPE>
__global__ __launch_bounds__(256) void ewa_periodic_direct_kernel(const DirectArgs a, const PlaneIO io) {
    using G = DirectGeom<T, SX>;
    constexpr int B = G::B, RW = G::RW;
    constexpr int K = kDirectK, R = kDirectR;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int tile_x, tile_y;
    swizzled_tile(tile_x, tile_y);
    const int nphase = a.px * a.py;
    const int item = tile_y * 4 + wave;
    const int ch = item / nphase;
    const int ph = item - ch * nphase;
    const int q = ph / a.px;
    const int p = ph - q * a.px;
    const int j = ch * R;  // first period-row of the chunk
    const int rows_valid = a.nj - j;
    if (rows_valid <= 0) return;  // wave-uniform
    const int i_lane = tile_x * (64 * K) + K * lane;
    const int cols_valid = a.ni - i_lane;
    if (cols_valid <= 0) return;

    const size_t frame = blockIdx.z;
    const char* sbase = static_cast<const char*>(io.src) + frame * io.src_frame_stride;
    const JINC_CONSTANT float* cs =
        (const JINC_CONSTANT float*)(a.coeffs + static_cast<size_t>(a.set[ph]) * (static_cast<size_t>(a.fs) * a.coeff_row));
    const uint32_t lane_off = static_cast<uint32_t>(a.start_x[p] + SX * i_lane) * static_cast<uint32_t>(sizeof(T));
    const int row0 = a.start_y[q] + a.sy * j;
    const int fs = a.fs;

    float acc[R][K];
#pragma unroll
    for (int jj = 0; jj < R; ++jj)
#pragma unroll
        for (int k = 0; k < K; ++k) acc[jj][k] = 0.f;

    // segments + coefficients of the step at kernel row fly, first tap flx
    auto fetch = [&](uint32_t (&raw)[R][RW], float (&cf)[B], int fly, int flx) __attribute__((always_inline)) {
#pragma unroll
        for (int jj = 0; jj < R; ++jj) {
            int r = row0 + a.sy * jj + fly;
            r = r < a.row_clamp ? r : a.row_clamp;  // rows of chains that do not exist (jj >= rows_valid)
            const char* rowp = sbase + static_cast<size_t>(r) * io.src_pitch + flx * static_cast<int>(sizeof(T));
            load_raw<RW>(rowp + lane_off, raw[jj]);
        }
        const JINC_CONSTANT float* c = cs + static_cast<size_t>(fly) * a.coeff_row + flx;
#pragma unroll
        for (int t = 0; t < B; ++t) cf[t] = c[t];  // wave-uniform -> SGPRs (the allocation has slack past the last row)
    };

    if constexpr (PIPE) {
        uint32_t raw_a[R][RW], raw_b[R][RW];
        float cf_a[B], cf_b[B];
        int ly = 0, lx = 0;
        fetch(raw_a, cf_a, 0, 0);
        while (true) {
            int nly = ly, nlx = lx + B;
            if (nlx >= fs) nlx = 0, ++nly;
            const bool more_b = nly < fs;
            if (more_b) fetch(raw_b, cf_b, nly, nlx);
            MacSelect<T, SX, B, RW, B>::run(fs - lx, acc, raw_a, cf_a);
            if (!more_b) break;
            ly = nly, lx = nlx;
            nlx = lx + B;
            if (nlx >= fs) nlx = 0, ++nly;
            const bool more_a = nly < fs;
            if (more_a) fetch(raw_a, cf_a, nly, nlx);
            MacSelect<T, SX, B, RW, B>::run(fs - lx, acc, raw_b, cf_b);
            if (!more_a) break;
            ly = nly, lx = nlx;
        }
    } else {
        for (int ly = 0; ly < fs; ++ly)
            for (int lx = 0; lx < fs; lx += B) {
                uint32_t raw[R][RW];
                float cf[B];
                fetch(raw, cf, ly, lx);
                MacSelect<T, SX, B, RW, B>::run(fs - lx, acc, raw, cf);
            }
    }

    const BufferRsrc drsrc = make_rsrc(static_cast<char*>(io.dst) + frame * io.dst_frame_stride,
                                       static_cast<uint32_t>(io.dst_pitch) * a.dst_h);
    const unsigned x0 = a.ix0 + a.px * i_lane + p;
    const int y0 = a.iy0 + a.py * j + q;
#pragma unroll
    for (int jj = 0; jj < R; ++jj) {
        if (jj < rows_valid) {  // wave-uniform
            const uint32_t soff = static_cast<uint32_t>(y0 + jj * a.py) * io.dst_pitch;
#pragma unroll
            for (int k = 0; k < K; ++k)
                if (k < cols_valid)
                    store_sample_buf<T>(drsrc, (x0 + k * a.px) * static_cast<uint32_t>(sizeof(T)), soff, acc[jj][k], io.peak);
        }
    }
}

// PIPE default: integer planes (their raw segments are 1/4 or 1/2 the size of the converted ones); float planes
// would need 2 x 4 x 20 VGPRs for the second register set.  JINC_DIRECT_PIPE=0/1 overrides for A/B runs.
template <typename T>
bool direct_pipe_default() {
    static const int env = [] {
        const char* e = std::getenv("JINC_DIRECT_PIPE");
        return e ? std::atoi(e) : -1;
    }();
    if (sizeof(T) == 4) return false;
    return env < 0 ? true : env != 0;
}

template <typename T, int SX>
int launch_direct_t(const DirectArgs& da, const PlaneIO& io, hipStream_t stream) {
    const int nchunks = (da.nj + kDirectR - 1) / kDirectR;
    const int items = nchunks * da.px * da.py;
    dim3 grid((da.ni + 64 * kDirectK - 1) / (64 * kDirectK), (items + 3) / 4, io.nframes);
    if constexpr (sizeof(T) != 4) {
        if (direct_pipe_default<T>()) {
            hipLaunchKernelGGL((ewa_periodic_direct_kernel<T, SX, true>), grid, dim3(256, 1, 1), 0, stream, da, io);
            return static_cast<int>(hipGetLastError());
        }
    }
    hipLaunchKernelGGL((ewa_periodic_direct_kernel<T, SX, false>), grid, dim3(256, 1, 1), 0, stream, da, io);
    return static_cast<int>(hipGetLastError());
}

template <typename T>
int launch_direct_sx(const DirectArgs& da, const PlaneIO& io, hipStream_t stream) {
    switch (da.sx) {
        case 1: return launch_direct_t<T, 1>(da, io, stream);
        case 2: return launch_direct_t<T, 2>(da, io, stream);
        case 3: return launch_direct_t<T, 3>(da, io, stream);
        case 4: return launch_direct_t<T, 4>(da, io, stream);
        default: return static_cast<int>(hipErrorInvalidValue);
    }
}

}  // namespace

bool direct_supported(int fs, int px, int py, int sx, int sy) {
    if (fs < 1 || sx < 1 || sx > 4 || sy < 1 || sy > 4) return false;
    return px >= 1 && py >= 1 && px <= 16 && py <= 16 && px * py <= 256;
}

int launch_direct(const DirectArgs& args, const PlaneIO& io, void* stream) {
    if (args.ni <= 0 || args.nj <= 0 || io.nframes <= 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (io.sample_bytes) {
        case 1: return launch_direct_sx<uint8_t>(args, io, s);
        case 2: return launch_direct_sx<uint16_t>(args, io, s);
        default: return launch_direct_sx<float>(args, io, s);
    }
}

}  // namespace jinc
