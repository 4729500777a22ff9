// kernel_direct.hip -- launchers of ewa_direct_kernel (kernel_direct_impl.inc): the row-strip mode and the round-1
// interior form (DirectShape 0) for every sample type and source step live in this unit; the row-walk interior forms
// (DirectShape 2 / 3) are instantiated per sample type and source step in kernel_direct_walk_*_sx*.hip.
#include <atomic>

#include "kernel_direct_impl.inc"

namespace jinc {
namespace {

template <typename T, int MODE>
int launch_direct_sx(const DirectArgs& da, const PlaneIO& io, hipStream_t stream) {
    switch (da.sx) {
        case 1: return launch_direct_shape<T, 1, MODE, 0>(da, io, stream);
        case 2: return launch_direct_shape<T, 2, MODE, 0>(da, io, stream);
        case 3: return launch_direct_shape<T, 3, MODE, 0>(da, io, stream);
        case 4: return launch_direct_shape<T, 4, MODE, 0>(da, io, stream);
        default: return static_cast<int>(hipErrorInvalidValue);
    }
}

template <int MODE>
int launch_direct_mode(const DirectArgs& args, const PlaneIO& io, hipStream_t s) {
    switch (io.sample_bytes) {
        case 1: return launch_direct_sx<uint8_t, MODE>(args, io, s);
        case 2: return launch_direct_sx<uint16_t, MODE>(args, io, s);
        default: return launch_direct_sx<float, MODE>(args, io, s);
    }
}

// Interior shape of a launch (DirectShape): row walk where its step scheme covers the filter size; 8 columns per lane
// for 8-bit down-scales where that still fills the device with waves.  set_direct_shape (test header: knob DIRECT_SHAPE)
// = 0 / 2 / 3 forces one (A/B runs).
std::atomic<int> g_last_shape{-1};    // shape of the most recent interior launch (test hook)
std::atomic<int> g_forced_shape{-1};  // -1: automatic, 0 / 2 / 3: forced (set_direct_shape)

int interior_shape(const DirectArgs& da, const PlaneIO& io) {
    const int forced = g_forced_shape.load(std::memory_order_relaxed);
    if (!walk_supported(da.fs) || forced == 0) return 0;
    if (forced == 2 || (forced == 3 && walk_wide_supported(da.fs, da.sx))) return forced;
    if (da.fs < kWalkMinTapsPeriodic) return 0;
    if (!walk_wide_supported(da.fs, da.sx)) return 2;
    const long long waves8 = (static_cast<long long>((da.ni + 7) / 8) * ((da.nj + 3) / 4) + 63) / 64 * da.px * da.py * io.nframes;
    // 8-bit only: 16-bit and float planes hold too many raw words per lane at 8 columns (measured 4K -> 1080p:
    // 8-bit 146 against 138 Gpix/s, 16-bit 71 against 73, float 31 against 35)
    const bool wide = io.sample_bytes == 1 && waves8 >= 6144;  // one full residency of the device (6 waves per SIMD)
    return wide ? 3 : 2;
}

// See load_raw: lane l reads dword l of a 2N-byte buffer through a descriptor of N bytes with soffset = N - 128 and with
// soffset = N + 256; the range check must zero lanes 32.. of the first read and all of the second.
__global__ void soffset_probe_kernel(const uint32_t* buf, uint32_t nbytes, uint32_t* out) {
    const BufferRsrc r = make_rsrc(const_cast<uint32_t*>(buf), nbytes);
    const uint32_t l = threadIdx.x;
    out[l] = __builtin_amdgcn_raw_buffer_load_b32(r, 4 * l, nbytes - 128, 0);
    out[64 + l] = __builtin_amdgcn_raw_buffer_load_b32(r, 4 * l, nbytes + 256, 0);
}
}  // namespace

void set_direct_shape(int shape) { g_forced_shape.store(shape == 0 || shape == 2 || shape == 3 ? shape : -1, std::memory_order_relaxed); }

int last_direct_shape() { return g_last_shape.load(std::memory_order_relaxed); }

int launch_soffset_probe(const uint32_t* buf, uint32_t nbytes, uint32_t* out, void* stream) {
    hipLaunchKernelGGL(soffset_probe_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), buf, nbytes, out);
    return static_cast<int>(hipGetLastError());
}

bool direct_supported(int fs, int px, int py, int sx, int sy) {
    if (fs < 1 || sx < 1 || sx > 4 || sy < 1 || sy > 4) return false;
    return px >= 1 && py >= 1 && px <= 16 && py <= 16 && px * py <= 256;
}

int launch_direct(const DirectArgs& args, const PlaneIO& io, void* stream) {
    if (args.ni <= 0 || args.nj <= 0 || io.nframes <= 0) return 0;
    const int shape = interior_shape(args, io);
    g_last_shape.store(shape, std::memory_order_relaxed);
    if (shape == 0) return launch_direct_mode<kDirectInterior>(args, io, static_cast<hipStream_t>(stream));
#define JINC_WALK_BY_SX(tag)                                                          \
    switch (args.sx) {                                                                \
        case 1: return launch_direct_walk_##tag##_sx1(args, io, stream, shape);       \
        case 2: return launch_direct_walk_##tag##_sx2(args, io, stream, shape);       \
        case 3: return launch_direct_walk_##tag##_sx3(args, io, stream, shape);       \
        case 4: return launch_direct_walk_##tag##_sx4(args, io, stream, shape);       \
        default: return static_cast<int>(hipErrorInvalidValue);                       \
    }
    switch (io.sample_bytes) {
        case 1: JINC_WALK_BY_SX(u8)
        case 2: JINC_WALK_BY_SX(u16)
        default: JINC_WALK_BY_SX(f32)
    }
#undef JINC_WALK_BY_SX
}

bool direct_runs_supported(int fs, int px, int py, int sx, int sy) { return direct_supported(fs, px, py, sx, sy) && walk_supported(fs); }

int launch_direct_runs(const DirectArgs& args, const PlaneIO& io, void* stream) {
    if (args.n_items <= 0 || io.nframes <= 0) return 0;
    if (!args.runs || !args.item_run || !walk_supported(args.fs)) return static_cast<int>(hipErrorInvalidValue);
#define JINC_RUNS_SX(tag)                                                      \
    switch (args.sx) {                                                         \
        case 1: return launch_direct_runs_##tag##_sx1(args, io, stream);       \
        case 2: return launch_direct_runs_##tag##_sx2(args, io, stream);       \
        case 3: return launch_direct_runs_##tag##_sx3(args, io, stream);       \
        case 4: return launch_direct_runs_##tag##_sx4(args, io, stream);       \
        default: return static_cast<int>(hipErrorInvalidValue);                \
    }
    switch (io.sample_bytes) {
        case 1: JINC_RUNS_SX(u8)
        case 2: JINC_RUNS_SX(u16)
        default: JINC_RUNS_SX(f32)
    }
#undef JINC_RUNS_SX
}

int launch_direct_row_strips(const DirectArgs& args, const PlaneIO& io, void* stream) {
    if (args.ni <= 0 || io.nframes <= 0) return 0;
    return launch_direct_mode<kDirectRowStrip>(args, io, static_cast<hipStream_t>(stream));
}

}  // namespace jinc
