// kernel_scan.hip -- does a batch of float planes hold nothing but finite samples?
//
// The periodic kernels may run float planes on the TRIMMED support (device_plan.cpp trim_periodic) only where every source
// sample is finite: a tap whose coefficient is 0.0f contributes sample * 0 = +-0 and may be left out -- unless the sample
// is an infinity or a NaN, whose product with 0 is a NaN the reference's chain propagates (ref
// /root/reference/src/JincResize.cpp:570-579 multiplies every tap).  This pass reads every source sample of the call once
// (HBM-bound: 0.3 % of C4's step) and raises flags[frame] when a frame's plane holds a non-finite one; the trimmed launch
// then skips that frame and the full-window launch behind it computes it (PeriodicArgs::frame_flags / run_when).
#include "device_common.hpp"

namespace jinc {
namespace {

constexpr int kScanThreads = 256;
constexpr int kScanRowsPerBlock = 8;

__global__ __launch_bounds__(kScanThreads) void finite_scan_kernel(const char* __restrict__ base, uint32_t pitch, size_t frame_stride, int w,
                                                                   int h, uint32_t* __restrict__ flags) {
    const size_t frame = blockIdx.z;
    const int y0 = blockIdx.y * kScanRowsPerBlock;
    const char* plane = base + frame * frame_stride;
    uint32_t bad = 0;
    for (int r = 0; r < kScanRowsPerBlock; ++r) {
        const int y = y0 + r;
        if (y >= h) break;
        const uint32_t* row = reinterpret_cast<const uint32_t*>(plane + static_cast<size_t>(y) * pitch);
        for (int x = blockIdx.x * kScanThreads + threadIdx.x; x < w; x += gridDim.x * kScanThreads)
            bad |= ((row[x] & 0x7f800000u) == 0x7f800000u) ? 1u : 0u;  // exponent all ones: infinity or NaN
    }
    if (__builtin_amdgcn_ballot_w64(bad != 0) != 0 && (threadIdx.x & 63) == 0) flags[frame] = 1u;  // (every writer writes 1)
}

}  // namespace

// flags[0 .. io.nframes) must be zero before the launch (the caller clears them on the same stream).
int launch_finite_scan(const PlaneIO& io, int w, int h, uint32_t* flags, void* stream) {
    if (w <= 0 || h <= 0 || io.nframes <= 0) return 0;
    const int bx = std::max(1, std::min(8, (w + kScanThreads * 4 - 1) / (kScanThreads * 4)));
    dim3 grid(bx, (h + kScanRowsPerBlock - 1) / kScanRowsPerBlock, io.nframes);
    hipLaunchKernelGGL(finite_scan_kernel, grid, dim3(kScanThreads), 0, static_cast<hipStream_t>(stream),
                       static_cast<const char*>(io.src), static_cast<uint32_t>(io.src_pitch), io.src_frame_stride, w, h, flags);
    return static_cast<int>(hipGetLastError());
}

}  // namespace jinc
