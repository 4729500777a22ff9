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

// The plane minus the rectangle [rx0, rx1) x [ry0, ry1): rows above and below it over the whole width, columns left and right of
// it over its height.  A few thousand samples per frame: one workgroup per frame walks them.
__global__ __launch_bounds__(kScanThreads) void finite_scan_outside_kernel(const char* __restrict__ base, uint32_t pitch, size_t frame_stride, int w,
                                                                           int h, int rx0, int ry0, int rx1, int ry1,
                                                                           uint32_t* __restrict__ flags) {
    const size_t frame = blockIdx.z;
    const char* plane = base + frame * frame_stride;
    uint32_t bad = 0;
    auto check = [&](int x, int y) {
        const uint32_t v = *reinterpret_cast<const uint32_t*>(plane + static_cast<size_t>(y) * pitch + static_cast<size_t>(x) * 4u);
        bad |= ((v & 0x7f800000u) == 0x7f800000u) ? 1u : 0u;
    };
    const int tid = blockIdx.x * kScanThreads + threadIdx.x, nthreads = gridDim.x * kScanThreads;
    const int rows_above = ry0, rows_below = h - ry1, cols_left = rx0, cols_right = w - rx1, mid = ry1 - ry0;
    for (long long i = tid; i < static_cast<long long>(rows_above) * w; i += nthreads) check(static_cast<int>(i % w), static_cast<int>(i / w));
    for (long long i = tid; i < static_cast<long long>(rows_below) * w; i += nthreads) check(static_cast<int>(i % w), ry1 + static_cast<int>(i / w));
    for (long long i = tid; i < static_cast<long long>(cols_left) * mid; i += nthreads) check(static_cast<int>(i % cols_left), ry0 + static_cast<int>(i / cols_left));
    for (long long i = tid; i < static_cast<long long>(cols_right) * mid; i += nthreads)
        check(rx1 + static_cast<int>(i % cols_right), ry0 + static_cast<int>(i / cols_right));
    if (__builtin_amdgcn_ballot_w64(bad != 0) != 0 && (threadIdx.x & 63) == 0) flags[frame] = 1u;
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

int launch_finite_scan_outside(const PlaneIO& io, int w, int h, int rx0, int ry0, int rx1, int ry1, uint32_t* flags, void* stream) {
    if (w <= 0 || h <= 0 || io.nframes <= 0) return 0;
    rx0 = std::max(0, std::min(rx0, w)), rx1 = std::max(rx0, std::min(rx1, w));
    ry0 = std::max(0, std::min(ry0, h)), ry1 = std::max(ry0, std::min(ry1, h));
    const long long outside = static_cast<long long>(w) * h - static_cast<long long>(rx1 - rx0) * (ry1 - ry0);
    if (outside <= 0) return 0;
    const int bx = static_cast<int>(std::max<long long>(1, std::min<long long>(16, outside / (kScanThreads * 16))));
    hipLaunchKernelGGL(finite_scan_outside_kernel, dim3(bx, 1, io.nframes), dim3(kScanThreads), 0, static_cast<hipStream_t>(stream),
                       static_cast<const char*>(io.src), static_cast<uint32_t>(io.src_pitch), io.src_frame_stride, w, h, rx0, ry0, rx1, ry1, flags);
    return static_cast<int>(hipGetLastError());
}

}  // namespace jinc
