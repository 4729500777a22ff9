// kernel_strip.hip -- ewa_strip_kernel: border rows and border columns of exactly periodic plans (filter sizes up to 9) with ONE
// register window per lane for the whole thickness of the strip.
//
// The border pixels of a plan are those whose window the reference shifted back inside the image (ref
// /root/reference/src/JincResize.cpp:395-418): every border ROW of the top strip reads the source rows [0, fs), every row of the
// bottom strip [src_h - fs, src_h) -- the rows of a strip differ in their coefficient sets only, and along the row the sets repeat
// with the interior's period (device_plan.cpp plan_direct: strips_ok).  So a lane that owns output column x of the strip loads its
// fs x fs window ONCE and walks the strip's rows with it: per row one coefficient set (wave-uniform: scalar loads into SGPRs) and
// the reference's chain, taps in (ly, lx) order, multiply and add un-fused (ref :570-579).  Border COLUMNS are the same thing
// transposed: lanes along y, the strip's columns share the source columns [0, fs) / [src_w - fs, src_w).
// Round 5: the kernels this replaces ran the columns at 0.10 of the VALU peak (the frame-lane kernel on 8-pixel-wide strips:
// 0.283 ms per 1024 C2 frames for 3.5 G operations) and the rows at 0.42 (ewa_direct_kernel, one output row per wave, every row
// re-fetching its window: 0.184 ms for 6.1 G).
//
// A workgroup = 256 lanes along the strip (four waves, 64 periods each).  It stages the fs source lines of the strip over its
// span as fp32 in LDS, laid out [line across the strip][position along it], so that for both orientations the lanes of a wave read
// consecutive words; a tap (ly, lx) is word [ly][o + lx] for row strips and [lx][o + ly] for column strips.
#include "device_common.hpp"
#include "knobs.h"

#pragma clang fp contract(off)

namespace jinc {
namespace {

constexpr int kStripLanes = 256;

template <int FS>
struct StripCfg {
    // source step 1 along the strip (up-scales: the plans of ewa_periodic_*; down-scales keep the direct kernel's strips)
    static constexpr int kAlong = kStripLanes + FS + 1;  // positions staged along the strip (+ 1 phase spread)
    static constexpr int kPitch = kAlong | 1;
};

template <typename T, int FS, int AXIS>
__global__ __launch_bounds__(kStripLanes) void ewa_strip_kernel(const StripArgs a, const PlaneIO io) {
    using Cfg = StripCfg<FS>;
    __shared__ float strip_lds[FS * Cfg::kPitch];
    const int lane = threadIdx.x;         // 0 .. 255: the workgroup's position along the strip, in periods
    const int g = blockIdx.y;             // which strip (top / bottom rows, left / right columns)
    const size_t frame = blockIdx.z;
    const int i_first = static_cast<int>(blockIdx.x) * kStripLanes;  // first period of the workgroup
    constexpr int along = Cfg::kAlong, pitch = Cfg::kPitch;
    const int a0 = a.min_start + i_first;                            // source position of staged word 0 along the strip
    const int origin = a.origin[g];                                  // first source line across the strip
    {   // stage: word [k][m] = source(line origin + k across, position a0 + m along), clamped to the plane like every kernel's halo;
        // all loads in front of the LDS writes
        const char* sbase = static_cast<const char*>(io.src) + frame * io.src_frame_stride;
        if constexpr (AXIS == 0) {
            constexpr int n = FS * along, kPer = (n + kStripLanes - 1) / kStripLanes;
            T staged[kPer];
#pragma unroll
            for (int r = 0; r < kPer; ++r) {
                const int e = min(r * kStripLanes + lane, n - 1);
                const int k = e / along, m = e - k * along;
                int gx = a0 + m, gy = origin + k;
                gx = gx < a.src_w ? gx : a.src_w - 1;
                gy = gy < a.src_h ? gy : a.src_h - 1;
                staged[r] = reinterpret_cast<const T*>(sbase + static_cast<size_t>(gy) * io.src_pitch)[gx];
            }
#pragma unroll
            for (int r = 0; r < kPer; ++r) {
                const int e = r * kStripLanes + lane;
                if (e < n) strip_lds[(e / along) * pitch + (e % along)] = to_float(staged[r]);
            }
        } else {
            // column strips: a lane takes whole source ROWS -- the fs samples of a row are one cache line, fetched with vector loads of
            // four samples (the last one overlapping: every load stays inside [origin, origin + fs)) instead of fs single samples by fs
            // different lanes: a third of the first form's requests, each row's line asked for once
            typedef T T4 __attribute__((ext_vector_type(4)));
            constexpr int kPer = (along + kStripLanes - 1) / kStripLanes;  // rows per lane (2: the second pass holds the halo)
            constexpr int kVec = (FS + 3) / 4;
            T4 staged[kPer][kVec];
            const BufferRsrc srsrc = make_rsrc(const_cast<char*>(sbase), static_cast<uint32_t>(io.src_pitch) * static_cast<uint32_t>(a.src_h - 1) +
                                                                             static_cast<uint32_t>(a.src_w) * static_cast<uint32_t>(sizeof(T)));
#pragma unroll
            for (int r = 0; r < kPer; ++r) {
                int gy = a0 + min(r * kStripLanes + lane, along - 1);
                gy = gy < a.src_h ? gy : a.src_h - 1;
                // raw buffer loads: one request of 4 / 8 / 2 x 8 bytes per four samples at any alignment (through a pointer the
                // compiler splits a 2-byte-aligned vector into single samples)
                const uint32_t rowoff = static_cast<uint32_t>(gy) * static_cast<uint32_t>(io.src_pitch) + static_cast<uint32_t>(origin) * static_cast<uint32_t>(sizeof(T));
#pragma unroll
                for (int v = 0; v < kVec; ++v) {
                    const int k0 = 4 * v + 4 <= FS ? 4 * v : FS - 4;  // (origin + fs <= src_w: host)
                    const uint32_t off = rowoff + static_cast<uint32_t>(k0 * sizeof(T));
                    if constexpr (sizeof(T) == 1) {
                        staged[r][v] = __builtin_bit_cast(T4, __builtin_amdgcn_raw_buffer_load_b32(srsrc, off, 0, 0));
                    } else if constexpr (sizeof(T) == 2) {
                        staged[r][v] = __builtin_bit_cast(T4, __builtin_amdgcn_raw_buffer_load_b64(srsrc, off, 0, 0));
                    } else {
                        // (single dwords: the compiler merges two 8-byte loads into one of 16 bytes, which at a 4-byte boundary does not
                        // return its dwords where they belong -- as the 16-byte stores of kernel_periodic.hip)
                        staged[r][v] = T4{__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srsrc, off, 0, 0)),
                                          __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srsrc, off + 4u, 0, 0)),
                                          __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srsrc, off + 8u, 0, 0)),
                                          __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srsrc, off + 12u, 0, 0))};
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < kPer; ++r) {
                const int m = r * kStripLanes + lane;
                if (m < along) {
#pragma unroll
                    for (int k = 0; k < FS; ++k) {
                        const int v = k / 4 < kVec - 1 || FS % 4 == 0 ? k / 4 : kVec - 1;
                        const int k0 = 4 * v + 4 <= FS ? 4 * v : FS - 4;
                        strip_lds[k * pitch + m] = to_float(staged[r][v][k - k0]);
                    }
                }
            }
        }
    }
    __syncthreads();
    const int i = i_first + lane;  // the lane's period
    if (i >= a.ni) return;         // no barrier below
    const BufferRsrc drsrc = make_rsrc(static_cast<char*>(io.dst) + frame * io.dst_frame_stride, static_cast<uint32_t>(io.dst_pitch) * a.dst_h);
    const int nlines = a.nlines[g], line0 = a.line0[g];
    const JINC_CONSTANT int32_t* sets = (const JINC_CONSTANT int32_t*)(a.sets) + static_cast<size_t>(a.set_base[g]) * a.P;
    for (int p = 0; p < a.P; ++p) {  // the phases along the strip: one window each
        const float* wp = strip_lds + (a.start[p] - a.min_start) + lane;
        float w[FS][FS];  // w[ly][lx]
#pragma unroll
        for (int k = 0; k < FS; ++k)
#pragma unroll
            for (int m = 0; m < FS; ++m) {
                const float v = wp[k * pitch + m];  // line k across the strip, position m along it
                if constexpr (AXIS == 0) w[k][m] = v; else w[m][k] = v;
            }
        const int c_along = a.i0 + a.P * i + p;  // the lane's output coordinate along the strip
        auto chain = [&](int line) {  // the reference's chain of output (line, c_along) on the lane's window
            const int set = sets[line * a.P + p];  // wave-uniform
            const JINC_CONSTANT float* cs = (const JINC_CONSTANT float*)(a.coeffs + static_cast<size_t>(set) * (FS * padded_row(FS)));
            float c[FS * FS];
#pragma unroll
            for (int k = 0; k < FS * FS; ++k) c[k] = cs[(k / FS) * padded_row(FS) + (k % FS)];
            float acc = 0.f;
#pragma unroll
            for (int ly = 0; ly < FS; ++ly)
#pragma unroll
                for (int lx = 0; lx < FS; ++lx) acc = acc + w[ly][lx] * c[ly * FS + lx];
            return acc;
        };
        if constexpr (AXIS == 0) {
            for (int line = 0; line < nlines; ++line)
                store_sample_buf<T>(drsrc, static_cast<uint32_t>(c_along) * static_cast<uint32_t>(sizeof(T)),
                                    static_cast<uint32_t>(line0 + line) * static_cast<uint32_t>(io.dst_pitch), chain(line), io.peak);
        } else {
            // column strips: the strip's columns are ADJACENT samples of the lane's output row -- four of them leave as one store
            // (a store per sample is 64 lanes x 64 different rows x one byte: the first form's columns took 0.49 ms per 1024 C2
            // frames against the frame-lane kernel's 0.28)
            const uint32_t soff = static_cast<uint32_t>(c_along) * static_cast<uint32_t>(io.dst_pitch);
            for (int l4 = 0; l4 < nlines; l4 += 4) {
                float r[4];
#pragma unroll
                for (int l = 0; l < 4; ++l) r[l] = l4 + l < nlines ? chain(l4 + l) : 0.f;  // (wave-uniform)
                const uint32_t xoff = static_cast<uint32_t>(line0 + l4) * static_cast<uint32_t>(sizeof(T));
                if (l4 + 4 <= nlines) {
                    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                    if constexpr (std::is_same_v<T, uint8_t>) {
                        uint32_t v = __builtin_amdgcn_cvt_pk_u8_f32(r[0], 0u, 0u);
                        v = __builtin_amdgcn_cvt_pk_u8_f32(r[1], 1u, v);
                        v = __builtin_amdgcn_cvt_pk_u8_f32(r[2], 2u, v);
                        v = __builtin_amdgcn_cvt_pk_u8_f32(r[3], 3u, v);
                        __builtin_amdgcn_raw_buffer_store_b32(v, drsrc, xoff, soff, 0);
                    } else if constexpr (std::is_same_v<T, uint16_t>) {
                        const u32x2 v = {round_sample(r[0], io.peak) | (round_sample(r[1], io.peak) << 16),
                                         round_sample(r[2], io.peak) | (round_sample(r[3], io.peak) << 16)};
                        __builtin_amdgcn_raw_buffer_store_b64(v, drsrc, xoff, soff, 0);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b64(u32x2{__builtin_bit_cast(uint32_t, r[0]), __builtin_bit_cast(uint32_t, r[1])}, drsrc, xoff, soff, 0);
                        __builtin_amdgcn_raw_buffer_store_b64(u32x2{__builtin_bit_cast(uint32_t, r[2]), __builtin_bit_cast(uint32_t, r[3])}, drsrc, xoff + 8u, soff, 0);
                    }
                } else {
#pragma unroll
                    for (int l = 0; l < 3; ++l)
                        if (l4 + l < nlines) store_sample_buf<T>(drsrc, xoff + static_cast<uint32_t>(l * sizeof(T)), soff, r[l], io.peak);
                }
            }
        }
    }
}

template <typename T, int FS>
int launch_strip_fs(const StripArgs& a, const PlaneIO& io, hipStream_t stream) {
    dim3 grid(static_cast<unsigned>((a.ni + kStripLanes - 1) / kStripLanes), static_cast<unsigned>(a.ngroups), static_cast<unsigned>(io.nframes));
    if (a.axis == 0)
        hipLaunchKernelGGL((ewa_strip_kernel<T, FS, 0>), grid, dim3(kStripLanes), 0, stream, a, io);
    else
        hipLaunchKernelGGL((ewa_strip_kernel<T, FS, 1>), grid, dim3(kStripLanes), 0, stream, a, io);
    return static_cast<int>(hipGetLastError());
}

template <typename T>
int launch_strip_t(const StripArgs& a, const PlaneIO& io, hipStream_t stream) {
    switch (a.fs) {
        case 5: return launch_strip_fs<T, 5>(a, io, stream);
        case 7: return launch_strip_fs<T, 7>(a, io, stream);
        case 9: return launch_strip_fs<T, 9>(a, io, stream);
        default: return static_cast<int>(hipErrorInvalidValue);
    }
}

}  // namespace

bool strip_supported(int fs, int period, int step, int spread) { return (fs == 5 || fs == 7 || fs == 9) && period >= 1 && period <= 16 && step == 1 && spread >= 0 && spread <= 1; }

int launch_strip(const StripArgs& args, const PlaneIO& io, void* stream) {
    if (args.ngroups <= 0 || args.ni <= 0 || io.nframes <= 0) return 0;
    if (!strip_supported(args.fs, args.P, args.S, args.spread) || args.ngroups > 4 || !args.sets) return static_cast<int>(hipErrorInvalidValue);
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (io.sample_bytes) {
        case 1: return launch_strip_t<uint8_t>(args, io, s);
        case 2: return launch_strip_t<uint16_t>(args, io, s);
        default: return launch_strip_t<float>(args, io, s);
    }
}

}  // namespace jinc
