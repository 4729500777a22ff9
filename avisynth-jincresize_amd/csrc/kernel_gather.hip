// kernel_gather.hip -- ewa_gather_kernel: any plan (border frame of periodic plans, everything for ratios
// without structure), plus the conversion test hook.  See device_common.hpp for the parity rules.
#include "device_common.hpp"
#include "knobs.h"

#pragma clang fp contract(off)

namespace jinc {
namespace {

// ------------------------------------------------------------------------------------------------
// Generic gather kernel
// ------------------------------------------------------------------------------------------------
struct GatherArgs {
    DevicePlan plan;
    PlaneIO io;
    RectList rects;
    int block_begin[5];
    int blocks_a[4];  // blocks along the lane axis of each rectangle
    int lane_axis[4]; // 0: lanes run along x (wide rectangles), 1: along y (border columns)
    int stride[4];    // lane stride P: lane l of an item handles coordinate origin + P*l + residue
    int lines[4];     // lines (rows for lane_axis 0, columns for lane_axis 1) per block
    int max_passes;   // uniform-coefficient passes per item before the per-lane fallback
};

constexpr int kGatherLdsFloats = 6144;  // 24 KB source tile per block

// One sequential chain of fs x fs taps with a run-time filter size: samples from the LDS tile (STAGED) or from
// memory, coefficients through `c` (wave-uniform constant-address-space pointer -> scalar loads into SGPRs, or a
// per-lane pointer), eight per step; rows are padded to a multiple of 4 floats and the allocation has slack, so a
// whole step may always be fetched.  Taps are visited in (ly, lx) raster order.
template <typename T, bool STAGED, typename CoeffPtr>
__device__ __forceinline__ float chain_runtime(float acc, const float* s_lds, int lds_pitch, const char* s_glb, int src_pitch,
                                               CoeffPtr c, int fs, int fsp) {
    for (int ly = 0; ly < fs; ++ly) {
        const T* g = reinterpret_cast<const T*>(s_glb);
        for (int lx = 0; lx < fs; lx += 8) {
            float cf[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) cf[t] = c[lx + t];
            const int n = fs - lx;  // wave-uniform
            if (n >= 8) {
#pragma unroll
                for (int t = 0; t < 8; ++t) acc = acc + (STAGED ? s_lds[lx + t] : to_float(g[lx + t])) * cf[t];
            } else {
#pragma unroll
                for (int t = 0; t < 7; ++t)
                    if (t < n) acc = acc + (STAGED ? s_lds[lx + t] : to_float(g[lx + t])) * cf[t];
            }
        }
        s_lds += lds_pitch;
        s_glb += src_pitch;
        c += fsp;
    }
    return acc;
}

// The same chain with the lane's private coefficients read from a lane-major copy: `c` points at the lane's first
// group of 4 taps, consecutive groups are 256 floats apart (64 lanes x 4), a kernel row has fsp / 4 groups.
template <typename T, bool STAGED>
__device__ __forceinline__ float chain_lane_major(float acc, const float* s_lds, int lds_pitch, const char* s_glb, int src_pitch,
                                                  const float* c, int fs, int fsp) {
    for (int ly = 0; ly < fs; ++ly) {
        const T* g = reinterpret_cast<const T*>(s_glb);
        for (int lx = 0; lx < fs; lx += 4) {
            const float4 v = *reinterpret_cast<const float4*>(c);
            const float cf[4] = {v.x, v.y, v.z, v.w};
            const int n = fs - lx;  // wave-uniform
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (t < n) acc = acc + (STAGED ? s_lds[lx + t] : to_float(g[lx + t])) * cf[t];
            c += 256;
        }
        s_lds += lds_pitch;
        s_glb += src_pitch;
    }
    return acc;
}

// Any plan.  A block covers 64*P coordinates along the lane axis x 4..32 lines; its source footprint is
// staged once in LDS as fp32.  Work items = (line, residue): the 64 lanes of an item are P apart, P
// being the plan's dominant phase period, so that (nearly) all lanes of an item share one coefficient
// set.  The item then runs a waterfall over the distinct sets actually present: the set of the first
// pending lane is made wave-uniform (readlane), its coefficients are fetched with scalar loads into
// SGPRs and every lane that uses this set runs its sequential chain.  For exactly periodic plans
// that is one pass; ratios whose phases drift (1.5x, 3x: the reference accumulates positions in
// float) add a pass per deviation.  There is no per-lane coefficient traffic.
template <typename T, int FS>
__global__ __launch_bounds__(256) void ewa_gather_kernel(const GatherArgs a) {
    __shared__ float tile[kGatherLdsFloats];
    const DevicePlan& p = a.plan;
    // Block id -> (tile, frame).  A batch of frames shares the plan, and a tile's coefficient sets (up to one per
    // pixel for plans without phase structure: 196 B per pixel at fs 7) are the dominant traffic of this kernel, so
    // all frames of a tile run back to back ON THE SAME XCD (workgroups are dealt round-robin over the 8 XCDs by
    // linear id): the sets are then fetched into that XCD's L2 once per batch instead of once per frame.
    const unsigned nf = static_cast<unsigned>(a.io.nframes);
    const unsigned xcd = blockIdx.x % 8u, slot = blockIdx.x / 8u;
    const int b = static_cast<int>((slot / nf) * 8u + xcd);
    if (b >= a.block_begin[4]) return;  // padding ids (tile count not a multiple of 8); whole block, before any barrier
    int r = 0;
    while (r + 1 < a.rects.n && b >= a.block_begin[r + 1]) ++r;
    const int local = b - a.block_begin[r];
    const int axis = a.lane_axis[r];
    const int P = a.stride[r];
    const int ba = local % a.blocks_a[r];  // block index along the lane axis
    const int bl = local / a.blocks_a[r];  // block index along the line axis
    const int rx0 = a.rects.x0[r], ry0 = a.rects.y0[r];
    const int rx1 = rx0 + a.rects.w[r], ry1 = ry0 + a.rects.h[r];
    // block extent in output pixels (inclusive last pixel), all wave-uniform
    const int nlines = a.lines[r];
    const int bx0 = axis == 0 ? rx0 + ba * 64 * P : rx0 + bl * nlines;
    const int by0 = axis == 0 ? ry0 + bl * nlines : ry0 + ba * 64 * P;
    const int bx1 = min(bx0 + (axis == 0 ? 64 * P : nlines), rx1) - 1;
    const int by1 = min(by0 + (axis == 0 ? nlines : 64 * P), ry1) - 1;

    const int fs = FS ? FS : p.fs;
    const int fsp = FS ? padded_row(FS) : padded_row(p.fs);
    const size_t frame = slot % nf;
    const char* sframe = static_cast<const char*>(a.io.src) + frame * a.io.src_frame_stride;
    char* dframe = static_cast<char*>(a.io.dst) + frame * a.io.dst_frame_stride;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // Source footprint of the block (window origins are non-decreasing in x and in y).
    const int tx0 = p.col_start[bx0], ty0 = p.row_start[by0];
    const int tw = p.col_start[bx1] + fs - tx0;
    const int th = p.row_start[by1] + fs - ty0;
    const int pitch = tw | 1;
    const bool staged = pitch * th <= kGatherLdsFloats;  // wave-uniform; huge down-scales read global memory
    if (staged) {
        // four rows per wave and pass, loads in front of the LDS writes (one memory round trip per pass, not per row)
        for (int r0 = wave; r0 < th; r0 += 16) {
            for (int c = lane; c < tw; c += 64) {
                T v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int rr = r0 + 4 * i < th ? r0 + 4 * i : th - 1;
                    v[i] = (reinterpret_cast<const T*>(sframe + static_cast<size_t>(ty0 + rr) * a.io.src_pitch) + tx0)[c];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (r0 + 4 * i < th) tile[(r0 + 4 * i) * pitch + c] = to_float(v[i]);
            }
        }
    }
    __syncthreads();

    const int nitems = nlines * P;
    for (int item = wave; item < nitems; item += 4) {
        const int line = item / P;
        const int res = item - line * P;
        const int x = axis == 0 ? bx0 + P * lane + res : bx0 + line;
        const int y = axis == 0 ? by0 + line : by0 + P * lane + res;
        const bool active = x <= bx1 && y <= by1;
        if (!__builtin_amdgcn_readfirstlane(__ballot(active) != 0)) continue;
        // number of this item in the rectangle's lane-major coefficient copy (see gather_item_count in kernels.h)
        const long long lane_item = a.rects.lane_item_base[r] + static_cast<long long>((bl * nlines + line) * P + res) * a.blocks_a[r] + ba;

        // Inactive lanes (block overhang) look up the block's first pixel so every address stays in range.
        const int qx = active ? x : bx0, qy = active ? y : by0;
        const int sx = p.col_start[qx];
        const int sy = p.row_start[qy];
        const int rc = p.row_class[qy];
        const int cc = p.col_class[qx];
        int set;
        if (rc < 0)
            set = p.brow_set[static_cast<size_t>(~rc) * p.dst_w + qx];
        else if (cc < 0)
            set = p.bcol_set[static_cast<size_t>(~cc) * p.dst_h + qy];
        else
            set = p.interior_set[rc * p.n_col_classes + cc];

        float acc = 0.f;
        if constexpr (FS != 0) {
            if (staged) {
                const float* s = tile + (sy - ty0) * pitch + (sx - tx0);
                constexpr bool kWindowInRegs = FS <= 9;  // small windows: read LDS once, reuse across passes
                float w[kWindowInRegs ? FS * FS : 1];
                if constexpr (kWindowInRegs) {
#pragma unroll
                    for (int ly = 0; ly < FS; ++ly)
#pragma unroll
                        for (int lx = 0; lx < FS; ++lx) w[ly * FS + lx] = s[ly * pitch + lx];
                }
                unsigned long long todo = __ballot(active);
                // at most a.max_passes uniform passes; items whose lanes nearly all own a private set (border
                // pixels of drifting ratios) finish with per-lane coefficient loads instead of 64 passes
                for (int pass = 0; todo && pass < a.max_passes; ++pass) {
                    const int leader = __ffsll(static_cast<long long>(todo)) - 1;
                    const int u = __builtin_amdgcn_readlane(set, leader);
                    const bool mine = active && set == u;
                    const JINC_CONSTANT float* cs =
                        (const JINC_CONSTANT float*)(p.coeffs + static_cast<size_t>(u) * (FS * padded_row(FS)));
                    if (mine) {
                        if constexpr (kWindowInRegs) {
#pragma unroll
                            for (int ly = 0; ly < FS; ++ly)
#pragma unroll
                                for (int lx = 0; lx < FS; ++lx)
                                    acc = acc + w[ly * FS + lx] * cs[ly * padded_row(FS) + lx];
                        } else {
                            const float* sr = s;
                            for (int ly = 0; ly < FS; ++ly) {
                                float c[FS];
#pragma unroll
                                for (int lx = 0; lx < FS; ++lx) c[lx] = cs[ly * padded_row(FS) + lx];
#pragma unroll
                                for (int lx = 0; lx < FS; ++lx) acc = acc + sr[lx] * c[lx];
                                sr += pitch;
                            }
                        }
                    }
                    todo &= ~__ballot(mine);
                }
                if (todo) {
                    // private sets: 16-byte per-lane coefficient loads -- from the lane-major copy when the host
                    // built one (consecutive lanes, consecutive addresses), else from the lane's own set
                    const bool lane_major = a.rects.lane_coeffs != nullptr;  // wave-uniform
                    const float* c = lane_major ? a.rects.lane_coeffs + static_cast<size_t>(lane_item) * (FS * padded_row(FS) * 64) + 4 * lane
                                                : p.coeffs + static_cast<size_t>(set) * (FS * padded_row(FS));
                    const int cstep = lane_major ? 256 : 4;  // floats between a lane's consecutive groups of 4 taps
                    if (active && ((todo >> lane) & 1ull)) {
                        const float* sr = s;
                        for (int ly = 0; ly < FS; ++ly) {
                            float cr[padded_row(FS)];
#pragma unroll
                            for (int k = 0; k < padded_row(FS) / 4; ++k) {
                                const float4 v = *reinterpret_cast<const float4*>(c + cstep * k);
                                cr[4 * k] = v.x, cr[4 * k + 1] = v.y, cr[4 * k + 2] = v.z, cr[4 * k + 3] = v.w;
                            }
#pragma unroll
                            for (int lx = 0; lx < FS; ++lx) acc = acc + sr[lx] * cr[lx];
                            sr += pitch;
                            c += cstep * (padded_row(FS) / 4);
                        }
                    }
                }
            }
        }
        if (FS == 0 || !staged) {
            // Run-time filter size (fs > 17: taps 9..16, strong down-scales) or a source footprint larger than the
            // LDS tile (samples then come through L1/L2): the same waterfall over the distinct coefficient sets of
            // the item, coefficients in SGPRs eight at a time.
            const float* s_lds = tile + (sy - ty0) * pitch + (sx - tx0);
            const char* s_glb = sframe + static_cast<size_t>(sy) * a.io.src_pitch + static_cast<size_t>(sx) * sizeof(T);
            unsigned long long todo = __ballot(active);
            for (int pass = 0; todo && pass < a.max_passes; ++pass) {
                const int leader = __ffsll(static_cast<long long>(todo)) - 1;
                const int u = __builtin_amdgcn_readlane(set, leader);
                const bool mine = active && set == u;
                const JINC_CONSTANT float* cs = (const JINC_CONSTANT float*)(p.coeffs + static_cast<size_t>(u) * fs * fsp);
                if (mine) {
                    if (staged)
                        acc = chain_runtime<T, true>(acc, s_lds, pitch, s_glb, a.io.src_pitch, cs, fs, fsp);
                    else
                        acc = chain_runtime<T, false>(acc, s_lds, pitch, s_glb, a.io.src_pitch, cs, fs, fsp);
                }
                todo &= ~__ballot(mine);
            }
            if (active && ((todo >> lane) & 1ull)) {  // private sets: per-lane coefficient loads
                if (a.rects.lane_coeffs != nullptr) {
                    const float* c = a.rects.lane_coeffs + static_cast<size_t>(lane_item) * (static_cast<size_t>(fs) * fsp * 64) + 4 * lane;
                    if (staged)
                        acc = chain_lane_major<T, true>(acc, s_lds, pitch, s_glb, a.io.src_pitch, c, fs, fsp);
                    else
                        acc = chain_lane_major<T, false>(acc, s_lds, pitch, s_glb, a.io.src_pitch, c, fs, fsp);
                } else {
                    const float* c = p.coeffs + static_cast<size_t>(set) * fs * fsp;
                    if (staged)
                        acc = chain_runtime<T, true>(acc, s_lds, pitch, s_glb, a.io.src_pitch, c, fs, fsp);
                    else
                        acc = chain_runtime<T, false>(acc, s_lds, pitch, s_glb, a.io.src_pitch, c, fs, fsp);
                }
            }
        }
        if (active) {
            T* d = reinterpret_cast<T*>(dframe + static_cast<size_t>(y) * a.io.dst_pitch) + x;
            store_sample<T>(d, acc, a.io.peak);
        }
    }
}

template <typename T, int FS>
int launch_gather_t(const GatherArgs& ga, int total_blocks, hipStream_t stream) {
    dim3 grid(static_cast<unsigned>((total_blocks + 7) / 8) * 8u * static_cast<unsigned>(ga.io.nframes), 1, 1), block(256, 1, 1);
    hipLaunchKernelGGL((ewa_gather_kernel<T, FS>), grid, block, 0, stream, ga);
    return static_cast<int>(hipGetLastError());
}

template <typename T>
int launch_gather_fs(const GatherArgs& ga, int total_blocks, hipStream_t stream) {
    switch (ga.plan.fs) {
        case 3: return launch_gather_t<T, 3>(ga, total_blocks, stream);
        case 4: return launch_gather_t<T, 4>(ga, total_blocks, stream);
        case 5: return launch_gather_t<T, 5>(ga, total_blocks, stream);
        case 6: return launch_gather_t<T, 6>(ga, total_blocks, stream);
        case 7: return launch_gather_t<T, 7>(ga, total_blocks, stream);
        case 8: return launch_gather_t<T, 8>(ga, total_blocks, stream);
        case 9: return launch_gather_t<T, 9>(ga, total_blocks, stream);
        case 10: return launch_gather_t<T, 10>(ga, total_blocks, stream);
        case 11: return launch_gather_t<T, 11>(ga, total_blocks, stream);
        case 12: return launch_gather_t<T, 12>(ga, total_blocks, stream);
        case 13: return launch_gather_t<T, 13>(ga, total_blocks, stream);
        case 14: return launch_gather_t<T, 14>(ga, total_blocks, stream);
        case 15: return launch_gather_t<T, 15>(ga, total_blocks, stream);
        case 16: return launch_gather_t<T, 16>(ga, total_blocks, stream);
        case 17: return launch_gather_t<T, 17>(ga, total_blocks, stream);
        default: return launch_gather_t<T, 0>(ga, total_blocks, stream);
    }
}


}  // namespace

namespace {
// Runs exactly the conversion + store code of the resampling kernels on caller-supplied sums.
template <typename T>
__global__ void convert_kernel(const float* in, T* out, int n, float peak) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const BufferRsrc rsrc = make_rsrc(out, static_cast<uint32_t>(n) * sizeof(T));
    if constexpr (std::is_same_v<T, uint16_t>) {  // the packed-pair kernels' path (round_pair_u16): elements 4k + 2 and 4k + 3 as one dword
        if ((i & 3) == 2 && i + 1 < n) {
            const uint32_t v = round_pair_u16(in[i], in[i + 1], peak);
            out[i] = static_cast<uint16_t>(v), out[i + 1] = static_cast<uint16_t>(v >> 16);
            return;
        }
        if ((i & 3) == 3) return;
    }
    if (i & 1)
        store_sample_buf<T>(rsrc, static_cast<uint32_t>(i) * sizeof(T), 0u, in[i], peak);  // periodic kernels' path
    else
        store_sample<T>(out + i, in[i], peak);                                             // gather kernel's path
}
}  // namespace

int launch_debug_convert(const float* in, void* out, int n, int sample_bytes, float peak, void* stream) {
    if (n <= 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((n + 255) / 256), block(256);
    switch (sample_bytes) {
        case 1: hipLaunchKernelGGL(convert_kernel<uint8_t>, grid, block, 0, s, in, static_cast<uint8_t*>(out), n, peak); break;
        case 2: hipLaunchKernelGGL(convert_kernel<uint16_t>, grid, block, 0, s, in, static_cast<uint16_t*>(out), n, peak); break;
        default: hipLaunchKernelGGL(convert_kernel<float>, grid, block, 0, s, in, static_cast<float*>(out), n, peak); break;
    }
    return static_cast<int>(hipGetLastError());
}

int launch_gather(const DevicePlan& plan, const PlaneIO& io, const RectList& rects, void* stream) {
    GatherArgs ga;
    ga.plan = plan;
    ga.io = io;
    ga.rects = rects;
    ga.max_passes = rects.private_sets ? 0 : 4;
    const int knob_passes = knobs::geti(JINC_KNOB_GATHER_PASSES, -1);  // A/B knob
    if (knob_passes >= 0) ga.max_passes = knob_passes;
    int total = 0;
    for (int r = 0; r < 4; ++r) {
        ga.block_begin[r] = total;
        ga.blocks_a[r] = 1;
        ga.lane_axis[r] = 0;
        ga.stride[r] = 1;
        ga.lines[r] = 4;
        if (r < rects.n && rects.w[r] > 0 && rects.h[r] > 0) {
            int axis, P;
            gather_rect_layout(plan, rects.w[r], rects.h[r], rects.unit_stride, axis, P);
            const int along = axis == 0 ? rects.w[r] : rects.h[r];
            const int across = axis == 0 ? rects.h[r] : rects.w[r];
            // More lines per block amortise the block's fixed cost (bounds, staging, barrier) and its halo;
            // thin rectangles (the border frame) get as many lines as they have.
            int nl = 4;
            while (nl < 32 && nl < across) nl *= 2;
            // ... but keep enough blocks in flight for the chip (256 CUs): thin border rectangles of a small
            // batch would otherwise collapse into a few hundred long-running blocks
            while (nl > 4 && static_cast<long long>((along + 64 * P - 1) / (64 * P)) * ((across + nl - 1) / nl) * io.nframes < 1024)
                nl /= 2;
            {   // ... as long as the block's source footprint still fits the LDS tile (down-scales widen it)
                const double rx = static_cast<double>(plan.src_w) / plan.dst_w, ry = static_cast<double>(plan.src_h) / plan.dst_h;
                const double r_along = axis == 0 ? rx : ry, r_across = axis == 0 ? ry : rx;
                const double w_along = 64.0 * P * r_along + plan.fs + 2;
                while (nl > 1 && w_along * (nl * r_across + plan.fs + 2) > 0.9 * kGatherLdsFloats) nl /= 2;
            }
            ga.lane_axis[r] = axis;
            ga.stride[r] = P;
            ga.lines[r] = nl;
            ga.blocks_a[r] = (along + 64 * P - 1) / (64 * P);
            total += ga.blocks_a[r] * ((across + nl - 1) / nl);
        }
    }
    ga.block_begin[4] = total;
    if (total == 0 || io.nframes <= 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (io.sample_bytes) {
        case 1: return launch_gather_fs<uint8_t>(ga, total, s);
        case 2: return launch_gather_fs<uint16_t>(ga, total, s);
        default: return launch_gather_fs<float>(ga, total, s);
    }
}


}  // namespace jinc
