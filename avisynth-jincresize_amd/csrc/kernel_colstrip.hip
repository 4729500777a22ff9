// kernel_colstrip.hip -- ewa_colstrip_kernel: the left / right border columns of exactly phase-periodic plans over
// the interior's row range (the corners, where every pixel owns a coefficient set, stay with the gather kernel).
// See device_common.hpp for the parity rules.
//
// A border column x clamps its window at the image edge, so it owns one coefficient set per row phase q -- the same
// for every period-row j.  An item is therefore (column x, phase q) with the 64 lanes of a wave on 64 consecutive
// period-rows: coefficients are wave-uniform (SGPRs), window origins are affine in the lane, and nothing is looked
// up per lane (the gather kernel, which serves arbitrary plans, needs five dependent loads per item and lane).
// Lanes along y cannot fetch their rows themselves without touching 64 cache lines per load, so the block first
// stages the strip's source footprint -- the rows of its 64 period-rows x the few source columns the strip's windows
// cover -- into LDS as fp32, reading rows with adjacent lanes, laid out [row % sy][column][row / sy] so that the
// lanes of an item (sy source rows apart) read consecutive words.
#include "device_common.hpp"

#pragma clang fp contract(off)

namespace jinc {
namespace {

constexpr int kColStripIdxPitch = 105;  // words per (plane, column): 64 lanes + (fs + phase spread) / sy <= 105, odd

template <typename T>
__global__ __launch_bounds__(256) void ewa_colstrip_kernel(const ColStripArgs a, const PlaneIO io) {
    extern __shared__ __attribute__((aligned(16))) float cs_tile[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j0 = blockIdx.x * 64;            // first period-row of the tile
    const int grp = blockIdx.y;                 // column group
    const size_t frame = blockIdx.z;
    const int side = grp >= a.groups[0] ? 1 : 0;
    const int g_in_side = grp - (side ? a.groups[0] : 0);
    const int xg0 = a.x0[side] + g_in_side * a.group_cols;
    const int xg1 = min(xg0 + a.group_cols, a.x0[side] + a.nx[side]);
    const int c0 = a.src_c0[side], wsrc = a.src_w[side];
    const int fs = a.fs, sy = a.sy;

    // ---- stage rows [row0, row0 + nrows) x columns [c0, c0 + wsrc) as fp32 ----
    const int row0 = a.min_sy + sy * j0;
    const int jn = min(64, a.nj - j0);          // period-rows of this tile
    const int nrows = sy * (jn - 1) + fs + a.spread_y;
    {
        const char* sbase = static_cast<const char*>(io.src) + frame * io.src_frame_stride;
        const int cshift = a.col_shift;         // lanes per staged row = 1 << cshift >= wsrc
        const int c = threadIdx.x & ((1 << cshift) - 1);
        const int rstep = 256 >> cshift;
        if (c < wsrc) {
            // four rows per thread and pass, loads in front of the LDS writes
            for (int r0 = threadIdx.x >> cshift; r0 < nrows; r0 += 4 * rstep) {
                T v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = r0 + i * rstep < nrows ? r0 + i * rstep : nrows - 1;
                    v[i] = reinterpret_cast<const T*>(sbase + static_cast<size_t>(row0 + r) * io.src_pitch)[c0 + c];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = r0 + i * rstep;
                    if (r < nrows) cs_tile[((r % sy) * wsrc + c) * kColStripIdxPitch + r / sy] = to_float(v[i]);
                }
            }
        }
    }
    __syncthreads();

    const BufferRsrc drsrc = make_rsrc(static_cast<char*>(io.dst) + frame * io.dst_frame_stride,
                                       static_cast<uint32_t>(io.dst_pitch) * a.plan.dst_h);
    const int nitems = (xg1 - xg0) * a.py;
    for (int item = wave; item < nitems; item += 4) {
        const int xi = item / a.py;
        const int q = item - xi * a.py;
        const int x = xg0 + xi;
        // wave-uniform look-ups (scalar loads)
        const int yr = a.iy0 + q;
        const int cc = ((const JINC_CONSTANT int32_t*)a.plan.col_class)[x];
        const int set = cc < 0 ? ((const JINC_CONSTANT int32_t*)a.plan.bcol_set)[static_cast<size_t>(~cc) * a.plan.dst_h + yr]
                               : ((const JINC_CONSTANT int32_t*)a.plan.interior_set)
                                     [static_cast<size_t>(((const JINC_CONSTANT int32_t*)a.plan.row_class)[yr]) * a.plan.n_col_classes + cc];
        const int cb = ((const JINC_CONSTANT int32_t*)a.plan.col_start)[x] - c0;  // window's first column in the tile
        const int rb = a.start_y[q] - a.min_sy;                                   // lane 0's first row in the tile
        const JINC_CONSTANT float* cs =
            (const JINC_CONSTANT float*)(a.coeffs + static_cast<size_t>(set) * (static_cast<size_t>(fs) * a.coeff_row));

        float acc = 0.f;
        int plane = rb % sy, idx0 = rb / sy;  // of kernel row ly, lane 0
        for (int ly = 0; ly < fs; ++ly) {
            const float* p = cs_tile + (plane * wsrc + cb) * kColStripIdxPitch + idx0 + lane;
            const JINC_CONSTANT float* crow = cs + static_cast<size_t>(ly) * a.coeff_row;
            for (int lx = 0; lx < fs; lx += 8) {
                float cf[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) cf[t] = crow[lx + t];  // the allocation has slack past the last row
                const int n = fs - lx;  // wave-uniform
                if (n >= 8) {
#pragma unroll
                    for (int t = 0; t < 8; ++t) acc = acc + p[t * kColStripIdxPitch] * cf[t];
                } else {
#pragma unroll
                    for (int t = 0; t < 7; ++t)
                        if (t < n) acc = acc + p[t * kColStripIdxPitch] * cf[t];
                }
                p += 8 * kColStripIdxPitch;
            }
            if (++plane == sy) plane = 0, ++idx0;
        }
        if (lane < jn) {
            const int y = a.iy0 + a.py * (j0 + lane) + q;
            store_sample_buf<T>(drsrc, static_cast<uint32_t>(y) * static_cast<uint32_t>(io.dst_pitch) + static_cast<uint32_t>(x) * sizeof(T),
                                0u, acc, io.peak);
        }
    }
}

template <typename T>
int launch_colstrip_t(const ColStripArgs& ca, const PlaneIO& io, hipStream_t stream) {
    const int wmax = ca.src_w[0] > ca.src_w[1] ? ca.src_w[0] : ca.src_w[1];
    const size_t lds = sizeof(float) * static_cast<size_t>(ca.sy) * wmax * kColStripIdxPitch;
    dim3 grid((ca.nj + 63) / 64, ca.groups[0] + ca.groups[1], io.nframes);
    hipLaunchKernelGGL((ewa_colstrip_kernel<T>), grid, dim3(256, 1, 1), lds, stream, ca, io);
    return static_cast<int>(hipGetLastError());
}

}  // namespace

// Fills the launch geometry; false if the strip does not fit this kernel (the gather kernel takes it then).
bool colstrip_configure(ColStripArgs& a) {
    if (a.sy < 1 || a.sy > 4 || a.nj < 1 || a.fs < 1) return false;
    if ((a.fs + a.spread_y + a.sy - 1) / a.sy + 64 > kColStripIdxPitch) return false;
    int wmax = 0;
    for (int s = 0; s < 2; ++s) {
        if (a.nx[s] <= 0) {
            a.groups[s] = 0;
            continue;
        }
        if (a.src_w[s] < 1 || a.src_w[s] > 64) return false;
        wmax = a.src_w[s] > wmax ? a.src_w[s] : wmax;
    }
    if (wmax == 0) return false;
    if (sizeof(float) * static_cast<size_t>(a.sy) * wmax * kColStripIdxPitch > 48 * 1024) return false;
    a.col_shift = 0;
    while ((1 << a.col_shift) < wmax) ++a.col_shift;
    // a block of 4 waves gets about two rounds of items: (columns of the group) x py
    a.group_cols = (8 + a.py - 1) / a.py;
    if (a.group_cols < 1) a.group_cols = 1;
    for (int s = 0; s < 2; ++s) a.groups[s] = a.nx[s] > 0 ? (a.nx[s] + a.group_cols - 1) / a.group_cols : 0;
    return true;
}

int launch_colstrip(const ColStripArgs& args, const PlaneIO& io, void* stream) {
    if (args.groups[0] + args.groups[1] <= 0 || args.nj <= 0 || io.nframes <= 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (io.sample_bytes) {
        case 1: return launch_colstrip_t<uint8_t>(args, io, s);
        case 2: return launch_colstrip_t<uint16_t>(args, io, s);
        default: return launch_colstrip_t<float>(args, io, s);
    }
}

}  // namespace jinc
