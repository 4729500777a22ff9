// kernel_quasi.hip -- ewa_quasi_kernel: interior kernel of quasi-periodic plans (ratios whose phase classes
// drift because the reference accumulates positions in float: 1.5x, 3x, 8/3 x ...).
// See device_common.hpp for the parity rules.
#include "device_common.hpp"

#pragma clang fp contract(off)

namespace jinc {
namespace {

// ------------------------------------------------------------------------------------------------
// Quasi-periodic interior kernel (ratios whose phases drift: 1.5x, 3x, ...)
// ------------------------------------------------------------------------------------------------
// The reference accumulates output positions in float, so for ratios such as 3/2 or 3 the quantised
// phase of a column/row is only NEARLY periodic: along 1920 columns a residue class changes its
// coefficient class about ten times.  The window ORIGINS, however, stay exactly affine per residue.
// This kernel keeps everything of ewa_periodic_kernel that depends on the origins only -- source tile
// staged once as fp32 in LDS, lanes = consecutive periods, fs x fs register window sliding down the
// tile with compile-time rotation (SY new source rows per output row) -- and looks the coefficient set
// up per (row, lane): row class from an LDS copy (wave-uniform), lane class loaded once per phase, set id
// from an LDS copy of the class-pair table.  The set of the previous row stays in SGPRs; a waterfall
// over the distinct sets of the wave reloads them only at the rare change points, so the common case is
// again 2 VALU instructions per tap with wave-uniform SGPR coefficients.
template <typename T, int FS, int SX, int SY>
__global__ __launch_bounds__(256) void ewa_quasi_kernel(const QuasiArgs a, const PlaneIO io) {
    extern __shared__ __attribute__((aligned(16))) float q_smem[];
    float* tile = q_smem;
    int* l_iset = reinterpret_cast<int*>(tile + a.lds_rows * a.lds_pitch);
    int* l_rc = l_iset + a.n_row_classes * a.n_col_classes;

    const int nthreads = blockDim.x;
    const int nwaves = nthreads >> 6;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile_rows = a.rg * FS;  // output rows (per row phase) of a tile
    int tile_x, tile_y;
    swizzled_tile(tile_x, tile_y);
    const int i0 = tile_x * 64;
    const int j0 = tile_y * tile_rows;
    const size_t frame = blockIdx.z;
    const int pitch = a.lds_pitch;

    {
        const int gx0 = a.min_sx + SX * i0;
        const int gy0 = a.min_sy + SY * j0;
        const char* sbase = static_cast<const char*>(io.src) + frame * io.src_frame_stride;
        for (int r = wave; r < a.lds_rows; r += nwaves) {
            int gy = gy0 + r;
            gy = gy < a.src_h ? gy : a.src_h - 1;
            const T* srow = reinterpret_cast<const T*>(sbase + static_cast<size_t>(gy) * io.src_pitch);
            for (int c = lane; c < a.lds_cols; c += 64) {
                int gx = gx0 + c;
                gx = gx < a.src_w ? gx : a.src_w - 1;
                // SX column planes per row (column c -> plane c % SX, index c / SX): the lanes of a wave are SX
                // source columns apart, so within a plane they read consecutive LDS words (no bank conflicts)
                tile[r * pitch + (c % SX) * a.lds_plane + c / SX] = to_float(srow[gx]);
            }
        }
        const int ntab = a.n_row_classes * a.n_col_classes;
        for (int k = threadIdx.x; k < ntab; k += nthreads) l_iset[k] = a.interior_set[k];
        // classes of the tile's output rows, in output order: entry py*jj + q <-> row iy0 + py*(j0+jj) + q
        const int nrows = a.py * tile_rows;
        for (int k = threadIdx.x; k < nrows; k += nthreads) {
            const int jj = k / a.py;
            l_rc[k] = (j0 + jj) < a.nj ? a.row_class[a.iy0 + a.py * j0 + k] : 0;
        }
    }
    __syncthreads();

    const BufferRsrc drsrc = make_rsrc(static_cast<char*>(io.dst) + frame * io.dst_frame_stride,
                                       static_cast<uint32_t>(io.dst_pitch) * a.dst_h);
    const int nphase = a.px * a.py;
    for (int ph = wave; ph < nphase; ph += nwaves) {
        const int q = ph / a.px;
        const int p = ph - q * a.px;
        const int ia = i0 + lane;
        const bool valid = ia < a.ni;
        const unsigned long long valid_mask = __ballot(valid);
        if (valid_mask == 0) continue;
        const unsigned x = a.ix0 + a.px * (valid ? ia : i0) + p;
        const int cc = a.col_class[x];  // one global load per phase and lane
        const uint32_t xoff = x * static_cast<uint32_t>(sizeof(T));
        // one base pointer per plane: source column off + lx (lx = SX*k + m) lives at bptr[m][k]
        const int off = a.start_x[p] - a.min_sx;
        const float* bptr[SX];
#pragma unroll
        for (int m = 0; m < SX; ++m)
            bptr[m] = tile + (a.start_y[q] - a.min_sy) * pitch + ((off + m) % SX) * a.lds_plane + (off + m) / SX + lane;

        float win[FS][FS];
#pragma unroll
        for (int r = 0; r < FS - SY; ++r)
#pragma unroll
            for (int lx = 0; lx < FS; ++lx) win[r][lx] = bptr[lx % SX][r * pitch + lx / SX];

        float cf[FS * FS];
#pragma unroll
        for (int k = 0; k < FS * FS; ++k) cf[k] = 0.f;
        int cur_set = -1;

        for (int g = 0; g < a.rg; ++g) {
            if (j0 + g * FS >= a.nj) break;  // wave-uniform
            const int grow = (SY * g * FS) * pitch;
#pragma unroll
            for (int t = 0; t < FS; ++t) {
                // window of output row t covers relative source rows SY*t .. SY*t+FS-1; the SY newest arrive now
#pragma unroll
                for (int k = 0; k < SY; ++k) {
                    const int rel = SY * t + FS - SY + k;
#pragma unroll
                    for (int lx = 0; lx < FS; ++lx) win[rel % FS][lx] = bptr[lx % SX][grow + rel * pitch + lx / SX];
                }
                const int jj = g * FS + t;
                if (j0 + jj < a.nj) {  // wave-uniform
                    const int rc = __builtin_amdgcn_readfirstlane(l_rc[a.py * jj + q]);
                    const int set = l_iset[rc * a.n_col_classes + cc];
                    float acc = 0.f;
                    unsigned long long todo = valid_mask;
                    while (todo) {
                        // lanes that use the set already sitting in SGPRs go first: a wave that straddles a
                        // change point then reloads one set per row instead of two
                        const unsigned long long cached = __ballot(set == cur_set) & todo;
                        const int leader = __ffsll(static_cast<long long>(cached ? cached : todo)) - 1;
                        const int u = __builtin_amdgcn_readlane(set, leader);
                        if (u != cur_set) {  // wave-uniform: only at the class change points of the drift
                            const JINC_CONSTANT float* cs =
                                (const JINC_CONSTANT float*)(a.coeffs + static_cast<size_t>(u) * (FS * padded_row(FS)));
#pragma unroll
                            for (int k = 0; k < FS * FS; ++k) cf[k] = cs[(k / FS) * padded_row(FS) + (k % FS)];
                            cur_set = u;
                        }
                        const bool mine = valid && set == u;
                        if (mine) {
#pragma unroll
                            for (int ly = 0; ly < FS; ++ly)
#pragma unroll
                                for (int lx = 0; lx < FS; ++lx)
                                    acc = acc + win[(SY * t + ly) % FS][lx] * cf[ly * FS + lx];
                        }
                        todo &= ~__ballot(mine);
                    }
                    if (valid) {
                        const int y = a.iy0 + a.py * (j0 + jj) + q;
                        store_sample_buf<T>(drsrc, xoff, static_cast<uint32_t>(y) * io.dst_pitch, acc, io.peak);
                    }
                }
            }
        }
    }
}


}  // namespace

bool quasi_supported(int fs, int px, int py, int sx, int sy, int n_col_classes, int n_row_classes) {
    if (fs != 7 && fs != 9) return false;                                  // register window; taps 3 and 4
    if (px < 1 || py < 1 || px > 16 || py > 16) return false;
    if (sx < 1 || sy < 1 || sx > 4 || sy > 4) return false;
    return static_cast<long long>(n_col_classes) * n_row_classes <= 4096;   // class-pair table held in LDS
}

bool quasi_configure(QuasiArgs& a, int fs, int spread_x, int spread_y) {
    const int nphase = a.px * a.py;
    a.nwaves = nphase % 4 == 0 ? 4 : (nphase % 3 == 0 ? 3 : (nphase % 2 == 0 ? 2 : (nphase >= 4 ? 4 : nphase)));
    a.lds_cols = a.sx * 64 + fs + spread_x;
    a.lds_plane = (a.lds_cols + a.sx - 1) / a.sx + 1;
    if (a.sx == 2 || a.sx == 4) a.lds_plane = ((a.lds_plane + 15) / 32) * 32 + (a.sx == 2 ? 16 : 8);  // planes on distinct banks
    a.lds_pitch = a.sx * a.lds_plane;
    size_t budget = 20 * 1024;  // A/B on 1.5x: 20 KB 180 Gpix/s, 30 KB 160, 40 KB 134 (occupancy beats tile size)
    if (const char* e = std::getenv("JINC_QUASI_LDS_KB")) budget = static_cast<size_t>(std::atoi(e)) * 1024;  // tuning knob
    for (int rg = 8; rg >= 1; --rg) {
        const int rows = a.sy * rg * fs + fs + spread_y;
        const size_t bytes = sizeof(float) * static_cast<size_t>(rows) * a.lds_pitch +
                             sizeof(int) * (static_cast<size_t>(a.n_col_classes) * a.n_row_classes + a.py * rg * fs);
        if (bytes <= budget || rg == 1) {
            a.rg = rg;
            a.lds_rows = rows;
            return true;
        }
    }
    return false;
}

namespace {
template <typename T, int FS, int SX, int SY>
int launch_quasi_t(const QuasiArgs& qa, const PlaneIO& io, hipStream_t stream) {
    const size_t lds = sizeof(float) * static_cast<size_t>(qa.lds_rows) * qa.lds_pitch +
                       sizeof(int) * (static_cast<size_t>(qa.n_col_classes) * qa.n_row_classes + qa.py * qa.rg * FS);
    const int tile_rows = qa.rg * FS;
    dim3 grid((qa.ni + 63) / 64, (qa.nj + tile_rows - 1) / tile_rows, io.nframes);
    hipLaunchKernelGGL((ewa_quasi_kernel<T, FS, SX, SY>), grid, dim3(64 * qa.nwaves, 1, 1), lds, stream, qa, io);
    return static_cast<int>(hipGetLastError());
}
template <typename T, int FS, int SX>
int launch_quasi_sy(const QuasiArgs& qa, const PlaneIO& io, hipStream_t stream) {
    switch (qa.sy) {
        case 1: return launch_quasi_t<T, FS, SX, 1>(qa, io, stream);
        case 2: return launch_quasi_t<T, FS, SX, 2>(qa, io, stream);
        case 3: return launch_quasi_t<T, FS, SX, 3>(qa, io, stream);
        default: return launch_quasi_t<T, FS, SX, 4>(qa, io, stream);
    }
}
template <typename T, int FS>
int launch_quasi_sx(const QuasiArgs& qa, const PlaneIO& io, hipStream_t stream) {
    switch (qa.sx) {
        case 1: return launch_quasi_sy<T, FS, 1>(qa, io, stream);
        case 2: return launch_quasi_sy<T, FS, 2>(qa, io, stream);
        case 3: return launch_quasi_sy<T, FS, 3>(qa, io, stream);
        default: return launch_quasi_sy<T, FS, 4>(qa, io, stream);
    }
}
template <typename T>
int launch_quasi_fs(const QuasiArgs& qa, int fs, const PlaneIO& io, hipStream_t stream) {
    switch (fs) {
        case 7: return launch_quasi_sx<T, 7>(qa, io, stream);
        case 9: return launch_quasi_sx<T, 9>(qa, io, stream);
        default: return static_cast<int>(hipErrorInvalidValue);
    }
}
}  // namespace

int launch_quasi(const QuasiArgs& args, int fs, const PlaneIO& io, void* stream) {
    if (args.ni <= 0 || args.nj <= 0 || io.nframes <= 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (io.sample_bytes) {
        case 1: return launch_quasi_fs<uint8_t>(args, fs, io, s);
        case 2: return launch_quasi_fs<uint16_t>(args, fs, io, s);
        default: return launch_quasi_fs<float>(args, fs, io, s);
    }
}


}  // namespace jinc
