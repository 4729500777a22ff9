// quasi_dispatch.cpp -- host side of the quasi-periodic kernel: eligibility, tile configuration, size dispatch.
#include <cstddef>
#include <cstdlib>

#include "kernels.h"
#include "knobs.h"

namespace jinc {

int launch_quasi_fs7(const QuasiArgs& args, const PlaneIO& io, void* stream);  // kernel_quasi_fs7.hip
int launch_quasi_fs9(const QuasiArgs& args, const PlaneIO& io, void* stream);  // kernel_quasi_fs9.hip
int launch_quasi_exact_fs7(const QuasiArgs& args, const PlaneIO& io, void* stream);  // kernel_quasi_exact_fs7.hip
int launch_quasi_exact_fs9(const QuasiArgs& args, const PlaneIO& io, void* stream);  // kernel_quasi_exact_fs9.hip
int launch_quasi_lane_fs7(const QuasiArgs& args, const PlaneIO& io, void* stream);   // kernel_quasi_lane_fs7.hip
int launch_quasi_lane_fs9(const QuasiArgs& args, const PlaneIO& io, void* stream);   // kernel_quasi_lane_fs9.hip

bool quasi_supported(int fs, int px, int py, int sx, int sy, int n_col_classes, int n_row_classes) {
    if (fs != 7 && fs != 9) return false;                                  // register window; taps 3 and 4
    if (px < 1 || py < 1 || px > 16 || py > 16) return false;
    if (sx < 1 || sy < 1 || sx > 4 || sy > 4) return false;
    return static_cast<long long>(n_col_classes) * n_row_classes <= 4096;   // class-pair table held in LDS
}

bool quasi_configure(QuasiArgs& a, int fs, int spread_x, int spread_y) {
    const int nphase = a.px * a.py;
    a.nwaves = nphase % 4 == 0 ? 4 : (nphase % 3 == 0 ? 3 : (nphase % 2 == 0 ? 2 : (nphase >= 4 ? 4 : nphase)));
    if (spread_x > a.sx) return false;  // the compile-time tile width allows a phase spread of one source step
    a.lds_cols = a.sx * 64 + fs + spread_x;
    a.lds_plane = quasi_plane_words(fs, a.sx);
    a.lds_pitch = quasi_pitch_words(fs, a.sx);
    // A/B on 1.5x, SGPR variants (8 waves/SIMD by registers): 20 KB 180 Gpix/s, 30 KB 160, 40 KB 134 -- occupancy beats tile
    // size.  The per-lane coefficient variant holds ~110 VGPRs (4 waves/SIMD), so a larger tile costs no occupancy and
    // amortises the per-phase set-up: 20 KB 249, 30 KB 284, 40 KB 271, 56 KB 220 Gpix/s.
    size_t budget = (a.exact == 2 ? 30 : 20) * 1024;
    if (knobs::is_set(JINC_KNOB_QUASI_LDS_KB)) budget = static_cast<size_t>(knobs::geti(JINC_KNOB_QUASI_LDS_KB, 0)) * 1024;  // tuning knob
    for (int rg = 8; rg >= 1; --rg) {
        const int rows = a.sy * rg * fs + fs + spread_y;
        const size_t bytes = sizeof(float) * static_cast<size_t>(rows) * a.lds_pitch +
                             sizeof(int) * (static_cast<size_t>(a.n_col_classes) * a.n_row_classes + a.py * rg * fs);
        if (bytes <= budget || (rg == 1 && bytes <= 60 * 1024)) {  // one workgroup may not exceed 64 KB of LDS
            a.rg = rg;
            a.lds_rows = rows;
            return true;
        }
    }
    return false;
}

int launch_quasi(const QuasiArgs& args, int fs, const PlaneIO& io, void* stream) {
    if (args.ni <= 0 || args.nj <= 0 || io.nframes <= 0) return 0;
    switch (fs) {
        case 7:
            return args.exact == 1 ? launch_quasi_exact_fs7(args, io, stream)
                                   : args.exact == 2 ? launch_quasi_lane_fs7(args, io, stream) : launch_quasi_fs7(args, io, stream);
        case 9:
            return args.exact == 1 ? launch_quasi_exact_fs9(args, io, stream)
                                   : args.exact == 2 ? launch_quasi_lane_fs9(args, io, stream) : launch_quasi_fs9(args, io, stream);
        default: return 1;  // hipErrorInvalidValue
    }
}

}  // namespace jinc
