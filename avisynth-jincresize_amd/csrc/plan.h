// plan.h -- compact EWA resampling plan (host side, built once per filter instance).
//
// The reference materialises, per output pixel, a 12-byte {start_x,start_y,coeff_offset} record and
// one private coefficient set per border pixel (generate_coeff_table_c,
// /root/reference/src/JincResize.cpp:336-533).  This builder produces the same per-pixel
// semantics in a separable, de-duplicated form sized for on-chip residency on MI355X:
//
//   * start_x depends only on x and start_y only on y  -> two int arrays (W + H entries);
//   * a coefficient set is fully determined by the pair (column distance profile, row distance
//     profile), a profile being the vector of squared tap distances (dx*dx) along one axis
//     -> sets are computed once per distinct pair and shared;
//   * interior pixels take the set of the FIRST raster-order interior pixel with the same
//     quantised phase (the reference's factor_map cache, ref :431-435, :517-518), which by
//     separability is (first interior column with that x-phase, first interior row with that y-phase);
//   * border pixels (any axis clamped) use un-quantised positions on BOTH axes (ref :485-486).
//
// The arithmetic that defines a coefficient (float position accumulation, float clamp/subtract,
// double multiply, round-half-even LUT index, float normalisation) follows the reference
// operation for operation; see plan.cpp.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "jinc_lut.h"

namespace jinc {

// generate_coeff_params without the scratch-sizing hints (ref :315-333).
struct TableGeometry {
    int quant_x = 256, quant_y = 256;
    int src_w = 0, src_h = 0, dst_w = 0, dst_h = 0;
    double radius = 0.0;
    double crop_left = 0.0, crop_top = 0.0, crop_w = 0.0, crop_h = 0.0;
};

struct PlanePlan {
    TableGeometry g;
    int fs = 0;  // filter_size (ref :356); a set is fs*fs floats, row-major, no padding

    // window origin per output column / row (EWAPixelCoeffMeta::start_x / start_y)
    std::vector<int32_t> col_start, row_start;
    // >= 0: interior class of the column / row;  < 0: ~(index into the border column / row list)
    std::vector<int32_t> col_class, row_class;
    int n_col_classes = 0, n_row_classes = 0;
    int n_bcols = 0, n_brows = 0;
    std::vector<int32_t> interior_set;  // [row_class * n_col_classes + col_class] -> set id
    std::vector<int32_t> bcol_set;      // [bcol * dst_h + y] -> set id, for every border column
    std::vector<int32_t> brow_set;      // [brow * dst_w + x] -> set id, for every border row
    std::vector<float> coeffs;          // num_sets * fs * fs
    int num_sets = 0;

    // Interior rectangle [ix0,ix1) x [iy0,iy1) and, when found, its phase periodicity:
    // col_class[x+px] == col_class[x] and col_start[x+px] == col_start[x] + sx (rows likewise).
    int ix0 = 0, ix1 = 0, iy0 = 0, iy1 = 0;
    bool periodic = false;
    int px = 0, py = 0, sx = 0, sy = 0;
    // Weaker structure that also holds for ratios whose phases drift (1.5x, 3x: the reference accumulates
    // positions in float): the window ORIGINS are exactly affine per residue, col_start[x+qpx] == col_start[x]+qsx
    // (rows likewise), while the phase classes may change now and then along an axis.
    bool quasi = false;
    int qpx = 0, qpy = 0, qsx = 0, qsy = 0;

    int set_of(int x, int y) const {
        const int rc = row_class[y], cc = col_class[x];
        if (rc < 0) return brow_set[static_cast<std::size_t>(~rc) * g.dst_w + x];
        if (cc < 0) return bcol_set[static_cast<std::size_t>(~cc) * g.dst_h + y];
        return interior_set[static_cast<std::size_t>(rc) * n_col_classes + cc];
    }
    const float* set_ptr(int set) const { return coeffs.data() + static_cast<std::size_t>(set) * fs * fs; }
};

// One rectangle of a drifting plan (quasi && !periodic): the pixels of column phase p and row phase q whose period-columns lie in
// one run of constant column class and whose period-rows lie in one run of constant row class.  They share ONE coefficient set and
// their windows advance by (qsx, qsy) source samples per period: inside the rectangle the plan is exactly periodic (the premise of
// the direct kernel; kernels.h DirectRun has this layout).
struct PlanRun {
    int32_t set = 0;
    int32_t x0 = 0, y0 = 0;    // output pixel of the first period (then every qpx-th column, qpy-th row)
    int32_t sx0 = 0, sy0 = 0;  // its window origin
    int32_t ni = 0, nj = 0;    // periods
    int32_t first_item = 0;    // items before this rectangle's first (an item = 64 lanes x 4 x 4 periods)
};
// The rectangles of the interior block [ix0, ix0 + qpx * ni) x [iy0, iy0 + qpy * nj), ni = (ix1 - ix0) / qpx (nj likewise), ordered by
// position, and the rectangle of every item.  False when the plan is not a drifting one or does not keep what `quasi` promises.
bool build_plan_runs(const PlanePlan& p, std::vector<PlanRun>& runs, std::vector<int32_t>& item_run);

// Throws std::runtime_error for geometry the reference handles only through out-of-bounds reads
// (source plane smaller than the filter footprint) or degenerate sizes.
PlanePlan build_plane_plan(const JincLut& lut, const TableGeometry& g);

}  // namespace jinc
