// kernels.h -- launch interface between the host filter (filter.cpp) and the gfx950 kernels
// (kernels.hip).  Plain structs of device pointers and sizes; no HIP types except the stream.
#pragma once
#include <cstddef>
#include <cstdint>

namespace jinc {

// Device-resident compact plan of one table (see plan.h for the meaning of each array).
struct DevicePlan {
    const int32_t* col_start = nullptr;
    const int32_t* row_start = nullptr;
    const int32_t* col_class = nullptr;
    const int32_t* row_class = nullptr;
    const int32_t* interior_set = nullptr;
    const int32_t* bcol_set = nullptr;
    const int32_t* brow_set = nullptr;
    const float* coeffs = nullptr;
    int src_w = 0, src_h = 0, dst_w = 0, dst_h = 0;
    int fs = 0;
    int n_col_classes = 0;
    // Dominant phase period of the output columns / rows (1..8): the lane stride of the gather kernel.
    int gather_period_x = 1, gather_period_y = 1;
};

// One plane of a batch of frames, device pointers, pitches/strides in bytes.
struct PlaneIO {
    const void* src = nullptr;
    void* dst = nullptr;
    int src_pitch = 0, dst_pitch = 0;
    size_t src_frame_stride = 0, dst_frame_stride = 0;
    int nframes = 1;
    int sample_bytes = 1;  // 1: uint8, 2: uint16, 4: float
    float peak = 255.f;    // clamp ceiling of integer planes (ref JincResize.cpp:582, :793)
};

// Up to four output rectangles handled by one gather launch (whole plane, or the border frame
// around the periodic interior).
struct RectList {
    int n = 0;
    int x0[4] = {0, 0, 0, 0}, y0[4] = {0, 0, 0, 0}, w[4] = {0, 0, 0, 0}, h[4] = {0, 0, 0, 0};
    // true for the border frame of a plan that is not exactly periodic: there nearly every pixel owns a
    // private coefficient set, so the kernel skips the uniform passes and loads coefficients per lane
    bool private_sets = false;
    // lane stride 1 instead of the plan's dominant period: for rectangles in which even pixels of one phase never
    // share a set (corners, border frame of drifting plans) the period stride only thins out the lanes
    bool unit_stride = false;
    // Optional lane-major copy of the coefficients of these rectangles' pixels (private_sets only): item i of
    // rectangle r (numbering: gather_item_index) owns fs * padded_row(fs) * 64 floats laid out
    // [kernel row][group of 4 taps][lane][4], so the 64 lanes of an item fetch their private coefficients with
    // contiguous 16-byte loads (1 KiB per wave and fetch) instead of touching 64 cache lines.
    const float* lane_coeffs = nullptr;
    long long lane_item_base[4] = {0, 0, 0, 0};
};

// How the gather kernel walks a rectangle: lane axis (0: lanes along x, 1: along y) and lane stride P; shared by
// launch_gather and by the host code that builds RectList::lane_coeffs.
inline void gather_rect_layout(const DevicePlan& plan, int w, int h, bool unit_stride, int& axis, int& P) {
    // Narrow rectangles (border columns, up to ~fs wide) put the lanes along y: a wave is then not mostly
    // idle, and its lanes share the border column's coefficient sets (along x every lane would own one).
    axis = (w < 64 && h > w) ? 1 : 0;
    // The lane stride exists to give the lanes of an item one coefficient set; where every pixel owns its set it only
    // thins out the lanes of narrow rectangles (a 33-pixel corner: 17 of 64 lanes at stride 2).
    P = unit_stride ? 1 : (axis == 0 ? plan.gather_period_x : plan.gather_period_y);
}
// Items of a rectangle are numbered (line * P + residue) * blocks_along + block, with line = row (axis 0) or column
// (axis 1) inside the rectangle and blocks_along = ceil(extent along the lane axis / (64 * P)).
inline long long gather_item_count(int w, int h, int axis, int P) {
    const int along = axis == 0 ? w : h, across = axis == 0 ? h : w;
    return static_cast<long long>(across) * P * ((along + 64 * P - 1) / (64 * P));
}

// Phase-periodic interior (see plan.h): output pixel (ix0 + px*i + p, iy0 + py*j + q) reads the
// source window at (start_x[p] + i, start_y[q] + j) with coefficient set set[q*px + p].
constexpr uint32_t span7_form(int ly, int q, uint32_t form) { return form << (2 * (2 * ly + q)); }  // (PeriodicArgs::kQuadSpan7Mpeg2)
struct PeriodicArgs {
    const float* coeffs = nullptr;
    int px = 1, py = 1;
    int ix0 = 0, iy0 = 0;
    int ni = 0, nj = 0;          // number of whole periods covered in x / y
    int start_x[8] = {0}, start_y[8] = {0};
    int min_sx = 0, min_sy = 0;  // min over phases of start_x / start_y
    int set[64] = {0};
    int src_w = 0, src_h = 0;
    int dst_h = 0;  // rows of the destination plane (bounds of the store descriptor)
    // 2x up-scales whose phases share their window origin (ewa_periodic_quad_kernel): coefficient pairs
    // quad[ly][q][8 pairs][p] = (set(p=0,q), set(p=1,q))[ly][lx], or nullptr when the plan has no such form
    const float* quad = nullptr;
    // ewa_periodic_quad2_kernel: bit 2 * ly + q set = taps 0 and 5 of kernel row ly carry zero coefficients for both phases p of q
    uint32_t quad_inner = 0;
    int quad_taps = 0;  // ewa_periodic_quad2_kernel: taps per kernel row when they differ from the row count (7: the 6 x 7 support), else 0
    // the 6 x 7 support: zero coefficients in front of / behind the span of kernel row ly at row phase q for BOTH phases p, two bits each
    // (capped at 3) at 4 * (2 * ly + q) and 4 * (2 * ly + q) + 2
    uint64_t quad_span7 = 0;
    // ... and the span FORM of every (ly, q) the kernel is instantiated for (kernel_periodic.hip quad2_row7_span: 0 = all seven taps, 1 =
    // taps 1 .. 6, 2 = taps 1 .. 5, 3 = taps 2 .. 5), two bits at 2 * (2 * ly + q): chroma planes sited as MPEG-2 at 2x with tap 3 --
    // q = 0: forms 3 1 0 0 0 2 for ly = 0 .. 5, q = 1: 2 0 0 0 1 3.  36 of 42 taps.
    static constexpr uint32_t kQuadSpan7Mpeg2 = span7_form(0, 0, 3) | span7_form(1, 0, 1) | span7_form(5, 0, 2) | span7_form(0, 1, 2) |
                                                span7_form(4, 1, 1) | span7_form(5, 1, 3);
    // (the spans of the 8-row x 9-column support -- chroma at tap 4 -- use the same field: 16 (ly, q) entries of four bits)
    // (which output row is q = 0 follows from the parity of the interior's first row: the same pattern with the row phases exchanged)
    static constexpr uint32_t kQuadSpan7Mpeg2Swapped = span7_form(0, 1, 3) | span7_form(1, 1, 1) | span7_form(5, 1, 2) | span7_form(0, 0, 2) |
                                                       span7_form(4, 0, 1) | span7_form(5, 0, 3);
    static constexpr uint32_t kQuadInnerTap3 = (1u << (2 * 5 + 0)) | (1u << (2 * 0 + 1));  // the mask the kernel is instantiated for
    // quad forms on the 8 x 8 support: taps kernel row ly of q leaves out per side (0 .. 3), two bits at 2 * (2 * ly + q)
    uint32_t quad_trim8 = 0;
    static const uint32_t kQuad8TrimTap4;  // the pattern the kernels are instantiated for (below)
    // trimmed support only (ewa_periodic_rows_kernel): row_trim[phase * 32 + ly] = taps kernel row ly of the phase leaves out on
    // EITHER side (min of its leading and trailing zero coefficients, at most 5), or nullptr
    const int32_t* row_trim = nullptr;
    int rows_ny = 0;  // ewa_periodic_rows_kernel on a support with fewer kernel rows than taps per row: the row count (0: fs rows)
    // ewa_periodic_rowpair_kernel (2x up-scales with 12 .. 17 taps per kernel row whose two phases p share their window origin):
    // coefficient pairs rowpair[(q * rowpair_ny + ly) * rowpair_stride + 2 * lx + p] = set(p, q)[ly][lx], and per q the taps each
    // kernel row leaves out on either side for BOTH p, three bits per kernel row (bits 3 * ly ...), and in bits 54 .. 58 / 59 .. 63 the
    // first kernel row executed / the one behind the last (0: all rowpair_ny) -- the rows outside are zero for both p; nullptr: no such form
    // Border rows as launches of the same kernel (device_plan.cpp plan_rowpair_rows): the border rows of one end of the plane are the
    // py <= kRowPairMaxPhases "row phases" of a launch with nj = 1, every one with its own coefficient pairs and all with the window
    // origin start_y[0] (py > 8: start_y[q] is not read).  rowpair_strip_phases > 0 marks such a launch: its grid's y counts groups of
    // that many phases (a workgroup = one group x one tile column: 17 rows in one workgroup were three rounds of its eight waves on
    // a chip the launch does not fill -- C3: 46 us per launch), not tile rows.
    static constexpr int kRowPairMaxPhases = 24;
    int rowpair_strip_phases = 0;
    const float* rowpair = nullptr;
    int rowpair_n = 0, rowpair_ny = 0, rowpair_stride = 0;
    uint64_t rowpair_trim[kRowPairMaxPhases] = {};
    // float planes on the trimmed support: frame_flags[frame] (kernel_scan.hip: 1 = the frame's plane holds a non-finite
    // sample) decides which of two launches computes a frame -- a launch returns at once for frames whose flag differs from
    // run_when.  nullptr: every frame.
    const uint32_t* frame_flags = nullptr;
    uint32_t run_when = 0;
    // run_when == kRunAllAndFlag: the launch computes EVERY frame and sets frame_flags[frame] = 1 where it stages a non-finite
    // sample (the trimmed launch of a float plane is its own finite-sample scan; kernel_periodic.hip NonFinite)
    static constexpr uint32_t kRunAllAndFlag = 2;
    // ewa_periodic_quad2_kernel / ewa_periodic_quad2x8_kernel on integer planes: the plane's border COLUMNS beside the interior rows, computed by the first / last
    // tile column out of the tile it has staged anyway (device_plan.cpp plan_edge_columns; as kernels of their own the columns were
    // bound by the scattered 64-byte lines of their source windows and stores, not by arithmetic).  Side s (0 left, 1 right): n[s]
    // output columns from x0[s] on, all with the source window whose first column is LDS column lds_col[s] (-1: the column in front
    // of the tile, staged into the pitch's spare word) of tile column tile_x[s]; coefficient rows
    // coeffs[((s * kMaxPerSide + k) * 2 + q) * (NR * NCP) + ly * NCP + lx] = kernel rows r0 .. r0 + NR - 1 (the interior's trimmed rows: 6,
    // or 8 for ewa_periodic_quad2x8_kernel) of column k's set at row phase q, NCP = the filter size rounded up to a multiple of 4.
    // coeffs == nullptr: the border kernels compute the columns.
    struct EdgeColumns {
        static constexpr int kMaxPerSide = 16;
        const float* coeffs = nullptr;
        int n[2] = {0, 0}, x0[2] = {0, 0}, lds_col[2] = {0, 0}, tile_x[2] = {0, 0};
    } edge;
};

// Quad forms on the 8 x 8 support: per (kernel row ly, q) the taps left out per side, two bits at 2 * (2 * ly + q).  The pattern
// the kernels are instantiated for is the 2x up-scale with tap 4 (blur 1 and 0.98): q = 0 rows 0 / 6 / 7 leave out 1 / 1 / 2 taps
// per side, q = 1 rows 0 / 1 / 7 leave out 2 / 1 / 1 -- 56 instead of 64 taps per sample.
constexpr uint32_t quad8_bits(int ly, int q, int t) { return static_cast<uint32_t>(t) << (2 * (2 * ly + q)); }
constexpr uint32_t kQuad8TrimTap4Value = quad8_bits(0, 0, 1) | quad8_bits(6, 0, 1) | quad8_bits(7, 0, 2) | quad8_bits(0, 1, 2) |
                                         quad8_bits(1, 1, 1) | quad8_bits(7, 1, 1);
inline constexpr uint32_t PeriodicArgs::kQuad8TrimTap4 = kQuad8TrimTap4Value;
// Does the plan leave out at least what the span forms of `pattern` leave out?  (form -> taps left out in front / behind)
// The 8-row x 9-column support (chroma planes sited as MPEG-2 at 2x with tap 4; ewa_periodic_quad2x8_kernel with nine taps per kernel
// row): span FORM of every (ly, q), three bits at 3 * (2 * ly + q) -- 0 = all nine taps, 1 = taps 1 .. 8, 2 = 1 .. 7, 3 = 2 .. 7, 4 =
// 2 .. 6 (kernel_periodic.hip quad2_row9_span).  Blur 1: q = 0 forms 4 2 1 0 0 0 2 3 for ly = 0 .. 7, q = 1: 3 2 0 0 0 1 2 4 -- 60 of 72.
constexpr uint64_t span9_form(int ly, int q, uint64_t form) { return form << (3 * (2 * ly + q)); }
constexpr uint64_t kQuadSpan9Mpeg2 = span9_form(0, 0, 4) | span9_form(1, 0, 2) | span9_form(2, 0, 1) | span9_form(6, 0, 2) | span9_form(7, 0, 3) |
                                     span9_form(0, 1, 3) | span9_form(1, 1, 2) | span9_form(5, 1, 1) | span9_form(6, 1, 2) | span9_form(7, 1, 4);
constexpr uint64_t kQuadSpan9Mpeg2Swapped = span9_form(0, 1, 4) | span9_form(1, 1, 2) | span9_form(2, 1, 1) | span9_form(6, 1, 2) | span9_form(7, 1, 3) |
                                            span9_form(0, 0, 3) | span9_form(1, 0, 2) | span9_form(5, 0, 1) | span9_form(6, 0, 2) | span9_form(7, 0, 4);
inline bool quad_span9_fits(uint64_t plan, uint64_t pattern) {  // plan: lead / trail per (ly, q) as in PeriodicArgs::quad_span7
    constexpr int lead[5] = {0, 1, 1, 2, 2}, trail[5] = {0, 0, 1, 1, 2};
    for (int k = 0; k < 16; ++k) {
        const int form = static_cast<int>((pattern >> (3 * k)) & 7u);
        if (form > 4) return false;
        if (static_cast<int>((plan >> (4 * k)) & 3u) < lead[form] || static_cast<int>((plan >> (4 * k + 2)) & 3u) < trail[form]) return false;
    }
    return true;
}
inline int quad_span9_taps(uint64_t pattern) {  // taps the pattern executes over the 16 (ly, q) rows: twice the taps per sample
    constexpr int n[5] = {9, 8, 7, 6, 5};
    int t = 0;
    for (int k = 0; k < 16; ++k) t += n[(pattern >> (3 * k)) & 7u];
    return t;
}
inline bool quad_span7_fits(uint64_t plan, uint32_t pattern) {
    constexpr int lead[4] = {0, 1, 1, 2}, trail[4] = {0, 0, 1, 1};
    for (int k = 0; k < 12; ++k) {
        const int form = static_cast<int>((pattern >> (2 * k)) & 3u);
        if (static_cast<int>((plan >> (4 * k)) & 3u) < lead[form] || static_cast<int>((plan >> (4 * k + 2)) & 3u) < trail[form]) return false;
    }
    return true;
}
inline int quad_span7_taps(uint32_t pattern) {  // taps per sample the pattern executes, both q averaged x 2 (12 (ly, q) rows)
    constexpr int n[4] = {7, 6, 5, 4};
    int t = 0;
    for (int k = 0; k < 12; ++k) t += n[(pattern >> (2 * k)) & 3u];
    return t;
}
inline bool quad8_pattern_fits(uint32_t plan, uint32_t pattern) {  // the plan leaves out at least what the pattern leaves out
    for (int k = 0; k < 16; ++k)
        if (((plan >> (2 * k)) & 3u) < ((pattern >> (2 * k)) & 3u)) return false;
    return true;
}

// Quasi-periodic interior: the window origins are affine per residue (output pixel (ix0 + px*i + p,
// iy0 + py*j + q) reads the source window at (start_x[p] + sx*i, start_y[q] + sy*j)) but the coefficient
// set is looked up per row and per lane, because the phase classes of drifting ratios change along the axes.
struct QuasiArgs {
    const float* coeffs = nullptr;
    const int32_t* col_class = nullptr;
    const int32_t* row_class = nullptr;
    const int32_t* interior_set = nullptr;
    int n_col_classes = 0, n_row_classes = 0;
    int px = 1, py = 1, sx = 1, sy = 1;
    int ix0 = 0, iy0 = 0, ni = 0, nj = 0;
    int start_x[16] = {0}, start_y[16] = {0};
    int min_sx = 0, min_sy = 0;
    int lds_cols = 0, lds_rows = 0, lds_pitch = 0;  // fp32 source tile staged per workgroup
    int lds_plane = 0;                              // words per column plane (sx planes per tile row)
    int rg = 1;                                     // row groups (of fs output rows) per tile
    int nwaves = 4;                                 // waves per workgroup (divides px*py evenly where possible)
    int exact = 0;                                  // plan is exactly periodic: one set per phase, no per-row lookup
    int phase_split = 1;                            // workgroups per (tile, frame): each takes every phase_split-th share of the phases
    int phase_set[256] = {0};                       // exact plans: set id of phase q*px + p
    int src_w = 0, src_h = 0, dst_h = 0;
};

// LDS row layout of the quasi-periodic kernel, a compile-time function of (filter size, source step) so that every
// LDS offset of its inner loop is an immediate: sx column planes of quasi_plane_words() words each.
constexpr int quasi_tile_cols(int fs, int sx) { return sx * 64 + fs + sx; }  // 64 lanes + window + phase spread <= sx
constexpr int quasi_plane_words(int fs, int sx) {
    const int w = (quasi_tile_cols(fs, sx) + sx - 1) / sx + 1;
    return (sx == 2 || sx == 4) ? ((w + 15) / 32) * 32 + (sx == 2 ? 16 : 8) : w;  // even steps: planes on distinct banks
}
constexpr int quasi_pitch_words(int fs, int sx) { return sx * quasi_plane_words(fs, sx); }

bool quasi_supported(int fs, int px, int py, int sx, int sy, int n_col_classes, int n_row_classes);
// Fills the tile geometry fields (lds_*, rg, nwaves) of `args`; returns false if no configuration fits.
bool quasi_configure(QuasiArgs& args, int fs, int spread_x, int spread_y);
int launch_quasi(const QuasiArgs& args, int fs, const PlaneIO& io, void* stream);

// Exactly periodic plans of any filter size and source step 1..4 (kernel_direct.hip; no LDS, row segments are read
// from memory in the source format with bounds-checked buffer loads).
//   interior  : output pixel (ix0 + px*i + p, iy0 + py*j + q), i < ni, j < nj, reads the source window at
//               (start_x[p] + sx*i, start_y[q] + sy*j) with coefficient set set[q*px + p];
//   row strips: the output rows line0[k] .. line0[k]+line_n[k]-1 (k = 0, 1: above / below the interior) over the
//               interior's column range; window row and coefficient set come from the plan tables per row and phase.
// The row walk of kernel_direct_impl.inc loads coefficient rows up to this many rows before / after a set without using
// them ((R-1) * sy = 3 * 4); upload_table keeps that much addressable memory around the coefficient array.
constexpr int kDirectCoeffSlackRows = 12;
// One rectangle of the runs form (plans whose window origins are affine per residue while the phase classes change now and
// then along an axis -- 1.5x, 3x, ...: PlanePlan::quasi): the pixels of column phase p and row phase q whose period-columns lie
// in one run of constant column class and whose period-rows lie in one run of constant row class.  They share ONE coefficient
// set, and their windows advance by (sx, sy) source samples per period -- inside the rectangle the plan is exactly periodic.
struct DirectRun {
    int32_t set = 0;
    int32_t x0 = 0, y0 = 0;    // output pixel of the first period (then every px-th column, py-th row)
    int32_t sx0 = 0, sy0 = 0;  // its window origin
    int32_t ni = 0, nj = 0;    // periods
    int32_t first_wave = 0;    // items before this rectangle's first (an item = 64 lanes x 4 x 4 periods)
};

struct DirectArgs {
    const float* coeffs = nullptr;
    int fs = 0, coeff_row = 0;   // filter size; floats per coefficient row on the device (padded_row(fs))
    int px = 1, py = 1, sx = 1, sy = 1;
    int ix0 = 0, iy0 = 0, ni = 0, nj = 0;
    int start_x[16] = {0}, start_y[16] = {0};
    int set[256] = {0};          // interior only
    int line0[2] = {0, 0}, line_n[2] = {0, 0};  // strips only
    DevicePlan plan;             // strips only: row/column tables
    uint32_t src_bytes = 0;      // readable bytes from the aligned-down base of one source plane (filter.cpp direct_src_bytes)
    int dst_h = 0;
    // runs form only: the rectangles, the rectangle of every item, items per frame
    const DirectRun* runs = nullptr;
    const int32_t* item_run = nullptr;
    int n_items = 0;
};
bool direct_supported(int fs, int px, int py, int sx, int sy);
// Probe of the hardware premise of kernel_direct.hip (the buffer range check covers the scalar offset): `buf` holds
// 2 * nbytes bytes with dword i = i, `out` 128 dwords.  See load_raw in kernel_direct.hip.
int launch_soffset_probe(const uint32_t* buf, uint32_t nbytes, uint32_t* out, void* stream);
int launch_direct(const DirectArgs& args, const PlaneIO& io, void* stream);
// Interior form launch_direct picks (kernel_direct_impl.inc DirectShape: 0 per-chain fetches, 2 row walk, 3 row walk with 8
// columns per lane); set_direct_shape forces one process-wide where it applies (tests, A/B runs; -1 = automatic),
// last_direct_shape reports the most recent launch's.
void set_direct_shape(int shape);
int last_direct_shape();
// row-walk interior forms (DirectShape 2 / 3), one translation unit per sample type and source step
// (kernel_direct_walk_*_sx*.hip)
#define JINC_DECLARE_DIRECT_WALK(tag)                                                            \
    int launch_direct_walk_##tag##_sx1(const DirectArgs&, const PlaneIO&, void* stream, int shape); \
    int launch_direct_walk_##tag##_sx2(const DirectArgs&, const PlaneIO&, void* stream, int shape); \
    int launch_direct_walk_##tag##_sx3(const DirectArgs&, const PlaneIO&, void* stream, int shape); \
    int launch_direct_walk_##tag##_sx4(const DirectArgs&, const PlaneIO&, void* stream, int shape);
JINC_DECLARE_DIRECT_WALK(u8)
JINC_DECLARE_DIRECT_WALK(u16)
JINC_DECLARE_DIRECT_WALK(f32)
#undef JINC_DECLARE_DIRECT_WALK
int launch_direct_row_strips(const DirectArgs& args, const PlaneIO& io, void* stream);
// Runs form (DirectRun): ewa_direct_kernel's row walk over the rectangles of a drifting plan, one launch per plane.
bool direct_runs_supported(int fs, int px, int py, int sx, int sy);
int launch_direct_runs(const DirectArgs& args, const PlaneIO& io, void* stream);
#define JINC_DECLARE_DIRECT_RUNS(tag)                                                 \
    int launch_direct_runs_##tag##_sx1(const DirectArgs&, const PlaneIO&, void* stream); \
    int launch_direct_runs_##tag##_sx2(const DirectArgs&, const PlaneIO&, void* stream); \
    int launch_direct_runs_##tag##_sx3(const DirectArgs&, const PlaneIO&, void* stream); \
    int launch_direct_runs_##tag##_sx4(const DirectArgs&, const PlaneIO&, void* stream);
JINC_DECLARE_DIRECT_RUNS(u8)
JINC_DECLARE_DIRECT_RUNS(u16)
JINC_DECLARE_DIRECT_RUNS(f32)
#undef JINC_DECLARE_DIRECT_RUNS

// Left / right border columns of an exactly periodic plan over the interior's row range (kernel_colstrip.hip):
// item = (output column x, row phase q), lanes = 64 consecutive period-rows, source footprint staged in LDS.
struct ColStripArgs {
    const float* coeffs = nullptr;
    int fs = 0, coeff_row = 0;
    int py = 1, sy = 1, iy0 = 0, nj = 0;
    int start_y[16] = {0};
    int min_sy = 0, spread_y = 0;          // min over phases of start_y, max - min
    int x0[2] = {0, 0}, nx[2] = {0, 0};    // output column runs: left, right
    int src_c0[2] = {0, 0}, src_w[2] = {0, 0};  // source columns their windows cover
    int groups[2] = {0, 0}, group_cols = 1;     // blocks per run, columns per block (colstrip_configure)
    int col_shift = 0;                          // staging: 1 << col_shift lanes per source row
    DevicePlan plan;
};
bool colstrip_configure(ColStripArgs& args);
int launch_colstrip(const ColStripArgs& args, const PlaneIO& io, void* stream);

// Frame-lane kernel (kernel_framelane.hip): batches of frames, any plan.  The 64 lanes of a wave are the same output
// pixel of 64 different frames, so coefficients are wave-uniform (SGPRs) whatever the plan's structure.
constexpr int kFrameLaneMaxTile = 32;  // output tile edge limit
// LDS bytes of a workgroup's tables in front of its tile: window origins of the tile's columns and rows, and the coefficient
// set of every pixel (row pitch kFrameLaneMaxTile) for the tile's (1 << ty_shift) rows.
constexpr int kFrameLaneTableBytes(int ty_shift) { return (2 * kFrameLaneMaxTile + kFrameLaneMaxTile * (1 << ty_shift)) * 4; }
constexpr int kFrameLanePosBytes(size_t sample_bytes) { return static_cast<int>(64 * sample_bytes + 4); }  // LDS bytes per source position
// Frame-pair form (kernel_framelane_pair.hip): 128 frames per workgroup, a lane owns two adjacent frames; 8-byte pad for
// float planes keeps a lane's pair 8-byte aligned at every position.
constexpr int kFrameLanePairFrames = 128;
constexpr int kFrameLanePairPosBytes(size_t sample_bytes) { return static_cast<int>(128 * sample_bytes + (sample_bytes == 4 ? 8 : 4)); }
struct FrameLaneArgs {
    DevicePlan plan;
    PlaneIO io;
    RectList rects;           // rectangles of the output plane to compute (x0/y0/w/h only)
    int block_begin[5] = {0, 0, 0, 0, 0};  // first tile of each rectangle; [4] = number of tiles
    int tiles_x[4] = {1, 1, 1, 1};
    int tx_shift = 5, ty_shift = 5;        // tile = (1 << tx_shift) x (1 << ty_shift) output pixels, tx >= 4, ty >= 4
    int threads = 512;
    int lds_bytes = 0;
    int vec_store_ok = 0;     // bit 0: destination base, pitch and frame stride are multiples of 4 samples; bit 1: of 16 bytes
    int variant = 0;          // 0: automatic (sliding-window form for filter sizes 5, 7, 8, 9), 1: row-segment form always
    int pair = 0;             // 1: configured for the frame-pair form (128 frames per workgroup)
    int subgroups = 0;        // ewa_framelane_sub_kernel: sub-groups per wave (2 / 4 / 8 = 32 / 16 / 8 frames per workgroup); set per launch
};
// Chooses the tile size for the rectangles `rects` of plane plan `p` (host arrays) so that every tile's source
// footprint fits the LDS budget and is at most 64 columns wide; false if no tile size fits (huge filter footprints).
struct PlanePlan;
bool framelane_configure(const PlanePlan& p, const RectList& rects, int sample_bytes, int nframes_hint, FrameLaneArgs& out);
int launch_framelane(const FrameLaneArgs& args, void* stream);
// Groups of fewer than 64 frames (kernel_framelane_sub.hip): a wave is args.subgroups sub-groups of 64 / subgroups lanes, each on
// its own output row of the tile; filter sizes 5, 7, 8, 9 (the sliding-window form's) on a tile configuration of framelane_configure.
bool framelane_sub_supported(int fs, int subgroups, int ty_shift);
int launch_framelane_sub(const FrameLaneArgs& args, void* stream);
// Frame-pair form: filter sizes 5 and 7 only (two windows per lane in 128 registers); false otherwise or if no tile fits.
bool framelane_pair_configure(const PlanePlan& p, const RectList& rects, int sample_bytes, int nframes_hint, FrameLaneArgs& out);
int launch_framelane_pair(const FrameLaneArgs& args, void* stream);

// Generic gather kernel: one lane per output pixel, any plan.  Returns a hipError_t value as int.
int launch_gather(const DevicePlan& plan, const PlaneIO& io, const RectList& rects, void* stream);

// True when a specialised periodic kernel exists for this filter size / period / sample type.
bool periodic_supported(int fs, int px, int py, int sx, int sy);
// variant: 0 = default choice per filter size, 1 = always the row-streamed kernel (A/B measurements)
int launch_periodic(const PeriodicArgs& args, int fs, const PlaneIO& io, void* stream, int variant = 0);
// kernel_strip.hip: border rows (axis 0: lanes along x) or border columns (axis 1: lanes along y) of an exactly periodic plan,
// one register window per lane for the strip's whole thickness.  A group = the lines (output rows | columns) of one strip: they
// share the source lines [origin, origin + fs) across the strip; along the strip output coordinate i0 + P * i + p reads the
// window at start[p] + i (source step 1) and line `l` of group g uses coefficient set sets[(set_base[g] + l) * P + p].
struct StripArgs {
    const float* coeffs = nullptr;
    const int32_t* sets = nullptr;  // device
    int fs = 0, axis = 0;
    int P = 1, S = 1;
    int i0 = 0, ni = 0;
    int start[16] = {0};
    int min_start = 0, spread = 0;
    int ngroups = 0;
    int line0[4] = {0, 0, 0, 0}, nlines[4] = {0, 0, 0, 0}, origin[4] = {0, 0, 0, 0}, set_base[4] = {0, 0, 0, 0};
    int src_w = 0, src_h = 0, dst_h = 0;
};
// Border columns on packed column pairs (kernel_colpair.hip): side s (0 left, 1 right) = n[s] output columns from x0[s] on, all with the
// source window columns origin[s] .. origin[s] + fs - 1; output row iy0 + py * j + q reads the source rows start_y[q] + j ...
// coeffs[(((s * kMaxGroups + g) * py + q) * fs + ly) * 4 fs + 4 lx + k] = set(x0[s] + 4 g + k, q)[ly][lx] (0 beyond the side's last column).
struct ColPairArgs {
    static constexpr int kMaxGroups = 6;  // groups of four columns per side
    const float* coeffs = nullptr;
    int fs = 0, py = 1, iy0 = 0, nj = 0;
    int start_y[4] = {0, 0, 0, 0};
    int min_sy = 0, spread = 0;
    int n[2] = {0, 0}, x0[2] = {0, 0}, origin[2] = {0, 0};
    int src_w = 0, src_h = 0, dst_h = 0;
};
bool colpair_supported(int fs, int py, int sy, int spread);
int launch_colpair(const ColPairArgs& args, const PlaneIO& io, void* stream);

bool strip_supported(int fs, int period, int step, int spread);
int launch_strip(const StripArgs& args, const PlaneIO& io, void* stream);
// kernel_rowpair.hip: the row-streamed kernel in packed phase-pair form (PeriodicArgs::rowpair; 12 .. 17 taps per kernel row)
bool rowpair_supported(int taps_per_row);
bool rowpair_strip_supported(int taps_per_row);  // ... as strips of border rows (PeriodicArgs::rowpair_strip_phases)
int launch_rowpair(const PeriodicArgs& args, const PlaneIO& io, void* stream);
// kernel_scan.hip: flags[frame] = 1 where the frame's float source plane (w x h samples) holds an infinity or a NaN
int launch_finite_scan(const PlaneIO& io, int w, int h, uint32_t* flags, void* stream);
// ... the same over the samples of the plane OUTSIDE the rectangle [rx0, rx1) x [ry0, ry1) only (what a trimmed launch that flags
// the samples it stages itself does not see: PeriodicArgs::kRunAllAndFlag)
int launch_finite_scan_outside(const PlaneIO& io, int w, int h, int rx0, int ry0, int rx1, int ry1, uint32_t* flags, void* stream);

// Compatibility modes: the summation order of the reference's SIMD paths (kernel_simdorder.hip).  order 1 = SSE4.1,
// 2 = AVX2, 3 = AVX-512; min_val = lower clamp of float source samples of this plane.
int launch_simd_order(const DevicePlan& plan, const PlaneIO& io, int order, float min_val, void* stream);

// Frame transport by the shader (kernel_blit.hip): entry = one plane of one frame, `rows` rows of `row_bytes` bytes from
// src (pitch src_pitch) to dst (pitch dst_pitch); either side may be host memory mapped into the device's address space.
// unit = 16 / 4 / 1: widest access both pointers and pitches allow (blit_unit).  The table lives in memory the device can
// read (the pipeline keeps it in pinned host memory); one launch moves entries [first, first + count).
struct BlitEntry {
    const void* src = nullptr;
    void* dst = nullptr;
    uint32_t src_pitch = 0, dst_pitch = 0, row_bytes = 0, rows = 0, unit = 1, pad = 0;
};
int blit_unit(const void* src, const void* dst, uint32_t src_pitch, uint32_t dst_pitch);
// max_rows / max_row_bytes: the largest plane among the entries (work is cut into slices of ~64 KB of rows); workgroups:
// how many run at once (the link bounds the kernel, the rest of the chip stays free for the resampling kernels).
int launch_blit_rows(const BlitEntry* table_device, int first, int count, uint32_t max_rows, uint32_t max_row_bytes, int workgroups,
                     void* stream);

// Measurement hook (kernel_probe.hip): `samplers` single-lane workgroups stamp the shader clock counter and the 100 MHz
// real-time counter until *stop_flag (device memory) becomes non-zero or max_seconds pass; out[2 k] = shader ticks,
// out[2 k + 1] = real-time ticks of sampler k.
int launch_clock_sampler(const int* stop_flag, unsigned long long* out, int samplers, double max_seconds, void* stream);

// Measurement hook: `blocks` x 256 lanes run iters x 8 (v_mul_f32 with an SGPR coefficient, v_add_f32 onto one chain);
// out holds blocks x 256 floats.
int launch_valu_pair_probe(float* out, int blocks, int iters, void* stream);

// Test hook: applies the kernels' float -> sample conversion (clamp, round-half-even, store) to n sums.
int launch_debug_convert(const float* in, void* out, int n, int sample_bytes, float peak, void* stream);

}  // namespace jinc
