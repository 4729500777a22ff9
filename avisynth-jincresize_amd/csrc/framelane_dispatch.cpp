// framelane_dispatch.cpp -- host side of the frame-lane kernel (kernel_framelane.hip): tile configuration.
#include <algorithm>
#include <cstddef>
#include <cstdlib>

#include "kernels.h"
#include "plan.h"
#include "knobs.h"

namespace jinc {

namespace {
// Largest source extent (window origins are non-decreasing) of a run of `t` consecutive outputs inside [i0, i1).
int max_extent(const std::vector<int32_t>& start, int i0, int i1, int t, int fs) {
    int best = 0;
    for (int a = i0; a < i1; a += t) {
        const int b = std::min(a + t, i1) - 1;
        best = std::max(best, start[b] + fs - start[a]);
    }
    return best;
}
}  // namespace

namespace {
// Tile choice and rectangle bookkeeping for one frame-lane form: `ps` LDS bytes per source position, `budget` bytes of LDS
// per workgroup, `frames_per_group` frames per workgroup; `col_weight` prices the window columns a strip loads per pixel
// (wide tiles) against the staged footprint per pixel.
bool configure_tiles(const PlanePlan& p, const RectList& rects, int ps, size_t budget, int frames_per_group, int nframes_hint,
                     double col_weight, FrameLaneArgs& out) {
    const int groups = std::max(1, (nframes_hint + frames_per_group - 1) / frames_per_group);
    // A/B knob FL_FILL_WEIGHT (default on): see the cost below (whole planes: no change of any choice measured; the 6-pixel
    // border frame of 1.5x with tap 4 at 256 frames: border kernel 2.02 -> 0.37 ms, step 259 -> 300 Gpix/s)
    const bool fill_weighted = knobs::flag(JINC_KNOB_FL_FILL_WEIGHT, true);

    bool found = false, found_enough = false;
    double best_cost = 0.0;
    for (int tys = 5; tys >= 2; --tys)
        for (int txs = 5; txs >= 2; --txs) {
            const int tx = 1 << txs, ty = 1 << tys;
            int max_tw = 0, max_th = 0;
            long long tiles = 0;
            for (int r = 0; r < rects.n; ++r) {
                if (rects.w[r] <= 0 || rects.h[r] <= 0) continue;
                max_tw = std::max(max_tw, max_extent(p.col_start, rects.x0[r], rects.x0[r] + rects.w[r], tx, p.fs));
                max_th = std::max(max_th, max_extent(p.row_start, rects.y0[r], rects.y0[r] + rects.h[r], ty, p.fs));
                tiles += static_cast<long long>((rects.w[r] + tx - 1) / tx) * ((rects.h[r] + ty - 1) / ty);
            }
            if (tiles <= 0 || tiles > (1ll << 30) || max_tw > 64) continue;
            // (+ 1 row: the sliding-window form pads the column pitch of its column-major tile to an odd number)
            const size_t bytes = kFrameLaneTableBytes(tys) + (static_cast<size_t>(max_tw) * (max_th + 1) + 8) * ps;
            if (bytes > budget) continue;
            double cost = static_cast<double>(max_tw) * max_th / (static_cast<double>(tx) * ty) + col_weight * max_tw / tx;
            if (fill_weighted) {  // thin rectangles (border frames): a tile's cost is spread over the pixels it really holds
                double area = 0.0;
                for (int r = 0; r < rects.n; ++r)
                    if (rects.w[r] > 0 && rects.h[r] > 0) area += static_cast<double>(rects.w[r]) * rects.h[r];
                cost *= static_cast<double>(tiles) * tx * ty / area;
            }
            const bool enough = tiles * groups >= 1024;  // >= 2 workgroups in flight per CU, twice over
            if (found && ((found_enough && !enough) || (enough == found_enough && cost >= best_cost))) continue;
            found = true;
            found_enough = enough;
            best_cost = cost;
            out.tx_shift = txs;
            out.ty_shift = tys;
            out.lds_bytes = static_cast<int>(bytes);
        }
    if (!found) return false;
    const int tx = 1 << out.tx_shift, ty = 1 << out.ty_shift;
    int total = 0;
    out.rects = RectList{};
    out.rects.n = rects.n;
    for (int r = 0; r < 4; ++r) {
        out.block_begin[r] = total;
        out.tiles_x[r] = 1;
        if (r < rects.n) {
            out.rects.x0[r] = rects.x0[r], out.rects.y0[r] = rects.y0[r], out.rects.w[r] = rects.w[r], out.rects.h[r] = rects.h[r];
            if (rects.w[r] > 0 && rects.h[r] > 0) {
                out.tiles_x[r] = (rects.w[r] + tx - 1) / tx;
                total += out.tiles_x[r] * ((rects.h[r] + ty - 1) / ty);
            }
        }
    }
    out.block_begin[4] = total;
    return total > 0;
}
}  // namespace

bool framelane_configure(const PlanePlan& p, const RectList& rects, int sample_bytes, int nframes_hint, FrameLaneArgs& out) {
    const int ps = kFrameLanePosBytes(static_cast<size_t>(sample_bytes));
    // LDS per workgroup: a larger tile has less halo (fewer staged samples per output pixel), a smaller one lets more
    // workgroups share a CU.  One workgroup may not exceed 64 KB of dynamic LDS.
    size_t budget = 64 * 1024;  // A/B (64 frames): 1.37x fs 7 48 KB = 64 KB; 5/6 down-scale fs 8 +28 % over 48 KB
    int variant = 0;
    variant = knobs::geti(JINC_KNOB_FL_VARIANT, variant);  // A/B knob: 1 = row-segment form always
    const bool window_form = variant != 1 && (p.fs == 5 || p.fs == 7 || p.fs == 8 || p.fs == 9);
    // fs 7: the 1024-thread shape of the sliding-window kernel (two workgroups per CU at 64 VGPRs = 8 waves per SIMD) with
    // tiles of up to 80 KB
    bool big = window_form && p.fs == 7;
    big = big && knobs::flag(JINC_KNOB_FL_1K, true);  // A/B knob
    // the other sliding-window forms (512 threads, two workgroups per CU either way) take up to 80 KB too: 1.5x with tap 4
    // (fs 9) gets 32 x 32 tiles, 46.6 -> 50.4 % of the VALU peak; the row-segment form is launched without the attribute
    if (big || (window_form && p.fs != 7)) budget = 80 * 1024;
    // the row-segment form (fs > 9: 80 VGPRs, 6 waves per SIMD) gains more from a third workgroup per CU than it loses to the
    // smaller tile's halo: 1.5x with tap 8, 64 frames: 40 KB 43 %, 48 KB 53 %, 56 / 64 KB 49 % of the VALU peak
    if (p.fs > 9) budget = 48 * 1024;
    const size_t cap = (big || (window_form && p.fs != 7)) ? 80 * 1024 : 64 * 1024;
    if (knobs::is_set(JINC_KNOB_FL_LDS_KB)) budget = static_cast<size_t>(knobs::geti(JINC_KNOB_FL_LDS_KB, 0)) * 1024;  // tuning knob
    budget = std::min(budget, cap);
    double colw = 0.5;  // (1.5x with tap 8: 56.7 -> 58.1 % over 0; the window forms' choices do not change up to 2)
    colw = knobs::get(JINC_KNOB_FL_COLW, colw);  // tuning knob: price of a strip's window columns
    if (!configure_tiles(p, rects, ps, budget, 64, nframes_hint, colw, out)) return false;
    const int tx = 1 << out.tx_shift, ty = 1 << out.ty_shift;
    const int units = (tx / 4) * (ty / 4);
    out.threads = 64 * std::min(8, std::max(1, units));
    if (big && units >= 16) out.threads = 1024;  // (small tiles keep the 512-thread shape)
    if (knobs::is_set(JINC_KNOB_FL_THREADS)) out.threads = std::min(out.threads, std::max(64, knobs::geti(JINC_KNOB_FL_THREADS, 0) / 64 * 64));  // A/B knob
    out.variant = variant;
    out.pair = 0;
    return true;
}

// Frame-pair form (kernel_framelane_pair.hip): 128 frames per workgroup in up to 80 KB of LDS (two workgroups per CU).
bool framelane_pair_configure(const PlanePlan& p, const RectList& rects, int sample_bytes, int nframes_hint, FrameLaneArgs& out) {
    if (p.fs != 5 && p.fs != 7) return false;
    const int ps = kFrameLanePairPosBytes(static_cast<size_t>(sample_bytes));
    size_t budget = 80 * 1024;
    if (knobs::is_set(JINC_KNOB_FLP_LDS_KB)) budget = std::min<size_t>(budget, static_cast<size_t>(knobs::geti(JINC_KNOB_FLP_LDS_KB, 0)) * 1024);  // tuning knob
    double colw = 0.5;
    colw = knobs::get(JINC_KNOB_FLP_COLW, colw);  // tuning knob: price of a strip's window columns
    if (!configure_tiles(p, rects, ps, budget, kFrameLanePairFrames, nframes_hint, colw, out)) return false;
    const int ty = 1 << out.ty_shift;
    out.threads = 64 * std::min(8, ty);  // a wave walks whole strips (output rows of the tile)
    if (knobs::is_set(JINC_KNOB_FLP_THREADS)) out.threads = std::min(out.threads, std::max(64, knobs::geti(JINC_KNOB_FLP_THREADS, 0) / 64 * 64));  // A/B knob
    out.variant = 0;
    out.pair = 1;
    return true;
}

}  // namespace jinc
