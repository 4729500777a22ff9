// plan.cpp -- see plan.h.  Compile with -ffp-contract=off, no -ffast-math, no -march flags:
// coefficient values must match the reference's table generator bit for bit
// ("ref:" = /root/reference/src/JincResize.cpp).
#include "plan.h"

#include <algorithm>

#include <cmath>
#include <cstring>
#include <stdexcept>
#include <string>
#include <unordered_map>

namespace jinc {
namespace {

// avs/minmax.h clamp as the reference uses it (JincResize.h:9): upper bound first, then lower.
inline float clamp_pos(float v, float lo, float hi) {
    v = v > hi ? hi : v;
    return v < lo ? lo : v;
}

// Interning table for per-axis distance profiles (fs squared distances each).
class ProfileTable {
public:
    explicit ProfileTable(int fs) : fs_(fs) {}
    int intern(const double* d2) {
        std::string key(reinterpret_cast<const char*>(d2), sizeof(double) * fs_);
        auto it = ids_.find(key);
        if (it != ids_.end()) return it->second;
        const int id = static_cast<int>(data_.size() / fs_);
        data_.insert(data_.end(), d2, d2 + fs_);
        ids_.emplace(std::move(key), id);
        return id;
    }
    const double* get(int id) const { return data_.data() + static_cast<size_t>(id) * fs_; }

private:
    int fs_;
    std::vector<double> data_;
    std::unordered_map<std::string, int> ids_;
};

// Everything the reference's inner loops derive from one axis coordinate.
struct AxisScan {
    int n = 0;
    std::vector<int32_t> start;       // meta start (ref :406-421)
    std::vector<uint8_t> border;      // axis contribution to is_border (ref :395-418)
    std::vector<int32_t> phase;       // quantised phase value (ref :426-427); interior entries only
    std::vector<int32_t> prof_border; // profile when the pixel is a border pixel (un-quantised position)
    std::vector<int32_t> prof_inner;  // profile when the pixel is interior (quantised position), else -1
};

// pos0/pos_step: float start and increment (ref :358-364); support: common float support (ref :355);
// step: filter_step of this axis (ref :349-350).
AxisScan scan_axis(int n, int src, int quant, int fs, float pos0, float pos_step, float support, double step,
                   ProfileTable& profiles) {
    AxisScan a;
    a.n = n;
    a.start.resize(n);
    a.border.resize(n);
    a.phase.assign(n, -1);
    a.prof_border.resize(n);
    a.prof_inner.assign(n, -1);
    std::vector<double> d2(fs);
    const float hi = static_cast<float>(src - 1);

    float pos = pos0;
    for (int i = 0; i < n; ++i) {
        bool border = false;
        int win_end = static_cast<int>(pos + support);  // ref :392-393
        if (win_end >= src) {
            win_end = src - 1;
            border = true;
        }
        int win_begin = win_end - fs + 1;
        if (win_begin < 0) {
            win_begin = 0;
            border = true;
        }
        a.start[i] = win_begin;
        a.border[i] = border;

        // Border-mode profile: un-quantised position against the (clamped) meta window (ref :485-486).
        {
            const float p = clamp_pos(pos, 0.f, hi);
            for (int l = 0; l < fs; ++l) {
                const double d = (p - (win_begin + l)) * step;  // float subtract, double multiply
                d2[l] = d * d;
            }
            a.prof_border[i] = profiles.intern(d2.data());
        }

        if (!border) {
            // ref :424-429, :446-451: interior sets are evaluated at the quantised position with a
            // window re-derived from it (the meta window above is NOT changed).
            const int q_int = static_cast<int>(pos * quant);
            a.phase[i] = q_int % quant;
            if (a.phase[i] < 0)  // cannot happen for an interior window (pos >= support - 1 > 0)
                throw std::runtime_error("JincResize: negative interior phase.");
            const float q_pos = static_cast<float>(q_int) / quant;
            const int q_begin = static_cast<int>(q_pos + support) - fs + 1;
            const float p = clamp_pos(q_pos, 0.f, hi);
            for (int l = 0; l < fs; ++l) {
                const double d = (p - (q_begin + l)) * step;
                d2[l] = d * d;
            }
            a.prof_inner[i] = profiles.intern(d2.data());
        }
        pos += pos_step;  // ref :524 / :527
    }
    return a;
}

// Finds the smallest period p (<= max_p) of (class, start) over the interior range [i0, i1).
bool find_period(const std::vector<int32_t>& cls, const std::vector<int32_t>& start, int i0, int i1, int max_p,
                 int& period, int& advance) {
    const int len = i1 - i0;
    for (int p = 1; p <= max_p && 2 * p <= len; ++p) {
        const int s = start[i0 + p] - start[i0];
        if (s < 0) continue;
        bool ok = true;
        for (int i = i0; i + p < i1; ++i) {
            if (cls[i + p] != cls[i] || start[i + p] - start[i] != s) {
                ok = false;
                break;
            }
        }
        if (ok) {
            period = p;
            advance = s;
            return true;
        }
    }
    return false;
}

// Smallest period p (<= max_p) under which the window origins alone are affine over [i0, i1).
bool find_affine_period(const std::vector<int32_t>& start, int i0, int i1, int max_p, int& period, int& advance) {
    const int len = i1 - i0;
    for (int p = 1; p <= max_p && 2 * p <= len; ++p) {
        const int s = start[i0 + p] - start[i0];
        if (s < 1) continue;
        bool ok = true;
        for (int i = i0; i + p < i1; ++i)
            if (start[i + p] - start[i] != s) {
                ok = false;
                break;
            }
        if (ok) {
            period = p;
            advance = s;
            return true;
        }
    }
    return false;
}

}  // namespace

PlanePlan build_plane_plan(const JincLut& lut, const TableGeometry& g) {
    if (g.src_w <= 0 || g.src_h <= 0 || g.dst_w <= 0 || g.dst_h <= 0)
        throw std::runtime_error("JincResize: plane dimensions must be positive.");
    if (!(g.crop_w > 0.0) || !(g.crop_h > 0.0))
        throw std::runtime_error("JincResize: source crop must have positive width and height.");

    PlanePlan plan;
    plan.g = g;

    // ref :349-356
    const double ratio_x = static_cast<double>(g.dst_w) / g.crop_w;
    const double ratio_y = static_cast<double>(g.dst_h) / g.crop_h;
    const double step_x = ratio_x < 1.0 ? ratio_x : 1.0;
    const double step_y = ratio_y < 1.0 ? ratio_y : 1.0;
    const float support_x = static_cast<float>(g.radius / step_x);
    const float support_y = static_cast<float>(g.radius / step_y);
    const float support = support_x > support_y ? support_x : support_y;
    const int fs_x = static_cast<int>(std::ceil(support_x * 2.0));
    const int fs_y = static_cast<int>(std::ceil(support_y * 2.0));
    const int fs = fs_x > fs_y ? fs_x : fs_y;
    plan.fs = fs;
    if (fs > g.src_w || fs > g.src_h)
        throw std::runtime_error(
            "JincResize: source plane is smaller than the filter footprint (the reference reads out of bounds here).");
    if (fs > 1024) throw std::runtime_error("JincResize: filter footprint too large.");

    // ref :358-364
    const float x0 = static_cast<float>(g.crop_left + (g.crop_w / g.dst_w - 1.0) / 2.0);
    const float x_step = static_cast<float>(g.crop_w / g.dst_w);
    const float y_step = static_cast<float>(g.crop_h / g.dst_h);
    const float y0 = static_cast<float>(g.crop_top + (g.crop_h - g.dst_h) / (g.dst_h * static_cast<int64_t>(2)));

    ProfileTable col_profiles(fs), row_profiles(fs);
    const AxisScan cols = scan_axis(g.dst_w, g.src_w, g.quant_x, fs, x0, x_step, support, step_x, col_profiles);
    const AxisScan rows = scan_axis(g.dst_h, g.src_h, g.quant_y, fs, y0, y_step, support, step_y, row_profiles);

    plan.col_start = cols.start;
    plan.row_start = rows.start;
    // The kernels size their source tiles from the first and last window origin of a block.
    for (int i = 1; i < g.dst_w; ++i)
        if (cols.start[i] < cols.start[i - 1]) throw std::runtime_error("JincResize: window origins are not monotonic in x.");
    for (int i = 1; i < g.dst_h; ++i)
        if (rows.start[i] < rows.start[i - 1]) throw std::runtime_error("JincResize: window origins are not monotonic in y.");

    // ---- coefficient sets, one per distinct (column profile, row profile) pair -------------------
    const double radius2 = g.radius * g.radius;  // ref :377
    constexpr double kRoundMagic = 6755399441055744.0;  // ref :284
    std::unordered_map<uint64_t, int32_t> pair_to_set;
    std::vector<float> tmp(static_cast<size_t>(fs) * fs);
    auto set_for = [&](int col_prof, int row_prof) -> int32_t {
        const uint64_t key = (static_cast<uint64_t>(static_cast<uint32_t>(row_prof)) << 32) | static_cast<uint32_t>(col_prof);
        auto it = pair_to_set.find(key);
        if (it != pair_to_set.end()) return it->second;
        const double* dx2 = col_profiles.get(col_prof);
        const double* dy2 = row_profiles.get(row_prof);
        float divider = 0.f;
        for (int ly = 0; ly < fs; ++ly) {  // ref :480-502
            for (int lx = 0; lx < fs; ++lx) {
                const int index =
                    static_cast<int>(std::llround((kLutSamples - 1) * (dx2[lx] + dy2[ly]) / radius2 + kRoundMagic));
                const float f = lut.factor(index);
                tmp[static_cast<size_t>(ly) * fs + lx] = f;
                divider += f;
            }
        }
        for (float& c : tmp) c /= divider;  // ref :505-514
        const int32_t id = plan.num_sets++;
        plan.coeffs.insert(plan.coeffs.end(), tmp.begin(), tmp.end());
        pair_to_set.emplace(key, id);
        return id;
    };

    // ---- interior classes: phase -> first interior index with that phase (raster first-come) ------
    auto classify = [](const AxisScan& a, int quant, std::vector<int32_t>& cls, int& n_classes, int& n_border,
                       std::vector<int32_t>& class_profile) {
        std::vector<int32_t> phase_to_class(quant, -1);
        cls.resize(a.n);
        n_classes = 0;
        n_border = 0;
        for (int i = 0; i < a.n; ++i) {
            if (a.border[i]) {
                cls[i] = ~(n_border++);
                continue;
            }
            int32_t& c = phase_to_class[a.phase[i]];
            if (c < 0) {
                c = n_classes++;
                class_profile.push_back(a.prof_inner[i]);  // profile of the first-come representative
            }
            cls[i] = c;
        }
    };
    std::vector<int32_t> col_class_profile, row_class_profile;
    classify(cols, g.quant_x, plan.col_class, plan.n_col_classes, plan.n_bcols, col_class_profile);
    classify(rows, g.quant_y, plan.row_class, plan.n_row_classes, plan.n_brows, row_class_profile);

    plan.interior_set.resize(static_cast<size_t>(plan.n_row_classes) * plan.n_col_classes);
    for (int r = 0; r < plan.n_row_classes; ++r)
        for (int c = 0; c < plan.n_col_classes; ++c)
            plan.interior_set[static_cast<size_t>(r) * plan.n_col_classes + c] =
                set_for(col_class_profile[c], row_class_profile[r]);

    // ---- border pixels: un-quantised profiles on both axes ---------------------------------------
    plan.brow_set.resize(static_cast<size_t>(plan.n_brows) * g.dst_w);
    plan.bcol_set.resize(static_cast<size_t>(plan.n_bcols) * g.dst_h);
    for (int y = 0; y < g.dst_h; ++y) {
        if (plan.row_class[y] >= 0) continue;
        int32_t* out = plan.brow_set.data() + static_cast<size_t>(~plan.row_class[y]) * g.dst_w;
        for (int x = 0; x < g.dst_w; ++x) out[x] = set_for(cols.prof_border[x], rows.prof_border[y]);
    }
    for (int x = 0; x < g.dst_w; ++x) {
        if (plan.col_class[x] >= 0) continue;
        int32_t* out = plan.bcol_set.data() + static_cast<size_t>(~plan.col_class[x]) * g.dst_h;
        for (int y = 0; y < g.dst_h; ++y) out[y] = set_for(cols.prof_border[x], rows.prof_border[y]);
    }

    // ---- interior rectangle and its periodicity ---------------------------------------------------
    auto interior_range = [](const std::vector<int32_t>& cls, int& i0, int& i1) {
        const int n = static_cast<int>(cls.size());
        i0 = 0;
        while (i0 < n && cls[i0] < 0) ++i0;
        i1 = i0;
        while (i1 < n && cls[i1] >= 0) ++i1;
        for (int i = i1; i < n; ++i)
            if (cls[i] >= 0) return false;  // interior not contiguous: treat everything as generic
        return true;
    };
    const bool cx = interior_range(plan.col_class, plan.ix0, plan.ix1);
    const bool cy = interior_range(plan.row_class, plan.iy0, plan.iy1);
    plan.periodic = false;
    if (cx && cy && plan.ix1 > plan.ix0 && plan.iy1 > plan.iy0) {
        constexpr int kMaxPeriod = 8;
        int px, py, sx, sy;
        if (find_period(plan.col_class, plan.col_start, plan.ix0, plan.ix1, kMaxPeriod, px, sx) &&
            find_period(plan.row_class, plan.row_start, plan.iy0, plan.iy1, kMaxPeriod, py, sy)) {
            plan.periodic = true;
            plan.px = px;
            plan.py = py;
            plan.sx = sx;
            plan.sy = sy;
        }
        int qpx, qpy, qsx, qsy;
        constexpr int kMaxAffinePeriod = 16;
        if (find_affine_period(plan.col_start, plan.ix0, plan.ix1, kMaxAffinePeriod, qpx, qsx) &&
            find_affine_period(plan.row_start, plan.iy0, plan.iy1, kMaxAffinePeriod, qpy, qsy)) {
            plan.quasi = true;
            plan.qpx = qpx;
            plan.qpy = qpy;
            plan.qsx = qsx;
            plan.qsy = qsy;
        }
    } else if (!cx || !cy) {
        plan.ix0 = plan.ix1 = plan.iy0 = plan.iy1 = 0;
    }
    return plan;
}

bool build_plan_runs(const PlanePlan& p, std::vector<PlanRun>& runs, std::vector<int32_t>& item_run) {
    runs.clear();
    item_run.clear();
    if (p.periodic || !p.quasi || p.qpx < 1 || p.qpy < 1) return false;
    const int px = p.qpx, py = p.qpy, sx = p.qsx, sy = p.qsy;
    const int ni = (p.ix1 - p.ix0) / px, nj = (p.iy1 - p.iy0) / py;
    if (ni < 1 || nj < 1) return false;
    struct AxisRun {
        int first, count, cls;
    };
    bool ok = true;
    auto axis_runs = [&](const std::vector<int32_t>& cls, const std::vector<int32_t>& start, int o0, int period, int step, int n, int phase) {
        std::vector<AxisRun> r;
        for (int i = 0; i < n; ++i) {
            const int x = o0 + period * i + phase;
            if (cls[x] < 0 || start[x] != start[o0 + phase] + step * i) ok = false;  // not what PlanePlan::quasi promises
            if (r.empty() || r.back().cls != cls[x])
                r.push_back({i, 1, cls[x]});
            else
                ++r.back().count;
        }
        return r;
    };
    for (int q = 0; q < py; ++q) {
        const std::vector<AxisRun> ry = axis_runs(p.row_class, p.row_start, p.iy0, py, sy, nj, q);
        for (int r = 0; r < px; ++r) {
            const std::vector<AxisRun> rx = axis_runs(p.col_class, p.col_start, p.ix0, px, sx, ni, r);
            if (!ok) {
                runs.clear();
                return false;
            }
            for (const AxisRun& b : ry)
                for (const AxisRun& a : rx) {
                    PlanRun d;
                    d.set = p.interior_set[static_cast<std::size_t>(b.cls) * p.n_col_classes + a.cls];
                    d.x0 = p.ix0 + px * a.first + r;
                    d.y0 = p.iy0 + py * b.first + q;
                    d.sx0 = p.col_start[d.x0];
                    d.sy0 = p.row_start[d.y0];
                    d.ni = a.count;
                    d.nj = b.count;
                    runs.push_back(d);
                }
        }
    }
    // rectangles of the same neighbourhood next to each other: consecutive items then read the same source rows
    std::sort(runs.begin(), runs.end(), [](const PlanRun& a, const PlanRun& b) { return a.y0 != b.y0 ? a.y0 < b.y0 : a.x0 < b.x0; });
    for (std::size_t k = 0; k < runs.size(); ++k) {
        const long long lanes = static_cast<long long>((runs[k].ni + 3) / 4) * ((runs[k].nj + 3) / 4);  // 4 x 4 periods per lane
        runs[k].first_item = static_cast<int32_t>(item_run.size());
        item_run.insert(item_run.end(), static_cast<std::size_t>((lanes + 63) / 64), static_cast<int32_t>(k));
    }
    return !item_run.empty() && item_run.size() <= 0x3fffffffu;
}

}  // namespace jinc
