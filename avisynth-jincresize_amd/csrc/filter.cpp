// filter.cpp -- host side of libjincresize_hip.so: the C ABI declared in include/jincresize_hip.h.
//
// Mirrors the reference plugin's filter life cycle for the accelerated path
// ("ref:" = /root/reference/src/JincResize.cpp):
//   jinc_filter_create      <- Create_JincResize      (ref :654-984)  same defaults, same checks in
//                                                      the same order, same error strings
//   jinc_filter_get_frame   <- process_frame call in JincResize_GetFrame (ref :615)
//   jinc_filter_free        <- free_JincResize         (ref :632-647)
//   jinc_alias_args         <- resizer()/resizer_jincresize<taps> (ref :1007-1040)
// There is no CPU fallback: without a HIP device every frame call fails loudly.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/jincresize_hip.h"
#include "jinc_lut.h"
#include "kernels.h"
#include "plan.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

struct HipError : std::runtime_error {
    explicit HipError(const std::string& what) : std::runtime_error(what) {}
};

void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess) throw HipError(std::string("JincResize: HIP error in ") + what + ": " + hipGetErrorString(e));
}

struct ArgError : std::runtime_error {
    explicit ArgError(const char* what) : std::runtime_error(what) {}
};

// One table's device-resident plan: a single allocation carved into the arrays of DevicePlan.
struct DeviceTable {
    void* blob = nullptr;
    size_t bytes = 0;
    jinc::DevicePlan plan;
    bool use_periodic = false;
    jinc::PeriodicArgs periodic;
    bool use_quasi = false;  // quasi-periodic interior kernel (affine window origins, drifting classes)
    jinc::QuasiArgs quasi;
    jinc::RectList border_rects;  // gather work when the periodic kernel covers the interior
    bool use_direct = false;      // exactly periodic, any filter size / source step (kernel_direct.hip)
    jinc::DirectArgs direct;      // interior
    jinc::DirectArgs row_strips;  // border rows of the same plan over the interior's columns (any interior kernel)
    bool strips_ok = false;       // the border rows / columns really repeat their coefficient sets per phase (plan_direct)
    bool use_colstrip = false;    // border columns over the interior's rows on kernel_colstrip.hip
    jinc::ColStripArgs col_strips;
    jinc::RectList corner_rects;  // ... then only the corners are left for the gather kernel
    jinc::RectList column_rects;  // otherwise: left / right columns, full height, on the gather kernel
    std::vector<void*> lane_blobs;  // lane-major coefficient copies of the private-set rectangles (RectList::lane_coeffs)
    jinc::RectList whole;         // gather work when it does not
    bool use_framelane = false;   // frame-lane kernel configured for the whole plane (batches of frames, any plan)
    jinc::FrameLaneArgs fl_whole;
    const char* last_kernel = "";  // interior kernel of the most recent call (reports)
};

struct EventPair {
    hipEvent_t start = nullptr, stop = nullptr;
};

struct DeviceFrameBuf {  // device staging planes of one in-flight frame (host-pointer entry points)
    void* src[4] = {nullptr, nullptr, nullptr, nullptr};
    void* dst[4] = {nullptr, nullptr, nullptr, nullptr};
    int src_pitch[4] = {0, 0, 0, 0};
    int dst_pitch[4] = {0, 0, 0, 0};
    hipStream_t stream = nullptr;  // slot 0 uses the filter's stream, further slots own theirs
    bool ready = false;
    bool busy = false;             // work enqueued and not yet waited for
    long long ticket = -1;
};

struct PinnedRange {  // a caller buffer registered with hipHostRegister (look-ahead pipeline, opt-in)
    char* base = nullptr;
    size_t bytes = 0;
    unsigned long long stamp = 0;
    long long ticket = -1;  // latest frame whose copies use this range (may still be in flight)
};

}  // namespace

struct jinc_filter {
    jinc_video_info vi_in{};
    jinc_video_info vi_out{};
    std::string cplace;
    int chroma_location = -1;
    float peak = 0.f;
    int planecount = 0;
    bool subsampled = false;
    jinc::JincLut lut;
    std::vector<jinc::PlanePlan> plans;  // [0] luma / all planes, [1] chroma of subsampled formats
    int kernel_mode = 0;
    int border_strips = 1;  // border rows/columns of exactly periodic plans on kernel_direct.hip (0: gather kernel)
    bool direct_premise = false;  // buffer_range_check_covers_soffset(device) == 1
    int simd_order = 0;  // 0: opt=0 results (default); 1 / 2 / 3: summation order of the reference's SSE4.1 / AVX2 / AVX-512 path
    int overlap_border = -1;  // -1: automatic (side stream when the border frame is heavy: fs > 9), 0: off, 1: on

    int device = -1;  // -1: host-only instance (plan inspection); frame calls fail
    hipStream_t stream = nullptr;
    std::vector<DeviceTable> tables;
    std::vector<DeviceFrameBuf> slots = std::vector<DeviceFrameBuf>(1);  // frames in flight (pipeline depth)
    long long next_ticket = 0;
    bool register_host = false;
    std::vector<PinnedRange> pinned;
    unsigned long long pin_clock = 0;
    bool profiling = false;
    std::vector<EventPair> ev_periodic, ev_gather;  // recorded, not yet collected
    // The border gather kernel (load/store-issue bound) runs on a side stream next to the periodic
    // interior kernel (VALU bound): fork/join with two reusable events.
    hipStream_t aux_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;

    ~jinc_filter() {
        if (device >= 0) {
            (void)hipSetDevice(device);
            for (auto& t : tables) {
                if (t.blob) (void)hipFree(t.blob);
                for (void* b : t.lane_blobs) (void)hipFree(b);
            }
            for (size_t s = 0; s < slots.size(); ++s) {
                for (int i = 0; i < 4; ++i) {
                    if (slots[s].src[i]) (void)hipFree(slots[s].src[i]);
                    if (slots[s].dst[i]) (void)hipFree(slots[s].dst[i]);
                }
                if (s > 0 && slots[s].stream) (void)hipStreamDestroy(slots[s].stream);
            }
            for (auto& p : pinned) (void)hipHostUnregister(p.base);
            for (auto* v : {&ev_periodic, &ev_gather})
                for (auto& e : *v) {
                    (void)hipEventDestroy(e.start);
                    (void)hipEventDestroy(e.stop);
                }
            if (ev_fork) (void)hipEventDestroy(ev_fork);
            if (ev_join) (void)hipEventDestroy(ev_join);
            if (aux_stream) (void)hipStreamDestroy(aux_stream);
            if (stream) (void)hipStreamDestroy(stream);
        }
    }

    int table_of_plane(int i) const { return (subsampled && (i == 1 || i == 2)) ? 1 : 0; }  // ref :552-558
    void plane_dims(const jinc_video_info& vi, int i, int& w, int& h) const {
        w = vi.width;
        h = vi.height;
        if (subsampled && (i == 1 || i == 2)) {
            w >>= vi.sub_w;
            h >>= vi.sub_h;
        }
    }
};

namespace {

bool is_yuv_subsampled(const jinc_video_info& vi, int sw, int sh) {
    return !vi.is_rgb && vi.num_components >= 3 && vi.sub_w == sw && vi.sub_h == sh;
}

std::string lower(std::string s) {
    for (auto& c : s) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
    return s;
}

// ---- Create_JincResize argument handling (ref :700-789), then geometry (ref :791-866) ------------
void configure(jinc_filter& f, const jinc_video_info& vi, const jinc_args& a) {
    auto has = [&](unsigned bit) { return (a.defined & bit) != 0; };

    if (!vi.is_planar) throw ArgError("JincResize: clip must be in planar format.");

    const int tap = has(JINC_ARG_TAP) ? a.tap : 3;
    if (tap < 1 || tap > 16) throw ArgError("JincResize: tap must be between 1..16.");

    const int quant_x = has(JINC_ARG_QUANT_X) ? a.quant_x : 256;
    if (quant_x < 1 || quant_x > 256) throw ArgError("JincResize: quant_x must be between 1..256.");
    const int quant_y = has(JINC_ARG_QUANT_Y) ? a.quant_y : 256;
    if (quant_y < 1 || quant_y > 256) throw ArgError("JincResize: quant_y must be between 1..256.");

    std::string cplace = (has(JINC_ARG_CPLACE) && a.cplace) ? a.cplace : "";
    if (!cplace.empty()) {
        cplace = lower(cplace);
        if (cplace != "mpeg2" && cplace != "mpeg1" && cplace != "topleft")
            throw ArgError("JincResize: cplace must be MPEG2, MPEG1 or topleft.");
    } else {
        if (a.frame0_chroma_location >= 0) {  // the property exists and is an integer (ref :730)
            switch (a.frame0_chroma_location) {
                case 0: cplace = "mpeg2"; break;
                case 1: cplace = "mpeg1"; break;
                case 2: cplace = "topleft"; break;
                default: throw ArgError("JincResize: invalid _ChromaLocation");
            }
        } else {
            cplace = "mpeg2";
        }
    }
    const bool is_420 = is_yuv_subsampled(vi, 1, 1);
    if (cplace == "topleft" && !is_420)
        throw ArgError("JincResize: topleft must be used only for 4:2:0 chroma subsampling.");

    const int opt = has(JINC_ARG_OPT) ? a.opt : -1;
    if (opt > 3) throw ArgError("JincResize: opt higher than 3 is not allowed.");
    if (opt == 3 && !a.cpu_has_avx512f) throw ArgError("JincResize: opt=3 requires AVX-512F.");
    if (opt == 2 && !a.cpu_has_avx2) throw ArgError("JincResize: opt=2 requires AVX2.");
    if (opt == 1 && !a.cpu_has_sse41) throw ArgError("JincResize: opt=1 requires SSE4.1.");

    const int threads = has(JINC_ARG_THREADS) ? a.threads : 0;
    if (threads < 0 || threads > 1) throw ArgError("JincResize: threads must be either 0 or 1.");

    double crop_left = has(JINC_ARG_SRC_LEFT) ? a.src_left : 0.0;
    double crop_width = has(JINC_ARG_SRC_WIDTH) ? a.src_width : static_cast<double>(vi.width);
    if (crop_width <= 0.0) crop_width = vi.width - crop_left + crop_width;
    double crop_top = has(JINC_ARG_SRC_TOP) ? a.src_top : 0.0;
    double crop_height = has(JINC_ARG_SRC_HEIGHT) ? a.src_height : static_cast<double>(vi.height);
    if (crop_height <= 0.0) crop_height = vi.height - crop_top + crop_height;

    double blur = has(JINC_ARG_BLUR) ? a.blur : 0.0;
    if (!blur) blur = 1.0;

    const int target_width = a.target_width;
    const int target_height = a.target_height;

    const double initial_factor = has(JINC_ARG_INITIAL_FACTOR) ? a.initial_factor : 1.50;
    if (initial_factor < 1.0) throw ArgError("JincResize: initial_factor must be eqaul to or greater than 1.0.");

    const int src_width = vi.width;
    const int src_height = vi.height;
    const int initial_capacity = has(JINC_ARG_INITIAL_CAPACITY)
                                     ? a.initial_capacity
                                     : std::max(target_width * target_height, src_width * src_height);
    if (initial_capacity <= 0) throw ArgError("JincResize: initial_capacity must be greater than 0.");

    // ---- ref :791-866 ----
    f.vi_in = vi;
    f.vi_out = vi;
    f.vi_out.width = target_width;
    f.vi_out.height = target_height;
    f.cplace = cplace;
    f.peak = vi.bits_per_component <= 16 ? static_cast<float>((1 << vi.bits_per_component) - 1) : 0.f;
    f.planecount = vi.num_components;
    const double radius = jinc::jinc_radius(tap);
    jinc::build_lut(f.lut, radius, blur);

    jinc::TableGeometry g;
    g.quant_x = quant_x;
    g.quant_y = quant_y;
    g.src_w = src_width;
    g.src_h = src_height;
    g.dst_w = target_width;
    g.dst_h = target_height;
    g.radius = radius;
    g.crop_left = crop_left;
    g.crop_top = crop_top;
    g.crop_w = crop_width;
    g.crop_h = crop_height;

    const bool is_444 = !vi.is_rgb && vi.sub_w == 0 && vi.sub_h == 0;
    f.subsampled = f.planecount > 1 && !(is_444 || vi.is_rgb);
    f.plans.push_back(jinc::build_plane_plan(f.lut, g));
    if (f.subsampled) {
        const double div_w = 1 << vi.sub_w;
        const double div_h = 1 << vi.sub_h;
        const double crop_left_uv =
            (cplace == "mpeg2" || cplace == "topleft")
                ? (0.5 * (1.0 - static_cast<double>(src_width) / target_width) + crop_left) / div_w
                : crop_left / div_w;
        const double crop_top_uv =
            (cplace == "topleft") ? (0.5 * (1.0 - static_cast<double>(src_height) / target_height) + crop_top) / div_h
                                  : crop_top / div_h;
        jinc::TableGeometry gc = g;
        gc.src_w = src_width >> vi.sub_w;
        gc.src_h = src_height >> vi.sub_h;
        gc.dst_w = target_width >> vi.sub_w;
        gc.dst_h = target_height >> vi.sub_h;
        gc.crop_left = crop_left_uv;
        gc.crop_top = crop_top_uv;
        gc.crop_w = crop_width / div_w;
        gc.crop_h = crop_height / div_h;
        f.plans.push_back(jinc::build_plane_plan(f.lut, gc));
    }

    // ref :617-625
    if (is_420 || is_yuv_subsampled(vi, 1, 0) || is_yuv_subsampled(vi, 2, 0))
        f.chroma_location = cplace == "mpeg2" ? 0 : (cplace == "mpeg1" ? 1 : 2);
    else
        f.chroma_location = -1;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Smallest batch the frame-lane kernel takes over from the gather kernel: its lanes are the frames of the batch, so a
// batch of n < 64 frames fills n of 64 lanes.
constexpr int kFrameLaneMinFrames = 16;

// Smallest stride P <= 8 such that at least 90 % of the interior coordinates keep their class when
// stepping by P (1 if there is none).  Exact for periodic plans; for drifting ratios (1.5x, 3x) it is
// the nominal period, and the gather kernel's waterfall absorbs the deviations.
int dominant_period(const std::vector<int32_t>& cls) {
    const int n = static_cast<int>(cls.size());
    for (int P = 1; P <= 8; ++P) {
        long long same = 0, total = 0;
        for (int i = 0; i + P < n; ++i) {
            if (cls[i] < 0 || cls[i + P] < 0) continue;
            ++total;
            same += cls[i] == cls[i + P];
        }
        if (total > 0 && same * 10 >= total * 9) return P;
    }
    return 1;
}

void upload_table(const jinc::PlanePlan& p, DeviceTable& t, hipStream_t stream) {
    struct Piece {
        const void* host;
        size_t bytes;
        size_t offset;
    };
    std::vector<Piece> pieces;
    size_t off = 0;
    auto add = [&](const void* host, size_t bytes) {
        off = align_up(off, 256);
        pieces.push_back({host, bytes, off});
        off += bytes;
        return pieces.size() - 1;
    };
    const size_t i_cs = add(p.col_start.data(), p.col_start.size() * 4);
    const size_t i_rs = add(p.row_start.data(), p.row_start.size() * 4);
    const size_t i_cc = add(p.col_class.data(), p.col_class.size() * 4);
    const size_t i_rc = add(p.row_class.data(), p.row_class.size() * 4);
    const size_t i_is = add(p.interior_set.data(), p.interior_set.size() * 4);
    const size_t i_bc = add(p.bcol_set.data(), p.bcol_set.size() * 4);
    const size_t i_br = add(p.brow_set.data(), p.brow_set.size() * 4);
    // device layout of a set: fs rows of padded_fs floats (row stride a multiple of 16 bytes, zero padded)
    const int fsp = (p.fs + 3) & ~3;
    std::vector<float> padded(static_cast<size_t>(p.num_sets) * p.fs * fsp, 0.f);
    for (int s = 0; s < p.num_sets; ++s)
        for (int ly = 0; ly < p.fs; ++ly)
            std::memcpy(&padded[(static_cast<size_t>(s) * p.fs + ly) * fsp], p.set_ptr(s) + static_cast<size_t>(ly) * p.fs,
                        sizeof(float) * p.fs);
    const size_t i_co = add(padded.data(), padded.size() * 4);
    t.bytes = align_up(off, 256) + 256;  // slack: kernel_direct.hip fetches whole coefficient blocks (<= 16 floats)
    hip_check(hipMalloc(&t.blob, t.bytes), "hipMalloc(plan)");
    char* base = static_cast<char*>(t.blob);
    for (const Piece& pc : pieces)
        if (pc.bytes) hip_check(hipMemcpyAsync(base + pc.offset, pc.host, pc.bytes, hipMemcpyHostToDevice, stream), "plan upload");
    hip_check(hipStreamSynchronize(stream), "plan upload sync");

    auto ptr_i = [&](size_t i) { return reinterpret_cast<const int32_t*>(base + pieces[i].offset); };
    t.plan.col_start = ptr_i(i_cs);
    t.plan.row_start = ptr_i(i_rs);
    t.plan.col_class = ptr_i(i_cc);
    t.plan.row_class = ptr_i(i_rc);
    t.plan.interior_set = ptr_i(i_is);
    t.plan.bcol_set = ptr_i(i_bc);
    t.plan.brow_set = ptr_i(i_br);
    t.plan.coeffs = reinterpret_cast<const float*>(base + pieces[i_co].offset);
    t.plan.src_w = p.g.src_w;
    t.plan.src_h = p.g.src_h;
    t.plan.dst_w = p.g.dst_w;
    t.plan.dst_h = p.g.dst_h;
    t.plan.fs = p.fs;
    t.plan.n_col_classes = p.n_col_classes;
    t.plan.gather_period_x = dominant_period(p.col_class);
    t.plan.gather_period_y = dominant_period(p.row_class);
}

jinc::RectList border_frame(const jinc::PlanePlan& p, int x_end, int y_end);

// Decides how the output plane is split between the periodic kernel and the gather kernel.
void plan_launches(const jinc::PlanePlan& p, DeviceTable& t) {
    const int W = p.g.dst_w, H = p.g.dst_h;
    t.whole = jinc::RectList{};
    t.whole.n = 1;
    t.whole.w[0] = W;
    t.whole.h[0] = H;
    // Without an exactly periodic interior the lanes of an item rarely share a coefficient set (drifting classes,
    // or no structure at all): the gather kernel then skips its uniform passes and fetches coefficients per lane
    // (1.5x tap 8: 16 -> 26 Gpix/s, 1.37x: 41 -> 65, 5/6: 47 -> 57).
    t.whole.private_sets = !p.periodic;
    t.use_periodic = false;
    if (!p.periodic || !jinc::periodic_supported(p.fs, p.px, p.py, p.sx, p.sy)) return;

    jinc::PeriodicArgs pa;
    pa.coeffs = t.plan.coeffs;
    pa.px = p.px;
    pa.py = p.py;
    pa.ix0 = p.ix0;
    pa.iy0 = p.iy0;
    pa.ni = (p.ix1 - p.ix0) / p.px;
    pa.nj = (p.iy1 - p.iy0) / p.py;
    if (pa.ni < 1 || pa.nj < 1) return;
    int min_sx = INT32_MAX, max_sx = INT32_MIN, min_sy = INT32_MAX, max_sy = INT32_MIN;
    for (int q = 0; q < p.px; ++q) {
        pa.start_x[q] = p.col_start[p.ix0 + q];
        min_sx = std::min(min_sx, pa.start_x[q]);
        max_sx = std::max(max_sx, pa.start_x[q]);
    }
    for (int q = 0; q < p.py; ++q) {
        pa.start_y[q] = p.row_start[p.iy0 + q];
        min_sy = std::min(min_sy, pa.start_y[q]);
        max_sy = std::max(max_sy, pa.start_y[q]);
    }
    // The kernel's LDS tile has room for a phase spread of one source sample per axis.
    if (max_sx - min_sx > 1 || max_sy - min_sy > 1) return;
    pa.min_sx = min_sx;
    pa.min_sy = min_sy;
    for (int q = 0; q < p.py; ++q)
        for (int r = 0; r < p.px; ++r)
            pa.set[q * p.px + r] = p.interior_set[static_cast<size_t>(p.row_class[p.iy0 + q]) * p.n_col_classes +
                                                  p.col_class[p.ix0 + r]];
    pa.src_w = p.g.src_w;
    pa.src_h = p.g.src_h;
    pa.dst_h = p.g.dst_h;
    t.periodic = pa;
    t.use_periodic = true;

    t.border_rects = border_frame(p, p.ix0 + p.px * pa.ni, p.iy0 + p.py * pa.nj);
}

// The up-to-four rectangles around the interior block [ix0, x_end) x [iy0, y_end) of the output plane.
jinc::RectList border_frame(const jinc::PlanePlan& p, int x_end, int y_end) {
    const int W = p.g.dst_w, H = p.g.dst_h;
    jinc::RectList r;
    auto add = [&](int x0, int y0, int w, int h) {
        if (w <= 0 || h <= 0) return;
        r.x0[r.n] = x0;
        r.y0[r.n] = y0;
        r.w[r.n] = w;
        r.h[r.n] = h;
        ++r.n;
    };
    add(0, 0, W, p.iy0);                     // top rows
    add(0, y_end, W, H - y_end);             // bottom rows
    add(0, p.iy0, p.ix0, y_end - p.iy0);     // left columns
    add(x_end, p.iy0, W - x_end, y_end - p.iy0);  // right columns
    return r;
}

// Quasi-periodic interior (see kernels.h): used when the plan is not exactly periodic but its window
// origins are affine per residue, or when forced for A/B runs.
void plan_quasi(const jinc::PlanePlan& p, DeviceTable& t) {
    t.use_quasi = false;
    int px, py, sx, sy;
    if (p.quasi) {
        px = p.qpx, py = p.qpy, sx = p.qsx, sy = p.qsy;
    } else if (p.periodic) {
        px = p.px, py = p.py, sx = p.sx, sy = p.sy;
    } else {
        return;
    }
    if (!jinc::quasi_supported(p.fs, px, py, sx, sy, p.n_col_classes, p.n_row_classes)) return;
    jinc::QuasiArgs qa;
    qa.coeffs = t.plan.coeffs;
    qa.col_class = t.plan.col_class;
    qa.row_class = t.plan.row_class;
    qa.interior_set = t.plan.interior_set;
    qa.n_col_classes = p.n_col_classes;
    qa.n_row_classes = p.n_row_classes;
    qa.px = px, qa.py = py, qa.sx = sx, qa.sy = sy;
    qa.exact = p.periodic ? 1 : 2;  // 1: one set per phase; 2: drifting classes, per-lane coefficient registers
    qa.ix0 = p.ix0, qa.iy0 = p.iy0;
    qa.ni = (p.ix1 - p.ix0) / px;
    qa.nj = (p.iy1 - p.iy0) / py;
    if (qa.ni < 1 || qa.nj < 1) return;
    int min_sx = INT32_MAX, max_sx = INT32_MIN, min_sy = INT32_MAX, max_sy = INT32_MIN;
    for (int k = 0; k < px; ++k) {
        qa.start_x[k] = p.col_start[p.ix0 + k];
        min_sx = std::min(min_sx, qa.start_x[k]);
        max_sx = std::max(max_sx, qa.start_x[k]);
    }
    for (int k = 0; k < py; ++k) {
        qa.start_y[k] = p.row_start[p.iy0 + k];
        min_sy = std::min(min_sy, qa.start_y[k]);
        max_sy = std::max(max_sy, qa.start_y[k]);
    }
    qa.min_sx = min_sx, qa.min_sy = min_sy;
    qa.src_w = p.g.src_w, qa.src_h = p.g.src_h, qa.dst_h = p.g.dst_h;
    if (p.periodic)
        for (int q = 0; q < py; ++q)
            for (int r = 0; r < px; ++r)
                qa.phase_set[q * px + r] = p.interior_set[static_cast<size_t>(p.row_class[p.iy0 + q]) * p.n_col_classes +
                                                          p.col_class[p.ix0 + r]];
    if (!jinc::quasi_configure(qa, p.fs, max_sx - min_sx, max_sy - min_sy)) return;
    t.quasi = qa;
    t.use_quasi = true;
    if (!t.use_periodic) {
        t.border_rects = border_frame(p, p.ix0 + px * qa.ni, p.iy0 + py * qa.nj);
        t.border_rects.private_sets = !p.periodic;
        t.border_rects.unit_stride = !p.periodic;
    }
}

// Exactly periodic plans: kernel_direct.hip can take the interior (it is the choice for down-scales and taps > 8,
// which the register/LDS kernels do not cover) and, for every interior kernel, the border rows and columns.
void plan_direct(const jinc::PlanePlan& p, DeviceTable& t) {
    t.use_direct = false;
    if (!p.periodic || !jinc::direct_supported(p.fs, p.px, p.py, p.sx, p.sy)) return;
    jinc::DirectArgs da;
    da.coeffs = t.plan.coeffs;
    da.fs = p.fs;
    da.coeff_row = (p.fs + 3) & ~3;
    da.px = p.px, da.py = p.py, da.sx = p.sx, da.sy = p.sy;
    da.ix0 = p.ix0, da.iy0 = p.iy0;
    da.ni = (p.ix1 - p.ix0) / p.px;
    da.nj = (p.iy1 - p.iy0) / p.py;
    if (da.ni < 1 || da.nj < 1) return;
    for (int k = 0; k < p.px; ++k) da.start_x[k] = p.col_start[p.ix0 + k];
    for (int k = 0; k < p.py; ++k) da.start_y[k] = p.row_start[p.iy0 + k];
    da.dst_h = p.g.dst_h;
    da.plan = t.plan;
    const int x_end = p.ix0 + p.px * da.ni, y_end = p.iy0 + p.py * da.nj;
    const int W = p.g.dst_w, H = p.g.dst_h;

    t.row_strips = da;
    t.row_strips.line0[0] = 0, t.row_strips.line_n[0] = p.iy0;
    t.row_strips.line0[1] = y_end, t.row_strips.line_n[1] = H - y_end;
    jinc::RectList c;
    auto add = [&](int x0, int y0, int w, int h) {
        if (w <= 0 || h <= 0) return;
        c.x0[c.n] = x0, c.y0[c.n] = y0, c.w[c.n] = w, c.h[c.n] = h;
        ++c.n;
    };
    add(0, 0, p.ix0, H);
    add(x_end, 0, W - x_end, H);
    t.column_rects = c;
    c = jinc::RectList{};
    add(0, 0, p.ix0, p.iy0);
    add(x_end, 0, W - x_end, p.iy0);
    add(0, y_end, p.ix0, H - y_end);
    add(x_end, y_end, W - x_end, H - y_end);
    c.private_sets = true;  // corner pixels own a coefficient set each
    c.unit_stride = true;
    t.corner_rects = c;

    jinc::ColStripArgs ca;
    ca.coeffs = t.plan.coeffs;
    ca.fs = p.fs, ca.coeff_row = da.coeff_row;
    ca.py = p.py, ca.sy = p.sy, ca.iy0 = p.iy0, ca.nj = da.nj;
    int min_sy = INT32_MAX, max_sy = INT32_MIN;
    for (int k = 0; k < p.py; ++k) {
        ca.start_y[k] = da.start_y[k];
        min_sy = std::min(min_sy, da.start_y[k]);
        max_sy = std::max(max_sy, da.start_y[k]);
    }
    ca.min_sy = min_sy, ca.spread_y = max_sy - min_sy;
    ca.x0[0] = 0, ca.nx[0] = p.ix0;
    ca.x0[1] = x_end, ca.nx[1] = W - x_end;
    for (int s = 0; s < 2; ++s) {
        if (ca.nx[s] <= 0) continue;
        int lo = INT32_MAX, hi = INT32_MIN;  // window origins are non-decreasing in x, but do not rely on it
        for (int x = ca.x0[s]; x < ca.x0[s] + ca.nx[s]; ++x) {
            lo = std::min(lo, p.col_start[x]);
            hi = std::max(hi, p.col_start[x] + p.fs);
        }
        ca.src_c0[s] = lo, ca.src_w[s] = hi - lo;
    }
    ca.plan = t.plan;
    t.use_colstrip = jinc::colstrip_configure(ca);
    t.col_strips = ca;

    // The strip kernels take ONE coefficient set per (border row, column phase) / (border column, row phase).  That
    // holds when the border pixels' coefficients repeat with the interior's period (integer ratios, exact down-scales)
    // -- but a plan can have a periodic interior and still private border sets: for 3/2 on a small frame the
    // interior classes have not drifted yet, while the reference computes every border pixel's coefficients from its
    // float-accumulated position, so no two are equal (found by the widened random sweep).  Check, do not assume.
    bool uniform = true;
    for (int y = 0; y < H && uniform; ++y) {
        if (y >= p.iy0 && y < y_end) continue;
        for (int r = 0; r < p.px && uniform; ++r) {
            const int s0 = p.set_of(p.ix0 + r, y);
            for (int i = 1; i < da.ni; ++i)
                if (p.set_of(p.ix0 + p.px * i + r, y) != s0) {
                    uniform = false;
                    break;
                }
        }
    }
    for (int x = 0; x < W && uniform; ++x) {
        if (x >= p.ix0 && x < x_end) continue;
        for (int q = 0; q < p.py && uniform; ++q) {
            const int s0 = p.set_of(x, p.iy0 + q);
            for (int j = 1; j < da.nj; ++j)
                if (p.set_of(x, p.iy0 + p.py * j + q) != s0) {
                    uniform = false;
                    break;
                }
        }
    }
    t.strips_ok = uniform;

    for (int q = 0; q < p.py; ++q)
        for (int r = 0; r < p.px; ++r)
            da.set[q * p.px + r] = p.interior_set[static_cast<size_t>(p.row_class[p.iy0 + q]) * p.n_col_classes +
                                                  p.col_class[p.ix0 + r]];
    t.direct = da;
    t.use_direct = true;
    if (!t.use_periodic && !t.use_quasi) t.border_rects = border_frame(p, x_end, y_end);  // fallback border (gather)
    if (!t.strips_ok) t.border_rects.private_sets = t.border_rects.unit_stride = true;  // coefficients per lane
}

// kernel_direct.hip passes the row offset of its segment fetches as the buffer instructions' scalar offset and relies
// on the hardware range check covering it (measured on gfx950; LLVM's intrinsic documentation says otherwise).  Checked
// once per device on the device itself; 1 = covered, 0 = not (the direct kernel is then not used), < 0 = HIP error.
int buffer_range_check_covers_soffset(int device) {
    static std::atomic<int> cache[64];  // 0: unknown, 1: not covered, 2: covered
    if (device < 0 || device >= 64) return 0;
    const int c = cache[device].load();
    if (c != 0) return c - 1;
    constexpr uint32_t N = 4096;
    std::vector<uint32_t> h(2 * N / 4), r(128, 0xFFFFFFFFu);
    for (uint32_t i = 0; i < h.size(); ++i) h[i] = i;
    uint32_t *d = nullptr, *o = nullptr;
    if (hipSetDevice(device) != hipSuccess || hipMalloc(&d, 2 * N) != hipSuccess) return -1;
    if (hipMalloc(&o, 128 * 4) != hipSuccess) {
        (void)hipFree(d);
        return -1;
    }
    bool ok = hipMemcpy(d, h.data(), 2 * N, hipMemcpyHostToDevice) == hipSuccess &&
              jinc::launch_soffset_probe(d, N, o, nullptr) == 0 && hipMemcpy(r.data(), o, 128 * 4, hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(d);
    (void)hipFree(o);
    if (!ok) return -1;
    bool covered = true;
    for (uint32_t l = 0; l < 64; ++l) {
        covered = covered && r[l] == (l < 32 ? (N - 128) / 4 + l : 0u);  // in range up to the descriptor's end, zero past it
        covered = covered && r[64 + l] == 0u;                            // scalar offset alone past the end
    }
    cache[device].store(covered ? 2 : 1);
    return covered ? 1 : 0;
}

// kernel_direct.hip fetches whole segments as naturally aligned dwords through a buffer resource that ends with the
// aligned dword holding the plane's last sample, so it cannot touch memory outside the plane's own dwords.  It needs
// 4-byte multiples for pitch and frame stride (the plane base may be anywhere) and 32-bit offsets.
bool direct_fetch_is_safe(size_t frame_stride, int nframes, uint64_t plane_bytes, int pitch, int fs) {
    if (plane_bytes + static_cast<uint64_t>(pitch) * (fs + 16) + 64 >= (1ull << 32)) return false;
    if (pitch % 4 != 0) return false;
    return nframes <= 1 || frame_stride % 4 == 0;
}
// Readable bytes from the aligned-down plane base: up to the end of the aligned dword that holds the last sample.
uint32_t direct_src_bytes(const void* base, uint64_t plane_bytes) {
    const uint64_t mis = reinterpret_cast<uintptr_t>(base) & 3u;
    return static_cast<uint32_t>((mis + plane_bytes + 3) & ~3ull);
}

// Rectangles whose pixels own private coefficient sets (border frame of drifting plans, corners of periodic plans):
// a lane-major copy of exactly those coefficients, in the gather kernel's item order, turns its per-lane coefficient
// fetches (64 cache lines per wave and fetch) into contiguous ones.  See RectList::lane_coeffs.
void attach_lane_coeffs(const jinc::PlanePlan& p, DeviceTable& t, jinc::RectList& rects, hipStream_t stream) {
    if (!rects.private_sets || rects.n <= 0) return;
    const int fs = p.fs, fsp = (p.fs + 3) & ~3;
    const size_t item_floats = static_cast<size_t>(fs) * fsp * 64;
    long long total = 0;
    for (int r = 0; r < rects.n; ++r) {
        int axis, P;
        jinc::gather_rect_layout(t.plan, rects.w[r], rects.h[r], rects.unit_stride, axis, P);
        rects.lane_item_base[r] = total;
        total += jinc::gather_item_count(rects.w[r], rects.h[r], axis, P);
    }
    if (total <= 0 || static_cast<unsigned long long>(total) * item_floats * sizeof(float) > (512ull << 20)) return;
    std::vector<float> buf(static_cast<size_t>(total) * item_floats, 0.f);
    for (int r = 0; r < rects.n; ++r) {
        int axis, P;
        jinc::gather_rect_layout(t.plan, rects.w[r], rects.h[r], rects.unit_stride, axis, P);
        const int along = axis == 0 ? rects.w[r] : rects.h[r], across = axis == 0 ? rects.h[r] : rects.w[r];
        const int blocks = (along + 64 * P - 1) / (64 * P);
        for (int line = 0; line < across; ++line)
            for (int res = 0; res < P; ++res)
                for (int ba = 0; ba < blocks; ++ba) {
                    float* item = buf.data() + static_cast<size_t>(rects.lane_item_base[r] + static_cast<long long>(line * P + res) * blocks + ba) * item_floats;
                    for (int l = 0; l < 64; ++l) {
                        const int coord = ba * 64 * P + P * l + res;
                        if (coord >= along) break;
                        const int x = rects.x0[r] + (axis == 0 ? coord : line), y = rects.y0[r] + (axis == 0 ? line : coord);
                        const float* src = p.set_ptr(p.set_of(x, y));
                        for (int ly = 0; ly < fs; ++ly)
                            for (int lx = 0; lx < fs; ++lx)
                                item[((static_cast<size_t>(ly) * (fsp / 4) + lx / 4) * 64 + l) * 4 + lx % 4] = src[ly * fs + lx];
                    }
                }
    }
    void* dev = nullptr;
    hip_check(hipMalloc(&dev, buf.size() * sizeof(float)), "hipMalloc(lane-major coefficients)");
    t.lane_blobs.push_back(dev);
    hip_check(hipMemcpyAsync(dev, buf.data(), buf.size() * sizeof(float), hipMemcpyHostToDevice, stream), "lane-major coefficient upload");
    hip_check(hipStreamSynchronize(stream), "lane-major coefficient upload sync");
    rects.lane_coeffs = static_cast<const float*>(dev);
}

void init_device(jinc_filter& f, int device) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) throw HipError("JincResize: no HIP device available.");
    if (device >= count) throw HipError("JincResize: HIP device index out of range.");
    hip_check(hipSetDevice(device), "hipSetDevice");
    f.device = device;
    hip_check(hipStreamCreateWithFlags(&f.stream, hipStreamNonBlocking), "hipStreamCreate");
    {   // The side stream carries the small border kernels: at the highest priority its workgroups are dispatched as
        // soon as slots free up instead of queueing behind the interior kernel, which can hold every wave slot.
        int least = 0, greatest = 0;
        hip_check(hipDeviceGetStreamPriorityRange(&least, &greatest), "hipDeviceGetStreamPriorityRange");
        hip_check(hipStreamCreateWithPriority(&f.aux_stream, hipStreamNonBlocking, greatest), "hipStreamCreateWithPriority");
    }
    hip_check(hipEventCreateWithFlags(&f.ev_fork, hipEventDisableTiming), "hipEventCreate");
    hip_check(hipEventCreateWithFlags(&f.ev_join, hipEventDisableTiming), "hipEventCreate");
    f.direct_premise = buffer_range_check_covers_soffset(device) == 1;
    f.tables.resize(f.plans.size());
    for (size_t i = 0; i < f.plans.size(); ++i) {
        upload_table(f.plans[i], f.tables[i], f.stream);
        plan_launches(f.plans[i], f.tables[i]);
        plan_quasi(f.plans[i], f.tables[i]);
        plan_direct(f.plans[i], f.tables[i]);
        attach_lane_coeffs(f.plans[i], f.tables[i], f.tables[i].border_rects, f.stream);
        attach_lane_coeffs(f.plans[i], f.tables[i], f.tables[i].corner_rects, f.stream);
        f.tables[i].use_framelane =
            jinc::framelane_configure(f.plans[i], f.tables[i].whole, f.vi_in.component_size, 64, f.tables[i].fl_whole);
        f.tables[i].fl_whole.plan = f.tables[i].plan;
    }
}

void ensure_slot(jinc_filter& f, DeviceFrameBuf& s, bool own_stream) {
    if (s.ready) return;
    const int sb = f.vi_in.component_size;
    try {
        for (int i = 0; i < f.planecount; ++i) {
            int sw, sh, dw, dh;
            f.plane_dims(f.vi_in, i, sw, sh);
            f.plane_dims(f.vi_out, i, dw, dh);
            s.src_pitch[i] = static_cast<int>(align_up(static_cast<size_t>(sw) * sb, 256));
            s.dst_pitch[i] = static_cast<int>(align_up(static_cast<size_t>(dw) * sb, 256));
            hip_check(hipMalloc(&s.src[i], static_cast<size_t>(s.src_pitch[i]) * sh), "hipMalloc(src plane)");
            hip_check(hipMalloc(&s.dst[i], static_cast<size_t>(s.dst_pitch[i]) * dh), "hipMalloc(dst plane)");
        }
        if (own_stream)
            hip_check(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking), "hipStreamCreate");
        else
            s.stream = f.stream;
    } catch (...) {  // a later allocation failed: give back what this call allocated (the next call starts over)
        for (int i = 0; i < 4; ++i) {
            if (s.src[i]) (void)hipFree(s.src[i]);
            if (s.dst[i]) (void)hipFree(s.dst[i]);
            s.src[i] = s.dst[i] = nullptr;
        }
        throw;
    }
    s.ready = true;
}

// Pins [p, p + bytes) once (cache keyed by address range, least recently used out) so that the async copies of the
// pipeline really are asynchronous.  The cache holds at least every range the frames in flight can reference
// (depth x planes x (src + dst)), and a range whose frame may still be in flight is never unregistered under its copy:
// the owning slot's stream is drained first.  Failure to register is not an error: the copy takes the pageable path.
void pin_host_range(jinc_filter& f, const void* p, size_t bytes, long long ticket) {
    char* c = const_cast<char*>(static_cast<const char*>(p));
    for (auto& r : f.pinned)
        if (c >= r.base && c + bytes <= r.base + r.bytes) {
            r.stamp = ++f.pin_clock;
            r.ticket = ticket;
            return;
        }
    const size_t capacity = std::max<size_t>(64, f.slots.size() * 8 + 8);
    if (f.pinned.size() >= capacity) {
        size_t lru = 0;
        for (size_t i = 1; i < f.pinned.size(); ++i)
            if (f.pinned[i].stamp < f.pinned[lru].stamp) lru = i;
        for (auto& s : f.slots)
            if (s.busy && s.ticket == f.pinned[lru].ticket) (void)hipStreamSynchronize(s.stream);  // its copies may still run
        (void)hipHostUnregister(f.pinned[lru].base);
        f.pinned.erase(f.pinned.begin() + lru);
    }
    if (hipHostRegister(c, bytes, hipHostRegisterDefault) == hipSuccess) {
        f.pinned.push_back({c, bytes, ++f.pin_clock, ticket});
    } else {
        (void)hipGetLastError();  // clear; e.g. the range overlaps memory somebody else has registered
    }
}

// Enqueues H2D -> kernels -> D2H of one frame on the slot's stream.
void submit_frame(jinc_filter& f, DeviceFrameBuf& s, const void* const src[4], const int src_pitch[4], void* const dst[4],
                  const int dst_pitch[4]);

void enqueue(jinc_filter& f, const void* const src[4], const int src_pitch[4], const size_t src_fs[4],
             void* const dst[4], const int dst_pitch[4], const size_t dst_fs[4], int nframes, hipStream_t stream) {
    const int sb = f.vi_in.component_size;
    // kernel_mode: 0 automatic, 1 gather only, 2.. A/B variants of the periodic kernels, 7 quasi-periodic
    // kernel wherever it applies (also for exactly periodic plans)
    auto wants_quasi = [&](const DeviceTable& t) {
        return t.use_quasi && (f.kernel_mode == 7 || f.kernel_mode == 8 || f.kernel_mode == 10 || (f.kernel_mode != 1 && !t.use_periodic));
    };
    auto wants_periodic = [&](const DeviceTable& t) {
        return t.use_periodic && f.kernel_mode != 1 && f.kernel_mode != 7 && f.kernel_mode != 8 && f.kernel_mode != 10;
    };
    // Is kernel_direct.hip usable for plane i (interior and border strips)?  See direct_fetch_is_safe().
    auto direct_ok = [&](const DeviceTable& t, int i) {
        if (!t.use_direct || f.kernel_mode == 1 || !f.direct_premise) return false;
        const uint64_t plane_bytes = static_cast<uint64_t>(src_pitch[i]) * (t.plan.src_h - 1) + static_cast<uint64_t>(t.plan.src_w) * sb;
        return direct_fetch_is_safe(src_fs ? src_fs[i] : 0, nframes, plane_bytes, src_pitch[i], t.plan.fs);
    };
    // kernel_mode 9: the direct kernel wherever it applies; otherwise it takes the interior of the exactly periodic
    // plans the register/LDS kernels do not cover.
    auto wants_direct = [&](const DeviceTable& t, int i) {
        if (!direct_ok(t, i)) return false;
        return f.kernel_mode == 9 || (!wants_periodic(t) && !wants_quasi(t));
    };
    // Frame-lane kernel (lanes = frames): the choice for batches whose plan has no phase structure for the other
    // interior kernels (they would run on the gather kernel with per-lane coefficient traffic); kernel_mode 11 forces
    // it for every plan and batch size.
    auto wants_framelane = [&](const DeviceTable& t, int i) {
        if (!t.use_framelane || f.kernel_mode == 1) return false;
        if (f.kernel_mode == 11) return true;
        if (f.kernel_mode != 0) return false;
        if (nframes < kFrameLaneMinFrames) return false;
        if (wants_periodic(t)) return false;
        // Drifting plans with many phases (DVD -> 1080p: 8 x 9) leave the quasi-periodic kernel little to share per phase:
        // measured at 64 frames 33 % of the VALU peak against 46 % here; with few phases (1.5x, 3x: 9) it stays ahead.
        if (wants_quasi(t)) return !f.plans[f.table_of_plane(i)].periodic && t.quasi.px * t.quasi.py > 16 && nframes >= 48;
        return !wants_direct(t, i);
    };
    bool any_periodic = false;
    for (int i = 0; i < f.planecount; ++i) {
        const DeviceTable& t = f.tables[f.table_of_plane(i)];
        any_periodic |= f.simd_order == 0 && !wants_framelane(t, i) && (wants_periodic(t) || wants_quasi(t) || wants_direct(t, i));
    }
    // A/B on MI355X with the strip border kernels: overlapping wins 11 % on C3 (fs 17), 3 % on C4 (fs 9) and 2 % on
    // C2 (fs 7) -- three small border launches per plane would otherwise sit serially in front of the interior.
    const bool want_overlap = f.overlap_border != 0;
    const bool fork = any_periodic && want_overlap;
    if (fork) {  // border work may start once everything already queued on `stream` is done
        hip_check(hipEventRecord(f.ev_fork, stream), "hipEventRecord(fork)");
        hip_check(hipStreamWaitEvent(f.aux_stream, f.ev_fork, 0), "hipStreamWaitEvent(fork)");
    }
    hipStream_t border_stream = fork ? f.aux_stream : stream;
    for (int i = 0; i < f.planecount; ++i) {
        DeviceTable& t = f.tables[f.table_of_plane(i)];
        if (!src[i] || !dst[i]) throw ArgError("JincResize: null plane pointer.");
        if (src_pitch[i] % sb || dst_pitch[i] % sb) throw ArgError("JincResize: plane pitch is not a multiple of the sample size.");
        if (reinterpret_cast<uintptr_t>(src[i]) % sb || reinterpret_cast<uintptr_t>(dst[i]) % sb)
            throw ArgError("JincResize: plane pointer is not aligned to the sample size.");
        if (src_fs && nframes > 1 && src_fs[i] % sb) throw ArgError("JincResize: frame stride is not a multiple of the sample size.");
        if (dst_fs && nframes > 1 && dst_fs[i] % sb) throw ArgError("JincResize: frame stride is not a multiple of the sample size.");
        if (static_cast<size_t>(src_pitch[i]) < static_cast<size_t>(t.plan.src_w) * sb ||
            static_cast<size_t>(dst_pitch[i]) < static_cast<size_t>(t.plan.dst_w) * sb)
            throw ArgError("JincResize: plane pitch is smaller than the row size.");
        if (static_cast<uint64_t>(dst_pitch[i]) * t.plan.dst_h >= (1ull << 32))
            throw ArgError("JincResize: destination plane larger than 4 GiB is not supported (32-bit store offsets).");
        jinc::PlaneIO io;
        io.src = src[i];
        io.dst = dst[i];
        io.src_pitch = src_pitch[i];
        io.dst_pitch = dst_pitch[i];
        io.src_frame_stride = src_fs ? src_fs[i] : 0;
        io.dst_frame_stride = dst_fs ? dst_fs[i] : 0;
        io.nframes = nframes;
        io.sample_bytes = sb;
        io.peak = f.peak;
        auto timed = [&](std::vector<EventPair>& sink, hipStream_t s, const char* what, auto&& launch) {
            EventPair ev;
            if (f.profiling) {
                hip_check(hipEventCreate(&ev.start), "hipEventCreate");
                hip_check(hipEventCreate(&ev.stop), "hipEventCreate");
                hip_check(hipEventRecord(ev.start, s), "hipEventRecord");
            }
            hip_check(static_cast<hipError_t>(launch(s)), what);
            if (f.profiling) {
                hip_check(hipEventRecord(ev.stop, s), "hipEventRecord");
                sink.push_back(ev);
            }
        };
        if (f.simd_order != 0) {  // compatibility modes (private switch): whole plane on kernel_simdorder.hip
            const float min_val = (i != 0 && !f.vi_in.is_rgb) ? -0.5f : 0.f;  // ref resize_plane_sse41.cpp:20
            t.last_kernel = "ewa_simd_order_kernel";
            timed(f.ev_gather, stream, "SIMD-order kernel launch",
                  [&](hipStream_t s) { return jinc::launch_simd_order(t.plan, io, f.simd_order, min_val, s); });
            continue;
        }
        if (wants_framelane(t, i)) {
            jinc::FrameLaneArgs fa = t.fl_whole;
            fa.io = io;
            const uintptr_t vec = static_cast<uintptr_t>(4 * sb);
            fa.vec_store_ok = reinterpret_cast<uintptr_t>(dst[i]) % vec == 0 && static_cast<uintptr_t>(dst_pitch[i]) % vec == 0 &&
                              (nframes <= 1 || io.dst_frame_stride % vec == 0);
            t.last_kernel = "ewa_framelane_kernel";
            timed(f.ev_periodic, stream, "frame-lane kernel launch", [&](hipStream_t s) { return jinc::launch_framelane(fa, s); });
            continue;
        }
        const bool direct = wants_direct(t, i);
        const bool quasi = !direct && wants_quasi(t);
        const bool periodic = !direct && !quasi && wants_periodic(t);
        t.last_kernel = direct ? "ewa_direct_kernel" : quasi ? "ewa_quasi_kernel" : periodic ? "ewa_periodic_kernel" : "ewa_gather_kernel";
        if (direct || periodic || quasi) {
            // border frame: rows on kernel_direct.hip + columns on the gather kernel, or the gather kernel for all of it
            const bool strips = f.border_strips != 0 && t.strips_ok && direct_ok(t, i);
            if (strips) {
                jinc::DirectArgs rs = t.row_strips;
                rs.src_bytes = direct_src_bytes(
                    src[i], static_cast<uint64_t>(src_pitch[i]) * (t.plan.src_h - 1) + static_cast<uint64_t>(t.plan.src_w) * sb);
                const bool colstrip = t.use_colstrip && f.border_strips != 2;
                // the corner kernel first: few workgroups with long latency-bound chains (per-lane coefficients); queued
                // last it would start when the interior kernel already holds every wave slot
                if (colstrip && t.corner_rects.n > 0)
                    timed(f.ev_gather, border_stream, "corner kernel launch",
                          [&](hipStream_t s) { return jinc::launch_gather(t.plan, io, t.corner_rects, s); });
                timed(f.ev_gather, border_stream, "border row kernel launch",
                      [&](hipStream_t s) { return jinc::launch_direct_row_strips(rs, io, s); });
                if (colstrip) {
                    timed(f.ev_gather, border_stream, "border column kernel launch",
                          [&](hipStream_t s) { return jinc::launch_colstrip(t.col_strips, io, s); });
                } else if (t.column_rects.n > 0) {
                    timed(f.ev_gather, border_stream, "border column kernel launch",
                          [&](hipStream_t s) { return jinc::launch_gather(t.plan, io, t.column_rects, s); });
                }
            } else if (t.border_rects.n > 0) {
                timed(f.ev_gather, border_stream, "border kernel launch",
                      [&](hipStream_t s) { return jinc::launch_gather(t.plan, io, t.border_rects, s); });
            }
            if (direct)
                timed(f.ev_periodic, stream, "direct periodic kernel launch", [&](hipStream_t s) {
                    jinc::DirectArgs da = t.direct;
                    da.src_bytes = direct_src_bytes(
                        src[i], static_cast<uint64_t>(src_pitch[i]) * (t.plan.src_h - 1) + static_cast<uint64_t>(t.plan.src_w) * sb);
                    return jinc::launch_direct(da, io, s);
                });
            else if (quasi)
                timed(f.ev_periodic, stream, "quasi-periodic kernel launch", [&](hipStream_t s) {
                    jinc::QuasiArgs qa = t.quasi;
                    if (f.kernel_mode == 8) qa.exact = 0;   // A/B: per-row lookup + waterfall over sets in SGPRs
                    if (f.kernel_mode == 10) qa.exact = 2;  // A/B: per-row lookup + per-lane coefficient registers
                    return jinc::launch_quasi(qa, t.plan.fs, io, s);
                });
            else
                timed(f.ev_periodic, stream, "periodic kernel launch", [&](hipStream_t s) {
                    return jinc::launch_periodic(t.periodic, t.plan.fs, io, s, f.kernel_mode >= 3 ? f.kernel_mode - 2 : 0);
                });
        } else {
            timed(f.ev_gather, stream, "gather kernel launch",
                  [&](hipStream_t s) { return jinc::launch_gather(t.plan, io, t.whole, s); });
        }
    }
    if (fork) {  // `stream` continues only after the border kernels have finished too
        hip_check(hipEventRecord(f.ev_join, f.aux_stream), "hipEventRecord(join)");
        hip_check(hipStreamWaitEvent(stream, f.ev_join, 0), "hipStreamWaitEvent(join)");
    }
}

void submit_frame(jinc_filter& f, DeviceFrameBuf& s, const void* const src[4], const int src_pitch[4], void* const dst[4],
                  const int dst_pitch[4]) {
    const int sb = f.vi_in.component_size;
    for (int i = 0; i < f.planecount; ++i) {
        if (!src[i] || !dst[i]) throw ArgError("JincResize: null plane pointer.");
        int sw, sh, dw, dh;
        f.plane_dims(f.vi_in, i, sw, sh);
        f.plane_dims(f.vi_out, i, dw, dh);
        if (f.register_host) {
            pin_host_range(f, src[i], static_cast<size_t>(src_pitch[i]) * (sh - 1) + static_cast<size_t>(sw) * sb, f.next_ticket);
            pin_host_range(f, dst[i], static_cast<size_t>(dst_pitch[i]) * (dh - 1) + static_cast<size_t>(dw) * sb, f.next_ticket);
        }
        hip_check(hipMemcpy2DAsync(s.src[i], s.src_pitch[i], src[i], src_pitch[i], static_cast<size_t>(sw) * sb, sh,
                                   hipMemcpyHostToDevice, s.stream),
                  "H2D copy");
    }
    enqueue(f, s.src, s.src_pitch, nullptr, s.dst, s.dst_pitch, nullptr, 1, s.stream);
    for (int i = 0; i < f.planecount; ++i) {
        int dw, dh;
        f.plane_dims(f.vi_out, i, dw, dh);
        hip_check(hipMemcpy2DAsync(dst[i], dst_pitch[i], s.dst[i], s.dst_pitch[i], static_cast<size_t>(dw) * sb, dh,
                                   hipMemcpyDeviceToHost, s.stream),
                  "D2H copy");
    }
}

template <typename Fn>
int guarded(Fn&& fn) {
    try {
        fn();
        g_last_error.clear();
        return JINC_OK;
    } catch (const ArgError& e) {
        return fail(JINC_ERR_INVALID_ARG, e.what());
    } catch (const HipError& e) {
        return fail(JINC_ERR_HIP, e.what());
    } catch (const std::bad_alloc&) {
        return fail(JINC_ERR_NOMEM, "JincResize: out of memory.");
    } catch (const std::exception& e) {
        return fail(JINC_ERR_UNSUPPORTED, e.what());
    }
}

const jinc::PlanePlan* table_or_null(const jinc_filter* f, int table) {
    if (!f || table < 0 || table >= static_cast<int>(f->plans.size())) return nullptr;
    return &f->plans[table];
}

}  // namespace

extern "C" {

int jinc_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return 0;
    return count;
}

int jinc_pick_device(void) {
    static std::atomic<unsigned> counter{0};
    const int n = jinc_device_count();
    if (n <= 0) return -1;
    return static_cast<int>(counter.fetch_add(1) % static_cast<unsigned>(n));
}

const char* jinc_last_error(void) { return g_last_error.c_str(); }

int jinc_filter_create(const jinc_video_info* vi, const jinc_args* args, int device, jinc_filter** out, char* err,
                       size_t err_len) {
    if (out) *out = nullptr;
    if (err && err_len) err[0] = '\0';
    if (!vi || !args || !out) return fail(JINC_ERR_INVALID_ARG, "JincResize: null argument.");
    std::unique_ptr<jinc_filter> f(new (std::nothrow) jinc_filter());
    if (!f) return fail(JINC_ERR_NOMEM, "JincResize: out of memory.");
    int rc = guarded([&] {
        configure(*f, *vi, *args);
        if (device >= 0) init_device(*f, device);
    });
    if (rc == JINC_ERR_HIP && device >= 0 && jinc_device_count() == 0) rc = JINC_ERR_NO_DEVICE;
    if (rc != JINC_OK) {
        if (err && err_len) {
            std::strncpy(err, g_last_error.c_str(), err_len - 1);
            err[err_len - 1] = '\0';
        }
        return rc;
    }
    *out = f.release();
    return JINC_OK;
}

void jinc_filter_free(jinc_filter* f) { delete f; }

int jinc_filter_output_info(const jinc_filter* f, jinc_video_info* out_vi) {
    if (!f || !out_vi) return fail(JINC_ERR_INVALID_ARG, "JincResize: null argument.");
    *out_vi = f->vi_out;
    return JINC_OK;
}

int jinc_filter_chroma_location(const jinc_filter* f) { return f ? f->chroma_location : -1; }

int jinc_filter_get_frame(jinc_filter* f, const void* const src[4], const int src_pitch[4], void* const dst[4],
                          const int dst_pitch[4]) {
    if (!f || !src || !dst || !src_pitch || !dst_pitch) return fail(JINC_ERR_INVALID_ARG, "JincResize: null argument.");
    if (f->device < 0) return fail(JINC_ERR_NO_DEVICE, "JincResize: filter was created without a HIP device (device < 0).");
    return guarded([&] {
        hip_check(hipSetDevice(f->device), "hipSetDevice");
        for (auto& s : f->slots)  // a synchronous frame must not overtake frames still in the pipeline
            if (s.busy) {
                hip_check(hipStreamSynchronize(s.stream), "stream sync");
                s.busy = false;
            }
        DeviceFrameBuf& s = f->slots[0];
        ensure_slot(*f, s, false);
        submit_frame(*f, s, src, src_pitch, dst, dst_pitch);
        hip_check(hipStreamSynchronize(s.stream), "stream sync");
    });
}

int jinc_filter_set_pipeline(jinc_filter* f, int depth, int register_host_buffers) {
    if (!f || depth < 1 || depth > 16) return fail(JINC_ERR_INVALID_ARG, "JincResize: pipeline depth must be 1..16.");
    if (f->device < 0) return fail(JINC_ERR_NO_DEVICE, "JincResize: filter was created without a HIP device (device < 0).");
    return guarded([&] {
        hip_check(hipSetDevice(f->device), "hipSetDevice");
        for (auto& s : f->slots)
            if (s.busy) {
                hip_check(hipStreamSynchronize(s.stream), "stream sync");
                s.busy = false;
            }
        if (static_cast<size_t>(depth) > f->slots.size()) f->slots.resize(depth);
        f->register_host = register_host_buffers != 0;
        if (!f->register_host) {
            for (auto& p : f->pinned) (void)hipHostUnregister(p.base);
            f->pinned.clear();
        }
    });
}

int jinc_filter_submit(jinc_filter* f, const void* const src[4], const int src_pitch[4], void* const dst[4],
                       const int dst_pitch[4], long long* ticket) {
    if (!f || !src || !dst || !src_pitch || !dst_pitch || !ticket) return fail(JINC_ERR_INVALID_ARG, "JincResize: null argument.");
    if (f->device < 0) return fail(JINC_ERR_NO_DEVICE, "JincResize: filter was created without a HIP device (device < 0).");
    return guarded([&] {
        hip_check(hipSetDevice(f->device), "hipSetDevice");
        const long long t = f->next_ticket;
        const size_t k = static_cast<size_t>(t % static_cast<long long>(f->slots.size()));
        DeviceFrameBuf& s = f->slots[k];
        ensure_slot(*f, s, k != 0);
        if (s.busy) {  // the slot's previous frame has to be finished before its buffers are reused
            hip_check(hipStreamSynchronize(s.stream), "stream sync");
            s.busy = false;
        }
        submit_frame(*f, s, src, src_pitch, dst, dst_pitch);
        s.busy = true;
        s.ticket = t;
        *ticket = t;
        ++f->next_ticket;
    });
}

int jinc_filter_wait(jinc_filter* f, long long ticket) {
    if (!f) return fail(JINC_ERR_INVALID_ARG, "JincResize: null argument.");
    if (f->device < 0) return fail(JINC_ERR_NO_DEVICE, "JincResize: filter was created without a HIP device (device < 0).");
    return guarded([&] {
        hip_check(hipSetDevice(f->device), "hipSetDevice");
        for (auto& s : f->slots)
            if (s.busy && s.ticket == ticket) {
                hip_check(hipStreamSynchronize(s.stream), "stream sync");
                s.busy = false;
                return;
            }
        // unknown or already completed ticket: nothing to wait for (its slot has been reused or waited on)
    });
}

int jinc_filter_process_device(jinc_filter* f, const void* const src[4], const int src_pitch[4],
                               const size_t src_frame_stride[4], void* const dst[4], const int dst_pitch[4],
                               const size_t dst_frame_stride[4], int nframes, void* hip_stream) {
    if (!f || !src || !dst || !src_pitch || !dst_pitch) return fail(JINC_ERR_INVALID_ARG, "JincResize: null argument.");
    if (f->device < 0) return fail(JINC_ERR_NO_DEVICE, "JincResize: filter was created without a HIP device (device < 0).");
    if (nframes < 1 || nframes > 65535) return fail(JINC_ERR_INVALID_ARG, "JincResize: nframes must be in 1..65535.");
    if (nframes > 1 && (!src_frame_stride || !dst_frame_stride))
        return fail(JINC_ERR_INVALID_ARG, "JincResize: frame strides are required for nframes > 1.");
    return guarded([&] {
        hip_check(hipSetDevice(f->device), "hipSetDevice");
        hipStream_t s = static_cast<hipStream_t>(hip_stream);  // NULL = the HIP null stream, ordered with the caller's default-stream work
        enqueue(*f, src, src_pitch, src_frame_stride, dst, dst_pitch, dst_frame_stride, nframes, s);
    });
}

int jinc_filter_sync(jinc_filter* f) {
    if (!f) return fail(JINC_ERR_INVALID_ARG, "JincResize: null argument.");
    if (f->device < 0) return fail(JINC_ERR_NO_DEVICE, "JincResize: filter was created without a HIP device (device < 0).");
    return guarded([&] {
        hip_check(hipSetDevice(f->device), "hipSetDevice");
        hip_check(hipStreamSynchronize(f->stream), "stream sync");
    });
}

int jinc_alias_args(int taps, const jinc_args* in, jinc_args* out) {
    if (!in || !out) return fail(JINC_ERR_INVALID_ARG, "JincResize: null argument.");
    if (taps != 3 && taps != 4 && taps != 6 && taps != 8)
        return fail(JINC_ERR_INVALID_ARG, "JincResize: alias tap count must be 3, 4, 6 or 8.");
    jinc_args a{};
    a.target_width = in->target_width;
    a.target_height = in->target_height;
    const unsigned forwarded = JINC_ARG_SRC_LEFT | JINC_ARG_SRC_TOP | JINC_ARG_SRC_WIDTH | JINC_ARG_SRC_HEIGHT |
                               JINC_ARG_QUANT_X | JINC_ARG_QUANT_Y | JINC_ARG_CPLACE | JINC_ARG_THREADS;
    a.defined = (in->defined & forwarded) | JINC_ARG_TAP;
    a.src_left = in->src_left;
    a.src_top = in->src_top;
    a.src_width = in->src_width;
    a.src_height = in->src_height;
    a.quant_x = in->quant_x;
    a.quant_y = in->quant_y;
    a.cplace = in->cplace;
    a.threads = in->threads;
    a.tap = taps;
    a.frame0_chroma_location = in->frame0_chroma_location;
    a.cpu_has_sse41 = in->cpu_has_sse41;
    a.cpu_has_avx2 = in->cpu_has_avx2;
    a.cpu_has_avx512f = in->cpu_has_avx512f;
    *out = a;
    return JINC_OK;
}

int jinc_filter_num_tables(const jinc_filter* f) { return f ? static_cast<int>(f->plans.size()) : 0; }

int jinc_filter_plan_info(const jinc_filter* f, int table, jinc_plan_info* out) {
    const jinc::PlanePlan* p = table_or_null(f, table);
    if (!p || !out) return fail(JINC_ERR_INVALID_ARG, "JincResize: bad table index.");
    jinc_plan_info i{};
    i.src_width = p->g.src_w;
    i.src_height = p->g.src_h;
    i.dst_width = p->g.dst_w;
    i.dst_height = p->g.dst_h;
    i.filter_size = p->fs;
    i.num_sets = p->num_sets;
    i.periodic = p->periodic ? 1 : 0;
    i.period_x = p->px;
    i.period_y = p->py;
    i.step_x = p->sx;
    i.step_y = p->sy;
    i.interior_x0 = p->ix0;
    i.interior_x1 = p->ix1;
    i.interior_y0 = p->iy0;
    i.interior_y1 = p->iy1;
    i.plan_bytes = static_cast<int64_t>(4) * (p->col_start.size() + p->row_start.size() + p->col_class.size() +
                                              p->row_class.size() + p->interior_set.size() + p->bcol_set.size() +
                                              p->brow_set.size() + p->coeffs.size());
    i.quasi = p->quasi ? 1 : 0;
    i.quasi_period_x = p->qpx;
    i.quasi_period_y = p->qpy;
    i.quasi_step_x = p->qsx;
    i.quasi_step_y = p->qsy;
    *out = i;
    return JINC_OK;
}

int jinc_filter_plan_pixel(const jinc_filter* f, int table, int x, int y, int* start_x, int* start_y, float* coeffs) {
    const jinc::PlanePlan* p = table_or_null(f, table);
    if (!p || x < 0 || y < 0 || x >= p->g.dst_w || y >= p->g.dst_h)
        return fail(JINC_ERR_INVALID_ARG, "JincResize: bad table index or pixel.");
    if (start_x) *start_x = p->col_start[x];
    if (start_y) *start_y = p->row_start[y];
    if (coeffs) std::memcpy(coeffs, p->set_ptr(p->set_of(x, y)), sizeof(float) * p->fs * p->fs);
    return JINC_OK;
}

int jinc_filter_plan_dump(const jinc_filter* f, int table, int* start_x, int* start_y, int* set_id) {
    const jinc::PlanePlan* p = table_or_null(f, table);
    if (!p) return fail(JINC_ERR_INVALID_ARG, "JincResize: bad table index.");
    if (start_x) std::memcpy(start_x, p->col_start.data(), sizeof(int) * p->g.dst_w);
    if (start_y) std::memcpy(start_y, p->row_start.data(), sizeof(int) * p->g.dst_h);
    if (set_id)
        for (int y = 0; y < p->g.dst_h; ++y)
            for (int x = 0; x < p->g.dst_w; ++x) set_id[static_cast<size_t>(y) * p->g.dst_w + x] = p->set_of(x, y);
    return JINC_OK;
}

int jinc_filter_plan_set(const jinc_filter* f, int table, int set, float* coeffs) {
    const jinc::PlanePlan* p = table_or_null(f, table);
    if (!p || !coeffs || set < 0 || set >= p->num_sets) return fail(JINC_ERR_INVALID_ARG, "JincResize: bad table or set index.");
    std::memcpy(coeffs, p->set_ptr(set), sizeof(float) * p->fs * p->fs);
    return JINC_OK;
}

int jinc_filter_lut(const jinc_filter* f, double* lut1024) {
    if (!f || !lut1024) return fail(JINC_ERR_INVALID_ARG, "JincResize: null argument.");
    std::memcpy(lut1024, f->lut.v.data(), sizeof(double) * jinc::kLutSamples);
    return JINC_OK;
}

int jinc_filter_set_profiling(jinc_filter* f, int enable) {
    if (!f) return fail(JINC_ERR_INVALID_ARG, "JincResize: null argument.");
    f->profiling = enable != 0;
    return JINC_OK;
}

int jinc_filter_kernel_times(jinc_filter* f, double* periodic_ms, int* periodic_launches, double* gather_ms,
                             int* gather_launches) {
    if (!f) return fail(JINC_ERR_INVALID_ARG, "JincResize: null argument.");
    if (f->device < 0) return fail(JINC_ERR_NO_DEVICE, "JincResize: filter was created without a HIP device (device < 0).");
    return guarded([&] {
        hip_check(hipSetDevice(f->device), "hipSetDevice");
        auto collect = [](std::vector<EventPair>& v, double* ms_out, int* n_out) {
            double total = 0.0;
            for (auto& e : v) {
                hip_check(hipEventSynchronize(e.stop), "hipEventSynchronize");
                float ms = 0.f;
                hip_check(hipEventElapsedTime(&ms, e.start, e.stop), "hipEventElapsedTime");
                total += ms;
                (void)hipEventDestroy(e.start);
                (void)hipEventDestroy(e.stop);
            }
            if (ms_out) *ms_out = total;
            if (n_out) *n_out = static_cast<int>(v.size());
            v.clear();
        };
        collect(f->ev_periodic, periodic_ms, periodic_launches);
        collect(f->ev_gather, gather_ms, gather_launches);
    });
}

int jinc_debug_convert(const float* sums, void* out, int n, int sample_bytes, float peak, int device) {
    if (!sums || !out || n < 0 || (sample_bytes != 1 && sample_bytes != 2 && sample_bytes != 4))
        return fail(JINC_ERR_INVALID_ARG, "JincResize: bad argument.");
    return guarded([&] {
        hip_check(hipSetDevice(device), "hipSetDevice");
        float* d_in = nullptr;
        void* d_out = nullptr;
        hip_check(hipMalloc(&d_in, sizeof(float) * (n + 1)), "hipMalloc");
        hip_check(hipMalloc(&d_out, static_cast<size_t>(sample_bytes) * (n + 1)), "hipMalloc");
        hip_check(hipMemcpy(d_in, sums, sizeof(float) * n, hipMemcpyHostToDevice), "hipMemcpy");
        hip_check(static_cast<hipError_t>(jinc::launch_debug_convert(d_in, d_out, n, sample_bytes, peak, nullptr)), "convert launch");
        hip_check(hipMemcpy(out, d_out, static_cast<size_t>(sample_bytes) * n, hipMemcpyDeviceToHost), "hipMemcpy");
        (void)hipFree(d_in);
        (void)hipFree(d_out);
    });
}

const char* jinc_filter_interior_kernel(const jinc_filter* f, int table) {
    if (!f || f->device < 0 || table < 0 || table >= static_cast<int>(f->tables.size())) return "";
    const DeviceTable& t = f->tables[table];
    const int m = f->kernel_mode;
    const bool quasi = t.use_quasi && (m == 7 || m == 8 || m == 10 || (m != 1 && !t.use_periodic));
    const bool periodic = t.use_periodic && m != 1 && m != 7 && m != 8 && m != 10;
    if (t.use_direct && m != 1 && (m == 9 || (!periodic && !quasi))) return "ewa_direct_kernel";
    if (quasi) return "ewa_quasi_kernel";
    if (periodic) {
        const int fs = t.plan.fs;
        if (m == 5 || m == 6) return fs == 7 ? "ewa_periodic_pk_kernel" : "ewa_periodic_kernel";
        if (m == 3 || (fs != 7 && fs != 9)) return "ewa_periodic_rows_kernel";
        return "ewa_periodic_kernel";
    }
    return "ewa_gather_kernel";
}

int jinc_filter_set_simd_order(jinc_filter* f, int order) {
    if (!f || order < 0 || order > 3) return fail(JINC_ERR_INVALID_ARG, "JincResize: SIMD order must be 0..3.");
    f->simd_order = order;
    return JINC_OK;
}

int jinc_debug_buffer_range_check(int device) {
    const int r = buffer_range_check_covers_soffset(device);
    if (r < 0) return fail(JINC_ERR_HIP, "JincResize: the buffer range-check probe could not run.");
    return r;
}

const char* jinc_filter_last_kernel(const jinc_filter* f, int table) {
    if (!f || f->device < 0 || table < 0 || table >= static_cast<int>(f->tables.size())) return "";
    return f->tables[table].last_kernel;
}

int jinc_filter_set_border_strips(jinc_filter* f, int enable) {
    if (!f) return fail(JINC_ERR_INVALID_ARG, "JincResize: null argument.");
    f->border_strips = enable < 0 ? 1 : enable > 2 ? 1 : enable;  // 2: rows as strips, columns on the gather kernel
    g_last_error.clear();
    return JINC_OK;
}

int jinc_filter_set_kernel_mode(jinc_filter* f, int mode) {
    if (!f || mode < 0 || mode > 11) return fail(JINC_ERR_INVALID_ARG, "JincResize: bad kernel mode.");
    f->kernel_mode = mode;
    return JINC_OK;
}

int jinc_filter_set_border_overlap(jinc_filter* f, int enable) {
    if (!f) return fail(JINC_ERR_INVALID_ARG, "JincResize: null argument.");
    f->overlap_border = enable < 0 ? -1 : (enable != 0);
    return JINC_OK;
}

}  // extern "C"
