// filter_args.cpp -- Create_JincResize's argument handling and geometry derivation for the accelerated path
// ("ref:" = /root/reference/src/JincResize.cpp :700-866): same defaults, same checks in the same order, same error strings.
#include "filter_internal.h"
#include <cctype>

namespace jinc {
namespace host {

namespace {
bool is_yuv_subsampled(const jinc_video_info& vi, int sw, int sh) {
    return !vi.is_rgb && vi.num_components >= 3 && vi.sub_w == sw && vi.sub_h == sh;
}

std::string lower(std::string s) {
    for (auto& c : s) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
    return s;
}

}  // namespace

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
        if (a.frame0_chroma_location != -1) {  // the property exists and is an integer (ref :730)
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
    f.copy_helpers = threads != 1;  // the reference: 1 = this thread only, 0 = every core (ref :901); here: the copies of pageable planes

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

    // ref :617-625.  The reference compares `d->cplace`, a member that nothing ever assigns (`new JincResize()` :676
    // leaves it empty; the string the arguments and frame 0 decide is the LOCAL `cplace` of :715, which drives the
    // chroma geometry above and nothing else), so its GetFrame always takes the `else` of :623-624 and writes 2.
    // chroma_location is what the reference binary writes; chroma_location_by_siting is what its source means to
    // write, kept behind jinc_filter_set_chroma_location_mode.
    if (is_420 || is_yuv_subsampled(vi, 1, 0) || is_yuv_subsampled(vi, 2, 0)) {
        f.chroma_location = 2;
        f.chroma_location_by_siting = cplace == "mpeg2" ? 0 : (cplace == "mpeg1" ? 1 : 2);
    } else {
        f.chroma_location = -1;
        f.chroma_location_by_siting = -1;
    }
}

}  // namespace host
}  // namespace jinc
