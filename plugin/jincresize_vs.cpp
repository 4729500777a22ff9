// jincresize_vs.cpp -- VapourSynth (API 4) front-end around libjincresize_hip.so ("a VapourSynth front-end
// -- the upstream lineage named in README.md:5 -- reusing the same HIP library").
//
// Registers jinc.JincResize and jinc.Jinc36Resize / Jinc64Resize / Jinc144Resize / Jinc256Resize with the AviSynth
// plugin's argument names (/root/reference/README.md:17-111, registration /root/reference/src/JincResize.cpp:1044-1108), hands
// them to the same C ABI (include/jincresize_hip.h) the AviSynth shell uses -- same defaults, same checks, same messages
// (jinc_filter_create) -- and implements the filter's getFrame on jinc_filter_get_frame.  All resampling happens behind that
// ABI on the GPU; there is no CPU path in this file.
//
// What differs from the AviSynth shell, by the host's nature:
//   * arguments are named map entries; floats arrive as doubles (no float32 rounding as in AVS_Value);
//   * planar RGB planes are ordered R, G, B (AviSynth: G, B, R) -- all three use the same table, so the order is immaterial;
//   * there is no alpha plane inside a VapourSynth format;
//   * one filter instance serves the whole graph: fmUnordered = one getFrame at a time, in any order (the instance is not
//     re-entrant, like the reference's MT_MULTI_INSTANCE instances).
//
// Build against a VapourSynth installation (its SDK header first on the include path):
//   g++ -std=c++17 -shared -fPIC plugin/jincresize_vs.cpp -Iinclude -I<vapoursynth sdk>/include
//       -Lavisynth-jincresize_amd/lib -ljincresize_hip -o libjincresize_vs.so
// In this repository it is compiled against plugin/compat/VapourSynth4.h (self-written declaration, see its verify-list)
// and tested with the mock host of tests/mock_vs/ (tests/test_plugin_vs_mock_host.py).
#include "VapourSynth4.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "jincresize_hip.h"

namespace {

struct Pending {  // one frame in flight in the look-ahead ring
    const VSFrame* src = nullptr;
    VSFrame* dst = nullptr;
    long long ticket = -1;
    int frame = -1;
};

struct Instance {
    VSNode* node = nullptr;
    VSVideoInfo vi{};            // output
    jinc_filter* filter = nullptr;
    int chroma_location = -1;    // value written to _ChromaLocation, -1: format without sub-sampled chroma
    int planes = 0;
    // look-ahead (JINCRESIZE_LOOKAHEAD > 1): the frames of a window [base, base + lookahead) in flight, as in the AviSynth shell
    int lookahead = 1, group = 0;
    std::vector<Pending> ring;   // slot k % lookahead holds frame k while it is in the window
    int base = 0, next_submit = 0, resubmit = -1;
    std::mutex mutex;
};

void plane_pointers(const VSAPI* vsapi, const Instance* d, const VSFrame* src, VSFrame* dst, const void* sp[4], int spitch[4], void* dp[4],
                    int dpitch[4]) {
    for (int i = 0; i < 4; ++i) {
        sp[i] = nullptr, dp[i] = nullptr, spitch[i] = 0, dpitch[i] = 0;
        if (i >= d->planes) continue;
        sp[i] = vsapi->getReadPtr(src, i);
        spitch[i] = static_cast<int>(vsapi->getStride(src, i));
        dp[i] = vsapi->getWritePtr(dst, i);
        dpitch[i] = static_cast<int>(vsapi->getStride(dst, i));
    }
}

// Look-ahead form: in arInitial frame n asks for the source frames n .. n + lookahead - 1 (VapourSynth's cache serves the
// overlap between neighbouring requests); in arAllFramesReady -- which fmParallelRequests serialises, in whatever order the
// frames become ready -- the window ring of the AviSynth shell (plugin/jincresize_avs.cpp get_frame_lookahead: any frame of
// the window in any order without draining, a frame is dropped only when the window moves past it) submits what the window
// still lacks from THIS call's frame context and waits for frame n only.  Every new submission k lies in n .. n + lookahead -
// 1, which is what this context requested.  The pipeline coalesces the frames in flight into batch launches.
const VSFrame* get_frame_lookahead(int n, Instance* d, VSFrameContext* frameCtx, VSCore* core, const VSAPI* vsapi) {
    std::lock_guard<std::mutex> lock(d->mutex);
    const int depth = d->lookahead;
    const int last = d->vi.numFrames - 1;
    auto slot = [&](int k) -> Pending& { return d->ring[static_cast<size_t>(k % depth)]; };
    auto drop = [&](Pending& p) {
        jinc_filter_wait(d->filter, p.ticket);
        vsapi->freeFrame(p.src);
        vsapi->freeFrame(p.dst);
        p = Pending{};
    };
    auto move_window = [&](int base) {
        for (Pending& p : d->ring)
            if (p.frame >= 0 && (p.frame < base || p.frame >= base + depth)) drop(p);
        d->base = base;
    };
    // a request just below the window (a frame that became ready late): served by itself, the window untouched (see the
    // AviSynth shell; this context holds frame n, it asked for it in arInitial)
    if (n < d->base && n > d->base - depth) {
        const VSFrame* src = vsapi->getFrameFilter(n, d->node, frameCtx);
        if (!src) {
            vsapi->setFilterError("JincResize: the source frame is not available.", frameCtx);
            return nullptr;
        }
        VSFrame* dst = vsapi->newVideoFrame(&d->vi.format, d->vi.width, d->vi.height, src, core);
        const void* sp[4];
        void* dp[4];
        int spitch[4], dpitch[4];
        plane_pointers(vsapi, d, src, dst, sp, spitch, dp, dpitch);
        long long ticket = -1;
        if (jinc_filter_submit(d->filter, sp, spitch, dp, dpitch, &ticket) != JINC_OK || jinc_filter_wait(d->filter, ticket) != JINC_OK) {
            vsapi->setFilterError(jinc_last_error(), frameCtx);
            vsapi->freeFrame(src);
            vsapi->freeFrame(dst);
            return nullptr;
        }
        if (d->chroma_location >= 0) vsapi->mapSetInt(vsapi->getFramePropertiesRW(dst), "_ChromaLocation", d->chroma_location, maReplace);
        vsapi->freeFrame(src);
        return dst;
    }
    if (n < d->base || n >= d->base + depth) {
        move_window(n);
        d->next_submit = n;
    } else if (slot(n).frame != n && n < d->next_submit) {
        d->resubmit = n;
    }
    if (n - d->base > depth - std::max(1, depth / 4)) move_window(n - (depth - std::max(1, depth / 4)));
    while (d->base < n && slot(d->base).frame != d->base) ++d->base;

    auto submit = [&](int k) -> int {  // 0 ok, 1 this context does not hold frame k, 2 failure (reported)
        Pending& p = slot(k);
        p.src = vsapi->getFrameFilter(k, d->node, frameCtx);
        if (!p.src) return 1;
        p.dst = vsapi->newVideoFrame(&d->vi.format, d->vi.width, d->vi.height, p.src, core);
        const void* sp[4];
        void* dp[4];
        int spitch[4], dpitch[4];
        plane_pointers(vsapi, d, p.src, p.dst, sp, spitch, dp, dpitch);
        if (jinc_filter_submit(d->filter, sp, spitch, dp, dpitch, &p.ticket) != JINC_OK) return 2;
        p.frame = k;
        return 0;
    };
    auto failed = [&](int k) -> const VSFrame* {
        Pending& p = slot(k);
        vsapi->setFilterError(jinc_last_error(), frameCtx);
        if (p.src) vsapi->freeFrame(p.src);
        if (p.dst) vsapi->freeFrame(p.dst);
        p = Pending{};
        return nullptr;
    };
    if (d->resubmit >= 0) {
        const int k = d->resubmit, rc = submit(k);
        d->resubmit = -1;
        if (rc == 2) return failed(k);
    }
    for (int k = std::max(d->next_submit, d->base); k <= std::min(std::min(d->base + depth - 1, n + depth - 1), last); ++k) {
        if (slot(k).frame == k) {
            d->next_submit = k + 1;
            continue;
        }
        const int rc = submit(k);
        if (rc == 1) break;
        if (rc == 2) return failed(k);
        d->next_submit = k + 1;
    }
    if (d->next_submit > last) jinc_filter_flush(d->filter);
    Pending& want = slot(n);
    if (want.frame != n) {
        vsapi->setFilterError("JincResize: the source frame is not available.", frameCtx);
        return nullptr;
    }
    if (jinc_filter_wait(d->filter, want.ticket) != JINC_OK) return failed(n);
    VSFrame* dst = want.dst;
    if (d->chroma_location >= 0) vsapi->mapSetInt(vsapi->getFramePropertiesRW(dst), "_ChromaLocation", d->chroma_location, maReplace);
    vsapi->freeFrame(want.src);
    want = Pending{};
    while (d->base < d->next_submit && slot(d->base).frame != d->base) ++d->base;
    return dst;
}

const VSFrame* VS_CC jinc_vs_get_frame(int n, int activationReason, void* instanceData, void**, VSFrameContext* frameCtx, VSCore* core,
                                       const VSAPI* vsapi) {
    Instance* d = static_cast<Instance*>(instanceData);
    if (activationReason == arInitial) {
        const int ahead = std::min(n + d->lookahead - 1, d->vi.numFrames - 1);
        for (int k = n; k <= std::max(n, ahead); ++k) vsapi->requestFrameFilter(k, d->node, frameCtx);
        return nullptr;
    }
    if (activationReason != arAllFramesReady) return nullptr;
    if (d->lookahead > 1) return get_frame_lookahead(n, d, frameCtx, core, vsapi);
    const VSFrame* src = vsapi->getFrameFilter(n, d->node, frameCtx);
    VSFrame* dst = vsapi->newVideoFrame(&d->vi.format, d->vi.width, d->vi.height, src, core);  // inherits the frame properties (ref :613)
    const void* sp[4] = {nullptr, nullptr, nullptr, nullptr};
    void* dp[4] = {nullptr, nullptr, nullptr, nullptr};
    int spitch[4] = {0, 0, 0, 0}, dpitch[4] = {0, 0, 0, 0};
    for (int i = 0; i < d->planes; ++i) {
        sp[i] = vsapi->getReadPtr(src, i);
        spitch[i] = static_cast<int>(vsapi->getStride(src, i));
        dp[i] = vsapi->getWritePtr(dst, i);
        dpitch[i] = static_cast<int>(vsapi->getStride(dst, i));
    }
    if (jinc_filter_get_frame(d->filter, sp, spitch, dp, dpitch) != JINC_OK) {  // no CPU fallback: the failure goes to the host
        vsapi->setFilterError(jinc_last_error(), frameCtx);
        vsapi->freeFrame(src);
        vsapi->freeFrame(dst);
        return nullptr;
    }
    if (d->chroma_location >= 0)  // ref :617-625
        vsapi->mapSetInt(vsapi->getFramePropertiesRW(dst), "_ChromaLocation", d->chroma_location, maReplace);
    vsapi->freeFrame(src);  // ref :627
    return dst;
}

void VS_CC jinc_vs_free(void* instanceData, VSCore*, const VSAPI* vsapi) {  // ref :632-647
    Instance* d = static_cast<Instance*>(instanceData);
    for (Pending& p : d->ring) {
        if (p.frame < 0) continue;
        jinc_filter_wait(d->filter, p.ticket);
        vsapi->freeFrame(p.src);
        vsapi->freeFrame(p.dst);
    }
    jinc_filter_free(d->filter);
    vsapi->freeNode(d->node);
    delete d;
}

// userData = 0: JincResize; 3 / 4 / 6 / 8: the alias with that tap count (ref :1007-1040: the alias forwards src_*, quant_*,
// cplace, threads and adds tap).
void VS_CC jinc_vs_create(const VSMap* in, VSMap* out, void* userData, VSCore* core, const VSAPI* vsapi) {
    const int alias_taps = static_cast<int>(reinterpret_cast<intptr_t>(userData));
    int err = 0;
    VSNode* node = vsapi->mapGetNode(in, "clip", 0, &err);
    if (err || !node) {
        vsapi->mapSetError(out, "JincResize: clip is required.");
        return;
    }
    auto fail = [&](const char* msg) {
        vsapi->mapSetError(out, msg);
        vsapi->freeNode(node);
    };
    const VSVideoInfo* vi = vsapi->getVideoInfo(node);
    if (vi->format.colorFamily == cfUndefined || vi->width <= 0 || vi->height <= 0)
        return fail("JincResize: clip must have a constant format and size.");

    jinc_video_info jvi;
    std::memset(&jvi, 0, sizeof jvi);
    jvi.width = vi->width;
    jvi.height = vi->height;
    jvi.bits_per_component = vi->format.bitsPerSample;
    jvi.component_size = vi->format.bytesPerSample;
    jvi.num_components = vi->format.numPlanes;
    jvi.is_planar = 1;  // every VapourSynth video format is planar
    jvi.is_rgb = vi->format.colorFamily == cfRGB;
    jvi.sub_w = vi->format.subSamplingW;
    jvi.sub_h = vi->format.subSamplingH;
    if (vi->format.sampleType == stFloat && vi->format.bitsPerSample != 32) return fail("JincResize: half-precision float clips are not supported.");

    jinc_args a;
    std::memset(&a, 0, sizeof a);
    auto has = [&](const char* key) { return vsapi->mapNumElements(in, key) > 0; };
    auto get_int = [&](const char* key, unsigned bit, int& field) {
        if (!has(key)) return;
        field = static_cast<int>(vsapi->mapGetInt(in, key, 0, &err));
        a.defined |= bit;
    };
    auto get_float = [&](const char* key, unsigned bit, double& field) {
        if (!has(key)) return;
        field = vsapi->mapGetFloat(in, key, 0, &err);
        a.defined |= bit;
    };
    a.target_width = static_cast<int>(vsapi->mapGetInt(in, "target_width", 0, &err));
    a.target_height = static_cast<int>(vsapi->mapGetInt(in, "target_height", 0, &err));
    get_float("src_left", JINC_ARG_SRC_LEFT, a.src_left);
    get_float("src_top", JINC_ARG_SRC_TOP, a.src_top);
    get_float("src_width", JINC_ARG_SRC_WIDTH, a.src_width);
    get_float("src_height", JINC_ARG_SRC_HEIGHT, a.src_height);
    get_int("quant_x", JINC_ARG_QUANT_X, a.quant_x);
    get_int("quant_y", JINC_ARG_QUANT_Y, a.quant_y);
    get_int("threads", JINC_ARG_THREADS, a.threads);
    std::string cplace;
    if (has("cplace")) {
        const char* s = vsapi->mapGetData(in, "cplace", 0, &err);
        cplace.assign(s ? s : "", static_cast<size_t>(std::max(0, vsapi->mapGetDataSize(in, "cplace", 0, &err))));
        a.cplace = cplace.c_str();
        a.defined |= JINC_ARG_CPLACE;
    }
    if (alias_taps == 0) {  // JincResize only
        get_int("tap", JINC_ARG_TAP, a.tap);
        get_float("blur", JINC_ARG_BLUR, a.blur);
        get_int("opt", JINC_ARG_OPT, a.opt);
        get_int("initial_capacity", JINC_ARG_INITIAL_CAPACITY, a.initial_capacity);
        get_float("initial_factor", JINC_ARG_INITIAL_FACTOR, a.initial_factor);
    }
    // cplace not given: the first frame's _ChromaLocation decides (ref :727-742); the ABI applies the rules
    a.frame0_chroma_location = -1;
    if (!(a.defined & JINC_ARG_CPLACE)) {
        char msg[256];
        if (const VSFrame* frame0 = vsapi->getFrame(0, node, msg, sizeof msg)) {
            const VSMap* props = vsapi->getFramePropertiesRO(frame0);
            if (vsapi->mapGetType(props, "_ChromaLocation") == ptInt) {
                const int64_t loc = vsapi->mapGetInt(props, "_ChromaLocation", 0, &err);
                a.frame0_chroma_location = (loc >= 0 && loc <= 2) ? static_cast<int>(loc) : 3;  // anything else: "invalid _ChromaLocation" (ref :737)
            }
            vsapi->freeFrame(frame0);
        }
    }
    __builtin_cpu_init();  // ref :748-756: opt = 1 / 2 / 3 are validated against the host CPU as in the reference
    a.cpu_has_sse41 = __builtin_cpu_supports("sse4.1") != 0;
    a.cpu_has_avx2 = __builtin_cpu_supports("avx2") != 0;
    a.cpu_has_avx512f = __builtin_cpu_supports("avx512f") != 0;

    jinc_args final_args = a;
    if (alias_taps != 0 && jinc_alias_args(alias_taps, &a, &final_args) != JINC_OK) return fail(jinc_last_error());
    if (alias_taps != 0) final_args.cplace = a.cplace;

    char msg[512];
    jinc_filter* filter = nullptr;
    if (jinc_filter_create(&jvi, &final_args, jinc_pick_device(), &filter, msg, sizeof msg) != JINC_OK) return fail(msg);

    Instance* d = new Instance;
    d->node = node;
    d->filter = filter;
    // default: what the reference binary writes (2 for every sub-sampled format, ref :617-625 with d->cplace never
    // assigned); JINCRESIZE_CHROMALOC=siting writes 0 / 1 / 2 by the siting actually used
    if (const char* e = std::getenv("JINCRESIZE_CHROMALOC"))
        if (std::strcmp(e, "siting") == 0) jinc_filter_set_chroma_location_mode(filter, JINC_CHROMA_LOCATION_BY_SITING);
    d->chroma_location = jinc_filter_chroma_location(filter);
    // JINCRESIZE_SIMD_ORDER=auto | 1 | 2 | 3: as in the AviSynth shell -- the summation order of the path the reference's
    // ladder would pick for this `opt` on this CPU (ref :897-899) instead of the opt=0 result; default off
    if (const char* e = std::getenv("JINCRESIZE_SIMD_ORDER")) {
        int order = 0;
        if (std::strcmp(e, "auto") == 0) {
            const int opt = (final_args.defined & JINC_ARG_OPT) ? final_args.opt : -1;
            order = opt == 3 ? 3 : ((a.cpu_has_avx2 && opt < 0) || opt == 2) ? 2 : ((a.cpu_has_sse41 && opt < 0) || opt == 1) ? 1 : 0;
        } else {
            order = std::max(0, std::min(3, std::atoi(e)));
        }
        if (order) jinc_filter_set_simd_order(filter, order);
    }
    d->planes = vi->format.numPlanes;
    d->vi = *vi;
    jinc_video_info out_vi;
    jinc_filter_output_info(filter, &out_vi);  // ref :791-792
    d->vi.width = out_vi.width;
    d->vi.height = out_vi.height;
    // JINCRESIZE_LOOKAHEAD / JINCRESIZE_GROUP / JINCRESIZE_PIN_FRAMES: as in the AviSynth shell
    if (const char* e = std::getenv("JINCRESIZE_LOOKAHEAD")) d->lookahead = std::max(1, std::min(256, std::atoi(e)));
    if (const char* e = std::getenv("JINCRESIZE_GROUP")) d->group = std::max(0, std::min(d->lookahead, std::atoi(e)));
    const char* pin = std::getenv("JINCRESIZE_PIN_FRAMES");
    const int pin_frames = !pin ? 0 : std::strcmp(pin, "runtime") == 0 ? 3 : (std::strcmp(pin, "pool") == 0 || std::atoi(pin) != 0) ? 2 : 0;  // 0: through the library's pinned buffers; 2: cached registrations; 3: handed to the runtime
    if (d->lookahead > 1) {
        if (jinc_filter_set_pipeline_group(filter, d->lookahead, d->group, pin_frames) != JINC_OK) d->lookahead = 1;
        d->ring.resize(static_cast<size_t>(d->lookahead));
    }
    if (d->lookahead == 1 && pin_frames) jinc_filter_set_pipeline(filter, 1, pin_frames);
    // depth 1: frame n of the output needs frame n of the input, nothing else; look-ahead asks for n .. n + depth - 1.
    // fmParallelRequests: arInitial from any thread, arAllFramesReady one call at a time (the instance is single-threaded).
    VSFilterDependency deps[1] = {{node, d->lookahead > 1 ? rpGeneral : rpStrictSpatial}};
    vsapi->createVideoFilter(out, alias_taps ? "JincAliasResize" : "JincResize", &d->vi, jinc_vs_get_frame, jinc_vs_free,
                             d->lookahead > 1 ? fmParallelRequests : fmUnordered, deps, 1, d, core);
}

}  // namespace

VS_EXTERNAL_API(void) VapourSynthPluginInit2(VSPlugin* plugin, const VSPLUGINAPI* vspapi) {
    vspapi->configPlugin("com.jincresize.mi355x", "jinc", "EWA Jinc resampler (JincResize) on MI355X", VS_MAKE_VERSION(2, 1), VAPOURSYNTH_API_VERSION, 0,
                         plugin);
    // the reference's parameter list (ref :1044-1060), as named map entries
    vspapi->registerFunction("JincResize",
                             "clip:vnode;target_width:int;target_height:int;src_left:float:opt;src_top:float:opt;src_width:float:opt;"
                             "src_height:float:opt;quant_x:int:opt;quant_y:int:opt;tap:int:opt;blur:float:opt;cplace:data:opt;threads:int:opt;"
                             "opt:int:opt;initial_capacity:int:opt;initial_factor:float:opt;",
                             "clip:vnode;", jinc_vs_create, nullptr, plugin);
    static const char kAliasArgs[] =
        "clip:vnode;target_width:int;target_height:int;src_left:float:opt;src_top:float:opt;src_width:float:opt;src_height:float:opt;"
        "quant_x:int:opt;quant_y:int:opt;cplace:data:opt;threads:int:opt;";  // ref :1061-1108
    static const struct { const char* name; intptr_t taps; } kAliases[] = {
        {"Jinc36Resize", 3}, {"Jinc64Resize", 4}, {"Jinc144Resize", 6}, {"Jinc256Resize", 8}};
    for (const auto& al : kAliases) vspapi->registerFunction(al.name, kAliasArgs, "clip:vnode;", jinc_vs_create, reinterpret_cast<void*>(al.taps), plugin);
}
