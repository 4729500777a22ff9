// jincresize_avs.cpp -- the AviSynth+ C-API plugin shell around libjincresize_hip.so.
//
// Registers JincResize and Jinc36/64/144/256Resize with the reference's parameter strings
// (/root/reference/src/JincResize.cpp:1042-1111), parses script arguments the way Create_JincResize does
// (:654-792), and implements the AVS_FilterInfo callbacks (GetFrame :603-630, cache hints :649-652, free :632-647)
// on top of the C ABI in include/jincresize_hip.h.  All resampling happens behind that ABI on the GPU; there is no CPU
// path in this file.
//
// Build against an AviSynth+ installation (its SDK header first on the include path):
//   g++ -std=c++17 -shared -fPIC plugin/jincresize_avs.cpp -Iinclude -I<avisynth sdk>/include
//       -Lavisynth-jincresize_amd/lib -ljincresize_hip -o libjincresize.so
// In this repository, which does not ship the SDK header, it is compiled against plugin/compat/avisynth_c.h (a
// self-written declaration of the API subset with the upstream struct layouts, see the list of points to verify at its
// top): __graft_entry__.build() emits plugin/lib/libjincresize.so that way (plugin/Makefile), and the mock-host tests
// (tests/test_plugin_mock_host.py) compile it together with tests/mock_avs/mock_host.cpp.  INTEGRATION.md section 6
// says what that does and does not prove.
#include "avisynth_c.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "jincresize_hip.h"

// ---- Self-check for the first build against a REAL AviSynth+ SDK (VERDICT r5 Next 9) ----------------
// Everything in this repository was compiled and tested against plugin/compat/avisynth_c.h, a declaration written from the
// API's public names because the SDK header is absent from the build image; the mock host of tests/mock_avs/ shares it, so a
// wrong field order or enum value there is invisible to the tests.  When the SDK's own avisynth_c.h is first on the include
// path (AVS_INC=... make -C plugin), the compat header's guard is undefined and this block compares what the shell was
// written and tested against with what the SDK declares: the build then fails HERE, with the name of the assumption, instead
// of producing a plugin that reads the host's structs at the wrong offsets.
#ifndef JINCRESIZE_COMPAT_AVISYNTH_C_H
#include <cstddef>
static_assert(offsetof(AVS_FilterInfo, child) == 0, "AVS_FilterInfo: child is expected first");
static_assert(offsetof(AVS_FilterInfo, vi) > offsetof(AVS_FilterInfo, child) && offsetof(AVS_FilterInfo, env) > offsetof(AVS_FilterInfo, vi) &&
                  offsetof(AVS_FilterInfo, get_frame) > offsetof(AVS_FilterInfo, env) &&
                  offsetof(AVS_FilterInfo, set_cache_hints) > offsetof(AVS_FilterInfo, get_frame) &&
                  offsetof(AVS_FilterInfo, free_filter) > offsetof(AVS_FilterInfo, set_cache_hints) &&
                  offsetof(AVS_FilterInfo, error) > offsetof(AVS_FilterInfo, free_filter) &&
                  offsetof(AVS_FilterInfo, user_data) > offsetof(AVS_FilterInfo, error),
              "AVS_FilterInfo: field order child, vi, env, get_frame, ..., set_cache_hints, free_filter, error, user_data expected");
static_assert(offsetof(AVS_VideoInfo, width) == 0 && offsetof(AVS_VideoInfo, height) == sizeof(int), "AVS_VideoInfo: width, height expected first");
static_assert(offsetof(AVS_VideoInfo, num_frames) > offsetof(AVS_VideoInfo, height) && offsetof(AVS_VideoInfo, pixel_type) > offsetof(AVS_VideoInfo, num_frames),
              "AVS_VideoInfo: num_frames before pixel_type expected");
static_assert(sizeof(AVS_Value) == 2 * sizeof(void*), "AVS_Value: two machine words expected (type + array_size, then the union)");
static_assert(AVS_PLANAR_Y == 1 && AVS_PLANAR_U == 2 && AVS_PLANAR_V == 4 && AVS_PLANAR_A == 16 && AVS_PLANAR_R == 32 && AVS_PLANAR_G == 64 && AVS_PLANAR_B == 128,
              "AVS_PLANAR_*: the plane ids the mock host was written with differ from the SDK's");
static_assert(AVS_CPUF_SSE4_1 == 0x400 && AVS_CPUF_AVX2 == 0x2000 && AVS_CPUF_AVX512F == 0x10000, "AVS_CPUF_*: values differ from the ones tested against");
static_assert(AVS_CACHE_GET_MTMODE == 509, "AVS_CACHE_GET_MTMODE: value differs from the one tested against");
static_assert(AVS_AEP_INTERFACE_BUGFIX == 51, "AVS_AEP_INTERFACE_BUGFIX: value differs from the one tested against");
#endif

namespace {

struct Pending {  // one frame in flight in the look-ahead ring
    AVS_VideoFrame* src = nullptr;
    AVS_VideoFrame* dst = nullptr;
    long long ticket = -1;
    int frame = -1;
};

struct Instance {
    jinc_filter* filter = nullptr;
    int chroma_location = -1;     // value written to _ChromaLocation, -1: format without sub-sampled chroma
    int lookahead = 1;            // frames in flight (JINCRESIZE_LOOKAHEAD, default 1 = the reference's synchronous GetFrame)
    int group = 0;                // of them coalesced into one launch (JINCRESIZE_GROUP, default 0 = lookahead / 2)
    std::vector<Pending> ring;    // slot k % lookahead holds frame k while it is in the window
    int base = 0;                 // the window [base, base + lookahead): lowest frame not yet served
    int next_submit = 0;          // frames of the window below this one have been submitted
    int resubmit = -1;            // a frame of the window that was served already and is wanted again
    std::mutex mutex;             // look-ahead state (hosts that honour MT_SERIALIZED never contend for it)
    int pin_frames = 0;           // JINCRESIZE_PIN_FRAMES: 0 (default) copied through the library's pinned buffers; 1 / "pool": pinned once, cached by address; "runtime": handed to the HIP runtime
    std::string error;            // storage for fi->error
};

// Argument positions of JincResize (ref :656-674).
enum : int { A_CLIP, A_WIDTH, A_HEIGHT, A_SRC_LEFT, A_SRC_TOP, A_SRC_WIDTH, A_SRC_HEIGHT, A_QUANT_X, A_QUANT_Y, A_TAP, A_BLUR,
             A_CPLACE, A_THREADS, A_OPT, A_INITIAL_CAPACITY, A_INITIAL_FACTOR };

const int kPlanesYuv[4] = {AVS_PLANAR_Y, AVS_PLANAR_U, AVS_PLANAR_V, AVS_PLANAR_A};   // processing order, ref :539-541
const int kPlanesRgb[4] = {AVS_PLANAR_G, AVS_PLANAR_B, AVS_PLANAR_R, AVS_PLANAR_A};

void plane_pointers(const AVS_VideoInfo* vi, AVS_VideoFrame* src, AVS_VideoFrame* dst, const void* sp[4], int spitch[4],
                    void* dp[4], int dpitch[4]) {
    const int* order = avs_is_rgb(vi) ? kPlanesRgb : kPlanesYuv;
    const int n = avs_num_components(vi);
    for (int i = 0; i < 4; ++i) {
        sp[i] = nullptr, dp[i] = nullptr, spitch[i] = 0, dpitch[i] = 0;
        if (i >= n) continue;
        sp[i] = avs_get_read_ptr_p(src, order[i]);
        spitch[i] = avs_get_pitch_p(src, order[i]);
        dp[i] = avs_get_write_ptr_p(dst, order[i]);
        dpitch[i] = avs_get_pitch_p(dst, order[i]);
    }
}

void finish_frame(AVS_FilterInfo* fi, Instance* inst, AVS_VideoFrame* src, AVS_VideoFrame* dst) {
    if (inst->chroma_location >= 0)  // ref :617-625: int property, mode 0 = replace
        avs_prop_set_int(fi->env, avs_get_frame_props_rw(fi->env, dst), "_ChromaLocation", inst->chroma_location, 0);
    avs_release_video_frame(src);    // ref :627
}

// A failed frame: the message goes to the host through fi->error (no CPU fallback) and NO frame is returned -- AviSynth+'s C
// interface throws on fi->error before it takes ownership of what get_frame returned (avisynth_c.cpp, C_VideoFilter::GetFrame), so a
// frame handed back here would leak, and its content would be undefined anyway.
AVS_VideoFrame* report(AVS_FilterInfo* fi, Instance* inst, AVS_VideoFrame* src, AVS_VideoFrame* dst) {
    inst->error = jinc_last_error();
    fi->error = inst->error.c_str();
    if (src) avs_release_video_frame(src);
    if (dst) avs_release_video_frame(dst);
    return nullptr;
}

// ref :603-630, synchronous form
AVS_VideoFrame* get_frame_sync(AVS_FilterInfo* fi, Instance* inst, int n) {
    AVS_VideoFrame* src = avs_get_frame(fi->child, n);
    if (!src) return nullptr;  // ref :610-611
    AVS_VideoFrame* dst = avs_new_video_frame_p(fi->env, &fi->vi, src);  // inherits the frame properties, ref :613
    const void* sp[4];
    void* dp[4];
    int spitch[4], dpitch[4];
    plane_pointers(&fi->vi, src, dst, sp, spitch, dp, dpitch);
    if (jinc_filter_get_frame(inst->filter, sp, spitch, dp, dpitch) != JINC_OK) return report(fi, inst, src, dst);
    finish_frame(fi, inst, src, dst);
    return dst;
}

// Look-ahead form: the frames of a WINDOW [base, base + depth) are in flight on
// the filter's pipeline, which coalesces consecutive frames into groups that share one set of kernel launches (batch
// kernels); each GetFrame still returns exactly frame n.
//
// Who calls: with look-ahead on, the filter answers MT_SERIALIZED (jinc_set_cache_hints), so under Prefetch(N) the host
// keeps ONE instance and its N worker threads call it one at a time -- but not in frame order: thread 2 may ask for frame
// 7 before thread 1 asks for frame 5.  The ring therefore serves any frame of the window in any order without draining:
//   * frame k lives in slot k % depth for as long as it is in the window;
//   * a served frame frees its slot; `base` moves up over served frames, and the window is refilled up to base + depth;
//   * a frame is dropped (waited for, its references returned) only when the window moves past it: the client skipped it
//     (SelectEven: the window trails the newest request by at most 3/4 of its depth) or jumped (a request a whole window
//     or more away moves the window there; frames of the old window that the new one still covers stay in flight);
//   * a request just below the window (a worker thread that lags) is served by itself and leaves the window alone;
//   * a frame of the window that was served already and is asked for again (a cache miss upstream) is fetched again.
// The instance mutex makes the function safe for hosts that ignore the MT mode; under MT_SERIALIZED it is uncontended.
AVS_VideoFrame* get_frame_lookahead(AVS_FilterInfo* fi, Instance* inst, int n) {
    std::lock_guard<std::mutex> lock(inst->mutex);
    const int depth = inst->lookahead;
    const int last = fi->vi.num_frames - 1;
    auto slot = [&](int k) -> Pending& { return inst->ring[static_cast<size_t>(k % depth)]; };
    auto drop = [&](Pending& p) {  // wait for a frame in flight and give its references back
        jinc_filter_wait(inst->filter, p.ticket);
        avs_release_video_frame(p.src);
        avs_release_video_frame(p.dst);
        p = Pending{};
    };
    auto move_window = [&](int base) {  // frames outside [base, base + depth) leave the ring
        for (Pending& p : inst->ring)
            if (p.frame >= 0 && (p.frame < base || p.frame >= base + depth)) drop(p);
        inst->base = base;
    };
    // A request just BELOW the window: under Prefetch(N) a slower worker thread asks for a frame the window has
    // trailed past (or served already).  Treated as a jump it moved the window back, dropped the newer frames in flight and had
    // them computed again -- with a look-ahead below twice the host's threads the ring thrashed.  It is served out of band
    // instead: fetched, submitted and waited for by itself, the ring untouched.  Only a request a whole window or more behind
    // (a seek) moves the window.
    if (n < inst->base && n > inst->base - depth) {
        AVS_VideoFrame* src = avs_get_frame(fi->child, n);
        if (!src) return nullptr;
        AVS_VideoFrame* dst = avs_new_video_frame_p(fi->env, &fi->vi, src);
        const void* sp[4];
        void* dp[4];
        int spitch[4], dpitch[4];
        plane_pointers(&fi->vi, src, dst, sp, spitch, dp, dpitch);
        long long ticket = -1;
        if (jinc_filter_submit(inst->filter, sp, spitch, dp, dpitch, &ticket) != JINC_OK || jinc_filter_wait(inst->filter, ticket) != JINC_OK)
            return report(fi, inst, src, dst);
        finish_frame(fi, inst, src, dst);
        return dst;
    }
    if (n < inst->base || n >= inst->base + depth) {  // a jump
        move_window(n);
        inst->next_submit = n;
    } else if (slot(n).frame != n && n < inst->next_submit) {  // served before, wanted again
        inst->resubmit = n;
    }
    // a client that skips frames: trail it, do not stall -- but only by what the look-ahead needs in front of n (a quarter of
    // the window): frames up to 3/4 of the window behind the newest request stay in flight for the threads that lag
    if (n - inst->base > depth - std::max(1, depth / 4)) move_window(n - (depth - std::max(1, depth / 4)));
    while (inst->base < n && slot(inst->base).frame != inst->base) ++inst->base;  // over frames served (or dropped) already

    auto submit = [&](int k) -> int {  // 0 ok, 1 the child has no such frame, 2 failure (reported)
        Pending& p = slot(k);
        p.src = avs_get_frame(fi->child, k);
        if (!p.src) return 1;
        p.dst = avs_new_video_frame_p(fi->env, &fi->vi, p.src);
        const void* sp[4];
        void* dp[4];
        int spitch[4], dpitch[4];
        plane_pointers(&fi->vi, p.src, p.dst, sp, spitch, dp, dpitch);
        if (jinc_filter_submit(inst->filter, sp, spitch, dp, dpitch, &p.ticket) != JINC_OK) return 2;
        p.frame = k;
        return 0;
    };
    auto failed = [&](int k) {
        Pending& p = slot(k);
        AVS_VideoFrame *dst = p.dst, *src = p.src;
        p = Pending{};
        return report(fi, inst, src, dst);
    };
    if (inst->resubmit >= 0) {
        const int k = inst->resubmit, rc = submit(k);
        inst->resubmit = -1;
        if (rc == 2) return failed(k);
    }
    for (int k = std::max(inst->next_submit, inst->base); k <= std::min(inst->base + depth - 1, last); ++k) {
        if (slot(k).frame == k) {  // still in flight from before a jump
            inst->next_submit = k + 1;
            continue;
        }
        const int rc = submit(k);
        if (rc == 1) break;
        if (rc == 2) return failed(k);
        inst->next_submit = k + 1;
    }
    if (inst->next_submit > last) jinc_filter_flush(inst->filter);  // end of the clip: the last frames leave without company
    Pending& want = slot(n);
    if (want.frame != n) return nullptr;  // the child had no frame n
    if (jinc_filter_wait(inst->filter, want.ticket) != JINC_OK) return failed(n);
    AVS_VideoFrame* dst = want.dst;
    finish_frame(fi, inst, want.src, dst);
    want = Pending{};
    while (inst->base < inst->next_submit && slot(inst->base).frame != inst->base) ++inst->base;
    return dst;
}

AVS_VideoFrame* AVSC_CC jinc_get_frame(AVS_FilterInfo* fi, int n) {
    Instance* inst = static_cast<Instance*>(fi->user_data);
    return inst->lookahead > 1 ? get_frame_lookahead(fi, inst, n) : get_frame_sync(fi, inst, n);
}

// ref :649-652: MT_MULTI_INSTANCE, one instance per worker thread -- the reference's answer, and this filter's at depth 1.
// With look-ahead on, N instances under Prefetch(N) would each see every N-th frame or so and each prefetch the frames the
// others compute; the answer is then MT_SERIALIZED: ONE instance sees the whole clip and keeps the GPU busy from its
// window, the host's worker threads take turns at it (get_frame_lookahead says what that means for the order of requests).
int AVSC_CC jinc_set_cache_hints(AVS_FilterInfo* fi, int cachehints, int) {
    if (cachehints != AVS_CACHE_GET_MTMODE) return 0;
    const Instance* inst = static_cast<const Instance*>(fi->user_data);
    return inst && inst->lookahead > 1 ? 3 /* MT_SERIALIZED */ : 2 /* MT_MULTI_INSTANCE */;
}

void AVSC_CC jinc_free(AVS_FilterInfo* fi) {  // ref :632-647
    Instance* inst = static_cast<Instance*>(fi->user_data);
    if (!inst) return;
    for (Pending& p : inst->ring) {
        if (p.frame < 0) continue;
        jinc_filter_wait(inst->filter, p.ticket);
        avs_release_video_frame(p.src);
        avs_release_video_frame(p.dst);
    }
    jinc_filter_free(inst->filter);
    delete inst;
    fi->user_data = nullptr;
}

AVS_Value AVSC_CC create_jincresize(AVS_ScriptEnvironment* env, AVS_Value args, void*) {
    AVS_FilterInfo* fi = nullptr;
    AVS_Clip* clip = avs_new_c_filter(env, &fi, avs_array_elt(args, A_CLIP), 1);  // ref :679
    AVS_VideoInfo* vi = &fi->vi;
    auto fail = [&](const char* msg) {  // ref :682-687
        avs_release_clip(clip);
        return avs_new_value_error(msg);
    };

    // AviSynth+ interface 9.2 (r3688) or later, ref :689-698
    static const char kTooOld[] = "JincResize: AviSynth+ version must be r3688 or later.";
    if (avs_check_version(env, 9) != 0) return fail(kTooOld);
    if (avs_check_version(env, 10) != 0 && avs_get_env_property(env, AVS_AEP_INTERFACE_BUGFIX) < 2) return fail(kTooOld);

    jinc_video_info jvi;
    std::memset(&jvi, 0, sizeof jvi);
    jvi.width = vi->width;
    jvi.height = vi->height;
    jvi.bits_per_component = avs_bits_per_component(vi);
    jvi.component_size = avs_component_size(vi);
    jvi.num_components = avs_num_components(vi);
    jvi.is_planar = avs_is_planar(vi) ? 1 : 0;
    jvi.is_rgb = avs_is_rgb(vi) ? 1 : 0;
    const bool has_chroma = jvi.is_planar && !jvi.is_rgb && jvi.num_components > 1;
    jvi.sub_w = has_chroma ? avs_get_plane_width_subsampling(vi, AVS_PLANAR_U) : 0;    // ref :833
    jvi.sub_h = has_chroma ? avs_get_plane_height_subsampling(vi, AVS_PLANAR_U) : 0;   // ref :834

    jinc_args a;
    std::memset(&a, 0, sizeof a);
    auto given = [&](int idx, unsigned bit) {
        const bool d = avs_defined(avs_array_elt(args, idx)) != 0;
        if (d) a.defined |= bit;
        return d;
    };
    a.target_width = avs_as_int(avs_array_elt(args, A_WIDTH));
    a.target_height = avs_as_int(avs_array_elt(args, A_HEIGHT));
    if (given(A_SRC_LEFT, JINC_ARG_SRC_LEFT)) a.src_left = avs_as_float(avs_array_elt(args, A_SRC_LEFT));
    if (given(A_SRC_TOP, JINC_ARG_SRC_TOP)) a.src_top = avs_as_float(avs_array_elt(args, A_SRC_TOP));
    if (given(A_SRC_WIDTH, JINC_ARG_SRC_WIDTH)) a.src_width = avs_as_float(avs_array_elt(args, A_SRC_WIDTH));
    if (given(A_SRC_HEIGHT, JINC_ARG_SRC_HEIGHT)) a.src_height = avs_as_float(avs_array_elt(args, A_SRC_HEIGHT));
    if (given(A_QUANT_X, JINC_ARG_QUANT_X)) a.quant_x = avs_as_int(avs_array_elt(args, A_QUANT_X));
    if (given(A_QUANT_Y, JINC_ARG_QUANT_Y)) a.quant_y = avs_as_int(avs_array_elt(args, A_QUANT_Y));
    if (given(A_TAP, JINC_ARG_TAP)) a.tap = avs_as_int(avs_array_elt(args, A_TAP));
    if (given(A_BLUR, JINC_ARG_BLUR)) a.blur = avs_as_float(avs_array_elt(args, A_BLUR));
    if (given(A_CPLACE, JINC_ARG_CPLACE)) a.cplace = avs_as_string(avs_array_elt(args, A_CPLACE));
    if (given(A_THREADS, JINC_ARG_THREADS)) a.threads = avs_as_int(avs_array_elt(args, A_THREADS));
    if (given(A_OPT, JINC_ARG_OPT)) a.opt = avs_as_int(avs_array_elt(args, A_OPT));
    if (given(A_INITIAL_CAPACITY, JINC_ARG_INITIAL_CAPACITY)) a.initial_capacity = avs_as_int(avs_array_elt(args, A_INITIAL_CAPACITY));
    if (given(A_INITIAL_FACTOR, JINC_ARG_INITIAL_FACTOR)) a.initial_factor = avs_as_float(avs_array_elt(args, A_INITIAL_FACTOR));

    // cplace not given: the first frame's _ChromaLocation decides (ref :727-742); the ABI applies the rules
    a.frame0_chroma_location = -1;
    if (!(a.defined & JINC_ARG_CPLACE) && jvi.is_planar) {
        // (the reference asks the new filter's clip, whose get_frame is still unset and passes through to the child)
        if (AVS_VideoFrame* frame0 = avs_get_frame(fi->child, 0)) {
            const AVS_Map* props = avs_get_frame_props_ro(env, frame0);
            if (avs_prop_get_type(env, props, "_ChromaLocation") == 'i') {
                const int64_t loc = avs_prop_get_int(env, props, "_ChromaLocation", 0, nullptr);
                a.frame0_chroma_location = (loc >= 0 && loc <= 2) ? static_cast<int>(loc) : 3;  // anything else: "invalid _ChromaLocation" (ref :737)
            }
            avs_release_video_frame(frame0);
        }
    }
    const int cpu = avs_get_cpu_flags(env);  // ref :748: opt = 1/2/3 are validated against the host CPU as before
    a.cpu_has_sse41 = (cpu & AVS_CPUF_SSE4_1) != 0;
    a.cpu_has_avx2 = (cpu & AVS_CPUF_AVX2) != 0;
    a.cpu_has_avx512f = (cpu & AVS_CPUF_AVX512F) != 0;

    // The message of a failed create must outlive this call (the host reads it from the returned value).
    static thread_local char err[512];
    jinc_filter* filter = nullptr;
    // round-robin over the node's GPUs: AviSynth creates one instance per worker thread (Prefetch(N))
    if (jinc_filter_create(&jvi, &a, jinc_pick_device(), &filter, err, sizeof err) != JINC_OK) return fail(err);

    Instance* inst = new Instance;
    inst->filter = filter;
    // default: what the reference binary writes (2 for every sub-sampled format, ref :617-625 with d->cplace never
    // assigned); JINCRESIZE_CHROMALOC=siting writes 0 / 1 / 2 by the siting actually used
    if (const char* e = std::getenv("JINCRESIZE_CHROMALOC"))
        if (std::strcmp(e, "siting") == 0) jinc_filter_set_chroma_location_mode(filter, JINC_CHROMA_LOCATION_BY_SITING);
    inst->chroma_location = jinc_filter_chroma_location(filter);
    // JINCRESIZE_SIMD_ORDER: by default every frame is the reference's opt=0 result, whatever `opt` says (the parity
    // target).  "auto" reproduces what the reference binary would have computed for this call on this host instead: the
    // summation order of the path its ladder picks (ref :897-899: opt=3 -> AVX-512; opt=2, or opt<0 on a CPU with AVX2 ->
    // AVX2; opt=1, or opt<0 with SSE4.1 -> SSE4.1; else the C path).  "1" / "2" / "3" force one order.  Slow kernel.
    if (const char* e = std::getenv("JINCRESIZE_SIMD_ORDER")) {
        int order = 0;
        if (std::strcmp(e, "auto") == 0) {
            const int opt = (a.defined & JINC_ARG_OPT) ? a.opt : -1;
            order = opt == 3 ? 3 : ((a.cpu_has_avx2 && opt < 0) || opt == 2) ? 2 : ((a.cpu_has_sse41 && opt < 0) || opt == 1) ? 1 : 0;
        } else {
            order = std::max(0, std::min(3, std::atoi(e)));
        }
        if (order) jinc_filter_set_simd_order(filter, order);
    }
    if (const char* e = std::getenv("JINCRESIZE_LOOKAHEAD")) inst->lookahead = std::max(1, std::min(256, std::atoi(e)));
    if (const char* e = std::getenv("JINCRESIZE_GROUP")) inst->group = std::max(0, std::min(inst->lookahead, std::atoi(e)));
    // JINCRESIZE_PIN_FRAMES=1 (or "pool"): the host's frame buffers are pinned in place, once, and the registrations cached by
    // address -- for a host whose frame pool stays mapped: asynchronous copies, results written by the
    // shader (process-wide registry in the library: the instances of a script share the host's frames).  Unset / 0: the CPU copies
    // the planes through pinned buffers of the library's own and the device never maps the host's pages.  "runtime": the planes go
    // to the HIP runtime as they are (the default of rounds 1 - 5).
    if (const char* e = std::getenv("JINCRESIZE_PIN_FRAMES"))
        inst->pin_frames = std::strcmp(e, "runtime") == 0 ? 3 : (std::strcmp(e, "pool") == 0 || std::atoi(e) != 0) ? 2 : 0;
    if (inst->lookahead > 1) {
        if (jinc_filter_set_pipeline_group(filter, inst->lookahead, inst->group, inst->pin_frames) != JINC_OK) inst->lookahead = 1;
        inst->ring.resize(static_cast<size_t>(inst->lookahead));
    }
    if (inst->lookahead == 1 && inst->pin_frames) jinc_filter_set_pipeline(filter, 1, inst->pin_frames);

    jinc_video_info out_vi;
    jinc_filter_output_info(filter, &out_vi);  // ref :791-792
    vi->width = out_vi.width;
    vi->height = out_vi.height;
    fi->user_data = inst;
    fi->get_frame = jinc_get_frame;              // ref :977
    fi->set_cache_hints = jinc_set_cache_hints;  // ref :978
    fi->free_filter = jinc_free;                 // ref :979
    AVS_Value v = avs_new_value_clip(clip);      // ref :974
    avs_release_clip(clip);                      // ref :981
    return v;
}

// Jinc36Resize / Jinc64Resize / Jinc144Resize / Jinc256Resize: the arguments that are defined travel by name, plus
// tap = 3 / 4 / 6 / 8, into JincResize (ref :1007-1040).
AVS_Value AVSC_CC create_alias(AVS_ScriptEnvironment* env, AVS_Value args, void* param) {
    static const char* const kNames[8] = {"src_left", "src_top", "src_width", "src_height", "quant_x", "quant_y", "cplace", "threads"};
    AVS_Value values[12];
    const char* names[12];
    int n = 0;
    for (int i = 0; i < 3; ++i) values[n] = avs_array_elt(args, i), names[n++] = nullptr;  // clip, width, height
    for (int i = 0; i < 8; ++i) {
        const AVS_Value v = avs_array_elt(args, 3 + i);
        if (avs_defined(v)) values[n] = v, names[n++] = kNames[i];
    }
    values[n] = avs_new_value_int(static_cast<int>(reinterpret_cast<intptr_t>(param)));
    names[n++] = "tap";
    return avs_invoke(env, "JincResize", avs_new_value_array(values, n), names);
}

}  // namespace

extern "C" AVSC_EXPORT const char* AVSC_CC avisynth_c_plugin_init(AVS_ScriptEnvironment* env) {
    avs_add_function(env, "JincResize",
                     "cii[src_left]f[src_top]f[src_width]f[src_height]f[quant_x]i[quant_y]i[tap]i[blur]f[cplace]s[threads]i[opt]i"
                     "[initial_capacity]i[initial_factor]f",
                     create_jincresize, nullptr);
    static const char kAliasParams[] = "cii[src_left]f[src_top]f[src_width]f[src_height]f[quant_x]i[quant_y]i[cplace]s[threads]i";
    static const struct { const char* name; intptr_t taps; } kAliases[] = {
        {"Jinc36Resize", 3}, {"Jinc64Resize", 4}, {"Jinc144Resize", 6}, {"Jinc256Resize", 8}};
    for (const auto& al : kAliases) avs_add_function(env, al.name, kAliasParams, create_alias, reinterpret_cast<void*>(al.taps));
    return "JincResize";
}
