// dispatch.cpp -- the kernel launches of one frame call: per plane, which interior kernel and which border kernels run,
// on which stream (the body of the process_frame call, ref /root/reference/src/JincResize.cpp:615, on device planes).
// Layout: Rules (the measured cross-over constants) -> Choice (the rules applied to one call: no launches) -> launch_plane (the
// launches of one plane under a Choice) -> enqueue_run (batch split, fork / join of the side stream, the plane loop).
#include <algorithm>
#include <cstdlib>
#include <mutex>

#include "filter_internal.h"
#include "knobs.h"

namespace jinc {
namespace host {

namespace {
// Cross-over points of the automatic kernel choice, each from a measurement on MI355X (file under profiles/):
struct Rules {
    // calls below this many output samples of plans the quasi-periodic kernel owns go to the gather kernel
    // (round2/single_frame_rates.log: 4/3x one frame per call 45 -> 68 Gpix/s)
    static constexpr double kQuasiMinSamples = 3.0e6;
    // border kernels move to the side stream from this many taps per call (experiments/overlap_small_batches.sh)
    static constexpr double kOverlapMinTaps = 1.0e9;
    // frame-lane forms against the single-frame kernels (round2/fl_threshold.log): gather-kernel plans from
    // kFrameLaneMinFrames (16) frames, filter sizes above 9 from 24, drifting plans with more than 16 phases from 36,
    // drifting fs-9 / fs-7 plans with a source step of 2 from 48 / 64
    static constexpr int kFrameLaneMinFramesBigFs = 24;
    // Groups of fewer than 64 frames on the frame-lane kernel's sub-group form (profiles/round4/fl_sub_ab.log, fl_small_ab.log):
    // 16 / 8 / 4 sub-groups per wave (4 / 8 / 16 frames per workgroup) up to these frame counts -- two sub-groups (32 frames) are
    // never ahead of four (1.37x at 32 frames 298 : 295 Gpix/s, at 24 frames 222 : 229; 5/6 at 32 frames 200 : 226) -- and the
    // 64-frame form above 48 (1.37x at 48 frames: 267 against 300 on three groups of 16; 5/6 216 : 236; 1.5x with tap 4 184 : 203;
    // at 56 frames four groups of 16 cost what 64 frames cost).  The form passes the gather kernel at 2 frames (1.37x: 1 frame 32.2
    // against 31.3 Gpix/s -- level, and a one-frame call keeps the plane pairs of launch_plane --, 2: 60 : 42, 3: 82 : 48, 4: 98 : 51,
    // 8: 169 : 57, 12: 206 : 60).
    static constexpr int kFlSub16MaxFrames = 4;
    static constexpr int kFlSub8MaxFrames = 8;
    static constexpr int kFlSub4MaxFrames = 48;
    static constexpr int kFlSub2MaxFrames = 0;   // (two sub-groups: kernel mode 16 and the JINC_FL_SUB knob only)
    static constexpr int kFlSubMinFrames = 2;
    // (against the runs form of the direct kernel -- drifting plans with fs >= 9 -- the frame-lane kernel is never chosen:
    // round3/runs_vs_auto.txt, 128 frames, border frame on the frame-lane kernel: DVD -> 1080p with tap 4 169 against 159 Gpix/s,
    // 5/2 with tap 6 135.5 against 124.7, 1.5x with tap 8 at 256 frames 88.9 against 79.7, with tap 4 265 against 255)
    // batches from this many frames on: the border frame of a runs-form plan on the frame-lane kernel instead of the gather kernel
    // (round3/runs_vs_auto.txt, 128 / 256 frames, same box: 1.5x with tap 8 82.6 -> 88.9 Gpix/s, 3x with tap 8 77 -> 86, with tap 4
    // 246 -> 272, 5/2 with tap 6 120 -> 135.5, DVD -> 1080p with tap 4 152 -> 169; 1.5x with tap 4, whose frame is 6 pixels wide:
    // 276 -> 300, at 64 frames 268 -> 288 -- once the tile choice prices a tile by the pixels it really holds, framelane_dispatch.cpp;
    // with the square tiles chosen before, it lost 2 .. 8 % there)
    // (by batch size, frame-lane against gather border: 1.5x with tap 4 +6 % at 16 frames, +9 % at 32; with tap 8 -8 % at 16, level
    // at 32, +3 % at 48; 3x with tap 4 -2 % / +3 % / +5 %; DVD -> 1080p with tap 4 +1 % / +5.5 % / +10 %)
    static constexpr int kRunsFrameLaneBorderMinFrames = 32;
    // ... and from 8 frames where the frame-lane kernel's sub-group form takes the border (filter sizes up to 9: tap 4;
    // round4/runs_border_small_ab.log: 1.5x with tap 4 -15 % at 4 frames, +7.5 % at 8, +10 % at 16, +9 % at 32; 3x -10 / +6.5 / +13 /
    // +12 %; DVD -> 1080p -6 / -0.7 / +4 / +8 %)
    static constexpr int kRunsFrameLaneBorderMinFramesSub = 8;
    // calls of exactly periodic plans below this many taps: one gather launch over the border frame instead of the strip kernels
    static constexpr double kStripBorderMinTaps = 5.0e9;
    // ... single-plane calls whose border columns the interior kernel takes itself (see wants_border_strips)
    static constexpr double kStripBorderMinTapsEdgeCols = 2.4e9;
    // multi-plane calls whose first plane has at most this many output samples run their other planes on the side stream
    static constexpr double kPlaneForkMaxSamples = 1.0e7;
    // filter sizes from which the runs form is the automatic choice at every call size; below (the quasi-periodic kernel's plans)
    // only for calls of at most this many output samples
    static constexpr int kRunsMinFilterSize = 9;
    static constexpr double kRunsSmallFsMaxSamples = 4.0e7, kRunsSmallFsMaxSamplesManyPhases = 8.0e7;
    // calls (per plane) below this many taps stay with the gather kernel
    static constexpr double kRunsMinTaps = 1.0e8;
    // border kernels also move to the side stream when the border frame alone holds this many taps per call (drifting plans
    // with large taps: every border pixel owns a set; 1.5x with tap 8, one frame: 20.5e6 border taps, 24.8 -> 29.0 Gpix/s;
    // with tap 4: 2.9e6, 55.6 -> 44.3 when forked; four frames: 11.7e6, 143 -> 129)
    static constexpr double kOverlapMinBorderTaps = 2.0e7;
    static constexpr int kFrameLaneMinFramesManyPhases = 36;
    static constexpr int kFrameLaneMinFramesStep2Fs9 = 48;
    static constexpr int kFrameLaneMinFramesStep2Fs7 = 64;
    // window kernels take half-height tiles below this many workgroups per launch (round2 small-call rules)
    static constexpr long long kHalfTileMaxWorkgroups = 6144;
    // fs-9 quad form (ewa_periodic_quad9_kernel) below this many full-tile workgroups per launch (C4: one frame per call)
    static constexpr long long kQuad9MaxWorkgroups = 4096;
    // border columns of exactly periodic plans on the frame-lane kernel from this many frames per call on (up to 32 frames its
    // sub-group form; round4/fl_cols_small_ab.log: level at 4 and 8 frames, 1080p -> 4K 8-bit +0.4 ... 3.6 % at 16, +1.2 at 32,
    // +3 at 48; 4:2:0 +6.5 % at 16)
    static constexpr int kFlColsMinFrames = 16;
    // float planes on the trimmed support (a cleared flag set, a scan of the plane's rim and two launches per plane, the second
    // returning at once) from this many taps per plane and call on: 1080p -> 4K float, one frame (4e8 taps) 69 against 59 ... 62
    // Gpix/s, eight frames 144 -> 164, 64 frames 174 -> 207 (round4/float_trim_ab.log)
    static constexpr double kFloatTrimMinTaps = 1.0e9;
    // ... and its two-periods-per-lane form on integer planes from this many workgroups per launch on
    // (round 4: 4096.  Round 5: this form computes the plane's border columns in its edge tiles and is the only one with the chroma planes'
    // 8 x 9 support, and it is ahead from two frames per call on: Jinc64 1080p -> 4K Y8 at 2 / 4 / 8 frames 236 -> 239 / 320 -> 332 / 353 ->
    // 455 Gpix/s, level at one; 16-bit 4:2:0 at 8 / 16 frames 194 -> 255 / 270 -> 300; round5/chroma_span9_ab.log)
    static constexpr long long kQuad2x8MinWorkgroups = 1024;
    // quad form on the trimmed 8 x 8 support (tap 4 at 2x) instead of the window kernel
    static constexpr bool kQuad8 = true;
    // two-periods-per-lane quad form on the trimmed 6 x 6 support from this many half-height workgroups per launch on
    static constexpr long long kQuad2MinWorkgroups = 256;
    // ... and its half-height tiles (24 period-rows) below this many full-tile workgroups per launch
    static constexpr long long kQuad2HalfTileMaxWorkgroups = 12000;  // (C2 at 16 frames: 734 against 718 Gpix/s; at 64: 803 against 827)
    // fs-7 quad form from this many periods (2 x 2 pixels each) per plane on: 1080p -> 4K has 2.05 M, 360p -> 720p 0.22 M
    static constexpr long long kQuadMinPeriods = 1000000;
    // workgroups a quasi-periodic launch aims for when it splits a tile's phases (fs 9 tiles cost more to stage)
    static constexpr long long kQuasiSplitTarget = 1024, kQuasiSplitTargetFs9 = 400;
};

// Everything the launches below assume about the caller's planes, checked BEFORE anything is queued: an error on plane 2
// must not leave plane 0's border kernels running on the side stream with no join recorded.
void validate_planes(const jinc_filter& f, const void* const src[4], const int src_pitch[4], const size_t src_fs[4],
                     void* const dst[4], const int dst_pitch[4], const size_t dst_fs[4], int nframes) {
    const int sb = f.vi_in.component_size;
    for (int i = 0; i < f.planecount; ++i) {
        const DeviceTable& t = f.tables[f.table_of_plane(i)];
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
    }
}

// Two planes of ONE frame that share a table (the chroma planes of a YUV frame) as a two-frame batch of one plane: what the second
// plane's pointers are from the first's.
struct PlanePair {
    size_t src_stride = 0, dst_stride = 0;
};

// Can planes i and i + 1 of one frame be addressed as frames 0 and 1 of plane i?  Same pitches, the second plane behind the first
// by the same kind of distance in source and destination, the distances multiples of 4 bytes (what the direct kernel asks of a
// frame stride) and of the sample size.
bool plane_pair(const void* const src[4], const int src_pitch[4], void* const dst[4], const int dst_pitch[4], int i, int sb, PlanePair& out) {
    if (src_pitch[i] != src_pitch[i + 1] || dst_pitch[i] != dst_pitch[i + 1]) return false;
    const uintptr_t s0 = reinterpret_cast<uintptr_t>(src[i]), s1 = reinterpret_cast<uintptr_t>(src[i + 1]);
    const uintptr_t d0 = reinterpret_cast<uintptr_t>(dst[i]), d1 = reinterpret_cast<uintptr_t>(dst[i + 1]);
    if (s1 <= s0 || d1 <= d0) return false;
    out.src_stride = s1 - s0;
    out.dst_stride = d1 - d0;
    return out.src_stride % 4 == 0 && out.dst_stride % 4 == 0 && out.src_stride % sb == 0 && out.dst_stride % sb == 0;
}

// Frames per call from which the border frame of a runs-form plan goes to the frame-lane kernel; A/B knob
// RUNS_FL_BORDER_FRAMES (0: never).
int runs_fl_border_min_frames(bool sub_form) {
    const int knob = knobs::geti(JINC_KNOB_RUNS_FL_BORDER_FRAMES, -1);
    if (knob >= 0) return knob == 0 ? INT32_MAX : knob;
    return sub_form ? Rules::kRunsFrameLaneBorderMinFramesSub : Rules::kRunsFrameLaneBorderMinFrames;
}

bool plane_fork_enabled() { return knobs::flag(JINC_KNOB_PLANE_FORK, true); }  // A/B knob PLANE_FORK (default: on)
bool plane_pair_enabled() { return knobs::flag(JINC_KNOB_PLANE_PAIR, true); }  // A/B knob PLANE_PAIR (default: on)
// A/B knob QUASI_SPLIT: workgroups per tile of the quasi-periodic kernel (0 / 1: no split)
int quasi_split_knob() { return knobs::geti(JINC_KNOB_QUASI_SPLIT, -1); }
}  // namespace

namespace {
void enqueue_run(jinc_filter& f, const void* const src[4], const int src_pitch[4], const size_t src_fs[4], void* const dst[4],
                 const int dst_pitch[4], const size_t dst_fs[4], int nframes, hipStream_t stream, bool may_split);
// test hook (jinc_debug_last_call): interior kernel of table 0 and frame count of the most recent enqueue in this process,
// for callers that cannot reach the filter handle (the plugin shell's instances)
std::atomic<const char*> g_last_interior_kernel{""};
std::atomic<int> g_last_call_frames{0};
std::mutex g_last_instance_mutex;
std::string g_last_instance;  // DeviceTable::last_instance of table 0, same call
}

void enqueue(jinc_filter& f, const void* const src[4], const int src_pitch[4], const size_t src_fs[4],
             void* const dst[4], const int dst_pitch[4], const size_t dst_fs[4], int nframes, hipStream_t stream) {
    validate_planes(f, src, src_pitch, src_fs, dst, dst_pitch, dst_fs, nframes);
    enqueue_run(f, src, src_pitch, src_fs, dst, dst_pitch, dst_fs, nframes, stream, true);
    g_last_interior_kernel.store(f.tables[0].last_kernel, std::memory_order_relaxed);
    g_last_call_frames.store(nframes, std::memory_order_relaxed);
    std::lock_guard<std::mutex> lock(g_last_instance_mutex);
    g_last_instance = f.tables[0].last_instance;
}

const char* last_interior_kernel_in_process() { return g_last_interior_kernel.load(std::memory_order_relaxed); }
int last_call_frames_in_process() { return g_last_call_frames.load(std::memory_order_relaxed); }
const char* last_interior_instance_in_process() {
    thread_local std::string copy;
    std::lock_guard<std::mutex> lock(g_last_instance_mutex);
    copy = g_last_instance;
    return copy.c_str();
}

namespace {
// What one call launches for each plane: the rules of the automatic kernel choice (and the forced kernel modes), evaluated for the
// call's planes, pitches and frame count.  No launches here.
struct Choice {
    const jinc_filter& f;
    const int* src_pitch;
    const size_t* src_fs;
    const int nframes;
    const int sb;
    double call_samples = 0.0;  // output samples of the whole call

    Choice(const jinc_filter& filter, const int src_pitch_[4], const size_t src_fs_[4], int nframes_)
        : f(filter), src_pitch(src_pitch_), src_fs(src_fs_), nframes(nframes_), sb(filter.vi_in.component_size) {
        for (int i = 0; i < f.planecount; ++i) {
            const DeviceTable& t = f.tables[f.table_of_plane(i)];
            call_samples += static_cast<double>(t.plan.dst_w) * t.plan.dst_h;
        }
        call_samples *= nframes;
    }

    // kernel_mode: 0 automatic, 1 gather only, 2.. A/B variants of the periodic kernels, 7 quasi-periodic
    // kernel wherever it applies (also for exactly periodic plans)
    // Calls of fewer than ~3e6 output samples (one 1080p frame, two 960 x 540 frames) leave the quasi-periodic kernel with a
    // few dozen workgroups of long serial phase loops even after the phase split below; the gather kernel's many small
    // blocks finish sooner: one frame per call 4/3x 45 -> 68 Gpix/s, 4/3x with tap 4 37 -> 46, 960 x 540 at 1.5x 16.5 -> 28
    // (two frames: 33 -> 44), 5/4x and 1.5x at 1080p equal; 1.5x with tap 4 loses 11 % (45 -> 40).
    bool quasi_declined(const DeviceTable& t) const {  // (... in favour of the gather kernel, not of the direct kernel)
        return f.kernel_mode == 0 && call_samples < Rules::kQuasiMinSamples && t.use_quasi && !t.use_periodic;
    }
    bool wants_quasi(const DeviceTable& t) const {
        if (quasi_declined(t)) return false;
        return t.use_quasi && (f.kernel_mode == 7 || f.kernel_mode == 8 || f.kernel_mode == 10 || (f.kernel_mode != 1 && !t.use_periodic));
    }
    bool wants_periodic(const DeviceTable& t) const {
        return t.use_periodic && f.kernel_mode != 1 && f.kernel_mode != 7 && f.kernel_mode != 8 && f.kernel_mode != 10;
    }
    // Is kernel_direct.hip usable for plane i (interior and border strips)?  See direct_fetch_is_safe().
    bool direct_ok(const DeviceTable& t, int i) const {
        if (!t.use_direct || f.kernel_mode == 1 || !f.direct_premise) return false;
        const uint64_t plane_bytes = static_cast<uint64_t>(src_pitch[i]) * (t.plan.src_h - 1) + static_cast<uint64_t>(t.plan.src_w) * sb;
        return direct_fetch_is_safe(src_fs ? src_fs[i] : 0, nframes, plane_bytes, src_pitch[i], t.plan.fs);
    }
    // kernel_mode 9: the direct kernel wherever it applies; otherwise it takes the interior of the exactly periodic
    // plans the register/LDS kernels do not cover.
    bool wants_direct(const DeviceTable& t, int i) const {
        if (!direct_ok(t, i) || quasi_declined(t)) return false;
        return f.kernel_mode == 9 || (!wants_periodic(t) && !wants_quasi(t));
    }
    // Runs form of the direct kernel (DeviceTable::use_runs): drifting plans with filter sizes from 9 on (taps 4..16 at 1.5x, 3x,
    // 5/2, 8/3 x 9/4 ...).  Above fs 9 the alternative is the gather kernel (1.5x with tap 8, one frame per call: 13.6 -> 24.8
    // Gpix/s, 16 frames: 26 -> 77); at fs 9 the quasi-periodic kernel (1.5x with tap 4: 1 / 16 / 64 frames per call 39.5 / 138 /
    // 244 -> 55 / 217 / 268; 3x: 35.6 / 146 -> 51.5 / 197).  kernel_mode 14: wherever the plan has runs.
    bool wants_runs(const DeviceTable& t, int i) const {
        if (!t.use_runs || !f.direct_premise || (f.kernel_mode != 0 && f.kernel_mode != 14)) return false;
        if (f.kernel_mode == 0 && t.plan.fs < Rules::kRunsMinFilterSize) {
            // fs 7 / 8 (the quasi-periodic kernel's plans): small and medium calls of plans with source steps >= 2 only --
            // round3/runs_vs_auto.txt: 1.5x at 4 / 16 frames per call 148 -> 173 / 277 -> 283 Gpix/s, at 64 the frame-lane kernel
            // is ahead (437 against 407); DVD -> 1080p (72 phases) at 1 / 4 / 16 frames 16.7 -> 19.5 / 62 -> 74 / 126 -> 141, at 64
            // 274 against 194; 3x (source step 1) loses from 4 frames on (467 -> 350 at 16)
            if (t.runs.sx < 2 || t.runs.sy < 2 || call_samples < Rules::kQuasiMinSamples) return false;
            if (call_samples > (t.runs.px * t.runs.py > 16 ? Rules::kRunsSmallFsMaxSamplesManyPhases : Rules::kRunsSmallFsMaxSamples)) return false;
        }
        // tiny calls: the gather kernel's single launch is over before border + interior launches of this form are
        // (640 x 360 -> 960 x 540 with tap 4, one frame, 42e6 taps: 21.5 against 15.6 Gpix/s; four frames: 48 against 54;
        // with tap 8, one frame, 150e6 taps: 7.4 against 8.8)
        if (f.kernel_mode == 0 && static_cast<double>(t.plan.dst_w) * t.plan.dst_h * t.plan.fs * t.plan.fs * nframes < Rules::kRunsMinTaps)
            return false;
        const uint64_t plane_bytes = static_cast<uint64_t>(src_pitch[i]) * (t.plan.src_h - 1) + static_cast<uint64_t>(t.plan.src_w) * sb;
        return direct_fetch_is_safe(src_fs ? src_fs[i] : 0, nframes, plane_bytes, src_pitch[i], t.plan.fs);
    }
    // Frame-lane kernel (lanes = frames): the choice for batches whose plan has no phase structure for the other
    // interior kernels (they would run on the gather kernel with per-lane coefficient traffic); kernel_mode 11 forces
    // it for every plan and batch size.
    bool wants_framelane(const DeviceTable& t, int i) const {
        if (!t.use_framelane || f.kernel_mode == 1) return false;
        if (f.kernel_mode == 11 || f.kernel_mode == 16 || (f.kernel_mode == 12 && t.use_framelane_pair)) return true;
        if (f.kernel_mode != 0) return false;
        // batch sizes from which the frame-lane kernels (rate proportional to the filled lanes) pass the single-frame
        // kernels, measured (profiles/round2/fl_threshold.log): against the gather kernel 11 frames for 1.37x, 14 for the
        // 5/6 down-scale, 21 for 1.5x with tap 8 (row-segment form); against the quasi-periodic kernel 33 frames for DVD ->
        // 1080p (72 phases), 42 for 1.5x with tap 4, ~50 for 1.5x
        const bool sub_form = t.fl_whole.variant != 1 && jinc::framelane_sub_supported(t.plan.fs, 2, t.fl_whole.ty_shift);
        if (nframes < (t.plan.fs > 9 ? Rules::kFrameLaneMinFramesBigFs : sub_form ? Rules::kFlSubMinFrames : kFrameLaneMinFrames)) return false;
        if (wants_periodic(t)) return false;
        if (f.kernel_mode == 0 && wants_runs(t, i)) return false;
        if (wants_quasi(t)) {
            // whole groups of 128 frames: the frame-pair form is ahead of the quasi-periodic kernel on every plan measured
            // (256 frames: 1.5x 62 against 52 % of the VALU peak, 3x 69 against 68 %, 4/3x 60 against 54 %)
            if (t.use_framelane_pair && nframes >= jinc::kFrameLanePairFrames) return true;
            if (f.plans[f.table_of_plane(i)].periodic) return false;
            if (t.quasi.px * t.quasi.py > 16 && nframes >= Rules::kFrameLaneMinFramesManyPhases) return true;
            // (... where the batch fills its 64-frame groups: the frame-lane kernels' time goes by groups -- 1.5x at 64 / 96 / 120
            // frames: 442 / 341 / 398 Gpix/s against 395 / 390 / 393 on the quasi-periodic kernel with the border on the frame-lane kernel)
            const int groups64 = (nframes + 63) / 64;
            if (nframes * 10 < groups64 * 64 * 9) return false;
            return (t.plan.fs == 7 || t.plan.fs == 9) && t.quasi.sx >= 2 && t.quasi.sy >= 2 &&
                   nframes >= (t.plan.fs == 9 ? Rules::kFrameLaneMinFramesStep2Fs9 : Rules::kFrameLaneMinFramesStep2Fs7);
        }
        return !wants_direct(t, i);
    }
    // Sub-groups per wave for `n` < 64 frames on the frame-lane kernel (kernel_framelane_sub.hip: a wave's lanes are n frames x
    // several output rows instead of 64 frames), or 0: the 64-frame form.  Kernel mode 16 forces the form, FL_SUB is the
    // A/B knob (0: never, 2 / 4 / 8 / 16: that many sub-groups).
    int fl_subgroups(const DeviceTable& t, int n) const {
        const int forced = knobs::geti(JINC_KNOB_FL_SUB, -1);
        if (forced == 0 || t.fl_whole.variant == 1) return 0;
        if (forced < 0 && (n >= 64 || f.kernel_mode == 11)) return 0;  // (the knob: batches of any size and kernel mode, for A/B)
        int g = 0;
        if (forced > 0)
            g = forced;
        else if (f.kernel_mode == 16)
            g = n <= 4 ? 16 : n <= 8 ? 8 : n <= 16 ? 4 : 2;
        else if (f.kernel_mode == 0)
            g = n <= Rules::kFlSub16MaxFrames ? 16 : n <= Rules::kFlSub8MaxFrames ? 8 : n <= Rules::kFlSub4MaxFrames ? 4 : n <= Rules::kFlSub2MaxFrames ? 2 : 0;
        while (g >= 2 && !jinc::framelane_sub_supported(t.plan.fs, g, t.fl_whole.ty_shift)) g /= 2;  // (tiles of fewer rows than sub-groups)
        return g >= 2 ? g : 0;
    }
    // quad form of the periodic kernel (2x up-scales whose phases share their window origin), where it measured ahead
    // The periodic family on the trimmed support (integer planes whose phase sets have a zero rim; device_plan.cpp
    // trim_periodic): the automatic choice wherever it exists; kernel modes 5 / 6 (the fs-7 packed A/B variant) and 15
    // (= the automatic choice on the full window, for A/B and tests: jinc_filter::full_window) keep the reference's window.
    // Float planes take it frame by frame: the trimmed launch computes every frame and flags those in which it staged an infinity
    // or a NaN, a full-window launch behind it computes the flagged frames again (calls of at least kFloatTrimMinTaps taps per plane).
    // launches that fill the chip with the 128 x 24 tiles of the two-periods-per-lane quad form
    bool quad2_fills(const DeviceTable& t) const {
        return static_cast<long long>((t.periodic.ni + 127) / 128) * ((t.periodic.nj + 23) / 24) * nframes >= Rules::kQuad2MinWorkgroups;
    }
    bool trimmed(const DeviceTable& t) const {
        if (t.trim_fs <= 0 || f.full_window || f.kernel_mode == 5 || f.kernel_mode == 6) return false;
        // the 6-row x 7-column support exists for the quad2 form only: where that form is not what runs, the full window
        if (t.trim_nx != t.trim_fs && !(f.kernel_mode == 13 || (f.kernel_mode == 0 && quad2_fills(t)))) return false;
        // ... and the 8-row x 9-column one (chroma at tap 4) for ewa_periodic_quad2x8_kernel only
        if (t.trim_nx != t.trim_fs && t.trim_fs == 8 && !(knobs::flag(JINC_KNOB_QUAD8, Rules::kQuad8) && quad2x8_chosen(t, f.vi_in.component_size))) return false;
        if (!t.trim_needs_finite) return true;
        // (Float planes: the trimmed launch is its own finite-sample scan -- launch_plane -- so what is left of the price is a cleared
        // flag set, a scan of the plane's rim and a second launch that returns at once: from kFloatTrimMinTaps taps per plane and call.
        // With a scan PASS in front, round 4's first form, the 6 x 6 support gained nothing on float planes: C2's geometry on float
        // RGB 174.2 against 174.5 Gpix/s; without it 174 -> 207, round4/float_trim_ab.log.)
        // A/B knobs FLOAT_TRIM_MIN_TAPS (taps per plane and call from which float planes trim) and FLOAT_TRIM_MIN_FS (smallest
        // trimmed support float planes take by themselves)
        const double min_taps = knobs::get(JINC_KNOB_FLOAT_TRIM_MIN_TAPS, Rules::kFloatTrimMinTaps);
        const int min_fs = knobs::geti(JINC_KNOB_FLOAT_TRIM_MIN_FS, 0);
        if (f.kernel_mode == 0 && t.trim_fs < min_fs) return false;
        return static_cast<double>(t.plan.dst_w) * t.plan.dst_h * t.plan.fs * t.plan.fs * nframes >= min_taps || f.kernel_mode != 0;
    }
    const jinc::PeriodicArgs& periodic_args(const DeviceTable& t) const { return trimmed(t) ? t.periodic_trim : t.periodic; }
    int periodic_fs(const DeviceTable& t) const { return trimmed(t) ? t.trim_fs : t.plan.fs; }
    bool quad_chosen(const DeviceTable& t) const {
        if (!periodic_args(t).quad) return false;
        if (f.kernel_mode == 13) return true;
        if (f.kernel_mode != 0) return false;
        if (periodic_fs(t) == 6) return quad2_fills(t);  // two periods per lane on the 6 x 6 (6 x 7) support: tiles of 128 x 48 periods (24 rows on small calls)
        if (periodic_fs(t) == 8) {  // one period per lane on the 8 x 8 support (tap 4 at 2x)
            return knobs::flag(JINC_KNOB_QUAD8, Rules::kQuad8);  // A/B knob: 0 = window kernel, 1 = quad form
        }
        const long long wgs = static_cast<long long>((t.periodic.ni + 63) / 64) * ((t.periodic.nj + 8 * t.plan.fs - 1) / (8 * t.plan.fs)) * nframes;
        // (fs 7: large planes only -- on 1280 x 720 the border kernels beside the denser interior become the step's tail:
        // C1 at 256 frames 492 -> 465 Gpix/s)
        const long long periods = static_cast<long long>(t.periodic.ni) * t.periodic.nj;
        return t.plan.fs == 7 ? (wgs >= Rules::kHalfTileMaxWorkgroups && periods >= Rules::kQuadMinPeriods) : wgs < Rules::kQuad9MaxWorkgroups;
    }

    // 8 x 8 support on integer planes: two periods per lane (ewa_periodic_quad2x8_kernel) where the launch fills the chip with its
    // 128 x 32 tiles -- Jinc64 at 2x on 8-bit 483 -> 504 Gpix/s, 16-bit 4:2:0 253 -> 265; float planes (C4) are level and stay with
    // one period per lane (round4/quad2x8_ab.log).  (Given quad_chosen(t) and periodic_fs(t) == 8.)
    bool quad2x8_chosen(const DeviceTable& t, int sample_bytes) const {
        const int two = knobs::geti(JINC_KNOB_QUAD2X8, -1);  // A/B knob: 0 / 1
        const long long wgs2 = static_cast<long long>((t.periodic.ni + 127) / 128) * ((t.periodic.nj + 31) / 32) * nframes;
        return two >= 0 ? two != 0 : (sample_bytes < 4 && wgs2 >= Rules::kQuad2x8MinWorkgroups);
    }

    // does any plane launch border kernels beside an interior kernel?
    bool any_border_frame() const {
        bool any = false;
        for (int i = 0; i < f.planecount; ++i) {
            const DeviceTable& t = f.tables[f.table_of_plane(i)];
            any |= f.simd_order == 0 && !wants_framelane(t, i) && (wants_periodic(t) || wants_quasi(t) || wants_direct(t, i) || wants_runs(t, i));
        }
        return any;
    }
    // A/B on MI355X with the strip border kernels: overlapping wins 11 % on C3 (fs 17), 3 % on C4 (fs 9) and 2 % on
    // C2 (fs 7) -- three small border launches per plane would otherwise sit serially in front of the interior.
    // ... but not for calls so small that the fork / join through events costs more than the border kernels in front of
    // the interior (measured, one frame per call: C2 114 against 103 Gpix/s without the side stream, 4K->1080p 27 against
    // 23; from ~1e9 taps per call the side stream wins: C2 at 4 frames 319 against 287, C3 at 1 frame 24 against 20).
    // -1: this automatic rule, 1: always, 0: never.
    bool wants_border_overlap() const {
        if (f.overlap_border >= 0) return f.overlap_border != 0;
        double taps = 0.0, border_taps = 0.0;
        for (int i = 0; i < f.planecount; ++i) {
            const DeviceTable& t = f.tables[f.table_of_plane(i)];
            taps += static_cast<double>(t.plan.dst_w) * t.plan.dst_h * t.plan.fs * t.plan.fs;
            if (wants_runs(t, i))
                for (int r = 0; r < t.border_rects.n; ++r)
                    border_taps += static_cast<double>(t.border_rects.w[r]) * t.border_rects.h[r] * t.plan.fs * t.plan.fs;
        }
        return taps * nframes >= Rules::kOverlapMinTaps || border_taps * nframes >= Rules::kOverlapMinBorderTaps;
    }
    // Border frame of exactly periodic plans: three strip launches per plane (corners, rows, columns) or ONE launch of the gather
    // kernel over the frame's four rectangles.  The strips are the leaner kernels (C3 at 32 frames: +11 %), but below ~5e9 taps per
    // call their launches cost more than they save (round3/border_strips_ab.txt, one frame per call: C2 122 -> 183 Gpix/s, C1 15.7 ->
    // 25.4, 1080p -> 4K 4:2:0 49 -> 74, 4K -> 1080p 27.7 -> 45, C3 24 -> 31; four frames: C2 346 -> 412, C1 59 -> 93; level from 3e9 ..
    // 7e9 taps on; tap 16 at 9e9 taps: -30 %).  f.border_strips: -1 this rule, 1 / 2 / 3 / 4 / 0 forced (tests, A/B; 3 = ewa_strip_kernel where configured, 4 = 3 with the columns in the interior kernel's edge tiles where configured).
    bool wants_border_strips() const {
        if (f.border_strips >= 0) return f.border_strips != 0;
        double taps = 0.0;
        for (int i = 0; i < f.planecount; ++i) {
            const DeviceTable& t = f.tables[f.table_of_plane(i)];
            taps += static_cast<double>(t.plan.dst_w) * t.plan.dst_h * t.plan.fs * t.plan.fs;
        }
        if (taps * nframes >= Rules::kStripBorderMinTaps) return true;
        // Single-plane calls whose border columns the interior kernel computes in its edge tiles (round 5): what is left of the strip
        // border is two small launches (rows, corners) against the gather launch over the whole frame -- ahead from half the size
        // (round5/small_call_border_ab.log: C2 at 6 / 8 / 12 frames per call +3 / +13 / +11 %, C1 at 64 / 96 +3.6 / +8.6 %; C2 at 4 and
        // C1 at 32 level / behind; calls with chroma planes -- three times the launches -- stay behind up to the limit above).
        if (f.planecount == 1 && f.border_strips < 0) {
            const DeviceTable& t = f.tables[f.table_of_plane(0)];
            const bool edge_form = t.use_edge_cols && t.strips_ok && (f.kernel_mode == 0 || f.kernel_mode == 13) && trimmed(t) && quad_chosen(t) &&
                                   (periodic_fs(t) == 6 || (periodic_fs(t) == 8 && quad2x8_chosen(t, f.vi_in.component_size))) &&
                                   knobs::flag(JINC_KNOB_EDGE_COLS, true) && knobs::geti(JINC_KNOB_ROWPAIR_SMALL, 0) != 1;
            if (edge_form) return taps * nframes >= Rules::kStripBorderMinTapsEdgeCols;
        }
        return false;
    }
    // Small calls with chroma planes (no border fork): the planes behind the first go to the side stream, interior and border, so
    // that luma and chroma run beside each other -- a single frame's planes fill the chip even less one by one.
    // Measured, one frame per call (round3/plane_fork_ab.txt): 1080p -> 4K 4:2:0 44.1 -> 49.6 Gpix/s, 4K -> 1080p 4:2:0 16-bit 8.7 -> 9.9,
    // DVD -> 1080p with tap 6 13.2 -> 15.6, four DVD frames with tap 3 57.7 -> 61.8; level where the first plane fills the chip
    // alone (four 4K frames) and -2 % on 8K float RGB planes, hence the limit on the first plane's samples.
    bool wants_plane_fork(bool border_fork) const {
        const jinc::DevicePlan& first = f.tables[f.table_of_plane(0)].plan;
        return !border_fork && f.planecount >= 2 && f.simd_order == 0 && plane_fork_enabled() &&
               static_cast<double>(first.dst_w) * first.dst_h * nframes <= Rules::kPlaneForkMaxSamples;
    }
};
}  // namespace

namespace {
// The launches of plane i: interior kernel on `plane_stream`, border kernels on `border_stream` (the same stream unless forked).
// `pair`: plane i + 1 rides along as a second frame (single-frame calls only).
void launch_plane(jinc_filter& f, const Choice& c, int i, const void* const src[4], const int src_pitch[4], const size_t src_fs[4],
                  void* const dst[4], const int dst_pitch[4], const size_t dst_fs[4], hipStream_t plane_stream, hipStream_t border_stream,
                  const PlanePair* pair) {
    const int sb = c.sb, nframes = c.nframes;
    DeviceTable& t = f.tables[f.table_of_plane(i)];
    jinc::PlaneIO io;
    io.src = src[i];
    io.dst = dst[i];
    io.src_pitch = src_pitch[i];
    io.dst_pitch = dst_pitch[i];
    io.src_frame_stride = src_fs ? src_fs[i] : 0;
    io.dst_frame_stride = dst_fs ? dst_fs[i] : 0;
    io.nframes = nframes;
    if (pair) io.src_frame_stride = pair->src_stride, io.dst_frame_stride = pair->dst_stride, io.nframes = 2;
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
        t.last_instance = t.last_kernel;
        timed(f.ev_gather, plane_stream, "SIMD-order kernel launch",
              [&](hipStream_t s) { return jinc::launch_simd_order(t.plan, io, f.simd_order, min_val, s); });
        return;
    }
    if (c.wants_framelane(t, i)) {
        auto aligned_to = [&](uintptr_t bytes) {
            return reinterpret_cast<uintptr_t>(dst[i]) % bytes == 0 && static_cast<uintptr_t>(dst_pitch[i]) % bytes == 0 &&
                   (io.nframes <= 1 || io.dst_frame_stride % bytes == 0);  // io.nframes: 2 when two planes travel as one call
        };
        // bit 0: packed stores of 4 samples, bit 1: 16-byte stores (8-bit planes in the frame-pair form)
        const int vec_ok = (aligned_to(static_cast<uintptr_t>(4 * sb)) ? 1 : 0) | (aligned_to(16) ? 2 : 0);
        // Whole groups of 128 frames go to the frame-pair form (two frames per lane: half the per-pixel coefficient
        // traffic and scalar work, packed multiplies / adds; measured at 256 frames against the 64-frame form: 1.37x
        // 55 against 42 % of the VALU peak, 1.5x 62 against 54 %, DVD -> 1080p 63 against 49 %, 15 / 8 61 against 52 %);
        // what is left of the batch, and every batch below 128 frames, to the 64-frame form.  kernel_mode 12 (tests,
        // A/B): the frame-pair form for the whole batch, 11: the 64-frame form for the whole batch.
        int npair = 0;
        if (t.use_framelane_pair && f.kernel_mode != 11 && f.kernel_mode != 16)
            npair = f.kernel_mode == 12 ? nframes : nframes / jinc::kFrameLanePairFrames * jinc::kFrameLanePairFrames;
        if (npair > 0) {
            jinc::FrameLaneArgs fa = t.fl_pair;
            fa.io = io;
            fa.io.nframes = npair;
            fa.vec_store_ok = vec_ok;
            t.last_kernel = "ewa_framelane_pair_kernel";
            timed(f.ev_periodic, plane_stream, "frame-pair kernel launch", [&](hipStream_t s) { return jinc::launch_framelane_pair(fa, s); });
        }
        // What is left (fewer than 128 frames): whole groups of 64 on the 64-frame form; a remainder of up to 32 frames on the
        // sub-group form (a wave = 4 / 8 / 16 / 32 frames x 16 / 8 / 4 / 2 output rows), which a batch of fewer than 64 frames
        // is as a whole -- the rate of the 64-frame form is proportional to its filled lanes (1.37x, 16 frames: 94 -> 210 Gpix/s).
        auto launch_part = [&](int first, int n, bool names_the_call) {
            jinc::FrameLaneArgs fa = t.fl_whole;
            fa.io = io;
            fa.io.src = static_cast<const char*>(io.src) + static_cast<size_t>(first) * io.src_frame_stride;
            fa.io.dst = static_cast<char*>(io.dst) + static_cast<size_t>(first) * io.dst_frame_stride;
            fa.io.nframes = n;
            fa.vec_store_ok = vec_ok;
            fa.subgroups = c.fl_subgroups(t, n);
            if (fa.subgroups) {
                if (names_the_call) t.last_kernel = "ewa_framelane_sub_kernel";
                timed(f.ev_periodic, plane_stream, "frame-lane sub-group kernel launch", [&](hipStream_t s) { return jinc::launch_framelane_sub(fa, s); });
                return;
            }
            if (names_the_call)
                t.last_kernel = (fa.threads == 1024 && t.plan.fs == 7) ? "ewa_framelane_win1k_kernel"
                                : (fa.variant != 1 && (t.plan.fs == 5 || t.plan.fs == 7 || t.plan.fs == 8 || t.plan.fs == 9))
                                    ? "ewa_framelane_win_kernel" : "ewa_framelane_kernel";
            timed(f.ev_periodic, plane_stream, "frame-lane kernel launch", [&](hipStream_t s) { return jinc::launch_framelane(fa, s); });
        };
        const int rest = nframes - npair;
        if (rest >= 64 && rest % 64 != 0 && c.fl_subgroups(t, rest % 64)) {
            launch_part(npair, rest / 64 * 64, npair == 0);
            launch_part(npair + rest / 64 * 64, rest % 64, false);
        } else if (rest > 0) {
            launch_part(npair, rest, npair == 0);
        }
        t.last_instance = t.last_kernel;
        return;
    }
    // Border rectangles on the frame-lane kernel with fewer than 64 frames: sub-groups per wave of its sub-group form (0: the
    // 64-frame form -- more than 32 frames, or a filter size / tile the form does not take).
    auto border_subgroups = [&](const jinc::FrameLaneArgs& fa, int n) {
        int g = n <= Rules::kFlSub16MaxFrames ? 16 : n <= Rules::kFlSub8MaxFrames ? 8 : n <= Rules::kFlSub4MaxFrames ? 4 : n <= Rules::kFlSub2MaxFrames ? 2 : 0;
        while (g >= 2 && !jinc::framelane_sub_supported(t.plan.fs, g, fa.ty_shift)) g /= 2;
        return g >= 2 && fa.variant != 1 ? g : 0;
    };
    // Border frame of a drifting plan (every border pixel owns a coefficient set): the gather kernel, or in batches the frame-lane
    // kernel (lanes = frames make the private sets scalar loads).
    auto drifting_border = [&]() {
        if (t.use_fl_border && t.border_rects.n > 0 && nframes >= runs_fl_border_min_frames(border_subgroups(t.fl_border, 32) != 0)) {
            auto aligned_to = [&](uintptr_t bytes) {
                return reinterpret_cast<uintptr_t>(dst[i]) % bytes == 0 && static_cast<uintptr_t>(dst_pitch[i]) % bytes == 0 &&
                       (io.nframes <= 1 || io.dst_frame_stride % bytes == 0);  // io.nframes: 2 when two planes travel as one call
            };
            jinc::FrameLaneArgs fa = t.fl_border;
            fa.io = io;
            fa.vec_store_ok = (aligned_to(static_cast<uintptr_t>(4 * sb)) ? 1 : 0) | (aligned_to(16) ? 2 : 0);
            fa.subgroups = border_subgroups(fa, nframes);
            if (fa.subgroups)
                timed(f.ev_gather, border_stream, "border frame-lane kernel launch (sub-groups)",
                      [&](hipStream_t s) { return jinc::launch_framelane_sub(fa, s); });
            else
                timed(f.ev_gather, border_stream, "border frame-lane kernel launch", [&](hipStream_t s) { return jinc::launch_framelane(fa, s); });
        } else if (t.border_rects.n > 0) {
            timed(f.ev_gather, border_stream, "border kernel launch",
                  [&](hipStream_t s) { return jinc::launch_gather(t.plan, io, t.border_rects, s); });
        }
    };
    if (c.wants_runs(t, i)) {
        t.last_kernel = "ewa_direct_runs_kernel";
        t.last_instance = t.last_kernel;
        drifting_border();
        timed(f.ev_periodic, plane_stream, "direct runs kernel launch", [&](hipStream_t s) {
            jinc::DirectArgs da = t.runs;
            da.src_bytes = direct_src_bytes(
                src[i], static_cast<uint64_t>(src_pitch[i]) * (t.plan.src_h - 1) + static_cast<uint64_t>(t.plan.src_w) * sb);
            return jinc::launch_direct_runs(da, io, s);
        });
        return;
    }
    const bool direct = c.wants_direct(t, i);
    const bool quasi = !direct && c.wants_quasi(t);
    const bool periodic = !direct && !quasi && c.wants_periodic(t);
    t.last_kernel = direct     ? "ewa_direct_kernel"
                    : quasi    ? "ewa_quasi_kernel"
                    : periodic ? (c.quad_chosen(t) ? (c.periodic_fs(t) == 6 ? "ewa_periodic_quad2_kernel" : c.periodic_fs(t) == 8 ? "ewa_periodic_quad8_kernel" : "ewa_periodic_quad_kernel")
                                  : (f.kernel_mode == 5 || f.kernel_mode == 6) && t.plan.fs == 7 ? "ewa_periodic_pk_kernel"
                                  : (f.kernel_mode == 3 || c.periodic_fs(t) < 6 || c.periodic_fs(t) > 9) ? "ewa_periodic_rows_kernel"
                                                                                                          : "ewa_periodic_kernel")
                               : "ewa_gather_kernel";
    t.last_instance = t.last_kernel;
    // the launcher's own word on what it launched (the periodic family picks instantiations by tile height, chord pattern, ...)
    auto take_note = [&]() {
        const char* kernel = nullptr;
        const char* inst = knobs::take_instance(&kernel);
        if (kernel) t.last_kernel = kernel, t.last_instance = inst;
    };
    // Round 5: the border columns inside the interior kernel's edge tiles (ewa_periodic_quad2_kernel / ..quad2x8.. on integer planes,
    // device_plan.cpp plan_edge_columns) wherever that kernel is what runs and the border form is not forced.  Knob EDGE_COLS = 0: the
    // border kernels as before.
    bool edge_fused = false;
    if (direct || periodic || quasi) {
        // border frame: rows on kernel_direct.hip + columns on the gather kernel, or the gather kernel for all of it
        const bool strips = c.wants_border_strips() && t.strips_ok && c.direct_ok(t, i);
        edge_fused = strips && periodic && t.use_edge_cols && (f.border_strips < 0 || f.border_strips == 4) && (f.kernel_mode == 0 || f.kernel_mode == 13) && c.trimmed(t) &&
                     c.quad_chosen(t) && (c.periodic_fs(t) == 6 || (c.periodic_fs(t) == 8 && c.quad2x8_chosen(t, sb))) &&
                     knobs::flag(JINC_KNOB_EDGE_COLS, true) && knobs::geti(JINC_KNOB_ROWPAIR_SMALL, 0) != 1;  // (that knob sends the launch to the row-pair kernel)
        if (strips) {
            jinc::DirectArgs rs = t.row_strips;
            rs.src_bytes = direct_src_bytes(
                src[i], static_cast<uint64_t>(src_pitch[i]) * (t.plan.src_h - 1) + static_cast<uint64_t>(t.plan.src_w) * sb);
            // In batches the border columns (full height: the corners with them) go to the frame-lane kernel: its lanes are 64
            // frames, so a border pixel's private coefficient set is a scalar load and the taps run from registers -- the
            // column-strip kernel reads LDS once per tap (C2 at 1024 frames: knob FL_COLS_FRAMES A/B, round4/fl_cols_ab.log).
            const int fl_cols_min_frames = knobs::geti(JINC_KNOB_FL_COLS_FRAMES, Rules::kFlColsMinFrames);  // A/B knob: 0 = never
            // Round 5: kernel_strip.hip takes the rows and the columns of filter sizes up to 9 (knob STRIP_LDS = 0: the kernels
            // below as before; kernel mode 3 and a forced border form keep them too, for the tests that compare the forms).  The
            // corners then stay with the gather kernel.
            // Rows: always (C2 at 1024 frames per call: 0.134 against 0.184 ms on ewa_direct_kernel's row strips).  Columns: below the
            // batch size from which the frame-lane kernel takes them (it stays ahead there: 0.283 ms, corners included, against
            // 0.300 + corners) -- i.e. instead of ewa_colstrip_kernel.  Knob STRIP_LDS: 1 = this rule, 2 = rows and columns always.
            const int lds_knob = knobs::geti(JINC_KNOB_STRIP_LDS, 1);
            const bool lds_strips = f.border_strips >= 3 || (lds_knob != 0 && f.border_strips < 0 && f.kernel_mode != 3);
            const bool strip_rows = lds_strips && t.use_strip_rows;
            // Columns on packed column pairs (ewa_colpair_kernel) wherever configured (odd filter sizes 7 .. 17 at source step 1): C3
            // 0.24 -> 0.08 ms per step against ewa_colstrip_kernel (+3.7 %), tap 6 at 2x +2.9 %, two 8K float frames per call +16 %,
            // level elsewhere (round5/colpair_ab.log).  Knob COLPAIR: 0 never, 1 (default) this rule, 3 filter sizes from 11 on only.
            const int colpair_knob = knobs::geti(JINC_KNOB_COLPAIR, 1);
            const bool pair_cols = !edge_fused && t.use_colpair && (f.border_strips < 0 || f.border_strips == 4) && f.kernel_mode != 3 &&
                                   (colpair_knob == 1 || colpair_knob == 2 || (colpair_knob == 3 && t.plan.fs >= 11));
            const bool strip_cols = !edge_fused && !pair_cols && lds_strips && t.use_strip_cols &&
                                    (f.border_strips >= 3 || lds_knob == 2 || !(t.use_fl_cols && fl_cols_min_frames > 0 && nframes >= fl_cols_min_frames));
            const bool fl_cols = !edge_fused && !pair_cols && !strip_cols && t.use_fl_cols && f.border_strips != 2 && fl_cols_min_frames > 0 && nframes >= fl_cols_min_frames &&
                                 (f.kernel_mode == 0 || f.kernel_mode == 13 || f.kernel_mode == 2);
            if (fl_cols) {
                auto aligned_to = [&](uintptr_t bytes) {
                    return reinterpret_cast<uintptr_t>(dst[i]) % bytes == 0 && static_cast<uintptr_t>(dst_pitch[i]) % bytes == 0 &&
                           (io.nframes <= 1 || io.dst_frame_stride % bytes == 0);
                };
                jinc::FrameLaneArgs fa = t.fl_cols;
                fa.io = io;
                fa.vec_store_ok = (aligned_to(static_cast<uintptr_t>(4 * sb)) ? 1 : 0) | (aligned_to(16) ? 2 : 0);
                fa.subgroups = border_subgroups(fa, nframes);  // (fewer than 64 frames: the sub-group form, as for whole planes)
                if (fa.subgroups)
                    timed(f.ev_gather, border_stream, "border column frame-lane kernel launch (sub-groups)",
                          [&](hipStream_t s) { return jinc::launch_framelane_sub(fa, s); });
                else
                    timed(f.ev_gather, border_stream, "border column frame-lane kernel launch", [&](hipStream_t s) { return jinc::launch_framelane(fa, s); });
            }
            // Rows of filter sizes 11 .. 17 at 2x: launches of ewa_periodic_rowpair_kernel (plan_rowpair_rows).  Knob ROWPAIR_ROWS = 0:
            // ewa_direct_kernel's row strips.
            const bool pair_rows = !strip_rows && !t.rowpair_rows.empty() && (f.border_strips < 0 || f.border_strips == 4) && f.kernel_mode != 3 &&
                                   knobs::flag(JINC_KNOB_ROWPAIR_ROWS, true);
            t.last_border = (strip_rows ? 16 : pair_rows ? 128 : 2) | (edge_fused ? 64 : pair_cols ? 256 : strip_cols ? 32 : fl_cols ? 8 : (t.use_colstrip && f.border_strips != 2) ? 4 : 1);
            if (strip_cols || strip_rows || edge_fused || pair_cols) {
                if (t.corner_rects.n > 0 && (strip_cols || edge_fused || pair_cols))
                    timed(f.ev_gather, border_stream, "corner kernel launch", [&](hipStream_t s) { return jinc::launch_gather(t.plan, io, t.corner_rects, s); });
                if (strip_rows)
                    timed(f.ev_gather, border_stream, "border row strip launch", [&](hipStream_t s) { return jinc::launch_strip(t.strip_rows, io, s); });
                if (strip_cols)
                    timed(f.ev_gather, border_stream, "border column strip launch", [&](hipStream_t s) { return jinc::launch_strip(t.strip_cols, io, s); });
                if (pair_cols)
                    timed(f.ev_gather, border_stream, "border column pair launch", [&](hipStream_t s) { return jinc::launch_colpair(t.colpair, io, s); });
            }
            const bool colstrip = !edge_fused && !pair_cols && !strip_cols && !fl_cols && t.use_colstrip && f.border_strips != 2;
            // the corner kernel first: few workgroups with long latency-bound chains (per-lane coefficients); queued
            // last it would start when the interior kernel already holds every wave slot.  (Measured again in round 3 with
            // the corners last: no difference on any of eight configurations -- in a long batch the border kernels cost their
            // stand-alone time whatever the order: C2 at 1024 frames 0.19 + 0.37 + 0.05 ms of 13.9; round3/corner_order_ab.txt)
            if (colstrip && t.corner_rects.n > 0)
                timed(f.ev_gather, border_stream, "corner kernel launch",
                      [&](hipStream_t s) { return jinc::launch_gather(t.plan, io, t.corner_rects, s); });
            if (pair_rows) {
                for (const jinc::PeriodicArgs& ra : t.rowpair_rows)
                    timed(f.ev_gather, border_stream, "border row pair launch", [&](hipStream_t s) {
                        const int rc = jinc::launch_rowpair(ra, io, s);
                        (void)knobs::take_instance();  // (the border's launch does not name the call)
                        return rc;
                    });
            } else if (!strip_rows)
                timed(f.ev_gather, border_stream, "border row kernel launch",
                      [&](hipStream_t s) { return jinc::launch_direct_row_strips(rs, io, s); });
            if (colstrip) {
                timed(f.ev_gather, border_stream, "border column kernel launch",
                      [&](hipStream_t s) { return jinc::launch_colstrip(t.col_strips, io, s); });
            } else if (!edge_fused && !pair_cols && !fl_cols && !strip_cols && t.column_rects.n > 0) {
                timed(f.ev_gather, border_stream, "border column kernel launch",
                      [&](hipStream_t s) { return jinc::launch_gather(t.plan, io, t.column_rects, s); });
            }
        } else {
            t.last_border = 1;
            drifting_border();  // (use_fl_border is set for drifting plans only: the others keep the gather kernel here)
        }
        if (direct)
            timed(f.ev_periodic, plane_stream, "direct periodic kernel launch", [&](hipStream_t s) {
                jinc::DirectArgs da = (t.direct_trim_fs > 0 && !f.full_window) ? t.direct_trim : t.direct;
                da.src_bytes = direct_src_bytes(
                    src[i], static_cast<uint64_t>(src_pitch[i]) * (t.plan.src_h - 1) + static_cast<uint64_t>(t.plan.src_w) * sb);
                return jinc::launch_direct(da, io, s);
            });
        else if (quasi)
            timed(f.ev_periodic, plane_stream, "quasi-periodic kernel launch", [&](hipStream_t s) {
                jinc::QuasiArgs qa = t.quasi;
                if (f.kernel_mode == 8) qa.exact = 0;   // A/B: per-row lookup + waterfall over sets in SGPRs
                if (f.kernel_mode == 10) qa.exact = 2;  // A/B: per-row lookup + per-lane coefficient registers
                {   // Calls that do not fill the chip split a tile's phases over several workgroups (each stages the tile
                    // again): one DVD -> 1080p frame is 72 + 2 x 20 workgroups of 18 phases per wave, 0.23 ms whether the
                    // call holds one frame or four.  Aim: ~1024 workgroups per launch (fs 9, whose tiles cost more to
                    // stage: ~400).  Measured, one frame per call: DVD -> 1080p 8.8 -> 21 Gpix/s, 3x 102 -> 156, 1.5x 49 -> 61,
                    // 1.5x with tap 4 38 -> 45; equal from 4 .. 16 frames per call on.
                    const int tile_rows = qa.rg * t.plan.fs;
                    const long long wgs = static_cast<long long>((qa.ni + 63) / 64) * ((qa.nj + tile_rows - 1) / tile_rows) * nframes;
                    const int per_wave = (qa.px * qa.py + qa.nwaves - 1) / std::max(1, qa.nwaves);
                    const long long target = t.plan.fs == 9 ? Rules::kQuasiSplitTargetFs9 : Rules::kQuasiSplitTarget;
                    int split = wgs > 0 ? static_cast<int>((target + wgs - 1) / wgs) : 1;
                    qa.phase_split = std::max(1, std::min(split, per_wave));
                    if (quasi_split_knob() >= 0) qa.phase_split = std::max(1, std::min(quasi_split_knob(), per_wave));
                }
                return jinc::launch_quasi(qa, t.plan.fs, io, s);
            });
        else
            timed(f.ev_periodic, plane_stream, "periodic kernel launch", [&](hipStream_t s) {
                int variant = (f.kernel_mode >= 3 && f.kernel_mode <= 6) ? f.kernel_mode - 2 : 0;
                // quad form (2x up-scales whose phases share their window origin: a lane computes a period's 2 x 2 pixels from
                // one window on packed multiplies / adds), chosen where it measured ahead (profiles/round3/quad_ab.log):
                // fs 7 on calls that fill the chip (C2 at 16 / 1024 frames 544 -> 560 / 591 -> 606 Gpix/s, 16-bit 4:2:0 356 ->
                // 365, float RGB 165 -> 175; a 4-frame call loses 5 %), fs 9 on calls that do not (C4, one frame per call: 80
                // -> 94 Gpix/s; 4 / 16 frames: equal).  Kernel mode 13 forces it, 2 excludes it.
                const bool quad = c.quad_chosen(t);
                // Small calls (single frames, short batches) take the window kernels' half-height tiles: twice the
                // workgroups for a launch that does not fill the chip (C2, one frame: 600 workgroups on 1536 slots,
                // kernel 27.2 -> 22.3 us; 4 frames: 323 -> 354 Gpix/s); long batches keep the full tiles (+2 %).
                const int pfs = c.periodic_fs(t);
                if (!quad && f.kernel_mode == 0 && pfs >= 6 && pfs <= 9) {
                    const int rows = pfs * (pfs <= 7 ? 8 : 9);  // period-rows of a full tile
                    const long long wgs = static_cast<long long>((t.periodic.ni + 63) / 64) * ((t.periodic.nj + rows - 1) / rows) * nframes;
                    if (wgs < Rules::kHalfTileMaxWorkgroups) variant = 2;
                }
                if (quad) {
                    const int cols = pfs == 6 ? 128 : 64;  // periods per tile row
                    const long long quad_wgs = static_cast<long long>((t.periodic.ni + cols - 1) / cols) * ((t.periodic.nj + 8 * pfs - 1) / (8 * pfs)) * nframes;
                    // (the quad forms' tile height follows from their own workgroup count alone -- the window kernels'
                    // half-tile rule above once left variant 2 standing under a chosen quad form)
                    variant = quad_wgs < (pfs == 6 ? Rules::kQuad2HalfTileMaxWorkgroups : Rules::kHalfTileMaxWorkgroups) ? 6 : 5;
                    const int force_rg = knobs::geti(JINC_KNOB_QUAD_RG, 0);  // A/B knob: 8 / 4 forces full / half-height tiles of the quad forms
                    if (force_rg == 8) variant = 5;
                    if (force_rg == 4) variant = 6;
                    if (pfs == 8 && c.quad2x8_chosen(t, sb)) variant = 7;  // two periods per lane
                }
                if (c.trimmed(t) && t.trim_needs_finite) {
                    // float plane: which frames hold nothing but finite samples?  Those run on the trimmed support; the others
                    // (flag 1) on the reference's full window in a second launch that returns at once for the rest.
                    // (a ring of flag sets, one per call in turn like the fork / join events: calls queued on different streams
                    // may overlap on the device, and a set must not be cleared under a launch that still reads it)
                    constexpr int kFlagSets = jinc_filter::kForkEvents;
                    if (f.finite_flags_frames < io.nframes) {
                        if (f.finite_flags) {
                            hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize(finite flags)");  // launches of earlier calls may still read the old sets
                            (void)hipFree(f.finite_flags);
                        }
                        f.finite_flags = nullptr, f.finite_flags_frames = 0;
                        hip_check(hipMalloc(reinterpret_cast<void**>(&f.finite_flags), sizeof(uint32_t) * kFlagSets * 4 * static_cast<size_t>(io.nframes)),
                                  "hipMalloc(finite flags)");
                        f.finite_flags_frames = io.nframes;
                    }
                    uint32_t* flags = f.finite_flags + (static_cast<size_t>(f.finite_flags_turn % kFlagSets) * 4 + static_cast<size_t>(i)) * f.finite_flags_frames;
                    hip_check(hipMemsetAsync(flags, 0, sizeof(uint32_t) * io.nframes, s), "hipMemsetAsync(finite flags)");
                    jinc::PeriodicArgs fin = t.periodic_trim, rest = t.periodic;
                    fin.frame_flags = rest.frame_flags = flags;
                    int rc = 0;
                    // A/B knob FLOAT_SCAN = 1: a scan pass over the whole source in front (round 4's first form)
                    const bool scan_pass = knobs::flag(JINC_KNOB_FLOAT_SCAN, false);
                    if (scan_pass) {
                        rc = jinc::launch_finite_scan(io, t.plan.src_w, t.plan.src_h, flags, s);
                        fin.run_when = 0, rest.run_when = 1;
                    } else {
                        // The trimmed launch is its own scan: it computes every frame and flags those in whose tiles it stages an
                        // infinity or a NaN (a compare per staged sample instead of a pass over the source: C4 218 us of 1.25 ms).  Its
                        // tiles stage at least the source rectangle below; what lies outside it -- the rim only the reference's
                        // zero-coefficient taps and the border pixels reach -- gets the little scan.
                        const int n = pfs;
                        rc = jinc::launch_finite_scan_outside(io, t.plan.src_w, t.plan.src_h, fin.min_sx, fin.min_sy, fin.min_sx + fin.ni + n - 1,
                                                              fin.min_sy + fin.nj + n - 1, flags, s);
                        fin.run_when = jinc::PeriodicArgs::kRunAllAndFlag, rest.run_when = 1;
                    }
                    if (rc) return rc;
                    rc = jinc::launch_periodic(fin, pfs, io, s, variant);
                    take_note();
                    if (rc) return rc;
                    rc = jinc::launch_periodic(rest, t.plan.fs, io, s, 0);
                    (void)knobs::take_instance();  // (the flagged frames' full-window launch does not name the call)
                    return rc;
                }
                jinc::PeriodicArgs pa = c.periodic_args(t);
                if (edge_fused) pa.edge = t.edge_cols;  // (the launch is a two-periods-per-lane quad form: edge_fused says so)
                const int rc = jinc::launch_periodic(pa, pfs, io, s, variant);
                take_note();
                return rc;
            });
    } else {
        timed(f.ev_gather, plane_stream, "gather kernel launch",
              [&](hipStream_t s) { return jinc::launch_gather(t.plan, io, t.whole, s); });
    }
}

}  // namespace

namespace {
void enqueue_run(jinc_filter& f, const void* const src[4], const int src_pitch[4], const size_t src_fs[4], void* const dst[4],
                 const int dst_pitch[4], const size_t dst_fs[4], int nframes, hipStream_t stream, bool may_split) {
    const Choice c(f, src_pitch, src_fs, nframes);
    // A batch of 128 k + r frames whose whole groups of 128 go to the frame-pair form: the r frames left over are a call of
    // their own, chosen by the same rules with their own frame count (as a 64-frame frame-lane launch a
    // remainder of 1..15 frames cost as much as 64 frames; 129 frames took ~1.5 x the time of 128).
    if (may_split && f.kernel_mode == 0 && f.simd_order == 0 && nframes > jinc::kFrameLanePairFrames && nframes % jinc::kFrameLanePairFrames != 0) {
        bool pair_somewhere = false;
        for (int i = 0; i < f.planecount; ++i) {
            const DeviceTable& t = f.tables[f.table_of_plane(i)];
            pair_somewhere |= t.use_framelane_pair && c.wants_framelane(t, i);
        }
        if (pair_somewhere) {
            const int whole = nframes / jinc::kFrameLanePairFrames * jinc::kFrameLanePairFrames;
            const void* s2[4] = {nullptr, nullptr, nullptr, nullptr};
            void* d2[4] = {nullptr, nullptr, nullptr, nullptr};
            for (int i = 0; i < f.planecount; ++i) {
                s2[i] = static_cast<const char*>(src[i]) + static_cast<size_t>(whole) * src_fs[i];
                d2[i] = static_cast<char*>(dst[i]) + static_cast<size_t>(whole) * dst_fs[i];
            }
            enqueue_run(f, src, src_pitch, src_fs, dst, dst_pitch, dst_fs, whole, stream, false);
            enqueue_run(f, s2, src_pitch, src_fs, d2, dst_pitch, dst_fs, nframes - whole, stream, false);
            return;
        }
    }
    const bool fork = c.any_border_frame() && c.wants_border_overlap();
    const bool plane_fork = c.wants_plane_fork(fork);
    const unsigned turn = f.fork_turn % jinc_filter::kForkEvents;
    ++f.finite_flags_turn;  // (float planes on the trimmed support: a flag set of its own per call)
    if (fork || plane_fork) {  // side-stream work may start once everything already queued on `stream` is done
        ++f.fork_turn;
        hip_check(hipEventRecord(f.ev_fork[turn], stream), "hipEventRecord(fork)");
        hip_check(hipStreamWaitEvent(f.aux_stream, f.ev_fork[turn], 0), "hipStreamWaitEvent(fork)");
    }
    for (int i = 0; i < f.planecount; ++i) {
        hipStream_t plane_stream = (plane_fork && i >= 1) ? f.aux_stream : stream;
        // One frame per call: a plane and its successor with the same table, pitches and spacing in source and destination (U and
        // V of a frame) go out as ONE two-frame launch per kernel -- two launches fewer per 4:2:0 frame (round3/plane_pair_ab.txt:
        // 1080p -> 4K 4:2:0 74 -> 110 Gpix/s, DVD -> 1080p 20.5 -> 31, 4K -> 1080p 4:2:0 16-bit 15.1 -> 24.2, C3 30 -> 37).
        PlanePair pair;
        const bool paired = nframes == 1 && f.simd_order == 0 && i + 1 < f.planecount && f.table_of_plane(i) == f.table_of_plane(i + 1) &&
                            !c.wants_framelane(f.tables[f.table_of_plane(i)], i) &&  // (forced on a single frame: its launches count frames)
                            plane_pair(src, src_pitch, dst, dst_pitch, i, c.sb, pair) && plane_pair_enabled();
        launch_plane(f, c, i, src, src_pitch, src_fs, dst, dst_pitch, dst_fs, plane_stream, fork ? f.aux_stream : plane_stream,
                     paired ? &pair : nullptr);
        if (paired) {
            f.tables[f.table_of_plane(i + 1)].last_kernel = f.tables[f.table_of_plane(i)].last_kernel;
            f.tables[f.table_of_plane(i + 1)].last_instance = f.tables[f.table_of_plane(i)].last_instance;
            ++i;
        }
    }
    if (fork || plane_fork) {  // `stream` continues only after the side stream's kernels have finished too
        hip_check(hipEventRecord(f.ev_join[turn], f.aux_stream), "hipEventRecord(join)");
        hip_check(hipStreamWaitEvent(stream, f.ev_join[turn], 0), "hipStreamWaitEvent(join)");
    }
}
}  // namespace

}  // namespace host
}  // namespace jinc
