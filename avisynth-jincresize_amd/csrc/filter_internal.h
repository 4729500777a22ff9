// filter_internal.h -- types and internal interfaces shared by the host translation units of libjincresize_hip.so:
//   filter_args.cpp   Create_JincResize's argument handling and geometry derivation (configure)
//   device_plan.cpp   device-resident plans: upload, launch planning for every kernel family (init_device)
//   dispatch.cpp      per-call kernel selection and launches (enqueue)
//   pipeline.cpp      frames in flight: device staging slots, pinned host ranges, H2D -> kernels -> D2H
//   filter.cpp        the C ABI of include/jincresize_hip.h
// Nothing here crosses the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/jincresize_hip.h"
#include "../../include/jincresize_hip_test.h"
#include "jinc_lut.h"
#include "kernels.h"
#include "plan.h"

namespace jinc {
namespace host {

struct HipError : std::runtime_error {
    explicit HipError(const std::string& what) : std::runtime_error(what) {}
};

inline void hip_check(hipError_t e, const char* what) {
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
    // The same interior on the TRIMMED support (integer planes only; device_plan.cpp trim_periodic): trim_fs x trim_fs taps
    // per sample, the bounding box of the phase sets' non-zero coefficients; 0 = the sets have no zero rim (or float samples).
    int trim_fs = 0;
    int trim_nx = 0;  // taps per kernel row of the trimmed support: trim_fs, or 7 with trim_fs 6 (6 rows x 7 columns: quad2 form only)
    double trim_rows_taps = 0;       // taps per sample ewa_periodic_rows_kernel executes on the trimmed support (per-row spans)
    bool trim_needs_finite = false;  // float planes: the trimmed launch takes the frames whose samples are all finite
    jinc::PeriodicArgs periodic_trim;
    bool use_quasi = false;  // quasi-periodic interior kernel (affine window origins, drifting classes)
    jinc::QuasiArgs quasi;
    jinc::RectList border_rects;  // gather work when the periodic kernel covers the interior
    bool use_direct = false;      // exactly periodic, any filter size / source step (kernel_direct.hip)
    jinc::DirectArgs direct;      // interior
    int direct_trim_fs = 0;       // > 0: the same interior on the trimmed support (integer planes; device_plan.cpp trim_direct)
    jinc::DirectArgs direct_trim;
    jinc::DirectArgs row_strips;  // border rows of the same plan over the interior's columns (any interior kernel)
    bool use_runs = false;        // drifting plan cut into rectangles of one coefficient set each: the direct kernel's runs form
    jinc::DirectArgs runs;
    bool use_fl_border = false;   // border frame of a drifting plan (runs form or quasi-periodic kernel) on the frame-lane kernel when the call is a batch
    jinc::FrameLaneArgs fl_border;
    bool strips_ok = false;       // the border rows / columns really repeat their coefficient sets per phase (plan_direct)
    bool use_colstrip = false;    // border columns over the interior's rows on kernel_colstrip.hip
    jinc::ColStripArgs col_strips;
    jinc::RectList corner_rects;  // ... then only the corners are left for the gather kernel
    jinc::RectList column_rects;  // otherwise: left / right columns, full height, on the gather kernel
    bool use_fl_cols = false;     // ... or, in batches, on the frame-lane kernel (lanes = frames: every pixel's set is a scalar load)
    jinc::FrameLaneArgs fl_cols;
    // ... or (round 5, filter sizes up to 9, source step 1) on kernel_strip.hip: one register window per lane for the strip's thickness
    bool use_strip_rows = false, use_strip_cols = false;
    std::vector<jinc::PeriodicArgs> rowpair_rows;  // border rows as launches of ewa_periodic_rowpair_kernel (plan_rowpair_rows); empty: none
    bool use_colpair = false;  // border columns on ewa_colpair_kernel (plan_colpair)
    jinc::ColPairArgs colpair;
    bool use_edge_cols = false;  // border columns inside ewa_periodic_quad2_kernel's edge tiles (plan_edge_columns)
    jinc::PeriodicArgs::EdgeColumns edge_cols;
    jinc::StripArgs strip_rows, strip_cols;
    std::vector<void*> lane_blobs;  // lane-major coefficient copies of the private-set rectangles (RectList::lane_coeffs)
    jinc::RectList whole;         // gather work when it does not
    bool use_framelane = false;   // frame-lane kernel configured for the whole plane (batches of frames, any plan)
    jinc::FrameLaneArgs fl_whole;
    bool use_framelane_pair = false;  // ... and its frame-pair form (128 frames per workgroup; filter sizes 5 and 7)
    jinc::FrameLaneArgs fl_pair;
    int last_border = 0;           // border kernels of the most recent call (test header: jinc_filter_last_border)
    const char* last_kernel = "";  // interior kernel of the most recent call (reports)
    std::string last_instance;     // ... with its template arguments where the launcher chooses between instantiations (knobs.h note_instance)
};

struct EventPair {
    hipEvent_t start = nullptr, stop = nullptr;
};

// Frames in flight (pipeline.cpp).  A group is a run of consecutively submitted frames that share ONE strided device
// buffer per plane and ONE set of kernel launches (enqueue(..., nframes = frames.size(), ...)).  Three conveyor belts, each
// a stream that carries one kind of work for ALL groups in submission order:
//   h2d_stream   one DMA copy per plane and frame, queued at submit            -> event h2d_ready (per group)
//   f.stream     the resampling kernels of a group behind its h2d_ready         -> event kernels_done (per group)
//   d2h_stream   the group's results to the callers' planes behind kernels_done -> events done[] (per share / per frame)
// so that group j's results travel while group j + 1 is computed and group j + 2 arrives.  Results travel by the shader
// (kernel_blit.hip: one launch per share of the group, straight into the pinned host planes) when every destination plane
// of the group is pinned, else by one DMA copy per plane and frame.  One belt per kind of work keeps the stages in order:
// with a stream per group, the transports of two groups ran side by side, finished together, released two groups' worth
// of submissions at once and the pipeline fell into a two-phase rhythm (profiles/round3/e2e_pipeline_*.log).
// The two event hops between the belts cost ~0.2 ms of latency per group (measured: C2 at one frame per launch 5 600 ->
// 2 700 frames/s), so groups of fewer than kBeltMinGroup frames keep the round-2 shape instead: a stream per group that
// carries its copies, kernels and results in order, no events; consecutive groups overlap because their streams differ.
constexpr int kBeltMinGroup = 8;
constexpr int kGroupShares = 4;  // completion granularity of a group whose results travel by the shader

struct GroupFrame {
    void* dst[4] = {nullptr, nullptr, nullptr, nullptr};      // host destination planes
    void* dst_dev[4] = {nullptr, nullptr, nullptr, nullptr};  // their device mapping (pinned), or nullptr
    int dst_pitch[4] = {0, 0, 0, 0};
    long long ticket = -1;
    int done_event = -1;  // index into FrameGroup::done, set at launch
    // Planes (bit i) whose result goes to the group's OWN pinned host buffer and is copied into dst[i] by the CPU once the frame's
    // event has fired (deliver_frame): the caller's memory is pageable and the GPU never maps it (register_host == 0).
    unsigned staged_out = 0;
    bool delivered = false;
    // Small pageable source planes (bit i) of a frame in a group of several: not copied at submit but with the rest of the group
    // when it is launched -- one job for the copy lanes, one DMA copy per plane for the whole group (launch_group).
    unsigned deferred_in = 0;
    const void* src[4] = {nullptr, nullptr, nullptr, nullptr};
    int src_pitch[4] = {0, 0, 0, 0};
    // A frame that travels alone (a group of one: jinc_filter_get_frame): its staged result planes leave the device in row bands,
    // an event behind each, and the CPU copies a band out while the next ones are still on the wire.
    struct Band {
        int plane, y0, y1, event;  // rows [y0, y1) of `plane`, complete when FrameGroup::band_done[event] has fired
    };
    std::vector<Band> bands;
};

struct FrameGroup {
    void* src[4] = {nullptr, nullptr, nullptr, nullptr};  // device planes of `capacity` frames, frame stride *_fs: parts, in plane
    void* dst[4] = {nullptr, nullptr, nullptr, nullptr};  // order, of ONE allocation each (src_base / dst_base) -- the chroma planes of
    void* src_base = nullptr;                             // a single-frame group then lie at the same kind of distance in source and
    void* dst_base = nullptr;                             // destination, which lets them go out as one launch (dispatch.cpp plane_pair)
    int src_pitch[4] = {0, 0, 0, 0};
    int dst_pitch[4] = {0, 0, 0, 0};
    size_t src_fs[4] = {0, 0, 0, 0};
    size_t dst_fs[4] = {0, 0, 0, 0};
    int capacity = 0;
    hipStream_t own_stream = nullptr;  // small groups: everything of the group in order on this stream
    hipEvent_t h2d_ready = nullptr, kernels_done = nullptr;
    jinc::BlitEntry* table = nullptr;  // [capacity x planes], pinned host memory the device reads (hipHostMalloc)
    // Pinned host buffers of the library's own (hipHostMalloc, allocated when the first pageable plane arrives), laid out like
    // src_base / dst_base: pageable planes are copied through them by the CPU, so that no mapping of the CALLER's pages ever
    // exists on the device -- neither one of this library nor one the HIP runtime makes behind a copy from pageable memory.
    char* host_src = nullptr;
    char* host_dst = nullptr;
    std::vector<hipEvent_t> band_done;  // groups of one frame: events of GroupFrame::bands (created with the first banded frame)
    std::vector<hipEvent_t> done;    // [capacity]
    std::vector<GroupFrame> frames;  // frames of the current use of the buffer, in submission order
    enum State { Idle, Filling, Launched, Failed } state = Idle;
    std::string error;               // Failed: what the launch reported (handed to every wait on its frames)
};

struct PinnedRange {  // a caller buffer registered with hipHostRegister (look-ahead pipeline, opt-in)
    char* base = nullptr;
    size_t bytes = 0;
    char* dev = nullptr;    // the range's address in the device's address space (hipHostGetDevicePointer), or nullptr
    bool adopted = false;   // pinned by the caller (jinc_filter_adopt_host_range): never unregistered or evicted here
    unsigned long long stamp = 0;
    long long ticket = -1;  // latest frame whose copies use this range (may still be in flight)
    unsigned long long id = 0;  // the process-wide registry's entry this instance holds a reference to (0: adopted range)
};

struct FailedTickets {  // a group whose launch failed: every wait on one of its frames reports `error`
    long long first = 0, last = -1;
    std::string error;
};

// Upper limits of the look-ahead pipeline: frames in flight per instance, and the device memory its staging may take.
constexpr int kMaxPipelineDepth = 256;
constexpr size_t kPipelineBudgetBytes = size_t(24) << 30;
constexpr size_t kHostStagingBudgetBytes = size_t(4) << 30;  // pinned host memory of an instance whose planes are pageable (register_host == 0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace host
}  // namespace jinc
struct jinc_filter;
namespace jinc {
namespace host {
void release_pipeline(jinc_filter& f);  // pipeline.cpp: waits for the frames in flight, frees group buffers, events and copy lanes

// Smallest batch the frame-lane kernel takes over from the gather kernel: its lanes are the frames of the batch, so a
// batch of n < 64 frames fills n of 64 lanes.
constexpr int kFrameLaneMinFrames = 16;

}  // namespace host
}  // namespace jinc

struct jinc_filter {
    jinc_video_info vi_in{};
    jinc_video_info vi_out{};
    std::string cplace;
    int chroma_location = -1;            // what the reference binary writes: 2 for sub-sampled chroma, else -1 (ref :617-625)
    int chroma_location_by_siting = -1;  // 0 / 1 / 2 by cplace: what the reference's source means to write
    int chroma_location_mode = 0;        // jinc_filter_set_chroma_location_mode
    float peak = 0.f;
    int planecount = 0;
    bool subsampled = false;
    jinc::JincLut lut;
    std::vector<jinc::PlanePlan> plans;  // [0] luma / all planes, [1] chroma of subsampled formats
    int kernel_mode = 0;
    bool full_window = false;  // kernel mode 15: no trimmed support (the reference's full window everywhere)
    uint32_t* finite_flags = nullptr;  // [kForkEvents sets][4 planes][finite_flags_frames]: kernel_scan.hip's verdict per plane and frame (float planes)
    int finite_flags_frames = 0;
    unsigned finite_flags_turn = 0;  // which of the kForkEvents flag sets the current call uses (advanced once per call: dispatch.cpp enqueue)
    int border_strips = -1;  // border frame of exactly periodic plans: -1 by call size (dispatch.cpp Rules), 1 strip kernels, 2 rows only, 0 gather kernel
    bool direct_premise = false;  // buffer_range_check_covers_soffset(device) == 1
    int simd_order = 0;  // 0: opt=0 results (default); 1 / 2 / 3: summation order of the reference's SSE4.1 / AVX2 / AVX-512 path
    int overlap_border = -1;  // -1: automatic (side stream unless the call is tiny: dispatch.cpp), 0: off, 1: on

    int device = -1;  // -1: host-only instance (plan inspection); frame calls fail
    hipStream_t stream = nullptr;
    std::vector<jinc::host::DeviceTable> tables;
    int pipeline_depth = 1;   // frames the client keeps in flight (jinc_filter_set_pipeline)
    int group_frames = 1;     // frames coalesced into one launch
    std::vector<jinc::host::FrameGroup> groups = std::vector<jinc::host::FrameGroup>(1);  // ring of group buffers
    hipStream_t h2d_stream = nullptr, d2h_stream = nullptr;  // belts for groups of >= kBeltMinGroup frames (else nullptr)
    int transport = -1;       // results to the host: -1 automatic (shader when pinned), 0 DMA copies always (A/B: knob PIPELINE_DMA)
    int open_group = -1;      // index of the group being filled, -1: none
    int last_group = 0;       // most recently opened group (the ring advances from here)
    long long next_ticket = 0;
    // How caller buffers travel (include/jincresize_hip.h): 0 (the default) copied by the CPU through pinned buffers of the
    // library's own -- the device never maps the caller's pages; 2 registered by this library and cached by address (a host whose
    // frame memory is a pool that stays mapped); 3 handed to the HIP runtime as they are (which maps the caller's pages itself
    // behind every copy: the default of rounds 1 - 5).
    int register_host = 0;
    bool copy_helpers = true;  // pageable planes may be copied by the helper threads of host_copy.cpp (the script's threads != 1)
    std::vector<jinc::host::PinnedRange> pinned;
    std::vector<jinc::host::FailedTickets> failed;  // groups whose launch failed and whose buffer has gone back into the ring
    unsigned long long pin_clock = 0;
    bool profiling = false;
    std::vector<jinc::host::EventPair> ev_periodic, ev_gather;  // recorded, not yet collected
    // The border gather kernel (load/store-issue bound) runs on a side stream next to the periodic
    // interior kernel (VALU bound): fork/join with two reusable events.
    hipStream_t aux_stream = nullptr;
    // (a ring of event pairs, one pair per call in turn: a call never re-records an event that a stream of an earlier call --
    // the look-ahead pipeline's groups have streams of their own -- may still be waiting on)
    static constexpr int kForkEvents = 16;
    hipEvent_t ev_fork[kForkEvents] = {}, ev_join[kForkEvents] = {};
    unsigned fork_turn = 0;

    ~jinc_filter() {
        if (device >= 0) {
            (void)hipSetDevice(device);
            for (auto& t : tables) {
                if (t.blob) (void)hipFree(t.blob);
                for (void* b : t.lane_blobs) (void)hipFree(b);
            }
            if (finite_flags) (void)hipFree(finite_flags);
            jinc::host::release_pipeline(*this);  // (also returns this instance's references to pinned host ranges)
            for (auto* v : {&ev_periodic, &ev_gather})
                for (auto& e : *v) {
                    (void)hipEventDestroy(e.start);
                    (void)hipEventDestroy(e.stop);
                }
            for (hipEvent_t e : ev_fork)
                if (e) (void)hipEventDestroy(e);
            for (hipEvent_t e : ev_join)
                if (e) (void)hipEventDestroy(e);
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

namespace jinc {
namespace host {

// filter.cpp: message of the last failure on the calling thread
int fail(int code, const std::string& msg);
// filter_args.cpp: Create_JincResize's argument handling (ref :700-789) and geometry (ref :791-866); throws ArgError
void configure(jinc_filter& f, const jinc_video_info& vi, const jinc_args& a);
// device_plan.cpp
void init_device(jinc_filter& f, int device);
int buffer_range_check_covers_soffset(int device);
bool direct_fetch_is_safe(size_t frame_stride, int nframes, uint64_t plane_bytes, int pitch, int fs);
uint32_t direct_src_bytes(const void* base, uint64_t plane_bytes);
// dispatch.cpp: kernel launches of one call (one plane loop) on `stream`
void enqueue(jinc_filter& f, const void* const src[4], const int src_pitch[4], const size_t src_fs[4], void* const dst[4],
             const int dst_pitch[4], const size_t dst_fs[4], int nframes, hipStream_t stream);
const char* last_interior_kernel_in_process();
int last_call_frames_in_process();
const char* last_interior_instance_in_process();
// pipeline.cpp: frames in flight on one instance
void configure_pipeline(jinc_filter& f, int depth, int group, int register_host);  // drains first; register_host: 0 / 2 / 3 as jinc_filter::register_host
long long submit_frame(jinc_filter& f, const void* const src[4], const int src_pitch[4], void* const dst[4], const int dst_pitch[4]);
void wait_frame(jinc_filter& f, long long ticket);  // flushes the open group if the frame is in it
void adopt_host_range(jinc_filter& f, void* base, size_t bytes);  // caller-pinned memory: usable for async copies and shader transport
void release_host_range(jinc_filter& f, void* base, size_t bytes);  // the caller is about to unpin / free it: drains, then forgets adopted ranges that touch it
void launch_open_group(jinc_filter& f);             // the frames submitted so far leave now (a client that knows no more are coming)
void drain_pipeline(jinc_filter& f);                // every submitted frame complete
void transport_counts(long long* by_shader, long long* by_dma, long long* pinned_ranges, bool reset);  // process-wide (test header)
// host_copy.cpp: rows between a pageable plane and a pinned buffer of the library's own; large planes are cut into row ranges for the
// process-wide helper threads when may_use_helpers (the calling thread always takes part and returns when every row has been copied).
void copy_plane_rows(char* dst, size_t dst_pitch, const char* src, size_t src_pitch, size_t row_bytes, int rows, bool may_use_helpers);
struct PlaneCopy {
    char* dst;
    const char* src;
    size_t dst_pitch, src_pitch, row_bytes;
    int rows;
};
void copy_planes(const PlaneCopy* jobs, size_t njobs, bool may_use_helpers);
int copy_lanes_cpus();  // CPUs the process may keep busy (affinity mask cut down to the cgroup's CFS quota): what the helper pool is sized by  // several planes as ONE job for the lanes (the frames of a group's share)
// device_plan.cpp: host <-> device copies through the library's pinned bounce buffer (synchronous; test hooks, plan tables)
void bounce_upload(void* dev, const void* host, size_t bytes, const char* what);
void bounce_download(void* host, const void* dev, size_t bytes, const char* what);
long long staged_frames();  // frames whose results went through the library's own pinned buffers since the last reset of transport_counts

}  // namespace host
}  // namespace jinc
