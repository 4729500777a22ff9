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
    bool use_framelane_pair = false;  // ... and its frame-pair form (128 frames per workgroup; filter sizes 5 and 7)
    jinc::FrameLaneArgs fl_pair;
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

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Smallest batch the frame-lane kernel takes over from the gather kernel: its lanes are the frames of the batch, so a
// batch of n < 64 frames fills n of 64 lanes.
constexpr int kFrameLaneMinFrames = 16;

}  // namespace host
}  // namespace jinc

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
    int overlap_border = -1;  // -1: automatic (side stream unless the call is tiny: dispatch.cpp), 0: off, 1: on

    int device = -1;  // -1: host-only instance (plan inspection); frame calls fail
    hipStream_t stream = nullptr;
    std::vector<jinc::host::DeviceTable> tables;
    std::vector<jinc::host::DeviceFrameBuf> slots = std::vector<jinc::host::DeviceFrameBuf>(1);  // frames in flight (pipeline depth)
    long long next_ticket = 0;
    bool register_host = false;
    std::vector<jinc::host::PinnedRange> pinned;
    unsigned long long pin_clock = 0;
    bool profiling = false;
    std::vector<jinc::host::EventPair> ev_periodic, ev_gather;  // recorded, not yet collected
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
// pipeline.cpp
void ensure_slot(jinc_filter& f, DeviceFrameBuf& s, bool own_stream);
void submit_frame(jinc_filter& f, DeviceFrameBuf& s, const void* const src[4], const int src_pitch[4], void* const dst[4],
                  const int dst_pitch[4]);

}  // namespace host
}  // namespace jinc
