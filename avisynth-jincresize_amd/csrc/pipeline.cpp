// pipeline.cpp -- frames in flight on one filter instance (frame transport around GetFrame).
//
// The reference's surface is per frame (JincResize_GetFrame handles frame n only, ref /root/reference/src/JincResize.cpp:603-630)
// and stays that way: jinc_filter_submit / _wait / _get_frame take and complete single frames.  What happens in between is
// coalesced: consecutively submitted frames are staged into ONE strided device buffer per plane (a FrameGroup) and leave
// as ONE enqueue(..., nframes = k, ...) -- so the batch kernels of dispatch.cpp (frame-lane / frame-pair forms, wide
// tiles) serve the host-pointer entry points too.
//   submit(frame)  H2D copy of the frame into slot k of the open group, queued at once (pageable planes: the CPU copies the rows
//                  into the group's own pinned buffer first, band by band); the group is launched when it holds `group_frames`
//                  frames;
//   launch         the kernels behind the group's copies, then the results towards the callers' planes: by the shader straight
//                  into the pinned host planes (kernel_blit.hip, kGroupShares launches + events per group) when every
//                  destination plane is pinned; pageable planes: contiguous DMA copies into the group's pinned buffer, one per
//                  plane and share of the group (a lone frame: per row band), an event per share;
//   wait(ticket)   launches the open group early if the frame sits in it, waits for that frame's event only, and copies the
//                  frame's rows from the pinned buffer into pageable destination planes (deliver_frame);
//   groups rotate through a ring of ceil(depth / group_frames) + 1 buffers; reusing a buffer waits for its previous use
//   (back-pressure).  Streams: see FrameGroup in filter_internal.h (three belts: arrivals, kernels, departures).
// Why the shader: one DMA copy per plane and frame costs ~17 us of engine turnaround besides the wire time (2 MB planes:
// one every 60 us = 35 GB/s of the link's 52 GB/s); a single launch per share of the group has no per-frame cost.
// Results do not depend on the grouping: frames are independent and every kernel computes a frame's samples the same way
// whatever the batch size (tests/test_gpu_parity.py, tests/test_pipeline_groups.py).
#include <atomic>
#include <cstdlib>
#include <mutex>

#include "filter_internal.h"
#include "knobs.h"

namespace jinc {
namespace host {

namespace {

// ---- Host ranges pinned by this library, process-wide ------------------------------------------------------------
// AviSynth's frame pool is shared by every filter instance of a script (MT_MULTI_INSTANCE under Prefetch(N): N instances
// of this filter, ref /root/reference/src/JincResize.cpp:649-652), so the same buffer reaches different instances in turn.
// hipHostRegister refuses a range that is registered already; with a cache per instance the second instance to see a buffer
// fell back to pageable copies for good.  One registry for the process instead: a range is registered once (portable:
// valid on every device), instances take and return references, the last reference unregisters.  Ranges somebody ELSE
// pinned (the host itself, another library) are recognised by asking for their device address and are never unregistered.
struct SharedPin {
    unsigned long long id = 0;
    char* base = nullptr;
    size_t bytes = 0;
    int refs = 0;
    bool owned = false;  // registered here (else: found pinned)
    bool dead = false;   // unregistered early because it turned out stale (see shared_pin_acquire); kept until its references are back
    unsigned long long devices = 0;  // bit d: an instance on device d has held a reference (whose work may run through the range's mapping)
};
// The registry outlives every static destructor (a leaked function-local static): a host may free its filters -- and
// with them release pins -- while the process's own statics are already being destroyed.
struct PinRegistry {
    std::mutex mutex;
    std::vector<SharedPin> pins;
    unsigned long long next_id = 1;
};
PinRegistry& pin_registry() {
    static PinRegistry& r = *new PinRegistry;
    return r;
}

// The devices of `mask` idle: before a registration that instances still hold references to is unregistered early (a transport
// kernel or an asynchronous copy of one of them may still be going through the range's device mapping).  Only devices whose
// instances have held the range are touched (a rank of a one-process-per-GPU job must not create contexts on the
// other seven), and never under the registry's lock.
void quiesce_devices(unsigned long long mask) {
    int cur = 0;
    if (!mask || hipGetDevice(&cur) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    for (int d = 0; d < 64; ++d)
        if ((mask >> d) & 1)
            if (hipSetDevice(d) == hipSuccess) (void)hipDeviceSynchronize();
    (void)hipSetDevice(cur);
    (void)hipGetLastError();
}
std::atomic<long long> g_frames_by_shader{0}, g_frames_by_dma{0};  // how results left the device, process-wide (test header)
std::atomic<long long> g_frames_staged{0};  // frames whose results went through the library's own pinned buffers (register_host == 0)

// A reference to a live registered range that contains [c, c + bytes), registering it if need be; false: not pinnable.
// STALE registrations (pin mode 2 only: mode 1 lets go of a range with its last frame in flight): the host may free memory
// this registry still holds pinned (instances keep references for as long as their caches do), and hand out the same
// addresses again in other sizes.  hipHostRegister refuses a range whose START lies inside an existing registration; if that
// registration is one of ours and does not contain the new range, it can only be stale -- two live buffers do not overlap --
// so it is marked dead (holders find out through shared_pin_alive and let go), unregistered once the devices that used it are
// idle, and the new range is registered in its place.  A range is taken for "pinned by somebody else" only if NONE of our
// registrations touches it (two stale ranges under its ends with a hole between them would pass the device-address probe and
// fault in the hole).  What this cannot see is a range that came back at the SAME address and size: that is the promise of mode 2.
bool shared_pin_acquire(char* c, size_t bytes, int device, unsigned long long* id, char** base, size_t* len) {
    const unsigned long long dev_bit = device >= 0 && device < 64 ? 1ull << device : 0;
    struct Stale {
        char* base;
        unsigned long long devices;
        bool held;
    };
    std::vector<Stale> stale;
    bool touches_ours = false;
    {
        std::lock_guard<std::mutex> lock(pin_registry().mutex);
        for (auto& e : pin_registry().pins)
            if (!e.dead && c >= e.base && c + bytes <= e.base + e.bytes) {
                ++e.refs;
                e.devices |= dev_bit;
                *id = e.id, *base = e.base, *len = e.bytes;
                return true;
            }
        if ([&] {
                std::lock_guard<std::mutex> one(knobs::host_registration_mutex());
                return hipHostRegister(c, bytes, hipHostRegisterPortable | hipHostRegisterMapped);
            }() == hipSuccess) {
            knobs::count_host_registration(+1);
            pin_registry().pins.push_back({pin_registry().next_id++, c, bytes, 1, true, false, dev_bit});
            *id = pin_registry().pins.back().id, *base = c, *len = bytes;
            return true;
        }
        (void)hipGetLastError();
        for (auto& e : pin_registry().pins) {
            if (e.dead || !e.owned) continue;
            if (c >= e.base && c < e.base + e.bytes) {  // our registration under the new range's start, not containing it: stale
                e.dead = true;  // from here on nobody takes a new reference to it
                stale.push_back({e.base, e.devices, e.refs > 0});
            } else if (c + bytes > e.base && c < e.base + e.bytes) {
                touches_ours = true;
            }
        }
    }
    // outside the lock: other threads' acquires and releases go on while the devices drain
    for (const Stale& st : stale) {
        if (st.held) quiesce_devices(st.devices);  // (holders learn of it only at their next pin_host_range: nothing of theirs may still be in flight)
        if ([&] {
                std::lock_guard<std::mutex> one(knobs::host_registration_mutex());
                return hipHostUnregister(st.base);
            }() == hipSuccess)
            knobs::count_host_registration(-1);
        (void)hipGetLastError();  // (it may already be gone with its memory: the sticky error must not meet the next launch)
    }
    bool owned = true;
    const bool again = !stale.empty() && [&] {
        std::lock_guard<std::mutex> one(knobs::host_registration_mutex());
        return hipHostRegister(c, bytes, hipHostRegisterPortable | hipHostRegisterMapped);
    }() == hipSuccess;
    if (again) knobs::count_host_registration(+1);
    if (!again) {
        (void)hipGetLastError();
        if (!stale.empty() || touches_ours) return false;
        // Pinned by the host application itself?  Then the WHOLE range lies inside one registered object: its first byte has a
        // device address, and the allocation that address belongs to reaches past the range's last byte (two foreign ranges under
        // the ends with unpinned pages between them must not pass: the transport kernel would fault in the hole).
        void* d0 = nullptr;
        hipDeviceptr_t obj = nullptr;
        size_t obj_bytes = 0;
        if (hipHostGetDevicePointer(&d0, c, 0) != hipSuccess || hipMemGetAddressRange(&obj, &obj_bytes, d0) != hipSuccess ||
            static_cast<char*>(d0) < static_cast<char*>(obj) ||
            static_cast<size_t>(static_cast<char*>(d0) - static_cast<char*>(obj)) + bytes > obj_bytes) {
            (void)hipGetLastError();
            return false;
        }
        owned = false;
    }
    std::lock_guard<std::mutex> lock(pin_registry().mutex);
    pin_registry().pins.push_back({pin_registry().next_id++, c, bytes, 1, owned, false, dev_bit});
    *id = pin_registry().pins.back().id, *base = c, *len = bytes;
    return true;
}

bool shared_pin_alive(unsigned long long id) {
    std::lock_guard<std::mutex> lock(pin_registry().mutex);
    for (const auto& e : pin_registry().pins)
        if (e.id == id) return !e.dead;
    return false;
}

void shared_pin_release(unsigned long long id) {
    char* unregister = nullptr;
    {
        std::lock_guard<std::mutex> lock(pin_registry().mutex);
        for (size_t i = 0; i < pin_registry().pins.size(); ++i)
            if (pin_registry().pins[i].id == id) {
                if (--pin_registry().pins[i].refs > 0) return;
                if (pin_registry().pins[i].owned && !pin_registry().pins[i].dead) unregister = pin_registry().pins[i].base;
                pin_registry().pins.erase(pin_registry().pins.begin() + static_cast<std::ptrdiff_t>(i));
                break;
            }
        // (unregistered under the lock: a concurrent acquire of the same range must either find the entry or find the range free)
        if (unregister) {
            std::lock_guard<std::mutex> one(knobs::host_registration_mutex());
            if (hipHostUnregister(unregister) == hipSuccess) knobs::count_host_registration(-1);
            else (void)hipGetLastError();
        }
    }
}

void release_group(FrameGroup& g) {  // (the belts are idle: callers drain first)
    if (g.src_base) (void)hipFree(g.src_base);
    if (g.dst_base) (void)hipFree(g.dst_base);
    g.src_base = g.dst_base = nullptr;
    for (int i = 0; i < 4; ++i) g.src[i] = g.dst[i] = nullptr;
    for (hipEvent_t e : g.done) (void)hipEventDestroy(e);
    g.done.clear();
    if (g.table) (void)hipHostFree(g.table);
    g.table = nullptr;
    for (hipEvent_t e : g.band_done) (void)hipEventDestroy(e);
    g.band_done.clear();
    if (g.host_src) (void)hipHostFree(g.host_src);
    if (g.host_dst) (void)hipHostFree(g.host_dst);
    g.host_src = g.host_dst = nullptr;
    if (g.own_stream) (void)hipStreamDestroy(g.own_stream);
    g.own_stream = nullptr;
    for (hipEvent_t* e : {&g.h2d_ready, &g.kernels_done}) {
        if (*e) (void)hipEventDestroy(*e);
        *e = nullptr;
    }
    g.capacity = 0;
    g.frames.clear();
    g.state = FrameGroup::Idle;
}

bool use_belts(const jinc_filter& f) { return f.group_frames >= kBeltMinGroup; }
// Streams of a group's three kinds of work: the filter's belts, or the group's own stream for all three.
hipStream_t h2d_of(const jinc_filter& f, const FrameGroup& g) { return g.own_stream ? g.own_stream : f.h2d_stream; }
hipStream_t kernels_of(const jinc_filter& f, const FrameGroup& g) { return g.own_stream ? g.own_stream : f.stream; }
hipStream_t d2h_of(const jinc_filter& f, const FrameGroup& g) { return g.own_stream ? g.own_stream : f.d2h_stream; }

size_t staging_bytes_per_frame(const jinc_filter& f) {
    const int sb = f.vi_in.component_size;
    size_t total = 0;
    for (int i = 0; i < f.planecount; ++i) {
        int sw, sh, dw, dh;
        f.plane_dims(f.vi_in, i, sw, sh);
        f.plane_dims(f.vi_out, i, dw, dh);
        total += align_up(align_up(static_cast<size_t>(sw) * sb, 256) * sh, 256) + align_up(align_up(static_cast<size_t>(dw) * sb, 256) * dh, 256);
    }
    return total;
}

// Device buffers, stream, events and transport table of a group, for f.group_frames frames (allocated on first use).
void ensure_group(jinc_filter& f, FrameGroup& g) {
    if (g.capacity == f.group_frames) return;
    release_group(g);
    const int sb = f.vi_in.component_size;
    const size_t cap = static_cast<size_t>(f.group_frames);
    try {
        for (int i = 0; i < f.planecount; ++i) {
            int sw, sh, dw, dh;
            f.plane_dims(f.vi_in, i, sw, sh);
            f.plane_dims(f.vi_out, i, dw, dh);
            g.src_pitch[i] = static_cast<int>(align_up(static_cast<size_t>(sw) * sb, 256));
            g.dst_pitch[i] = static_cast<int>(align_up(static_cast<size_t>(dw) * sb, 256));
            g.src_fs[i] = align_up(static_cast<size_t>(g.src_pitch[i]) * sh, 256);
            g.dst_fs[i] = align_up(static_cast<size_t>(g.dst_pitch[i]) * dh, 256);
        }
        size_t src_total = 0, dst_total = 0;
        for (int i = 0; i < f.planecount; ++i) src_total += g.src_fs[i] * cap, dst_total += g.dst_fs[i] * cap;
        hip_check(hipMalloc(&g.src_base, src_total), "hipMalloc(src planes of a frame group)");
        hip_check(hipMalloc(&g.dst_base, dst_total), "hipMalloc(dst planes of a frame group)");
        size_t so = 0, dof = 0;
        for (int i = 0; i < f.planecount; ++i) {
            g.src[i] = static_cast<char*>(g.src_base) + so;
            g.dst[i] = static_cast<char*>(g.dst_base) + dof;
            so += g.src_fs[i] * cap;
            dof += g.dst_fs[i] * cap;
        }
        if (use_belts(f)) {
            hip_check(hipEventCreateWithFlags(&g.h2d_ready, hipEventDisableTiming), "hipEventCreate");
            hip_check(hipEventCreateWithFlags(&g.kernels_done, hipEventDisableTiming), "hipEventCreate");
        } else {
            hip_check(hipStreamCreateWithFlags(&g.own_stream, hipStreamNonBlocking), "hipStreamCreate");
        }
        hip_check(hipHostMalloc(reinterpret_cast<void**>(&g.table), sizeof(jinc::BlitEntry) * cap * 4, hipHostMallocDefault),
                  "hipHostMalloc(transport table)");
        g.done.reserve(cap);
        for (size_t k = 0; k < cap; ++k) {
            hipEvent_t e = nullptr;
            hip_check(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
            g.done.push_back(e);
        }
    } catch (...) {  // a later allocation failed: give back what this call allocated (the next call starts over)
        release_group(g);
        throw;
    }
    g.capacity = f.group_frames;
}

// ---- Pageable planes: through pinned buffers of the library's own ---------------------------------------------------
// The default (register_host == 0).  Rounds 1 - 5 handed pageable planes to hipMemcpy2DAsync as they were; the runtime maps the
// caller's pages into the device behind such a copy (2 MB in 46 us = 45 GB/s from "pageable" memory, first touch 77 - 126 us:
// profiles/round6/pageable_rect_copy_probe.log) and the copy engine or a blit kernel reads them in place.  Full test runs of
// round 6 ended in GPU memory access faults on HEAP addresses inside exactly those copies (4 of the first 13 runs, each time with no
// registration of this library alive; profiles/round6/README.md), under every allocator setting tried, and the cause could not
// be isolated (the probe's unmap / remap / register-over-it / many-streams scenarios all pass).  A library a video host loads
// must not be able to take the host down with a GPU fault, so by default the device never sees the caller's pages at all:
// the CPU copies source rows into host_src at submit and result rows out of host_dst once the frame's event has fired; the
// DMA engines move whole contiguous planes between these buffers and the group's device buffers.  The price is the CPU's copy
// rate; hosts that want the link's rate pin their frame memory (modes 2 / adopt_host_range) or accept the runtime's mapping (3).
void ensure_host_staging(jinc_filter& f, FrameGroup& g) {
    if (g.host_src && g.host_dst) return;
    const size_t cap = static_cast<size_t>(g.capacity);
    size_t src_total = 0, dst_total = 0;
    for (int i = 0; i < f.planecount; ++i) src_total += g.src_fs[i] * cap, dst_total += g.dst_fs[i] * cap;
    if (!g.host_src) hip_check(hipHostMalloc(reinterpret_cast<void**>(&g.host_src), src_total, hipHostMallocDefault), "hipHostMalloc(source planes of a frame group)");
    if (!g.host_dst) hip_check(hipHostMalloc(reinterpret_cast<void**>(&g.host_dst), dst_total, hipHostMallocDefault), "hipHostMalloc(result planes of a frame group)");
}

char* host_src_plane(const FrameGroup& g, int i, size_t k) {
    return g.host_src + (static_cast<const char*>(g.src[i]) - static_cast<const char*>(g.src_base)) + g.src_fs[i] * k;
}
char* host_dst_plane(const FrameGroup& g, int i, size_t k) {
    return g.host_dst + (static_cast<const char*>(g.dst[i]) - static_cast<const char*>(g.dst_base)) + g.dst_fs[i] * k;
}

// The frame's event has fired: its staged result planes go to the caller's planes (once).
void deliver_frame(jinc_filter& f, FrameGroup& g, size_t k) {
    GroupFrame& fr = g.frames[k];
    if (fr.delivered) return;
    fr.delivered = true;
    const int sb = f.vi_in.component_size;
    if (!fr.bands.empty()) {  // (a frame that travelled alone: band by band, behind each band's event)
        for (const GroupFrame::Band& b : fr.bands) {
            hip_check(hipEventSynchronize(g.band_done[static_cast<size_t>(b.event)]), "hipEventSynchronize(band done)");
            int dw, dh;
            f.plane_dims(f.vi_out, b.plane, dw, dh);
            copy_plane_rows(static_cast<char*>(fr.dst[b.plane]) + static_cast<size_t>(fr.dst_pitch[b.plane]) * b.y0, static_cast<size_t>(fr.dst_pitch[b.plane]),
                            host_dst_plane(g, b.plane, k) + static_cast<size_t>(g.dst_pitch[b.plane]) * b.y0, static_cast<size_t>(g.dst_pitch[b.plane]),
                            static_cast<size_t>(dw) * sb, b.y1 - b.y0, f.copy_helpers);
        }
        return;
    }
    // Every frame that became complete with this one (the same share of the group, the same event) is delivered with it, as ONE
    // job for the copy lanes: small planes do not fill six lanes one by one, a share's worth of them does, and the client is
    // going to ask for the others next (their destination planes stay the caller's until each frame's own wait returns, as
    // the contract of jinc_filter_submit says).
    std::vector<PlaneCopy> jobs;
    for (size_t j = 0; j < g.frames.size(); ++j) {
        GroupFrame& other = g.frames[j];
        if (j != k && (other.delivered || other.done_event != fr.done_event || !other.bands.empty())) continue;
        other.delivered = true;
        for (int i = 0; i < f.planecount; ++i) {
            if (!((other.staged_out >> i) & 1)) continue;
            int dw, dh;
            f.plane_dims(f.vi_out, i, dw, dh);
            jobs.push_back({static_cast<char*>(other.dst[i]), host_dst_plane(g, i, j), static_cast<size_t>(other.dst_pitch[i]), static_cast<size_t>(g.dst_pitch[i]),
                            static_cast<size_t>(dw) * sb, dh});
        }
    }
    if (!jobs.empty()) copy_planes(jobs.data(), jobs.size(), f.copy_helpers);
}

// Row bands a staged plane of `bytes` bytes is cut into on its way between the CPU's copy and the DMA engine, so that one works
// while the other does (2 MiB per band, at most 4; A/B knob STAGE_BANDS, 1 = whole planes).
int stage_bands(size_t bytes, int rows) {
    const int most = std::max(1, knobs::geti(JINC_KNOB_STAGE_BANDS, 4));
    return static_cast<int>(std::max<size_t>(1, std::min<size_t>({static_cast<size_t>(most), bytes >> 21, static_cast<size_t>(std::max(rows, 1))})));
}

// The group's previous use is over: every frame of it complete (or failed), the buffer free for the next run of frames.
void finish_group(jinc_filter& f, FrameGroup& g) {
    // Launched: the event of the group's last frame closes everything the group has queued (the departures belt is in
    // order).  Filling / Failed: its arrivals may still be on the wire.
    if (g.state == FrameGroup::Launched && !g.frames.empty()) {
        hip_check(hipEventSynchronize(g.done[static_cast<size_t>(g.frames.back().done_event)]), "hipEventSynchronize(group done)");
        for (size_t k = 0; k < g.frames.size(); ++k) deliver_frame(f, g, k);  // frames nobody waited for arrive all the same
    } else if (g.state != FrameGroup::Idle && h2d_of(f, g))
        hip_check(hipStreamSynchronize(h2d_of(f, g)), "stream sync");
}

void retire_group(jinc_filter& f, FrameGroup& g) {
    finish_group(f, g);
    // the buffer goes back into the ring; what its frames' waits must still be told stays behind (a failed launch is
    // reported by every wait on a frame of that group, also after the ring has rotated)
    if (g.state == FrameGroup::Failed && !g.frames.empty()) {
        if (f.failed.size() >= 64) f.failed.erase(f.failed.begin());
        f.failed.push_back({g.frames.front().ticket, g.frames.back().ticket, g.error});
    }
    g.frames.clear();
    g.error.clear();
    g.state = FrameGroup::Idle;
}

// Workgroups of the transport kernel.  The link bounds it, and next to a transport-bound plan 48 workgroups keep the link
// full (C2, 1.37x: 24 workgroups lose 6 %); next to a kernel-bound plan (many taps per output byte: 1.5x with tap 8)
// every transport wave takes issue slots from the resampling kernel and 24 are the better trade (17.8 against 16.5
// thousand frames/s; profiles/round3/blit_workgroups.log).  A/B knob: BLIT_WORKGROUPS.
int blit_workgroups(const jinc_filter& f) {
    const int knob = knobs::geti(JINC_KNOB_BLIT_WORKGROUPS, 0);
    if (knob > 0) return knob;
    int taps_per_byte = 0;
    for (const auto& p : f.plans) taps_per_byte = std::max(taps_per_byte, p.fs * p.fs / f.vi_in.component_size);
    return taps_per_byte >= 100 ? 24 : 48;
}

int group_shares() {  // A/B knob GROUP_SHARES (default kGroupShares)
    const int n = knobs::geti(JINC_KNOB_GROUP_SHARES, 0);
    return n > 0 ? n : kGroupShares;
}

// Diagnosis only (knob PIPELINE_SKIP = 1: no H2D copies, 2: no kernels -- wrong results, same transport otherwise): which
// stage slows which.
int debug_skip() { return knobs::geti(JINC_KNOB_PIPELINE_SKIP, 0); }

bool dma_forced() { return knobs::flag(JINC_KNOB_PIPELINE_DMA, false); }

// The small pageable source planes the group's frames left for the launch (submit_frame): into the group's pinned buffer as ONE
// job for the copy lanes, then one DMA copy per plane and run of consecutive frames.
void stage_deferred_sources(jinc_filter& f, FrameGroup& g) {
    const int sb = f.vi_in.component_size;
    std::vector<PlaneCopy> jobs;
    for (size_t k = 0; k < g.frames.size(); ++k)
        for (int i = 0; i < f.planecount; ++i)
            if ((g.frames[k].deferred_in >> i) & 1) {
                int sw, sh;
                f.plane_dims(f.vi_in, i, sw, sh);
                jobs.push_back({host_src_plane(g, i, k), static_cast<const char*>(g.frames[k].src[i]), static_cast<size_t>(g.src_pitch[i]),
                                static_cast<size_t>(g.frames[k].src_pitch[i]), static_cast<size_t>(sw) * sb, sh});
            }
    if (jobs.empty()) return;
    copy_planes(jobs.data(), jobs.size(), f.copy_helpers);
    for (int i = 0; i < f.planecount; ++i)
        for (size_t k = 0; k < g.frames.size();) {
            if (!((g.frames[k].deferred_in >> i) & 1)) {
                ++k;
                continue;
            }
            size_t end = k;
            while (end < g.frames.size() && ((g.frames[end].deferred_in >> i) & 1)) ++end;
            hip_check(hipMemcpyAsync(static_cast<char*>(g.src[i]) + g.src_fs[i] * k, host_src_plane(g, i, k), g.src_fs[i] * (end - k), hipMemcpyHostToDevice,
                                     h2d_of(f, g)),
                      "H2D copy (staged, group)");
            for (size_t j = k; j < end; ++j) g.frames[j].deferred_in &= ~(1u << i);
            k = end;
        }
}

// Kernels of the group's frames in one call, then the results to the callers' planes and the events their waits block
// on.  A failure marks the group Failed; every wait on one of its frames reports it.
void launch_group(jinc_filter& f, FrameGroup& g) {
    const int n = static_cast<int>(g.frames.size());
    const int sb = f.vi_in.component_size;
    const int planes = f.planecount;
    if (f.open_group >= 0 && &g == &f.groups[static_cast<size_t>(f.open_group)]) f.open_group = -1;
    try {
        const bool belts = !g.own_stream;
        hipStream_t d2h = d2h_of(f, g);
        stage_deferred_sources(f, g);
        if (belts) {  // the kernels read what the arrivals belt has copied
            hip_check(hipEventRecord(g.h2d_ready, f.h2d_stream), "hipEventRecord(arrivals)");
            hip_check(hipStreamWaitEvent(f.stream, g.h2d_ready, 0), "hipStreamWaitEvent(arrivals)");
        }
        if (debug_skip() != 2) enqueue(f, g.src, g.src_pitch, g.src_fs, g.dst, g.dst_pitch, g.dst_fs, n, kernels_of(f, g));
        if (belts) {  // the departures belt reads what the kernels have written
            hip_check(hipEventRecord(g.kernels_done, f.stream), "hipEventRecord(kernels)");
            hip_check(hipStreamWaitEvent(d2h, g.kernels_done, 0), "hipStreamWaitEvent(kernels)");
        }
        bool by_shader = f.transport != 0 && !dma_forced();
        for (const GroupFrame& fr : g.frames)
            for (int i = 0; i < planes; ++i) by_shader &= fr.dst_dev[i] != nullptr;
        if (by_shader) {
            uint32_t max_rows = 0, max_row_bytes = 0;
            for (int k = 0; k < n; ++k)
                for (int i = 0; i < planes; ++i) {
                    int dw, dh;
                    f.plane_dims(f.vi_out, i, dw, dh);
                    const GroupFrame& fr = g.frames[static_cast<size_t>(k)];
                    jinc::BlitEntry& e = g.table[static_cast<size_t>(k) * planes + i];
                    e.src = static_cast<const char*>(g.dst[i]) + g.dst_fs[i] * k;
                    e.dst = fr.dst_dev[i];
                    e.src_pitch = static_cast<uint32_t>(g.dst_pitch[i]);
                    e.dst_pitch = static_cast<uint32_t>(fr.dst_pitch[i]);
                    e.row_bytes = static_cast<uint32_t>(dw) * sb;
                    e.rows = static_cast<uint32_t>(dh);
                    e.unit = static_cast<uint32_t>(jinc::blit_unit(e.src, e.dst, e.src_pitch, e.dst_pitch));
                    max_rows = std::max(max_rows, e.rows);
                    max_row_bytes = std::max(max_row_bytes, e.row_bytes);
                }
            // shares of the group, each one launch + one event: a client collecting frames in order gets the first ones
            // while the later shares are still on the wire
            const int shares = std::min(n, group_shares());
            for (int s = 0; s < shares; ++s) {
                const int k0 = n * s / shares, k1 = n * (s + 1) / shares;
                hip_check(static_cast<hipError_t>(jinc::launch_blit_rows(g.table, k0 * planes, (k1 - k0) * planes, max_rows, max_row_bytes,
                                                                         blit_workgroups(f), d2h)),
                          "transport kernel launch");
                hip_check(hipEventRecord(g.done[static_cast<size_t>(s)], d2h), "hipEventRecord(share done)");
                for (int k = k0; k < k1; ++k) g.frames[static_cast<size_t>(k)].done_event = s;
            }
            g_frames_by_shader += n;
        } else {
            const unsigned every_plane = (1u << planes) - 1;
            bool all_staged = true;
            for (const GroupFrame& fr : g.frames) all_staged &= fr.staged_out == every_plane;
            if (all_staged && n == 1) {
                // a frame that travels alone: row bands with an event each (deliver_frame copies band j out while band j + 1 arrives)
                GroupFrame& fr = g.frames[0];
                fr.bands.clear();
                for (int i = 0; i < planes; ++i) {
                    int dw, dh;
                    f.plane_dims(f.vi_out, i, dw, dh);
                    const int nb = stage_bands(static_cast<size_t>(g.dst_pitch[i]) * dh, dh);
                    for (int b = 0; b < nb; ++b) {
                        const int y0 = dh * b / nb, y1 = dh * (b + 1) / nb;
                        const size_t off = static_cast<size_t>(g.dst_pitch[i]) * y0;
                        hip_check(hipMemcpyAsync(host_dst_plane(g, i, 0) + off, static_cast<const char*>(g.dst[i]) + off,
                                                 static_cast<size_t>(g.dst_pitch[i]) * (y1 - y0), hipMemcpyDeviceToHost, d2h),
                                  "D2H copy (staged band)");
                        const size_t e = fr.bands.size();
                        while (g.band_done.size() <= e) {
                            hipEvent_t ev = nullptr;
                            hip_check(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate");
                            g.band_done.push_back(ev);
                        }
                        hip_check(hipEventRecord(g.band_done[e], d2h), "hipEventRecord(band done)");
                        fr.bands.push_back({i, y0, y1, static_cast<int>(e)});
                    }
                }
                hip_check(hipEventRecord(g.done[0], d2h), "hipEventRecord(frame done)");
                fr.done_event = 0;
                g_frames_staged += 1;
            } else if (all_staged) {
                // every result plane of the group goes to the library's own pinned buffer, which is laid out like the device
                // buffer: one contiguous DMA copy per plane and share of the group, one event per share
                const int shares = std::min(n, group_shares());
                for (int s = 0; s < shares; ++s) {
                    const int k0 = n * s / shares, k1 = n * (s + 1) / shares;
                    for (int i = 0; i < planes; ++i)
                        hip_check(hipMemcpyAsync(host_dst_plane(g, i, static_cast<size_t>(k0)), static_cast<const char*>(g.dst[i]) + g.dst_fs[i] * k0,
                                                 g.dst_fs[i] * static_cast<size_t>(k1 - k0), hipMemcpyDeviceToHost, d2h),
                                  "D2H copy (staged)");
                    hip_check(hipEventRecord(g.done[static_cast<size_t>(s)], d2h), "hipEventRecord(share done)");
                    for (int k = k0; k < k1; ++k) g.frames[static_cast<size_t>(k)].done_event = s;
                }
                g_frames_staged += n;
            } else {
                for (int k = 0; k < n; ++k) {
                    GroupFrame& fr = g.frames[static_cast<size_t>(k)];
                    for (int i = 0; i < planes; ++i) {
                        int dw, dh;
                        f.plane_dims(f.vi_out, i, dw, dh);
                        const char* from = static_cast<const char*>(g.dst[i]) + g.dst_fs[i] * k;
                        if ((fr.staged_out >> i) & 1)
                            hip_check(hipMemcpyAsync(host_dst_plane(g, i, static_cast<size_t>(k)), from, static_cast<size_t>(g.dst_pitch[i]) * dh,
                                                     hipMemcpyDeviceToHost, d2h),
                                      "D2H copy (staged)");
                        else  // registered by the host or by this library, or mode 3: the runtime's business
                            hip_check(hipMemcpy2DAsync(fr.dst[i], fr.dst_pitch[i], from, g.dst_pitch[i], static_cast<size_t>(dw) * sb, dh,
                                                       hipMemcpyDeviceToHost, d2h),
                                      "D2H copy");
                    }
                    hip_check(hipEventRecord(g.done[static_cast<size_t>(k)], d2h), "hipEventRecord(frame done)");
                    fr.done_event = k;
                    (fr.staged_out ? g_frames_staged : g_frames_by_dma) += 1;
                }
            }
        }
        g.state = FrameGroup::Launched;
    } catch (const std::exception& e) {
        // whatever was queued must not touch the caller's planes after the error is reported
        (void)hipStreamSynchronize(kernels_of(f, g));
        if (d2h_of(f, g)) (void)hipStreamSynchronize(d2h_of(f, g));
        g.state = FrameGroup::Failed;
        g.error = e.what();
        throw;
    }
}

// Pins [p, p + bytes) once (cache keyed by address range, least recently used out) so that the copies of the pipeline
// really are asynchronous and the shader can reach the planes; returns p's address in the device's address space, or
// nullptr if the range could not be pinned (the frame then takes the DMA / pageable path -- not an error).  The cache
// holds at least every range the frames in flight can reference (frames x planes x (src + dst)), and a range whose frame
// may still be in flight is never unregistered under its transfer: its group is finished first.
// Round 6 built and withdrew another mode -- registrations made at submit, held by the frames in flight that use the range and
// given back with the last of them: logically the safe form (no registration outlives a buffer the host may release), and it
// passed every test of its own, buffers unmapped and mapped again at the same addresses between frames included.  Three of the
// nine full test runs that had it ended in GPU memory access faults inside the runtime's own copies from pageable memory, with no
// registration of this library alive at the time; later a run without it and a script with cached registrations faulted too
// (profiles/round6/README.md; the code: profiles/experiments/pin_while_in_flight.diff), which is why pageable planes now avoid
// the runtime's mapping altogether (ensure_host_staging above).  register_host == 2 means cached registrations, as in rounds 3-5.
char* pin_host_range(jinc_filter& f, const void* p, size_t bytes, long long ticket) {
    char* c = const_cast<char*>(static_cast<const char*>(p));
    for (size_t i = 0; i < f.pinned.size(); ++i) {
        PinnedRange& r = f.pinned[i];
        if (c < r.base || c + bytes > r.base + r.bytes) continue;
        if (!r.adopted && !shared_pin_alive(r.id)) {  // the registry found the range stale and let go of it: so does this instance
            shared_pin_release(r.id);
            f.pinned.erase(f.pinned.begin() + static_cast<std::ptrdiff_t>(i));
            break;
        }
        r.stamp = ++f.pin_clock;
        r.ticket = ticket;
        return r.dev ? r.dev + (c - r.base) : nullptr;
    }
    if (f.register_host != 2) return nullptr;  // only ranges the caller pinned are known: this one is pageable
    const size_t in_flight = f.groups.size() * static_cast<size_t>(f.group_frames);
    const size_t capacity = std::max<size_t>(64, in_flight * 8 + 8);
    size_t own = 0;
    for (const auto& r : f.pinned) own += r.adopted ? 0 : 1;
    if (own >= capacity) {
        size_t lru = f.pinned.size();
        for (size_t i = 0; i < f.pinned.size(); ++i)
            if (!f.pinned[i].adopted && (lru == f.pinned.size() || f.pinned[i].stamp < f.pinned[lru].stamp)) lru = i;
        const long long lru_ticket = f.pinned[lru].ticket;
        const unsigned long long lru_id = f.pinned[lru].id;
        for (auto& g : f.groups)  // its transfers may still run (a group being filled has its H2D copies queued already)
            for (size_t k = 0; k < g.frames.size(); ++k)
                if (g.frames[k].ticket == lru_ticket) {
                    if (g.state == FrameGroup::Filling || g.state == FrameGroup::Launched) finish_group(f, g);
                    break;
                }
        for (size_t i = 0; i < f.pinned.size(); ++i)
            if (!f.pinned[i].adopted && f.pinned[i].id == lru_id) {
                shared_pin_release(lru_id);
                f.pinned.erase(f.pinned.begin() + static_cast<std::ptrdiff_t>(i));
                break;
            }
    }
    char* base = nullptr;
    size_t len = 0;
    unsigned long long id = 0;
    if (!shared_pin_acquire(c, bytes, f.device, &id, &base, &len)) return nullptr;  // e.g. the range straddles memory somebody else has registered
    void* dev = nullptr;
    if (hipHostGetDevicePointer(&dev, base, 0) != hipSuccess) {
        (void)hipGetLastError();
        dev = nullptr;
    }
    f.pinned.push_back({base, len, static_cast<char*>(dev), false, ++f.pin_clock, ticket, id});
    return dev ? static_cast<char*>(dev) + (c - base) : nullptr;
}

}  // namespace

namespace {
void release_belts(jinc_filter& f) {
    if (f.h2d_stream) (void)hipStreamDestroy(f.h2d_stream);
    if (f.d2h_stream) (void)hipStreamDestroy(f.d2h_stream);
    f.h2d_stream = f.d2h_stream = nullptr;
}

// The arrivals and departures belts exist while groups are large enough to use them.  The departures belt runs at the
// highest priority: its kernel is bound by the link and needs few waves, but those must not queue behind the resampling
// kernels of the next group (measured: JINC_D2H_PRIORITY, profiles/round3/).
void ensure_belts(jinc_filter& f) {
    if (!use_belts(f)) {
        release_belts(f);
        return;
    }
    if (f.h2d_stream && f.d2h_stream) return;
    release_belts(f);  // (a half-made pair from a failed attempt)
    hipStream_t h2d = nullptr, d2h = nullptr;
    try {  // both or none: a missing departures belt would put transport, copies and events on the null stream
        hip_check(hipStreamCreateWithFlags(&h2d, hipStreamNonBlocking), "hipStreamCreate(arrivals)");
        int least = 0, greatest = 0;
        hip_check(hipDeviceGetStreamPriorityRange(&least, &greatest), "hipDeviceGetStreamPriorityRange");
        const int mode = knobs::geti(JINC_KNOB_D2H_PRIORITY, 1);  // A/B knob: 0 lowest, 1 highest (default), 2 normal
        if (mode == 2)
            hip_check(hipStreamCreateWithFlags(&d2h, hipStreamNonBlocking), "hipStreamCreate(departures)");
        else
            hip_check(hipStreamCreateWithPriority(&d2h, hipStreamNonBlocking, mode == 0 ? least : greatest), "hipStreamCreate(departures)");
    } catch (...) {
        if (h2d) (void)hipStreamDestroy(h2d);
        if (d2h) (void)hipStreamDestroy(d2h);
        throw;
    }
    f.h2d_stream = h2d;
    f.d2h_stream = d2h;
}
}  // namespace

void adopt_host_range(jinc_filter& f, void* base, size_t bytes) {
    char* c = static_cast<char*>(base);
    for (const auto& r : f.pinned)
        if (c >= r.base && c + bytes <= r.base + r.bytes) return;  // known already
    void* dev = nullptr;
    if (hipHostGetDevicePointer(&dev, c, 0) != hipSuccess) {
        (void)hipGetLastError();
        throw ArgError("JincResize: the host range is not pinned for this device (hipHostRegister / hipHostMalloc it first).");
    }
    // Neighbours that continue each other in host AND device addresses become one range: a registrar that pins a frame
    // pool piece by piece (jinc_batch_process: 16 frames at a time, clipped to what is pinned already) hands over pieces,
    // and a plane that begins in one piece and ends in the next must still lie inside ONE range to travel by the shader.
    PinnedRange nr{c, bytes, static_cast<char*>(dev), true, ++f.pin_clock, -1, 0};
    for (bool merged = true; merged;) {
        merged = false;
        for (size_t i = 0; i < f.pinned.size(); ++i) {
            PinnedRange& r = f.pinned[i];
            if (!r.adopted || !r.dev || !nr.dev) continue;
            const bool after = r.base + r.bytes == nr.base && r.dev + r.bytes == nr.dev;    // nr continues r
            const bool before = nr.base + nr.bytes == r.base && nr.dev + nr.bytes == r.dev;  // r continues nr
            if (!after && !before) continue;
            if (after) nr.base = r.base, nr.dev = r.dev;
            nr.bytes += r.bytes;
            f.pinned.erase(f.pinned.begin() + static_cast<std::ptrdiff_t>(i));
            merged = true;
            break;
        }
    }
    f.pinned.push_back(nr);
}

void release_host_range(jinc_filter& f, void* base, size_t bytes) {
    drain_pipeline(f);  // nothing of this instance may still travel through the range's mapping
    char* c = static_cast<char*>(base);
    f.pinned.erase(std::remove_if(f.pinned.begin(), f.pinned.end(),
                                  [&](const PinnedRange& r) { return r.adopted && c < r.base + r.bytes && c + bytes > r.base; }),
                   f.pinned.end());
}

void transport_counts(long long* by_shader, long long* by_dma, long long* pinned_ranges, bool reset) {
    if (by_shader) *by_shader = g_frames_by_shader.load();
    if (by_dma) *by_dma = g_frames_by_dma.load();
    if (pinned_ranges) {
        std::lock_guard<std::mutex> lock(pin_registry().mutex);
        *pinned_ranges = static_cast<long long>(pin_registry().pins.size());
    }
    if (reset) g_frames_by_shader = 0, g_frames_by_dma = 0, g_frames_staged = 0;
}

long long staged_frames() { return g_frames_staged.load(); }

void release_pipeline(jinc_filter& f) {
    for (hipStream_t s : {f.h2d_stream, f.stream, f.d2h_stream})
        if (s) (void)hipStreamSynchronize(s);
    for (auto& g : f.groups) {
        if (g.own_stream) (void)hipStreamSynchronize(g.own_stream);
        if (g.state == FrameGroup::Launched)  // frames in flight when the instance goes: their results still arrive
            for (size_t k = 0; k < g.frames.size(); ++k) {
                try {
                    deliver_frame(f, g, k);
                } catch (const std::exception&) {  // (a device error: nothing to deliver, nobody to tell)
                }
            }
        release_group(g);
    }
    release_belts(f);
    for (auto& p : f.pinned)  // this instance's references to the process-wide registry; the last one unregisters
        if (!p.adopted) shared_pin_release(p.id);
    f.pinned.clear();
}

void launch_open_group(jinc_filter& f) {
    if (f.open_group < 0) return;
    FrameGroup& g = f.groups[static_cast<size_t>(f.open_group)];
    f.open_group = -1;
    if (g.state == FrameGroup::Filling && !g.frames.empty()) launch_group(f, g);
}

void drain_pipeline(jinc_filter& f) {
    launch_open_group(f);
    for (auto& g : f.groups) finish_group(f, g);
}

void configure_pipeline(jinc_filter& f, int depth, int group, int register_host) {
    drain_pipeline(f);
    for (auto& g : f.groups) retire_group(f, g);
    f.failed.clear();  // the explicit reset: failures of the previous configuration are not carried over
    depth = std::max(1, std::min(depth, kMaxPipelineDepth));
    // Frames per launch unless the caller says otherwise: half the frames in flight, so that one group computes while the
    // client still collects the previous one -- from 8 frames in flight on; below that single frames on a stream each
    // (the round-2 shape) overlap better than pairs (host to host, one thread, round4/e2e_small_depths.log: depth 4 in pairs
    // -4 ... -13 %; depth 8 in fours +2 % on 1080p -> 4K, +10 % on 1.37x; depth 12 in sixes +5 % / +24 %: plans without phase
    // structure run a group of 3 or more frames on the frame-lane kernel's sub-group form instead of a gather launch per frame).
    if (group <= 0) group = depth >= 8 ? depth / 2 : 1;
    group = std::min(group, depth);
    const size_t per_frame = std::max<size_t>(1, staging_bytes_per_frame(f));
    auto ring_of = [&](int g) { return (depth + g - 1) / g + 1; };
    // (pageable planes are staged in pinned HOST memory of the same size: a smaller budget there)
    const size_t budget = register_host == 0 ? kHostStagingBudgetBytes : kPipelineBudgetBytes;
    while (group > 1 && per_frame * static_cast<size_t>(group) * static_cast<size_t>(ring_of(group)) > budget) group /= 2;
    if (group != f.group_frames)
        for (auto& g : f.groups) release_group(g);
    f.pipeline_depth = depth;
    f.group_frames = group;
    const size_t ring = static_cast<size_t>(ring_of(group));
    while (f.groups.size() > ring) {
        release_group(f.groups.back());
        f.groups.pop_back();
    }
    f.groups.resize(ring);
    f.open_group = -1;
    f.last_group = 0;
    f.register_host = register_host == 0 || register_host == 3 ? register_host : 2;
    if (f.register_host != 2) {  // cached registrations go (the pipeline is drained); ranges the caller pinned stay known
        for (auto& p : f.pinned)
            if (!p.adopted) shared_pin_release(p.id);
        f.pinned.erase(std::remove_if(f.pinned.begin(), f.pinned.end(), [](const PinnedRange& r) { return !r.adopted; }), f.pinned.end());
    }
    ensure_belts(f);
}

long long submit_frame(jinc_filter& f, const void* const src[4], const int src_pitch[4], void* const dst[4], const int dst_pitch[4]) {
    const int sb = f.vi_in.component_size;
    for (int i = 0; i < f.planecount; ++i) {
        if (!src[i] || !dst[i]) throw ArgError("JincResize: null plane pointer.");
        int sw, dw, h;
        f.plane_dims(f.vi_in, i, sw, h);
        f.plane_dims(f.vi_out, i, dw, h);
        if (static_cast<size_t>(src_pitch[i]) < static_cast<size_t>(sw) * sb || static_cast<size_t>(dst_pitch[i]) < static_cast<size_t>(dw) * sb)
            throw ArgError("JincResize: plane pitch is smaller than the row size.");
    }
    if (f.open_group < 0) {  // open the next buffer of the ring; its previous frames have to be finished first
        const int next = (f.last_group + 1) % static_cast<int>(f.groups.size());
        FrameGroup& g = f.groups[static_cast<size_t>(next)];
        retire_group(f, g);
        ensure_belts(f);
        ensure_group(f, g);
        g.state = FrameGroup::Filling;
        f.open_group = f.last_group = next;
    }
    FrameGroup& g = f.groups[static_cast<size_t>(f.open_group)];
    const size_t k = g.frames.size();
    const long long ticket = f.next_ticket;
    GroupFrame fr;
    fr.ticket = ticket;
    // Frames whose source planes are small do not fill the copy lanes one by one, and one DMA copy per plane and frame costs more
    // in engine turnaround than on the wire: in a group of four or more they wait for the launch (A/B knob STAGE_DEFER_KB: frames
    // up to this many KiB of source, default 1536; 0 = never).
    size_t frame_src_bytes = 0;
    for (int i = 0; i < f.planecount; ++i) {
        int sw, sh;
        f.plane_dims(f.vi_in, i, sw, sh);
        frame_src_bytes += static_cast<size_t>(sw) * sb * sh;
    }
    const bool defer_small = g.capacity >= 4 && frame_src_bytes <= static_cast<size_t>(std::max(0, knobs::geti(JINC_KNOB_STAGE_DEFER_KB, 1536))) << 10;
    for (int i = 0; i < f.planecount; ++i) {
        int sw, sh, dw, dh;
        f.plane_dims(f.vi_in, i, sw, sh);
        f.plane_dims(f.vi_out, i, dw, dh);
        bool src_mapped = f.register_host == 3;  // (mode 3: whatever the plane is, the runtime takes it as it is)
        if (f.register_host == 2 || !f.pinned.empty()) {
            src_mapped |= pin_host_range(f, src[i], static_cast<size_t>(src_pitch[i]) * (sh - 1) + static_cast<size_t>(sw) * sb, ticket) != nullptr;
            fr.dst_dev[i] = pin_host_range(f, dst[i], static_cast<size_t>(dst_pitch[i]) * (dh - 1) + static_cast<size_t>(dw) * sb, ticket);
        }
        if (!src_mapped || (!fr.dst_dev[i] && f.register_host != 3)) ensure_host_staging(f, g);
        if (!fr.dst_dev[i] && f.register_host != 3) fr.staged_out |= 1u << i;
        char* dev_plane = static_cast<char*>(g.src[i]) + g.src_fs[i] * k;
        if (debug_skip() == 1) {
        } else if (src_mapped) {
            hip_check(hipMemcpy2DAsync(dev_plane, g.src_pitch[i], src[i], src_pitch[i], static_cast<size_t>(sw) * sb, sh, hipMemcpyHostToDevice,
                                       h2d_of(f, g)),
                      "H2D copy");
        } else if (defer_small) {  // pageable and small, in company: copied with the rest of the group at its launch
            fr.src[i] = src[i];
            fr.src_pitch[i] = src_pitch[i];
            fr.deferred_in |= 1u << i;
        } else {  // pageable: the CPU copies the rows into the group's pinned buffer (its previous use is over: retire_group)
            char* staged = host_src_plane(g, i, k);
            const int nb = stage_bands(static_cast<size_t>(g.src_pitch[i]) * sh, sh);  // (band j travels while the CPU copies band j + 1)
            for (int b = 0; b < nb; ++b) {
                const int y0 = sh * b / nb, y1 = sh * (b + 1) / nb;
                const size_t off = static_cast<size_t>(g.src_pitch[i]) * y0;
                copy_plane_rows(staged + off, static_cast<size_t>(g.src_pitch[i]), static_cast<const char*>(src[i]) + static_cast<size_t>(src_pitch[i]) * y0,
                                static_cast<size_t>(src_pitch[i]), static_cast<size_t>(sw) * sb, y1 - y0, f.copy_helpers);
                hip_check(hipMemcpyAsync(dev_plane + off, staged + off, static_cast<size_t>(g.src_pitch[i]) * (y1 - y0), hipMemcpyHostToDevice, h2d_of(f, g)),
                          "H2D copy (staged)");
            }
        }
        fr.dst[i] = dst[i];
        fr.dst_pitch[i] = dst_pitch[i];
    }
    g.frames.push_back(fr);
    ++f.next_ticket;
    if (static_cast<int>(g.frames.size()) >= g.capacity) launch_group(f, g);
    return ticket;
}

void wait_frame(jinc_filter& f, long long ticket) {
    for (auto& g : f.groups) {
        for (size_t k = 0; k < g.frames.size(); ++k) {
            if (g.frames[k].ticket != ticket) continue;
            if (g.state == FrameGroup::Filling) launch_group(f, g);  // the client wants this frame now: no more company
            if (g.state == FrameGroup::Failed) throw HipError(g.error);
            if (g.frames[k].bands.empty() || g.frames[k].delivered)
                hip_check(hipEventSynchronize(g.done[static_cast<size_t>(g.frames[k].done_event)]), "hipEventSynchronize(frame done)");
            deliver_frame(f, g, k);  // (a banded frame: waits band by band; its last band's event is the frame's)
            return;
        }
    }
    for (const auto& r : f.failed)
        if (ticket >= r.first && ticket <= r.last) throw HipError(r.error);
    if (ticket < 0 || ticket >= f.next_ticket) throw ArgError("JincResize: no frame was submitted under this ticket.");
    // long completed ticket: nothing to wait for (its group buffer has been reused, which waited for it)
}

}  // namespace host
}  // namespace jinc
