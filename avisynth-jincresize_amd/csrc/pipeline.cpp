// pipeline.cpp -- frames in flight on one filter instance: device staging slots, pinned host ranges, and the
// H2D -> kernels -> D2H sequence of one frame (frame transport around GetFrame).
#include "filter_internal.h"

namespace jinc {
namespace host {

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
namespace {
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

}  // namespace

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

}  // namespace host
}  // namespace jinc
