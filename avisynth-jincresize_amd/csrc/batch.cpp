// batch.cpp -- jinc_batch_*: one clip's frames sharded over the HIP devices of a node (BASELINE.json
// configs[4]: "batch of 512 independent frames sharded across 8 x MI355X, per-device streams").
//
// Frames are independent units (JincResize_GetFrame touches frame n only, ref /root/reference/src/JincResize.cpp:603-630)
// and the plan is read-only, so the shard needs no exchange between devices: frame n goes to device n mod G; every
// device owns a replica of the plan (one jinc_filter) with `streams` frames in flight through the look-ahead pipeline of
// pipeline.cpp -- which coalesces them into groups of streams / 2 frames per launch, so a device's share of the clip runs
// on the batch kernels -- driven by one host thread per device.  No RCCL, no peer traffic.
//
// Host memory: with register_host_buffers the caller's planes are pinned HERE, once for all devices, by a registrar thread
// that runs ahead of the submitting threads: the planes of kPinChunk frames at a time are rounded out to pages, sorted and
// merged where they touch (frames allocated one after the other usually do), so that one hipHostRegister covers many
// planes -- a call costs ~60 us plus ~5 us per MiB (profiles/round3/hostreg_probe.log), per plane it was the larger part
// of a C2 frame's 160 us on the link.  The filters adopt the pinned ranges (jinc_filter_adopt_host_range).
// Built on the public C ABI (jinc_filter_create / _set_pipeline / _adopt_host_range / _submit / _flush / _wait) and
// hipHostRegister only.
#include <hip/hip_runtime_api.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/jincresize_hip.h"

namespace {
struct HostRange {
    char* base = nullptr;
    size_t bytes = 0;
};
constexpr int kPinChunk = 16;  // frames whose planes are pinned together
}  // namespace

struct jinc_batch {
    std::vector<jinc_filter*> filters;  // one per device
    std::vector<int> devices;
    int streams = 2;
    int planes = 0;
    bool register_host = false;
    jinc_video_info vi_in{}, vi_out{};
    std::vector<HostRange> pinned;  // registered here (portable: every device), unregistered by jinc_batch_free
    std::mutex pin_mutex;
};

namespace {
thread_local std::string g_batch_error;
int batch_fail(int code, const std::string& msg, char* err = nullptr, size_t err_len = 0) {
    g_batch_error = msg;
    if (err && err_len) {
        std::strncpy(err, msg.c_str(), err_len - 1);
        err[err_len - 1] = '\0';
    }
    return code;
}

void plane_dims(const jinc_video_info& vi, int i, int& w, int& h) {  // as jinc_filter::plane_dims (ref :546-550)
    w = vi.width;
    h = vi.height;
    const bool subsampled = vi.num_components >= 3 && !vi.is_rgb && (vi.sub_w || vi.sub_h);
    if (subsampled && (i == 1 || i == 2)) {
        w >>= vi.sub_w;
        h >>= vi.sub_h;
    }
}

// The planes of frames [n0, n1), rounded out to pages, merged where they touch or overlap, minus what is pinned already.
std::vector<HostRange> ranges_to_pin(jinc_batch& b, int n0, int n1, const void* const* src_planes, const int src_pitch[4],
                                     void* const* dst_planes, const int dst_pitch[4]) {
    static const uintptr_t page = static_cast<uintptr_t>(sysconf(_SC_PAGESIZE) > 0 ? sysconf(_SC_PAGESIZE) : 4096);
    std::vector<std::pair<uintptr_t, uintptr_t>> spans;
    auto add = [&](const void* p, int pitch, int w, int h, int sb) {
        if (!p || h <= 0) return;
        const uintptr_t a = reinterpret_cast<uintptr_t>(p), e = a + static_cast<size_t>(pitch) * (h - 1) + static_cast<size_t>(w) * sb;
        spans.emplace_back(a / page * page, (e + page - 1) / page * page);
    };
    for (int n = n0; n < n1; ++n)
        for (int i = 0; i < b.planes && i < 4; ++i) {
            int w, h;
            plane_dims(b.vi_in, i, w, h);
            add(src_planes[static_cast<size_t>(n) * 4 + i], src_pitch[i], w, h, b.vi_in.component_size);
            plane_dims(b.vi_out, i, w, h);
            add(dst_planes[static_cast<size_t>(n) * 4 + i], dst_pitch[i], w, h, b.vi_in.component_size);
        }
    std::sort(spans.begin(), spans.end());
    std::vector<std::pair<uintptr_t, uintptr_t>> merged;
    for (const auto& s : spans) {
        if (!merged.empty() && s.first <= merged.back().second) merged.back().second = std::max(merged.back().second, s.second);
        else merged.push_back(s);
    }
    std::vector<HostRange> out;
    std::lock_guard<std::mutex> lock(b.pin_mutex);
    for (auto m : merged) {
        for (const HostRange& p : b.pinned) {  // clip against what an earlier chunk pinned (a shared page at most)
            const uintptr_t pa = reinterpret_cast<uintptr_t>(p.base), pe = pa + p.bytes;
            if (m.first >= pa && m.first < pe) m.first = std::min(pe, m.second);
            if (m.second > pa && m.second <= pe) m.second = std::max(pa, m.first);
        }
        if (m.second > m.first) out.push_back({reinterpret_cast<char*>(m.first), static_cast<size_t>(m.second - m.first)});
    }
    return out;
}
}  // namespace

extern "C" {

const char* jinc_batch_last_error(void) { return g_batch_error.c_str(); }

int jinc_batch_create(const jinc_video_info* vi, const jinc_args* args, int ndevices, int streams_per_device, int register_host_buffers,
                      jinc_batch** out, char* err, size_t err_len) {
    if (out) *out = nullptr;
    if (err && err_len) err[0] = '\0';
    if (!vi || !args || !out) return batch_fail(JINC_ERR_INVALID_ARG, "JincResize: null argument.");
    const int avail = jinc_device_count();
    if (avail <= 0) return batch_fail(JINC_ERR_NO_DEVICE, "JincResize: no HIP device available.", err, err_len);
    if (ndevices <= 0 || ndevices > avail) ndevices = avail;
    if (streams_per_device < 1 || streams_per_device > 256)
        return batch_fail(JINC_ERR_INVALID_ARG, "JincResize: frames in flight per device must be 1..256.", err, err_len);
    jinc_batch* b = new (std::nothrow) jinc_batch();
    if (!b) return batch_fail(JINC_ERR_NOMEM, "JincResize: out of memory.");
    b->streams = streams_per_device;
    b->planes = vi->num_components;
    b->register_host = register_host_buffers != 0;
    b->vi_in = *vi;
    for (int d = 0; d < ndevices; ++d) {
        jinc_filter* f = nullptr;
        int rc = jinc_filter_create(vi, args, d, &f, err, err_len);
        if (rc == JINC_OK) rc = jinc_filter_set_pipeline(f, streams_per_device, 0);  // host memory is pinned here, not per instance
        if (rc != JINC_OK) {
            if (f) {
                if (err && err_len) {
                    std::strncpy(err, jinc_last_error(), err_len - 1);
                    err[err_len - 1] = '\0';
                }
                jinc_filter_free(f);
            }
            g_batch_error = err && err_len ? err : jinc_last_error();
            for (jinc_filter* g : b->filters) jinc_filter_free(g);
            delete b;
            return rc;
        }
        b->filters.push_back(f);
        b->devices.push_back(d);
    }
    jinc_filter_output_info(b->filters[0], &b->vi_out);
    *out = b;
    return JINC_OK;
}

void jinc_batch_free(jinc_batch* b) {
    if (!b) return;
    for (jinc_filter* f : b->filters) jinc_filter_free(f);  // (waits for frames in flight)
    for (const HostRange& r : b->pinned) (void)hipHostUnregister(r.base);
    delete b;
}

int jinc_batch_devices(const jinc_batch* b) { return b ? static_cast<int>(b->filters.size()) : 0; }

int jinc_shard_device(int frame, int ndevices) { return (frame < 0 || ndevices <= 0) ? -1 : frame % ndevices; }

int jinc_batch_device_of_frame(const jinc_batch* b, int n) {
    if (!b || b->filters.empty() || n < 0) return -1;
    return b->devices[static_cast<size_t>(jinc_shard_device(n, static_cast<int>(b->filters.size())))];
}

int jinc_batch_process(jinc_batch* b, int nframes, const void* const* src_planes, const int src_pitch[4], void* const* dst_planes,
                       const int dst_pitch[4]) {
    if (!b || !src_planes || !dst_planes || !src_pitch || !dst_pitch || nframes < 0)
        return batch_fail(JINC_ERR_INVALID_ARG, "JincResize: null argument.");
    const int G = static_cast<int>(b->filters.size());
    std::atomic<int> status{JINC_OK};
    std::mutex err_mutex;
    std::string first_error;

    // Registrar: pins the chunks in frame order, ahead of the workers; pinned_chunks = how many are done.
    const int nchunks = b->register_host ? (nframes + kPinChunk - 1) / kPinChunk : 0;
    std::mutex chunk_mutex;
    std::condition_variable chunk_cv;
    int pinned_chunks = 0;
    std::vector<std::vector<HostRange>> chunk_ranges(static_cast<size_t>(nchunks));  // what each chunk added (for adoption)
    std::thread registrar;
    if (nchunks > 0)
        registrar = std::thread([&] {
            (void)hipSetDevice(b->devices[0]);
            for (int c = 0; c < nchunks; ++c) {
                const int n0 = c * kPinChunk, n1 = std::min(nframes, n0 + kPinChunk);
                std::vector<HostRange> todo = ranges_to_pin(*b, n0, n1, src_planes, src_pitch, dst_planes, dst_pitch), done;
                for (const HostRange& r : todo) {
                    if (hipHostRegister(r.base, r.bytes, hipHostRegisterPortable) == hipSuccess) done.push_back(r);
                    else (void)hipGetLastError();  // e.g. pinned by somebody else already: those planes take the pageable path
                }
                {
                    std::lock_guard<std::mutex> lock(b->pin_mutex);
                    b->pinned.insert(b->pinned.end(), done.begin(), done.end());
                }
                {
                    std::lock_guard<std::mutex> lock(chunk_mutex);
                    chunk_ranges[static_cast<size_t>(c)] = std::move(done);
                    pinned_chunks = c + 1;
                }
                chunk_cv.notify_all();
            }
        });

    auto worker = [&](int d) {
        jinc_filter* f = b->filters[static_cast<size_t>(d)];
        std::vector<long long> tickets;  // frames in flight on this device, oldest first
        auto wait_oldest = [&]() {
            const int rc = jinc_filter_wait(f, tickets.front());
            tickets.erase(tickets.begin());
            return rc;
        };
        int rc = JINC_OK, adopted_chunks = 0;
        if (nchunks > 0) {  // ranges pinned by earlier calls
            std::vector<HostRange> known;
            {
                std::lock_guard<std::mutex> lock(b->pin_mutex);
                known = b->pinned;
            }
            for (const HostRange& r : known) (void)jinc_filter_adopt_host_range(f, r.base, r.bytes);
        }
        for (int n = d; n < nframes && rc == JINC_OK && status.load() == JINC_OK; n += G) {  // jinc_shard_device(n, G) == d
            if (nchunks > 0 && n / kPinChunk >= adopted_chunks) {  // the frame's planes have to be pinned and known to this instance
                std::unique_lock<std::mutex> lock(chunk_mutex);
                chunk_cv.wait(lock, [&] { return pinned_chunks > n / kPinChunk; });
                for (; adopted_chunks <= n / kPinChunk; ++adopted_chunks)
                    for (const HostRange& r : chunk_ranges[static_cast<size_t>(adopted_chunks)])
                        (void)jinc_filter_adopt_host_range(f, r.base, r.bytes);  // failure: pageable path for those planes
            }
            if (static_cast<int>(tickets.size()) >= b->streams) rc = wait_oldest();
            if (rc != JINC_OK) break;
            const void* s[4] = {nullptr, nullptr, nullptr, nullptr};
            void* t[4] = {nullptr, nullptr, nullptr, nullptr};
            for (int i = 0; i < b->planes && i < 4; ++i) {
                s[i] = src_planes[static_cast<size_t>(n) * 4 + i];
                t[i] = dst_planes[static_cast<size_t>(n) * 4 + i];
            }
            long long ticket = -1;
            rc = jinc_filter_submit(f, s, src_pitch, t, dst_pitch, &ticket);
            if (rc == JINC_OK) tickets.push_back(ticket);
        }
        if (rc == JINC_OK) rc = jinc_filter_flush(f);  // the device's last frames do not wait for company
        while (!tickets.empty()) {  // drain, also after an error: buffers must not be in use when we return
            const int w = wait_oldest();
            if (rc == JINC_OK) rc = w;
        }
        if (rc != JINC_OK) {
            int expected = JINC_OK;
            if (status.compare_exchange_strong(expected, rc)) {
                std::lock_guard<std::mutex> lock(err_mutex);
                first_error = jinc_last_error();  // thread-local of this worker
            }
        }
    };
    std::vector<std::thread> threads;
    for (int d = 1; d < G; ++d) threads.emplace_back(worker, d);
    worker(0);
    for (auto& t : threads) t.join();
    if (registrar.joinable()) registrar.join();
    if (status.load() != JINC_OK) return batch_fail(status.load(), first_error);
    g_batch_error.clear();
    return JINC_OK;
}

}  // extern "C"
