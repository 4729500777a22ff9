// batch.cpp -- jinc_batch_*: one clip's frames sharded over the HIP devices of a node (BASELINE.json
// configs[4]: "batch of 512 independent frames sharded across 8 x MI355X, per-device streams").
//
// Frames are independent units (JincResize_GetFrame touches frame n only, ref /root/reference/src/JincResize.cpp:603-630)
// and the plan is read-only, so the shard needs no exchange between devices: frame n goes to device n mod G; every
// device owns a replica of the plan (one jinc_filter) with `streams` frames in flight through the look-ahead pipeline of
// pipeline.cpp -- which coalesces them into groups of streams / 2 frames per launch, so a device's share of the clip runs
// on the batch kernels -- driven by one host thread per device.  No RCCL, no peer traffic.
// Built on the public C ABI only (jinc_filter_create / _set_pipeline / _submit / _flush / _wait).
#include <algorithm>
#include <atomic>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/jincresize_hip.h"

struct jinc_batch {
    std::vector<jinc_filter*> filters;  // one per device
    std::vector<int> devices;
    int streams = 2;
    int planes = 0;
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
    for (int d = 0; d < ndevices; ++d) {
        jinc_filter* f = nullptr;
        int rc = jinc_filter_create(vi, args, d, &f, err, err_len);
        if (rc == JINC_OK) rc = jinc_filter_set_pipeline(f, streams_per_device, register_host_buffers);
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
    *out = b;
    return JINC_OK;
}

void jinc_batch_free(jinc_batch* b) {
    if (!b) return;
    for (jinc_filter* f : b->filters) jinc_filter_free(f);
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
    auto worker = [&](int d) {
        jinc_filter* f = b->filters[static_cast<size_t>(d)];
        std::vector<long long> tickets;  // frames in flight on this device, oldest first
        auto wait_oldest = [&]() {
            const int rc = jinc_filter_wait(f, tickets.front());
            tickets.erase(tickets.begin());
            return rc;
        };
        int rc = JINC_OK;
        for (int n = d; n < nframes && rc == JINC_OK && status.load() == JINC_OK; n += G) {  // jinc_shard_device(n, G) == d
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
    if (status.load() != JINC_OK) return batch_fail(status.load(), first_error);
    g_batch_error.clear();
    return JINC_OK;
}

}  // extern "C"
