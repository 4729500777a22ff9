// batch.cpp -- jinc_batch_*: one clip's frames sharded over the HIP devices of a node (BASELINE.json
// configs[4]: "batch of 512 independent frames sharded across 8 x MI355X, per-device streams").
//
// Frames are independent units (JincResize_GetFrame touches frame n only, ref /root/reference/src/JincResize.cpp:603-630)
// and the plan is read-only, so the shard needs no exchange between devices: frame n goes to device n mod G; every
// device owns a replica of the plan (one jinc_filter) with `streams` frames in flight through the look-ahead pipeline of
// pipeline.cpp -- which coalesces them into groups of streams / 2 frames per launch, so a device's share of the clip runs
// on the batch kernels -- driven by one host thread per device.  No RCCL, no peer traffic.
//
// Host memory: with register_host_buffers the caller's planes are pinned HERE (portable registrations: every device) by
// registrar threads that run ahead of the submitting threads: the planes of kPinChunk frames at a time are rounded out to
// pages, sorted and merged where they touch (frames allocated one after the other usually do), so that one hipHostRegister
// covers many planes -- a call costs ~60 us plus ~5 us per MiB (profiles/round3/hostreg_probe.log), per plane it was the
// larger part of a C2 frame's 160 us on the link.  The filters adopt the pinned ranges (jinc_filter_adopt_host_range).
// Round 6: ONE registrar served every device (~0.9 ms per 16-frame C2 chunk = 18 k frames/s in total,
// three devices' worth) -- now one registrar per device, dealing the chunks round-robin, and every worker / registrar of
// device d runs on the CPUs of d's NUMA node (sysfs: /sys/bus/pci/devices/<bdf>/numa_node; jinc_batch_set_affinity(b, 0)
// switches the binding off).  Registrations stay until jinc_batch_free (a caller that re-uses its planes pays once; ending
// them with every call was tried in round 6 and withdrawn with the per-frame form: registration churn, pipeline.cpp).
// Built on the public C ABI (jinc_filter_create / _set_pipeline / _adopt_host_range / _release_host_range / _submit /
// _flush / _wait) and hipHostRegister only.
#include <hip/hip_runtime_api.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/jincresize_hip.h"
#include "../../include/jincresize_hip_test.h"
#include "knobs.h"

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
    int register_host = 0;  // 0: pageable planes, copied by the CPU through the instances' own pinned buffers; 3: pageable planes handed to the HIP runtime as they are; 2: the calls' planes are pinned and stay so until jinc_batch_free (1, pinned for the duration of a call, was built and withdrawn in round 6: registration churn, see pipeline.cpp pin_host_range)
    bool affinity = true;   // workers and registrars of device d run on the CPUs of d's NUMA node
    int registrars = 0;     // 0: one per device; > 0: that many (test header: several registrars on a one-device box)
    std::atomic<int> refused{0};  // ranges hipHostRegister would not take (their planes travel pageable), since creation
    std::string first_refusal;    // what the first of them was told (under pin_mutex)
    std::vector<std::vector<int>> device_cpus;  // [device index] CPUs of its NUMA node; empty: unknown / no NUMA -> no binding
    jinc_video_info vi_in{}, vi_out{};
    std::vector<HostRange> pinned;  // registered here (portable: every device), unregistered by jinc_batch_free (mode 2) or at the end of the call (mode 1)
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

// "0-15,64-79" -> CPU numbers (the kernel's cpulist format)
std::vector<int> parse_cpulist(const char* text) {
    std::vector<int> out;
    const char* p = text;
    while (*p) {
        while (*p == ',' || *p == ' ' || *p == '\n') ++p;
        if (*p < '0' || *p > '9') break;
        char* end = nullptr;
        const long a = std::strtol(p, &end, 10);
        long b2 = a;
        p = end;
        if (*p == '-') {
            b2 = std::strtol(p + 1, &end, 10);
            p = end;
        }
        for (long c = a; c <= b2 && c - a < 4096; ++c) out.push_back(static_cast<int>(c));
    }
    return out;
}

std::string read_small_file(const std::string& path) {
    std::string out;
    if (FILE* fp = std::fopen(path.c_str(), "r")) {
        char buf[4096];
        const size_t n = std::fread(buf, 1, sizeof(buf) - 1, fp);
        buf[n] = '\0';
        out = buf;
        std::fclose(fp);
    }
    return out;
}

// CPUs of the NUMA node the PCI device `bdf` ("0000:c1:00.0") hangs on, read under `sysfs_root` ("/sys"); empty when the
// device or its node is unknown (numa_node = -1: a machine without NUMA) -- the caller then leaves the affinity alone.
std::vector<int> numa_cpus_of_pci_device(const std::string& sysfs_root, const std::string& bdf) {
    std::string lower = bdf;
    for (char& ch : lower) ch = static_cast<char>(ch >= 'A' && ch <= 'Z' ? ch - 'A' + 'a' : ch);
    const std::string node_text = read_small_file(sysfs_root + "/bus/pci/devices/" + lower + "/numa_node");
    if (node_text.empty()) return {};
    const long node = std::strtol(node_text.c_str(), nullptr, 10);
    if (node < 0) return {};
    return parse_cpulist(read_small_file(sysfs_root + "/devices/system/node/node" + std::to_string(node) + "/cpulist").c_str());
}

void bind_this_thread(const std::vector<int>& cpus) {
    if (cpus.empty()) return;
    cpu_set_t allowed, want;
    CPU_ZERO(&want);
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return;
    int n = 0;
    for (int c : cpus)
        if (c >= 0 && c < CPU_SETSIZE && CPU_ISSET(c, &allowed)) {  // never outside what the process was given (cgroup cpuset)
            CPU_SET(c, &want);
            ++n;
        }
    if (n > 0) (void)sched_setaffinity(0, sizeof(want), &want);
}

// The planes of frames [n0, n1) as byte spans, merged where they (nearly) touch or overlap; without the spans an earlier range covers.
std::vector<HostRange> ranges_to_pin(jinc_batch& b, int n0, int n1, const void* const* src_planes, const int src_pitch[4],
                                     void* const* dst_planes, const int dst_pitch[4]) {
    // EXACT byte spans, not rounded out to pages: the runtime looks host pointers up byte by byte, and a rounded-out range would
    // take in whatever else lives in the first and last page -- another buffer of the host that begins there and runs past the
    // range's end can then no longer be copied from (hipErrorInvalidValue: seen on the library's own coefficient upload while a
    // batch held page-rounded ranges, profiles/round6/hostreg_semantics.log 4b / 4d).  Planes that follow each other with less
    // than kMergeGap bytes between them (allocator headers, the padding of a frame pool) still become one registration: what
    // lies in such a gap is wholly inside the range and stays valid for as long as the range is registered.
    constexpr uintptr_t kMergeGap = 4096;
    std::vector<std::pair<uintptr_t, uintptr_t>> spans;
    auto add = [&](const void* p, int pitch, int w, int h, int sb) {
        if (!p || h <= 0) return;
        const uintptr_t a = reinterpret_cast<uintptr_t>(p), e = a + static_cast<size_t>(pitch) * (h - 1) + static_cast<size_t>(w) * sb;
        spans.emplace_back(a, e);
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
        if (!merged.empty() && s.first <= merged.back().second + kMergeGap) merged.back().second = std::max(merged.back().second, s.second);
        else merged.push_back(s);
    }
    std::vector<HostRange> out;
    std::lock_guard<std::mutex> lock(b.pin_mutex);
    // Every span is registered WHOLE, so that each plane lies inside one registered object: the runtime refuses a copy whose host
    // range starts in one registered object and runs past its end (profiles/round6/hostreg_semantics.log, 4b / 4d / 6b).
    // Left out: a span that an earlier range already contains (the same buffers again), and a span that would START where an
    // earlier range starts (the runtime keys its objects by their start).
    for (const auto& m : merged) {
        bool skip = false;
        for (const HostRange& p : b.pinned) {
            const uintptr_t pa = reinterpret_cast<uintptr_t>(p.base), pe = pa + p.bytes;
            if ((m.first >= pa && m.second <= pe) || m.first == pa) {
                skip = true;
                break;
            }
        }
        if (!skip) out.push_back({reinterpret_cast<char*>(m.first), static_cast<size_t>(m.second - m.first)});
    }
    // claimed at once, in chunk order (jinc_batch_process settles every chunk's list before the registrars start)
    b.pinned.insert(b.pinned.end(), out.begin(), out.end());
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
    b->register_host = register_host_buffers == 0 || register_host_buffers == 3 ? register_host_buffers : 2;
    b->vi_in = *vi;
    for (int d = 0; d < ndevices; ++d) {
        jinc_filter* f = nullptr;
        int rc = jinc_filter_create(vi, args, d, &f, err, err_len);
        if (rc == JINC_OK) rc = jinc_filter_set_pipeline(f, streams_per_device, b->register_host == 3 ? 3 : 0);  // host memory is pinned here, not per instance
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
        char bdf[64] = {0};
        std::vector<int> cpus;
        if (hipDeviceGetPCIBusId(bdf, static_cast<int>(sizeof(bdf)), d) == hipSuccess) cpus = numa_cpus_of_pci_device("/sys", bdf);
        else (void)hipGetLastError();
        b->device_cpus.push_back(cpus);
    }
    jinc_filter_output_info(b->filters[0], &b->vi_out);
    *out = b;
    return JINC_OK;
}

void jinc_batch_free(jinc_batch* b) {
    if (!b) return;
    for (jinc_filter* f : b->filters) jinc_filter_free(f);  // (waits for frames in flight)
    for (const HostRange& r : b->pinned) {
        std::lock_guard<std::mutex> one(jinc::knobs::host_registration_mutex());
        if (hipHostUnregister(r.base) == hipSuccess) jinc::knobs::count_host_registration(-1);
        else (void)hipGetLastError();  // (the sticky error must not meet the caller's next launch)
    }
    delete b;
}

int jinc_batch_set_affinity(jinc_batch* b, int on) {
    if (!b) return batch_fail(JINC_ERR_INVALID_ARG, "JincResize: null argument.");
    b->affinity = on != 0;
    return JINC_OK;
}

int jinc_batch_device_cpus(const jinc_batch* b, int device_index, int* cpus, int max_cpus) {
    if (!b || device_index < 0 || device_index >= static_cast<int>(b->device_cpus.size())) return -1;
    const auto& v = b->device_cpus[static_cast<size_t>(device_index)];
    for (int i = 0; i < max_cpus && i < static_cast<int>(v.size()); ++i)
        if (cpus) cpus[i] = v[static_cast<size_t>(i)];
    return static_cast<int>(v.size());
}

int jinc_debug_batch_set_registrars(jinc_batch* b, int n) {
    if (!b || n < 0 || n > 64) return batch_fail(JINC_ERR_INVALID_ARG, "JincResize: registrars must be 0..64.");
    b->registrars = n;
    return JINC_OK;
}

int jinc_debug_batch_refused(jinc_batch* b, char* first, size_t first_len) {
    if (!b) return -1;
    std::lock_guard<std::mutex> lock(b->pin_mutex);
    if (first && first_len) {
        std::strncpy(first, b->first_refusal.c_str(), first_len - 1);
        first[first_len - 1] = '\0';
    }
    return b->refused.load();
}

long long jinc_debug_host_registrations(void) { return jinc::knobs::live_host_registrations(); }

int jinc_debug_numa_cpus(const char* sysfs_root, const char* bdf, int* cpus, int max_cpus) {
    if (!sysfs_root || !bdf) return -1;
    const std::vector<int> v = numa_cpus_of_pci_device(sysfs_root, bdf);
    for (int i = 0; i < max_cpus && i < static_cast<int>(v.size()); ++i)
        if (cpus) cpus[i] = v[static_cast<size_t>(i)];
    return static_cast<int>(v.size());
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

    // Registrars: one per device, chunk c belongs to registrar c mod R; chunk_done[c]: pinned and its ranges recorded.
    const int nchunks = b->register_host == 2 ? (nframes + kPinChunk - 1) / kPinChunk : 0;
    const int R = std::min(b->registrars > 0 ? b->registrars : G, nchunks);
    std::mutex chunk_mutex;
    std::condition_variable chunk_cv;
    std::vector<char> chunk_done(static_cast<size_t>(nchunks), 0);
    std::vector<std::vector<HostRange>> chunk_ranges(static_cast<size_t>(nchunks));  // what each chunk added (for adoption)
    // What each chunk has to register is settled HERE, in chunk order, before the registrars start (which range is "earlier"
    // must not depend on the order the threads get going); the registrars only make the expensive calls side by side.  Every
    // plane of chunk c lies inside one range of chunk c (or of an earlier chunk that already contains it), which the worker has
    // adopted by the time it submits the plane's frame.
    std::vector<HostRange> known_before;  // what earlier calls registered (this call's claims are adopted chunk by chunk, once they ARE registered)
    if (nchunks > 0) {
        std::lock_guard<std::mutex> lock(b->pin_mutex);
        known_before = b->pinned;
    }
    std::vector<std::vector<HostRange>> chunk_todo(static_cast<size_t>(nchunks));
    for (int c = 0; c < nchunks; ++c)
        chunk_todo[static_cast<size_t>(c)] =
            ranges_to_pin(*b, c * kPinChunk, std::min(nframes, (c + 1) * kPinChunk), src_planes, src_pitch, dst_planes, dst_pitch);
    std::vector<std::thread> registrars;
    for (int r = 0; r < R; ++r)
        registrars.emplace_back([&, r] {
            if (b->affinity) bind_this_thread(b->device_cpus[static_cast<size_t>(r % G)]);
            (void)hipSetDevice(b->devices[static_cast<size_t>(r % G)]);
            for (int c = r; c < nchunks && status.load() == JINC_OK; c += R) {
                const std::vector<HostRange>& todo = chunk_todo[static_cast<size_t>(c)];
                std::vector<HostRange> done, refused;
                for (const HostRange& rg : todo) {
                    // (one registration call at a time in the process: filter_internal.h host_registration_mutex)
                    if ([&] {
                            std::lock_guard<std::mutex> one(jinc::knobs::host_registration_mutex());
                            return hipHostRegister(rg.base, rg.bytes, hipHostRegisterPortable);
                        }() == hipSuccess) {
                        jinc::knobs::count_host_registration(+1);
                        done.push_back(rg);
                    }
                    else {
                        const hipError_t why = hipGetLastError();  // e.g. pinned by somebody else already: those planes take the pageable path
                        refused.push_back(rg);
                        if (b->refused.fetch_add(1) == 0) {
                            std::lock_guard<std::mutex> lock(b->pin_mutex);
                            char where[96];
                            std::snprintf(where, sizeof(where), " (%p, %zu bytes, chunk %d)", static_cast<void*>(rg.base), rg.bytes, c);
                            b->first_refusal = std::string(hipGetErrorString(why)) + where;
                        }
                    }
                }
                if (!refused.empty()) {  // the claim of a range that could not be registered is withdrawn
                    std::lock_guard<std::mutex> lock(b->pin_mutex);
                    b->pinned.erase(std::remove_if(b->pinned.begin(), b->pinned.end(),
                                                   [&](const HostRange& p) {
                                                       for (const HostRange& rg : refused)
                                                           if (rg.base == p.base && rg.bytes == p.bytes) return true;
                                                       return false;
                                                   }),
                                    b->pinned.end());
                }
                {
                    std::lock_guard<std::mutex> lock(chunk_mutex);
                    chunk_ranges[static_cast<size_t>(c)] = std::move(done);
                    chunk_done[static_cast<size_t>(c)] = 1;
                }
                chunk_cv.notify_all();
            }
            {   // (a registrar that stops early because of an error elsewhere must not leave workers waiting, nor claims behind)
                std::vector<HostRange> unclaimed;
                {
                    std::lock_guard<std::mutex> lock(chunk_mutex);
                    for (int c = r; c < nchunks; c += R)
                        if (!chunk_done[static_cast<size_t>(c)]) {
                            chunk_done[static_cast<size_t>(c)] = 1;
                            unclaimed.insert(unclaimed.end(), chunk_todo[static_cast<size_t>(c)].begin(), chunk_todo[static_cast<size_t>(c)].end());
                        }
                }
                if (!unclaimed.empty()) {
                    std::lock_guard<std::mutex> lock(b->pin_mutex);
                    b->pinned.erase(std::remove_if(b->pinned.begin(), b->pinned.end(),
                                                   [&](const HostRange& p) {
                                                       for (const HostRange& rg : unclaimed)
                                                           if (rg.base == p.base && rg.bytes == p.bytes) return true;
                                                       return false;
                                                   }),
                                    b->pinned.end());
                }
            }
            chunk_cv.notify_all();
        });

    auto worker = [&](int d) {
        if (b->affinity) bind_this_thread(b->device_cpus[static_cast<size_t>(d)]);
        jinc_filter* f = b->filters[static_cast<size_t>(d)];
        std::vector<long long> tickets;  // frames in flight on this device, oldest first
        std::string message;  // of the first failure on this worker (later successful calls clear the thread's last error)
        auto note = [&](int r) {
            if (r != JINC_OK && message.empty()) message = jinc_last_error();
            return r;
        };
        auto wait_oldest = [&]() {
            const int rc = note(jinc_filter_wait(f, tickets.front()));
            tickets.erase(tickets.begin());
            return rc;
        };
        int rc = JINC_OK, adopted_chunks = 0;
        for (const HostRange& r : known_before) (void)jinc_filter_adopt_host_range(f, r.base, r.bytes);  // ranges pinned by earlier calls
        for (int n = d; n < nframes && rc == JINC_OK && status.load() == JINC_OK; n += G) {  // jinc_shard_device(n, G) == d
            if (nchunks > 0 && n / kPinChunk >= adopted_chunks) {  // the frame's planes have to be pinned and known to this instance
                std::unique_lock<std::mutex> lock(chunk_mutex);
                for (; adopted_chunks <= n / kPinChunk; ++adopted_chunks) {  // in chunk order (ranges that continue each other merge)
                    chunk_cv.wait(lock, [&] { return chunk_done[static_cast<size_t>(adopted_chunks)] != 0; });
                    for (const HostRange& r : chunk_ranges[static_cast<size_t>(adopted_chunks)])
                        (void)jinc_filter_adopt_host_range(f, r.base, r.bytes);  // failure: pageable path for those planes
                }
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
            rc = note(jinc_filter_submit(f, s, src_pitch, t, dst_pitch, &ticket));
            if (rc == JINC_OK) tickets.push_back(ticket);
        }
        if (rc == JINC_OK) rc = note(jinc_filter_flush(f));  // the device's last frames do not wait for company
        while (!tickets.empty()) {  // drain, also after an error: buffers must not be in use when we return
            const int w = wait_oldest();
            if (rc == JINC_OK) rc = w;
        }
        if (rc != JINC_OK) {
            int expected = JINC_OK;
            if (status.compare_exchange_strong(expected, rc)) {
                std::lock_guard<std::mutex> lock(err_mutex);
                first_error = message;
            }
        }
    };
    // (every worker on a thread of its own: the calling thread keeps its affinity)
    std::vector<std::thread> threads;
    for (int d = 0; d < G; ++d) threads.emplace_back(worker, d);
    for (auto& t : threads) t.join();
    for (auto& t : registrars) t.join();
    if (status.load() != JINC_OK) return batch_fail(status.load(), first_error);
    g_batch_error.clear();
    return JINC_OK;
}

}  // extern "C"
