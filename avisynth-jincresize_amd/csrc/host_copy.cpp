// host_copy.cpp -- rows between the caller's pageable planes and the library's own pinned buffers (pipeline.cpp, the default
// way of treating host memory from round 6 on: the device never maps the caller's pages).
//
// One CPU thread copies ~25 GB/s on the hosts of this pool; a 1080p -> 4K frame is 15.5 MB, so the copies of a frame would take
// longer than its transfers (45 - 55 GB/s) and its kernels together.  The reference spreads its work over every core of the host
// unless the script says threads = 1 (ref /root/reference/src/JincResize.cpp:758-760, :901: threads selects the single-threaded instantiation of resize_plane_*, 0 the parallel one); here
// the same argument decides whether plane copies may use helper threads: threads = 1 keeps every copy on the caller's thread.
// The helpers are a small process-wide pool (kHelpers threads per caller that copies at the same time, started on demand, shared by every instance,
// idle on a condition variable otherwise); a copy is cut into row ranges, the calling thread takes the first and waits for the
// rest.  Only memcpy runs on the helpers: no HIP call, no access to an instance.
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <fstream>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>

#include <sched.h>
#include <unistd.h>

#include "filter_internal.h"
#include "knobs.h"

namespace jinc {
namespace host {

namespace {

constexpr int kHelpers = 5;                    // + the calling thread: six lanes on hosts with 12 CPUs or more (C2, 8 frames in flight: 1 / 2 / 4 / 6 / 8 lanes = 1 961 / 3 186 / 4 600 / 4 847 / 5 450 frames/s, profiles/round6/host_modes_knobs.log)
constexpr size_t kParallelFrom = 512 << 10;    // bytes: below this one thread is done before the helpers have woken up

struct Piece {
    char* dst;
    const char* src;
    size_t dst_pitch, src_pitch, row_bytes;
    int rows;
    std::atomic<int>* left;  // pieces of the same copy still running
};

struct Pool {
    std::mutex mutex;
    std::condition_variable work, done;
    std::deque<Piece> queue;
    int threads = 0;
    int callers = 0;  // copies being cut up right now (jinc_batch_process: one worker thread per device)
};

// Leaked on purpose, like the pin registry: instances may be freed -- and copy -- while the process's statics are being destroyed.
Pool& pool() {
    static Pool& p = *new Pool;
    return p;
}

void rows(char* dst, size_t dst_pitch, const char* src, size_t src_pitch, size_t row_bytes, int n) {
    if (dst_pitch == row_bytes && src_pitch == row_bytes) {
        std::memcpy(dst, src, row_bytes * static_cast<size_t>(n));
        return;
    }
    for (int y = 0; y < n; ++y) std::memcpy(dst + dst_pitch * y, src + src_pitch * y, row_bytes);
}

void helper() {
    // a helper serves every instance of the process: it runs where the process's main thread may run, not where the thread that
    // happened to start it was bound (jinc_batch_process binds each device's worker to that device's NUMA node)
    cpu_set_t everywhere;
    if (sched_getaffinity(getpid(), sizeof(everywhere), &everywhere) == 0) (void)sched_setaffinity(0, sizeof(everywhere), &everywhere);
    Pool& p = pool();
    std::unique_lock<std::mutex> lock(p.mutex);
    for (;;) {
        p.work.wait(lock, [&] { return !p.queue.empty(); });
        const Piece piece = p.queue.front();
        p.queue.pop_front();
        lock.unlock();
        rows(piece.dst, piece.dst_pitch, piece.src, piece.src_pitch, piece.row_bytes, piece.rows);
        lock.lock();
        if (piece.left->fetch_sub(1) == 1) p.done.notify_all();
    }
}

// CPUs this process may keep busy: its affinity mask, cut down to the CFS quota of its cgroup where there is one (a container
// with 256 CPUs in the mask and cpu.max = 16 cores is throttled as a whole once more than 16 threads run: eight client threads
// with five helpers each fell from 3 812 to 2 347 C2 frames/s that way, profiles/round6/e2e_get_frame_modes.log).  Read once.
int usable_cpus() {
    static const int n = [] {
        int cpus = static_cast<int>(std::thread::hardware_concurrency());
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) cpus = CPU_COUNT(&set);
        auto first_line = [](const std::string& path) {
            std::ifstream in(path);
            std::string line;
            if (in) std::getline(in, line);
            return line;
        };
        double quota = 0.0;  // cores; 0: none found
        auto consider = [&](double q) {
            if (q > 0.0 && (quota == 0.0 || q < quota)) quota = q;
        };
        std::ifstream groups("/proc/self/cgroup");
        for (std::string line; groups && std::getline(groups, line);) {
            const size_t a = line.find(':'), b = a == std::string::npos ? a : line.find(':', a + 1);
            if (b == std::string::npos) continue;
            const std::string controllers = line.substr(a + 1, b - a - 1);
            std::string rel = line.substr(b + 1);
            const bool v2 = controllers.empty();
            if (!v2 && controllers.find("cpu") == std::string::npos) continue;
            for (;;) {  // this group and its parents
                if (v2) {
                    std::istringstream f(first_line("/sys/fs/cgroup" + rel + "/cpu.max"));
                    std::string q;
                    double period = 0.0;
                    if (f >> q >> period && q != "max" && period > 0.0) consider(std::atof(q.c_str()) / period);
                } else {
                    for (const char* ctl : {"cpu", "cpu,cpuacct"}) {
                        const std::string base = std::string("/sys/fs/cgroup/") + ctl + rel;
                        const double q = std::atof(first_line(base + "/cpu.cfs_quota_us").c_str());
                        const double period = std::atof(first_line(base + "/cpu.cfs_period_us").c_str());
                        if (q > 0.0 && period > 0.0) consider(q / period);
                    }
                }
                if (rel.empty() || rel == "/") break;
                const size_t cut = rel.find_last_of('/');
                rel = cut == std::string::npos || cut == 0 ? "/" : rel.substr(0, cut);
            }
        }
        if (quota > 0.0) cpus = std::min(cpus, std::max(1, static_cast<int>(quota + 0.5)));
        return std::max(cpus, 1);
    }();
    return n;
}

int helpers_wanted() {
    const int knob = knobs::geti(JINC_KNOB_COPY_THREADS, -1);  // A/B: total lanes, 1 = the calling thread only
    if (knob >= 1) return std::min(knob - 1, 15);
    const int cpus = usable_cpus();
    return cpus >= 12 ? kHelpers : cpus >= 8 ? 3 : cpus >= 4 ? 1 : 0;
}

}  // namespace

int copy_lanes_cpus() { return usable_cpus(); }

void copy_planes(const PlaneCopy* jobs, size_t njobs, bool may_use_helpers) {
    size_t total = 0;
    for (size_t j = 0; j < njobs; ++j) total += jobs[j].row_bytes * static_cast<size_t>(std::max(jobs[j].rows, 0));
    const int want = may_use_helpers && total >= kParallelFrom ? helpers_wanted() : 0;
    if (want <= 0) {
        for (size_t j = 0; j < njobs; ++j) rows(jobs[j].dst, jobs[j].dst_pitch, jobs[j].src, jobs[j].src_pitch, jobs[j].row_bytes, jobs[j].rows);
        return;
    }
    Pool& p = pool();
    std::atomic<int> left{0};
    {
        std::lock_guard<std::mutex> lock(p.mutex);
        // the pool grows with the callers that copy at the same time (each brings a thread of its own): `want` helpers per caller,
        // at most half the CPUs this process may use all told (callers included)
        ++p.callers;
        const int most = std::max(knobs::geti(JINC_KNOB_COPY_THREADS, -1) >= 1 ? want : 0, usable_cpus() / 2 - p.callers);
        const int pool_size = std::min({want * p.callers, most, 64});
        while (p.threads < pool_size) {
            try {
                std::thread(helper).detach();
            } catch (...) {
                break;  // no more threads to be had: fewer lanes
            }
            ++p.threads;
        }
        // pieces of about total / lanes bytes (a plane larger than that is cut into row ranges, smaller planes go whole), at least
        // 128 KiB each: several small planes -- the frames of a group's share -- spread over the lanes as well as one large plane does
        const int lanes = std::min(want, p.threads) + 1;
        const size_t piece_bytes = std::max<size_t>(total / static_cast<size_t>(lanes), 128 << 10);
        for (size_t j = 0; j < njobs; ++j) {
            const PlaneCopy& c = jobs[j];
            if (c.rows <= 0 || c.row_bytes == 0) continue;
            const size_t bytes = c.row_bytes * static_cast<size_t>(c.rows);
            const int parts = static_cast<int>(std::max<size_t>(1, std::min<size_t>((bytes + piece_bytes - 1) / piece_bytes, static_cast<size_t>(c.rows))));
            for (int k = 0; k < parts; ++k) {
                const int y0 = c.rows * k / parts, y1 = c.rows * (k + 1) / parts;
                p.queue.push_back({c.dst + c.dst_pitch * y0, c.src + c.src_pitch * y0, c.dst_pitch, c.src_pitch, c.row_bytes, y1 - y0, &left});
                ++left;
            }
        }
    }
    p.work.notify_all();
    // the caller works through its own pieces like a helper would (whatever the helpers do not get to is done here) and waits for
    // the pieces helpers are still busy with
    std::unique_lock<std::mutex> lock(p.mutex);
    while (left.load() > 0) {
        auto it = std::find_if(p.queue.begin(), p.queue.end(), [&](const Piece& q) { return q.left == &left; });
        if (it == p.queue.end()) {
            p.done.wait(lock, [&] { return left.load() == 0; });
            break;
        }
        const Piece piece = *it;
        p.queue.erase(it);
        lock.unlock();
        rows(piece.dst, piece.dst_pitch, piece.src, piece.src_pitch, piece.row_bytes, piece.rows);
        lock.lock();
        left.fetch_sub(1);
    }
    --p.callers;
}

void copy_plane_rows(char* dst, size_t dst_pitch, const char* src, size_t src_pitch, size_t row_bytes, int nrows, bool may_use_helpers) {
    const PlaneCopy one{dst, src, dst_pitch, src_pitch, row_bytes, nrows};
    copy_planes(&one, 1, may_use_helpers);
}

}  // namespace host
}  // namespace jinc
