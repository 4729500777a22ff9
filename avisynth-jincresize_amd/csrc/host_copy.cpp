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
#include <mutex>
#include <thread>

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

int helpers_wanted() {
    const int knob = knobs::geti(JINC_KNOB_COPY_THREADS, -1);  // A/B: total lanes, 1 = the calling thread only
    if (knob >= 1) return std::min(knob - 1, 15);
    const unsigned hw = std::thread::hardware_concurrency();
    return hw >= 12 ? kHelpers : hw >= 8 ? 3 : hw >= 4 ? 1 : 0;
}

}  // namespace

void copy_plane_rows(char* dst, size_t dst_pitch, const char* src, size_t src_pitch, size_t row_bytes, int nrows, bool may_use_helpers) {
    const size_t total = row_bytes * static_cast<size_t>(std::max(nrows, 0));
    const int want = may_use_helpers && total >= kParallelFrom ? helpers_wanted() : 0;
    if (want <= 0 || nrows < 2 * (want + 1)) {
        rows(dst, dst_pitch, src, src_pitch, row_bytes, nrows);
        return;
    }
    Pool& p = pool();
    std::atomic<int> left{0};
    int mine = 0;
    {
        std::lock_guard<std::mutex> lock(p.mutex);
        // the pool grows with the callers that copy at the same time (each brings a thread of its own): `want` helpers per caller,
        // at most half the host's CPUs all told
        ++p.callers;
        const int most = std::max(want, static_cast<int>(std::thread::hardware_concurrency() / 2) - p.callers);
        const int pool_size = std::min({want * p.callers, most, 64});
        while (p.threads < pool_size) {
            try {
                std::thread(helper).detach();
            } catch (...) {
                break;  // no more threads to be had: fewer lanes
            }
            ++p.threads;
        }
        const int lanes = std::min(want, p.threads) + 1;
        mine = nrows / lanes;
        int y = mine;
        for (int k = 1; k < lanes; ++k) {
            const int n = k + 1 == lanes ? nrows - y : nrows / lanes;
            p.queue.push_back({dst + dst_pitch * y, src + src_pitch * y, dst_pitch, src_pitch, row_bytes, n, &left});
            ++left;
            y += n;
        }
        if (lanes == 1) mine = nrows;
    }
    p.work.notify_all();
    rows(dst, dst_pitch, src, src_pitch, row_bytes, mine);
    std::unique_lock<std::mutex> lock(p.mutex);
    // a helper that is busy with another instance's copy leaves pieces in the queue: the caller takes them itself rather than wait
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

}  // namespace host
}  // namespace jinc
