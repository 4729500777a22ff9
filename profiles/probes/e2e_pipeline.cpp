// e2e_pipeline.cpp -- host-to-host rate of the look-ahead pipeline from ONE C++ host thread (no Python in the loop):
// submit / wait through the C ABI with registered caller buffers, plus the average host time spent inside submit and wait.
// Build: g++ -O2 -std=c++17 profiles/probes/e2e_pipeline.cpp -Iinclude -Lavisynth-jincresize_amd/lib -ljincresize_hip
//        -Wl,-rpath,$PWD/avisynth-jincresize_amd/lib -o profiles/probes/e2e_pipeline
// usage: e2e_pipeline <name> <src_w> <src_h> <dst_w> <dst_h> <tap> <depth> <group> [seconds] [register]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <vector>

#include "jincresize_hip.h"
#include "jincresize_hip_test.h"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    if (argc < 9) return std::fprintf(stderr, "usage: %s name sw sh dw dh tap depth group [seconds] [register]\n", argv[0]), 2;
    const char* name = argv[1];
    const int sw = std::atoi(argv[2]), sh = std::atoi(argv[3]), dw = std::atoi(argv[4]), dh = std::atoi(argv[5]);
    const int tap = std::atoi(argv[6]), depth = std::atoi(argv[7]), group = std::atoi(argv[8]);
    const double seconds = argc > 9 ? std::atof(argv[9]) : 2.0;
    const int reg = argc > 10 ? std::atoi(argv[10]) : 1;
    jinc_video_info vi{sw, sh, 8, 1, 1, 1, 0, 0, 0};
    jinc_args a{};
    a.target_width = dw, a.target_height = dh, a.tap = tap, a.defined = JINC_ARG_TAP;
    a.frame0_chroma_location = -1;
    jinc_filter* f = nullptr;
    char err[256];
    if (jinc_filter_create(&vi, &a, 0, &f, err, sizeof err) != JINC_OK) return std::fprintf(stderr, "%s\n", err), 1;
    if (jinc_filter_set_pipeline_group(f, depth, group, reg) != JINC_OK) return std::fprintf(stderr, "%s\n", jinc_last_error()), 1;
    const int sp = (sw + 63) & ~63, dp = (dw + 63) & ~63, nbuf = depth + 1;
    std::vector<unsigned char*> src(nbuf), dst(nbuf);
    unsigned s = 12345;
    for (int k = 0; k < nbuf; ++k) {
        if (posix_memalign(reinterpret_cast<void**>(&src[k]), 4096, static_cast<size_t>(sp) * sh)) return 1;
        if (posix_memalign(reinterpret_cast<void**>(&dst[k]), 4096, static_cast<size_t>(dp) * dh)) return 1;
        for (size_t i = 0; i < static_cast<size_t>(sp) * sh; ++i) s = s * 1664525u + 1013904223u, src[k][i] = static_cast<unsigned char>(s >> 8);
        std::memset(dst[k], 0, static_cast<size_t>(dp) * dh);
    }
    std::deque<long long> tickets;
    double t_submit = 0, t_wait = 0;
    long n = 0;
    auto submit = [&](long k) {
        const void* s4[4] = {src[k % nbuf], nullptr, nullptr, nullptr};
        void* d4[4] = {dst[k % nbuf], nullptr, nullptr, nullptr};
        const int sp4[4] = {sp, 0, 0, 0}, dp4[4] = {dp, 0, 0, 0};
        long long t = -1;
        const double t0 = now();
        if (jinc_filter_submit(f, s4, sp4, d4, dp4, &t) != JINC_OK) std::fprintf(stderr, "%s\n", jinc_last_error()), std::exit(1);
        t_submit += now() - t0;
        tickets.push_back(t);
    };
    auto wait_oldest = [&]() {
        const double t0 = now();
        if (jinc_filter_wait(f, tickets.front()) != JINC_OK) std::fprintf(stderr, "%s\n", jinc_last_error()), std::exit(1);
        t_wait += now() - t0;
        tickets.pop_front();
    };
    for (int k = 0; k < 2 * nbuf; ++k) {  // warm-up: allocations, registration, clocks
        submit(k);
        if (static_cast<int>(tickets.size()) >= depth) wait_oldest();
    }
    while (!tickets.empty()) wait_oldest();
    t_submit = t_wait = 0;
    int frames_per_call = 0;
    const char* kernel = "";
    const double t0 = now();
    while (now() - t0 < seconds) {
        submit(n++);
        if (static_cast<int>(tickets.size()) >= depth) wait_oldest();
        if (n == 3L * depth) kernel = jinc_debug_last_call(&frames_per_call);  // steady state, not the final flush
    }
    while (!tickets.empty()) wait_oldest();
    const double el = now() - t0;
    std::printf("{\"config\": \"%s\", \"pipeline_depth\": %d, \"frames_per_launch\": %d, \"kernel\": \"%s\", \"frames_in_that_call\": %d, "
                "\"registered\": %d, \"frames_per_s\": %.1f, \"Mpix_per_s\": %.1f, \"host_GB_per_s\": %.2f, \"submit_us\": %.1f, \"wait_us\": %.1f}\n",
                name, depth, jinc_filter_pipeline_group(f), kernel, frames_per_call, reg, n / el, n / el * dw * dh / 1e6,
                n / el * (static_cast<double>(sw) * sh + static_cast<double>(dw) * dh) / 1e9, t_submit / n * 1e6, t_wait / n * 1e6);
    jinc_filter_free(f);
    return 0;
}
