#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include "jincresize_hip_test.h"
int main() {
    struct Case { int sw, sh, tw, th, tap, bits, comp, planes, subw, subh; double left, top, w, h; unsigned def; };
    std::vector<Case> cases = {
        {640, 360, 1280, 720, 3, 8, 1, 1, 0, 0, 0, 0, 0, 0, JINC_ARG_TAP},
        {320, 180, 480, 270, 3, 8, 1, 1, 0, 0, 0, 0, 0, 0, JINC_ARG_TAP},
        {37, 23, 91, 50, 3, 8, 1, 1, 0, 0, 1.3, 0.7, 33.1, 20.2, JINC_ARG_TAP | JINC_ARG_SRC_LEFT | JINC_ARG_SRC_TOP | JINC_ARG_SRC_WIDTH | JINC_ARG_SRC_HEIGHT},
        {320, 180, 640, 360, 8, 16, 2, 3, 1, 1, 0, 0, 0, 0, JINC_ARG_TAP},
        {64, 48, 40, 30, 3, 8, 1, 1, 0, 0, 0, 0, 0, 0, JINC_ARG_TAP},
        {640, 480, 64, 48, 3, 8, 1, 1, 0, 0, 0, 0, 0, 0, JINC_ARG_TAP},
        {5, 5, 20, 20, 3, 8, 1, 1, 0, 0, 0, 0, 0, 0, JINC_ARG_TAP},
        {720, 480, 1920, 1080, 16, 32, 4, 3, 0, 0, 0, 0, 0, 0, JINC_ARG_TAP},
    };
    for (auto& c : cases) {
        jinc_video_info vi{c.sw, c.sh, c.bits, c.comp, c.planes, 1, 0, c.subw, c.subh};
        jinc_args a{};
        a.target_width = c.tw; a.target_height = c.th; a.tap = c.tap; a.defined = c.def;
        a.src_left = c.left; a.src_top = c.top; a.src_width = c.w; a.src_height = c.h; a.frame0_chroma_location = -1;
        jinc_filter* f = nullptr; char err[256];
        int rc = jinc_filter_create(&vi, &a, -1, &f, err, sizeof err);
        if (rc != 0) { std::printf("create rc=%d: %s\n", rc, err); continue; }
        for (int t = 0; t < jinc_filter_num_tables(f); ++t) {
            jinc_plan_info info; jinc_filter_plan_info(f, t, &info);
            std::vector<int> sx(info.dst_width), sy(info.dst_height), ids((size_t)info.dst_width * info.dst_height);
            jinc_filter_plan_dump(f, t, sx.data(), sy.data(), ids.data());
            std::vector<float> cf((size_t)info.filter_size * info.filter_size);
            long long sum = 0;
            for (int s = 0; s < info.num_sets; s += 97) { jinc_filter_plan_set(f, t, s, cf.data()); sum += (long long)(cf[0] * 1e6); }
            int a0, b0; jinc_filter_plan_pixel(f, t, info.dst_width - 1, info.dst_height - 1, &a0, &b0, cf.data());
            std::printf("%dx%d->%dx%d table %d fs %d sets %d periodic %d quasi %d (%lld)\n", c.sw, c.sh, c.tw, c.th, t, info.filter_size, info.num_sets, info.periodic, info.quasi, sum);
        }
        jinc_filter_free(f);
    }
    // host_copy.cpp: plane copies cut into row ranges for the process-wide helper threads, from several callers at once
    {
        std::vector<std::thread> callers;
        int wrong = 0;
        for (int t = 0; t < 4; ++t)
            callers.emplace_back([t, &wrong] {
                const int rows = 1080 + t, row_bytes = 3840, spitch = 3840 + 64 * (t & 1), dpitch = 3904;
                std::vector<unsigned char> src(size_t(rows) * spitch), dst(size_t(rows) * dpitch, 0xEE);
                for (size_t i = 0; i < src.size(); ++i) src[i] = static_cast<unsigned char>(i * 31 + t);
                for (int round = 0; round < 3; ++round) {
                    if (jinc_debug_copy_rows(dst.data(), dpitch, src.data(), spitch, row_bytes, rows, round != 2) != 0) __atomic_add_fetch(&wrong, 1, __ATOMIC_RELAXED);
                    for (int y = 0; y < rows; ++y)
                        if (std::memcmp(&dst[size_t(y) * dpitch], &src[size_t(y) * spitch], row_bytes) != 0 || dst[size_t(y) * dpitch + row_bytes] != 0xEE)
                            __atomic_add_fetch(&wrong, 1, __ATOMIC_RELAXED);
                }
            });
        for (auto& c : callers) c.join();
        std::printf("plane copies on the helper threads: %d wrong\n", wrong);
        if (wrong) return 1;
    }
    return 0;
}
