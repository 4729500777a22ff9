// duplex_probe.cpp -- does the host link run both directions at once, and for which kind of host memory?
// H2D alone, D2H alone and both together (two streams), for hipHostMalloc'ed, hipHostRegister'ed and pageable buffers,
// 1-D copies and 2-D copies (row = pitch), 8 MiB per copy, 64 copies per direction in flight.
// Build: hipcc -O2 profiles/probes/duplex_probe.cpp -o profiles/probes/duplex_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const size_t row = 3840, rows = 2160, bytes = row * rows;
    const int n = 64;
    void *d_in = nullptr, *d_out = nullptr;
    CK(hipMalloc(&d_in, bytes));
    CK(hipMalloc(&d_out, bytes));
    hipStream_t s1, s2, s3;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
    for (int kind = 0; kind < 3; ++kind) {
        void *h_in = nullptr, *h_out = nullptr;
        const char* name = kind == 0 ? "hipHostMalloc" : kind == 1 ? "hipHostRegister" : "pageable";
        if (kind == 0) {
            CK(hipHostMalloc(&h_in, bytes, hipHostMallocDefault));
            CK(hipHostMalloc(&h_out, bytes, hipHostMallocDefault));
        } else {
            if (posix_memalign(&h_in, 4096, bytes) || posix_memalign(&h_out, 4096, bytes)) return 1;
        }
        std::memset(h_in, 1, bytes);
        std::memset(h_out, 2, bytes);
        if (kind == 1) {
            CK(hipHostRegister(h_in, bytes, hipHostRegisterDefault));
            CK(hipHostRegister(h_out, bytes, hipHostRegisterDefault));
        }
        for (int two_d = 0; two_d < 2; ++two_d) {
            auto h2d = [&](hipStream_t s) {
                if (two_d) CK(hipMemcpy2DAsync(d_in, row, h_in, row, row, rows, hipMemcpyHostToDevice, s));
                else CK(hipMemcpyAsync(d_in, h_in, bytes, hipMemcpyHostToDevice, s));
            };
            auto d2h = [&](hipStream_t s) {
                if (two_d) CK(hipMemcpy2DAsync(h_out, row, d_out, row, row, rows, hipMemcpyDeviceToHost, s));
                else CK(hipMemcpyAsync(h_out, d_out, bytes, hipMemcpyDeviceToHost, s));
            };
            double r[4] = {0, 0, 0, 0};
            for (int mode = 0; mode < 4; ++mode) {  // 0: H2D, 1: D2H, 2: both on two streams, 3: D2H split over two streams
                for (int rep = 0; rep < 2; ++rep) {
                    CK(hipDeviceSynchronize());
                    const double t0 = now();
                    for (int k = 0; k < n; ++k) {
                        if (mode == 0 || mode == 2) h2d(s1);
                        if (mode == 1 || mode == 2) d2h(s2);
                        if (mode == 3) d2h(k & 1 ? s2 : s3);
                    }
                    CK(hipDeviceSynchronize());
                    const double el = now() - t0;
                    r[mode] = (mode == 2 ? 2.0 : 1.0) * n * bytes / el / 1e9;
                }
            }
            std::printf("{\"host_memory\": \"%s\", \"copy\": \"%s\", \"h2d_GBps\": %.1f, \"d2h_GBps\": %.1f, \"both_sum_GBps\": %.1f, \"d2h_two_streams_GBps\": %.1f}\n",
                        name, two_d ? "2D" : "1D", r[0], r[1], r[2], r[3]);
            std::fflush(stdout);
        }
        if (kind == 1) { CK(hipHostUnregister(h_in)); CK(hipHostUnregister(h_out)); }
        if (kind == 0) { CK(hipHostFree(h_in)); CK(hipHostFree(h_out)); } else { std::free(h_in); std::free(h_out); }
    }
    return 0;
}
