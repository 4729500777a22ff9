// hostreg_semantics_probe.cpp -- what hipHostRegister / hipHostGetDevicePointer / hipMemcpy do with ranges that share pages,
// touch, or are only partly registered (round 6: the pin registry's assumptions, csrc/pipeline.cpp).  No kernel touches host memory.
// build: hipcc -O2 hostreg_semantics_probe.cpp -o hostreg_semantics_probe -lpthread
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static const char* S(hipError_t e) { (void)hipGetLastError(); return hipGetErrorName(e); }

int main() {
    (void)hipSetDevice(0);
    const size_t page = 4096;
    char* buf = nullptr;
    if (posix_memalign(reinterpret_cast<void**>(&buf), page, 64 * page)) return 1;
    memset(buf, 1, 64 * page);
    void* dev = nullptr;
    (void)hipMalloc(&dev, 64 * page);
    // 1. two ranges that touch inside one page
    char* a = buf + 100;
    const size_t la = 3 * page + 500;  // ends inside page 3
    char* b = a + la;                  // starts in page 3, same page as a's end
    const size_t lb = 2 * page;
    std::printf("1a register A [%zu, %zu)            : %s\n", size_t(a - buf), size_t(a - buf) + la, S(hipHostRegister(a, la, hipHostRegisterPortable | hipHostRegisterMapped)));
    std::printf("1b register B [%zu, %zu) touching A  : %s\n", size_t(b - buf), size_t(b - buf) + lb, S(hipHostRegister(b, lb, hipHostRegisterPortable | hipHostRegisterMapped)));
    // 2. device pointers inside / outside the registered bytes
    void* d = nullptr;
    std::printf("2a devptr of A's first byte            : %s\n", S(hipHostGetDevicePointer(&d, a, 0)));
    std::printf("2b devptr of a byte BEFORE A, same page: %s\n", S(hipHostGetDevicePointer(&d, buf + 10, 0)));
    std::printf("2c devptr of a byte after B, same page : %s\n", S(hipHostGetDevicePointer(&d, b + lb + 8, 0)));
    std::printf("2d devptr two pages after B            : %s\n", S(hipHostGetDevicePointer(&d, b + lb + 2 * page, 0)));
    // 3. the allocation a device pointer lies in
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    (void)hipHostGetDevicePointer(&d, a + 10, 0);
    hipError_t e = hipMemGetAddressRange(&base, &size, d);
    std::printf("3  hipMemGetAddressRange(devptr(A+10))  : %s base-offset %td size %zu (A: offset 0 size %zu)\n", S(e), (char*)base - (char*)d + 10 - 0, size, la);
    hipPointerAttribute_t at;
    e = hipPointerGetAttributes(&at, a + 10);
    std::printf("3b hipPointerGetAttributes(A+10)        : %s type %d host %p dev %p\n", S(e), e == hipSuccess ? int(at.type) : -1, e == hipSuccess ? at.hostPointer : nullptr, e == hipSuccess ? at.devicePointer : nullptr);
    // 4. copies from host ranges relative to the registrations
    std::printf("4a memcpy H2D inside A                  : %s\n", S(hipMemcpy(dev, a + 16, page, hipMemcpyHostToDevice)));
    std::printf("4b memcpy H2D from A into B (crossing)  : %s\n", S(hipMemcpy(dev, a + la - 100, 1000, hipMemcpyHostToDevice)));
    std::printf("4c memcpy H2D starting before A into A  : %s\n", S(hipMemcpy(dev, buf + 10, 1000, hipMemcpyHostToDevice)));
    std::printf("4d memcpy H2D from B past its end       : %s\n", S(hipMemcpy(dev, b + lb - 100, 1000, hipMemcpyHostToDevice)));
    std::printf("4e memcpy2D H2D rows crossing A -> B    : %s\n", S(hipMemcpy2D(dev, 512, a + la - 1024, 512, 256, 8, hipMemcpyHostToDevice)));
    // 5. unregister from another thread, register again
    std::thread([&] { std::printf("5a unregister A from another thread    : %s\n", S(hipHostUnregister(a))); }).join();
    std::printf("5b register A again                     : %s\n", S(hipHostRegister(a, la, hipHostRegisterPortable)));
    std::printf("5c register a range overlapping A's bytes: %s\n", S(hipHostRegister(a + page, 2 * page, hipHostRegisterPortable)));
    std::printf("5d register a range CONTAINING A and B  : %s\n", S(hipHostRegister(buf, 10 * page, hipHostRegisterPortable)));
    std::printf("5e unregister A                         : %s\n", S(hipHostUnregister(a)));
    std::printf("5f unregister B                         : %s\n", S(hipHostUnregister(b)));
    std::printf("5g unregister B again                   : %s\n", S(hipHostUnregister(b)));
    std::printf("5h memcpy H2D where A was               : %s\n", S(hipMemcpy(dev, a + 16, page, hipMemcpyHostToDevice)));
    // 6. page-rounded neighbours registered side by side from four threads
    std::vector<std::thread> ts;
    hipError_t res[8];
    for (int t = 0; t < 8; ++t) ts.emplace_back([&, t] { (void)hipSetDevice(0); res[t] = hipHostRegister(buf + 16 * page + t * 4 * page, 4 * page, hipHostRegisterPortable); (void)hipGetLastError(); });
    for (auto& t : ts) t.join();
    for (int t = 0; t < 8; ++t) std::printf("6  neighbour %d registered in parallel  : %s\n", t, hipGetErrorName(res[t]));
    std::printf("6b memcpy2D across the 8 neighbours     : %s\n", S(hipMemcpy2D(dev, 4096, buf + 16 * page + 100, 4096, 4000, 7, hipMemcpyHostToDevice)));
    for (int t = 0; t < 8; ++t) std::printf("6c unregister neighbour %d             : %s\n", t, S(hipHostUnregister(buf + 16 * page + t * 4 * page)));
    (void)hipFree(dev);
    free(buf);
    return 0;
}
