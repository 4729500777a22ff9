// soffset_probe.hip -- does the gfx950 buffer range check cover the instruction's scalar offset (soffset)?
// A 2N-byte allocation holds the pattern i at dword i; the descriptor covers only the first N bytes.  Lane l reads a dword
//   (a) at voffset = 4*l, soffset = 0          -> in range, expected l
//   (b) at voffset = N + 4*l, soffset = 0      -> out of range through the per-lane offset, expected 0
//   (c) at voffset = 4*l, soffset = N          -> same address as (b), but reached through soffset
//   (e) at voffset = 4*l, soffset = N + 256    -> soffset alone beyond num_records (does "num_records - soffset" wrap?)
// (c) == 0 means soffset takes part in the range check; (c) == N/4 + l means it does not (LLVM: "soffset ... excluded from
// bounds checking").  All reads stay inside the allocation.
//   hipcc --offload-arch=gfx950 -O2 -o soffset_probe soffset_probe.hip && ./soffset_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(const unsigned* buf, unsigned nbytes, unsigned* out) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(buf), 0, nbytes, 0x00020000);
    const unsigned l = threadIdx.x;
    out[l] = __builtin_amdgcn_raw_buffer_load_b32(r, 4 * l, 0, 0);
    out[64 + l] = __builtin_amdgcn_raw_buffer_load_b32(r, nbytes + 4 * l, 0, 0);
    out[128 + l] = __builtin_amdgcn_raw_buffer_load_b32(r, 4 * l, nbytes, 0);
    out[192 + l] = __builtin_amdgcn_raw_buffer_load_b32(r, 4 * l, nbytes - 128, 0);  // straddles the end: lanes 32.. are past it
    out[256 + l] = __builtin_amdgcn_raw_buffer_load_b32(r, 4 * l, nbytes + 256, 0);  // (e) soffset LARGER than num_records
}

int main() {
    const unsigned N = 4096;
    std::vector<unsigned> h(2 * N / 4);
    for (unsigned i = 0; i < h.size(); ++i) h[i] = i;
    unsigned *d, *o;
    hipMalloc(&d, 2 * N);
    hipMalloc(&o, 320 * 4);
    hipMemcpy(d, h.data(), 2 * N, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d, N, o);
    std::vector<unsigned> r(320);
    hipMemcpy(r.data(), o, 320 * 4, hipMemcpyDeviceToHost);
    printf("(a) in range            lane 0/5/63: %u %u %u\n", r[0], r[5], r[63]);
    printf("(b) voffset past end    lane 0/5/63: %u %u %u\n", r[64], r[64 + 5], r[64 + 63]);
    printf("(c) soffset past end    lane 0/5/63: %u %u %u   (dword index of that address: %u)\n", r[128], r[128 + 5], r[128 + 63], N / 4);
    printf("(d) soffset = N-128     lane 0/31/32/63: %u %u %u %u\n", r[192], r[192 + 31], r[192 + 32], r[192 + 63]);
    printf("(e) soffset = N+256     lane 0/5/63: %u %u %u   (in range would read %u..)\n", r[256], r[256 + 5], r[256 + 63], (N + 256) / 4);
    printf("verdict: soffset is %s by the buffer range check\n", r[128 + 5] == 0 ? "COVERED" : "NOT covered");
    return 0;
}
