// kernel_blit.hip -- frame transport by the shader instead of the DMA engines (pipeline.cpp): ONE launch moves the planes
// of a whole group of frames between the group's device buffer and the callers' host planes (pinned with hipHostRegister,
// addressed through their device mapping), where the DMA path needs one copy per plane and frame and pays ~17 us of engine
// turnaround per copy (a 2 MB copy every 60 us = 35 GB/s of the link's 52 GB/s; profiles/round3/e2e_pipeline_*.log).
// Not part of the hot path's arithmetic: bytes in, the same bytes out.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace jinc {
namespace {

// Rows [r0, r1) of entry e, by the whole workgroup.
template <typename V>
__device__ __forceinline__ void move_rows(const BlitEntry& e, uint32_t r0, uint32_t r1) {
    constexpr uint32_t U = sizeof(V);
    const uint32_t ppr = e.row_bytes / U;  // whole units per row
    const uint32_t total = ppr * (r1 - r0);
    const uint32_t stride = blockDim.x;
    const char* __restrict__ src = static_cast<const char*>(e.src) + static_cast<size_t>(r0) * e.src_pitch;
    char* __restrict__ dst = static_cast<char*>(e.dst) + static_cast<size_t>(r0) * e.dst_pitch;
    uint32_t p = threadIdx.x;
    // four independent units per lane and pass: loads first, then stores (the link wants many requests in flight)
    for (; p + 3 * stride < total; p += 4 * stride) {
        V v[4];
        uint32_t off[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t q = p + k * stride, row = q / ppr, col = q - row * ppr;
            v[k] = *reinterpret_cast<const V*>(src + static_cast<size_t>(row) * e.src_pitch + col * U);
            off[k] = row * e.dst_pitch + col * U;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) *reinterpret_cast<V*>(dst + off[k]) = v[k];
    }
    for (; p < total; p += stride) {
        const uint32_t row = p / ppr, col = p - row * ppr;
        *reinterpret_cast<V*>(dst + static_cast<size_t>(row) * e.dst_pitch + col * U) =
            *reinterpret_cast<const V*>(src + static_cast<size_t>(row) * e.src_pitch + col * U);
    }
    const uint32_t tail = e.row_bytes - ppr * U;  // bytes of a row beyond its whole units
    if (tail) {
        for (uint32_t q = threadIdx.x; q < tail * (r1 - r0); q += stride) {
            const uint32_t row = q / tail, b = ppr * U + (q - row * tail);
            dst[static_cast<size_t>(row) * e.dst_pitch + b] = src[static_cast<size_t>(row) * e.src_pitch + b];
        }
    }
}

// A FEW workgroups (the link, not the shader, bounds this kernel: ~50 workgroups keep it full, and the resampling
// kernels of the next group of frames run beside it on the other compute units) walk over work items = (entry, slice of
// `rows_per_item` rows), item = blockIdx.x, blockIdx.x + gridDim.x, ...
__global__ __launch_bounds__(256) void blit_rows_kernel(const BlitEntry* __restrict__ table, uint32_t entries, uint32_t rows_per_item,
                                                        uint32_t items_per_entry) {
    for (uint32_t item = blockIdx.x; item < entries * items_per_entry; item += gridDim.x) {
        const uint32_t ei = item / items_per_entry, slice = item - ei * items_per_entry;
        const BlitEntry e = table[ei];
        const uint32_t r0 = slice * rows_per_item;
        if (r0 >= e.rows) continue;
        const uint32_t r1 = r0 + rows_per_item < e.rows ? r0 + rows_per_item : e.rows;
        if (e.unit == 16) move_rows<uint4>(e, r0, r1);
        else if (e.unit == 4) move_rows<uint32_t>(e, r0, r1);
        else move_rows<uint8_t>(e, r0, r1);
    }
}

}  // namespace

int blit_unit(const void* src, const void* dst, uint32_t src_pitch, uint32_t dst_pitch) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) | src_pitch | dst_pitch;
    return a % 16 == 0 ? 16 : a % 4 == 0 ? 4 : 1;
}

int launch_blit_rows(const BlitEntry* table_device, int first, int count, uint32_t max_rows, uint32_t max_row_bytes, int workgroups,
                     void* stream) {
    if (count <= 0 || max_rows == 0) return hipSuccess;
    // slices of ~64 KB: small enough to spread a share's planes evenly over the workgroups, large enough to amortise the loop
    uint32_t rows_per_item = 65536u / (max_row_bytes ? max_row_bytes : 1u);
    rows_per_item = rows_per_item < 1u ? 1u : rows_per_item;
    const uint32_t items_per_entry = (max_rows + rows_per_item - 1) / rows_per_item;
    const uint32_t items = static_cast<uint32_t>(count) * items_per_entry;
    const uint32_t grid = items < static_cast<uint32_t>(workgroups) ? items : static_cast<uint32_t>(workgroups);
    hipLaunchKernelGGL(blit_rows_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), table_device + first,
                       static_cast<uint32_t>(count), rows_per_item, items_per_entry);
    return hipGetLastError();
}

}  // namespace jinc
