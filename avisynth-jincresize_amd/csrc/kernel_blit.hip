// kernel_blit.hip -- frame transport by the shader instead of the DMA engines (pipeline.cpp): ONE launch moves the planes
// of a whole group of frames between the group's device buffer and the callers' host planes (pinned with hipHostRegister,
// addressed through their device mapping), where the DMA path needs one copy per plane and frame and pays ~17 us of engine
// turnaround per copy (a 2 MB copy every 60 us = 35 GB/s of the link's 52 GB/s; profiles/round3/e2e_pipeline_*.log).
// Not part of the hot path's arithmetic: bytes in, the same bytes out.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "kernels.h"
#include "knobs.h"

namespace jinc {
namespace {

// Rows [r0, r1) of entry e, by the whole workgroup: a wave takes every 4th group of 4 rows, its lanes walk along the rows
// in 64 x sizeof(V) byte steps.  No divisions, a handful of integer instructions per 16 bytes: the kernel runs beside
// VALU-bound resampling kernels and must not queue for the vector unit (a first form that cut the plane into units with
// q / units_per_row, q % units_per_row per access lost a third of its rate next to the frame-lane kernel).
template <typename V>
__device__ __forceinline__ void move_rows(const BlitEntry& e, uint32_t r0, uint32_t r1) {
    constexpr uint32_t U = sizeof(V);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const char* __restrict__ src = static_cast<const char*>(e.src);
    char* __restrict__ dst = static_cast<char*>(e.dst);
    const uint32_t whole = e.row_bytes / U * U;  // bytes of a row in whole units
    for (uint32_t r = r0 + 4 * wave; r < r1; r += 4 * waves) {
        for (uint32_t c = lane * U; c < whole; c += 64 * U) {
            V v[4];
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k)  // four rows in flight per lane: loads first, then stores
                if (r + k < r1) v[k] = *reinterpret_cast<const V*>(src + static_cast<size_t>(r + k) * e.src_pitch + c);
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k)
                if (r + k < r1) *reinterpret_cast<V*>(dst + static_cast<size_t>(r + k) * e.dst_pitch + c) = v[k];
        }
        if (whole + lane < e.row_bytes)  // bytes of a row beyond its whole units (fewer than sizeof(V) <= 64)
            for (uint32_t k = 0; k < 4 && r + k < r1; ++k)
                dst[static_cast<size_t>(r + k) * e.dst_pitch + whole + lane] = src[static_cast<size_t>(r + k) * e.src_pitch + whole + lane];
    }
}

// A FEW workgroups (the link, not the shader, bounds this kernel: ~50 workgroups keep it full, and the resampling
// kernels of the next group of frames run beside it on the other compute units) walk over work items = (entry, slice of
// `rows_per_item` rows), item = blockIdx.x, blockIdx.x + gridDim.x, ...
__global__ __launch_bounds__(256) void blit_rows_kernel(const BlitEntry* __restrict__ table, uint32_t entries, uint32_t rows_per_item,
                                                        uint32_t items_per_entry, int wave_priority) {
    if (wave_priority) __builtin_amdgcn_s_setprio(3);
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

namespace {
int blit_wave_priority() { return knobs::geti(JINC_KNOB_BLIT_SETPRIO, 1); }  // A/B knob BLIT_SETPRIO (default 1)
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
                       static_cast<uint32_t>(count), rows_per_item, items_per_entry, blit_wave_priority());
    return hipGetLastError();
}

}  // namespace jinc
