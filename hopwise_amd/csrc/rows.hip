// Row gather / scatter-add used on both sides of the row-sharded table exchange:
// the owner gathers requested rows into a send buffer; on the way back it adds returned row-gradients
// into its gradient shard.
#include "hw_common.h"

namespace hw {

// one 16-lane group per row, 16-byte chunks when dim % 4 == 0
template <bool VEC>
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ table, int64_t n_table, int dim,
                                                          const int64_t* __restrict__ idx, int64_t n,
                                                          float* __restrict__ out, int32_t* oob) {
    const int grp = threadIdx.x >> 4, gl = threadIdx.x & 15;
    for (int64_t i = (int64_t)blockIdx.x * 16 + grp; i < n; i += (int64_t)gridDim.x * 16) {
        const int r = clamp_row(idx[i], n_table, gl == 0 ? oob : nullptr);
        if (VEC) {
            const f32x4* __restrict__ s = reinterpret_cast<const f32x4*>(table + (size_t)r * dim);
            f32x4* __restrict__ d = reinterpret_cast<f32x4*>(out + (size_t)i * dim);
            for (int c = gl; c < dim / 4; c += 16) d[c] = s[c];
        } else {
            for (int c = gl; c < dim; c += 16) out[(size_t)i * dim + c] = table[(size_t)r * dim + c];
        }
    }
}

// dword-per-lane so that each atomic wave-instruction touches contiguous bytes of a row
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(float* __restrict__ table, int64_t n_table, int dim,
                                                               const int64_t* __restrict__ idx, int64_t n,
                                                               const float* __restrict__ src) {
    const int w = wave_in_block(), lane = threadIdx.x & 63;
    for (int64_t i = (int64_t)blockIdx.x * 4 + w; i < n; i += (int64_t)gridDim.x * 4) {
        const int64_t r = idx[i];
        if (r < 0 || r >= n_table) continue;
        for (int c = lane; c < dim; c += 64) atomicAdd(table + (size_t)r * dim + c, src[(size_t)i * dim + c]);
    }
}

}  // namespace hw

using namespace hw;

extern "C" int hw_gather_rows(const float* table, int64_t n_table_rows, int32_t dim, const int64_t* idx, int64_t n,
                              float* out, int32_t* oob_counter, void* stream) {
    if (!table || !idx || !out || n_table_rows < 1 || dim < 1 || n < 0) return HW_ERR_INVALID_ARG;
    if (n == 0) return HW_OK;
    int64_t grid = (n + 15) / 16;
    if (grid > 16384) grid = 16384;
    HW_CLEAR_ERR();
    const bool vec = (dim % 4 == 0) && ((((uintptr_t)table | (uintptr_t)out) & 15) == 0);
    if (vec)
        hipLaunchKernelGGL((gather_rows_kernel<true>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, table,
                           n_table_rows, dim, idx, n, out, oob_counter);
    else
        hipLaunchKernelGGL((gather_rows_kernel<false>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, table,
                           n_table_rows, dim, idx, n, out, oob_counter);
    return HW_LAUNCH_RC();
}

extern "C" int hw_scatter_add_rows(float* table, int64_t n_table_rows, int32_t dim, const int64_t* idx, int64_t n,
                                   const float* src, void* stream) {
    if (!table || !idx || !src || n_table_rows < 1 || dim < 1 || n < 0) return HW_ERR_INVALID_ARG;
    if (n == 0) return HW_OK;
    int64_t grid = (n + 3) / 4;
    if (grid > 16384) grid = 16384;
    HW_CLEAR_ERR();
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, table,
                       n_table_rows, dim, idx, n, src);
    return HW_LAUNCH_RC();
}

// Graph-safe stream-ordered fill (hw_common.h: fill_bytes_async): what a caller that captures the engine's launches in a HIP
// graph uses instead of hipMemsetAsync, whose memset NODES replay a stale pattern on this stack.
extern "C" int hw_fill_bytes(void* dst, int32_t byte_value, int64_t n_bytes, void* stream) {
    if (n_bytes < 0 || (n_bytes > 0 && !dst)) return HW_ERR_INVALID_ARG;
    HW_CLEAR_ERR();
    return hw::fill_bytes_async(dst, byte_value, (size_t)n_bytes, (hipStream_t)stream) == hipSuccess ? HW_OK : HW_ERR_LAUNCH;
}
