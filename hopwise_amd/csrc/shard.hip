// Device-side routing of a row-sharded table lookup.
//
// A rank looks up `n` ids in a table whose rows are sharded owner = id % P, local row = id / P.  Everything the
// exchange needs is produced here, on the device, so the host never reads a count:
//
//   mark      every id sets one bit of a bitmap laid out OWNER-MAJOR: bit  owner * seg_bits + local
//   scan      one workgroup per owner: popcount-scan of its segment -> exclusive prefix per word, the number of
//             distinct ids the owner is asked for (send_counts[owner]) and the list of their local rows in ascending
//             order (send_local[owner * cap + j]); more than `cap` distinct ids for one owner raises *overflow
//   remap     inv[i] = owner * cap + rank of ids[i] among the owner's distinct ids
//
// The exchange itself is three fixed-size all-to-alls (counts, cap local rows per peer, cap rows per peer), and the
// buffer the rows come back into IS the compact table: row owner * cap + j.  Rows past a peer's count are never
// written nor indexed.  Dedup is exact (a bitmap, not a hash) and the order is deterministic (ascending id inside an
// owner), so a fixed batch exchanges the same bytes in the same places every time.
//
// Owner side: shard_gather copies, for every requesting peer p, rows recv_local[p * cap + j], j < recv_counts[p], into
// the send buffer; shard_scatter_add adds returned row gradients into the gradient shard (float atomics: two peers
// may ask for the same row).
#include "hw_common.h"

namespace hw {

__global__ __launch_bounds__(256) void shard_mark_kernel(const int64_t* __restrict__ ids, int64_t n, int64_t n_rows,
                                                         int P, int64_t seg_bits, uint32_t* __restrict__ bitmap,
                                                         int32_t* oob) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t id = clamp_row(ids[i], n_rows, oob);
        const int64_t bit = (id % P) * seg_bits + id / P;
        atomicOr(bitmap + (bit >> 5), 1u << (bit & 31));
    }
}

// block-wide exclusive scan of one int per thread (1024 threads = 16 waves); returns the exclusive prefix, total in `sum`
__device__ __forceinline__ int block_exclusive_scan(int v, int* lds16, int& sum) {
    const int lane = threadIdx.x & 63, w = wave_in_block();
    int incl = v;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const int up = __shfl_up(incl, m, HW_WAVE);
        if (lane >= m) incl += up;
    }
    if (lane == 63) lds16[w] = incl;
    __syncthreads();
    int base = 0, tot = 0;
    const int nw = blockDim.x >> 6;
    for (int k = 0; k < nw; ++k) {
        const int s = lds16[k];
        if (k < w) base += s;
        tot += s;
    }
    __syncthreads();
    sum = tot;
    return base + incl - v;
}

__global__ __launch_bounds__(1024) void shard_scan_kernel(const uint32_t* __restrict__ bitmap, int64_t seg_words, int cap,
                                                          int32_t* __restrict__ word_prefix,
                                                          int32_t* __restrict__ send_counts, int64_t count_stride,
                                                          int32_t* __restrict__ send_local, int64_t local_stride,
                                                          int32_t* overflow) {
    __shared__ int lds16[16];
    const int owner = blockIdx.x;
    const uint32_t* seg = bitmap + (int64_t)owner * seg_words;
    int32_t* pre = word_prefix + (int64_t)owner * seg_words;
    int running = 0;
    for (int64_t w0 = 0; w0 < seg_words; w0 += blockDim.x) {
        const int64_t w = w0 + threadIdx.x;
        uint32_t bits = w < seg_words ? seg[w] : 0u;
        int tot;
        const int ex = running + block_exclusive_scan(__popc(bits), lds16, tot);
        if (w < seg_words) pre[w] = ex;
        int j = ex;
        while (bits) {  // distinct local rows in ascending order
            const int b = __ffs(bits) - 1;
            bits &= bits - 1;
            if (j < cap) send_local[(int64_t)owner * local_stride + j] = (int32_t)(w * 32 + b);
            ++j;
        }
        running += tot;
    }
    if (threadIdx.x == 0) {
        send_counts[(int64_t)owner * count_stride] = running < cap ? running : cap;
        if (running > cap) atomicMax(overflow, running);
    }
}

__global__ __launch_bounds__(256) void shard_remap_kernel(const int64_t* __restrict__ ids, int64_t n, int64_t n_rows, int P,
                                                          int64_t seg_bits, int cap, int64_t inv_stride,
                                                          const uint32_t* __restrict__ bitmap,
                                                          const int32_t* __restrict__ word_prefix,
                                                          int64_t* __restrict__ inv) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t id = clamp_row(ids[i], n_rows, nullptr);
        const int64_t owner = id % P;
        const int64_t bit = owner * seg_bits + id / P;
        const uint32_t word = bitmap[bit >> 5];
        int j = word_prefix[bit >> 5] + __popc(word & ((1u << (bit & 31)) - 1u));
        if (j >= cap) j = cap - 1;  // overflow is reported by the scan; keep the index inside the compact table
        inv[i] = owner * inv_stride + j;
    }
}

// one 16-lane group per row; rows past a peer's count are skipped.  ACC: atomic add into the shard (gradient path).
// The exchange buffer holds, per peer p, `peer_rows` rows; this table's rows sit at row_offset inside a peer's chunk:
// buf row of (p, j) = p * peer_rows + row_offset + j  (peer_rows = n_tables * cap packs every table of a lookup into ONE
// message per peer; n_tables = 1, row_offset = 0 is the plain [P * cap, d] buffer).
template <bool VEC, bool ACC>
__global__ __launch_bounds__(256) void shard_rows_kernel(float* __restrict__ shard, int64_t n_local, int dim,
                                                         const int32_t* __restrict__ local, int64_t local_stride,
                                                         const int32_t* __restrict__ counts, int64_t count_stride, int P,
                                                         int cap, float* __restrict__ buf, int64_t peer_rows,
                                                         int64_t row_offset, int32_t* oob) {
    const int grp = threadIdx.x >> 4, gl = threadIdx.x & 15;
    const int64_t total = (int64_t)P * cap;
    for (int64_t i = (int64_t)blockIdx.x * 16 + grp; i < total; i += (int64_t)gridDim.x * 16) {
        const int p = (int)(i / cap), j = (int)(i - (int64_t)p * cap);
        if (j >= counts[(int64_t)p * count_stride]) continue;
        const int r = clamp_row(local[(int64_t)p * local_stride + j], n_local, gl == 0 ? oob : nullptr);
        float* __restrict__ s = shard + (size_t)r * dim;
        float* __restrict__ b = buf + (size_t)((int64_t)p * peer_rows + row_offset + j) * dim;
        if (ACC) {
            for (int c = gl; c < dim; c += 16) atomicAdd(s + c, b[c]);
        } else if (VEC) {
            for (int c = gl; c < dim / 4; c += 16) reinterpret_cast<f32x4*>(b)[c] = reinterpret_cast<const f32x4*>(s)[c];
        } else {
            for (int c = gl; c < dim; c += 16) b[c] = s[c];
        }
    }
}

}  // namespace hw

using namespace hw;

extern "C" int64_t hw_shard_plan_workspace_bytes(int64_t n_rows, int32_t n_ranks) {
    if (n_rows < 1 || n_ranks < 1) return 0;
    const int64_t seg_words = ((n_rows + n_ranks - 1) / n_ranks + 31) / 32;
    return 2 * seg_words * n_ranks * (int64_t)sizeof(uint32_t);  // bitmap + word prefixes
}

extern "C" int hw_shard_plan(const int64_t* ids, int64_t n, int64_t n_rows, int32_t n_ranks, int32_t cap, void* workspace,
                             int64_t* inv, int64_t inv_stride, int32_t* send_counts, int64_t count_stride,
                             int32_t* send_local, int64_t local_stride, int32_t* overflow, int32_t* oob_counter,
                             void* stream) {
    if (!workspace || !send_counts || !send_local || !overflow || n < 0 || n_rows < 1 || n_ranks < 1 || n_ranks > 1024 ||
        cap < 1 || (n > 0 && (!ids || !inv)) || inv_stride < cap || count_stride < 1 || local_stride < cap)
        return HW_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int64_t seg_words = ((n_rows + n_ranks - 1) / n_ranks + 31) / 32, seg_bits = seg_words * 32;
    uint32_t* bitmap = (uint32_t*)workspace;
    int32_t* prefix = (int32_t*)(bitmap + seg_words * n_ranks);
    HW_CLEAR_ERR();
    if (hw::fill_bytes_async(bitmap, 0, seg_words * n_ranks * sizeof(uint32_t), st) != hipSuccess) return HW_ERR_LAUNCH;
    int64_t grid = (n + 255) / 256;
    if (grid > 4096) grid = 4096;
    if (n > 0)
        hipLaunchKernelGGL(shard_mark_kernel, dim3((unsigned)grid), dim3(256), 0, st, ids, n, n_rows, (int)n_ranks,
                           seg_bits, bitmap, oob_counter);
    hipLaunchKernelGGL(shard_scan_kernel, dim3((unsigned)n_ranks), dim3(1024), 0, st, bitmap, seg_words, (int)cap, prefix,
                       send_counts, count_stride, send_local, local_stride, overflow);
    if (n > 0)
        hipLaunchKernelGGL(shard_remap_kernel, dim3((unsigned)grid), dim3(256), 0, st, ids, n, n_rows, (int)n_ranks,
                           seg_bits, (int)cap, inv_stride, bitmap, prefix, inv);
    return HW_LAUNCH_RC();
}

static int shard_rows_launch(float* shard, int64_t n_local, int32_t dim, const int32_t* local, int64_t local_stride,
                             const int32_t* counts, int64_t count_stride, int32_t n_ranks, int32_t cap, float* buf,
                             int64_t peer_rows, int64_t row_offset, int32_t* oob, void* stream, bool acc) {
    if (!shard || !local || !counts || !buf || n_local < 1 || dim < 1 || n_ranks < 1 || cap < 1 || local_stride < cap ||
        count_stride < 1 || row_offset < 0 || peer_rows < row_offset + cap)
        return HW_ERR_INVALID_ARG;
    int64_t grid = ((int64_t)n_ranks * cap + 15) / 16;
    if (grid > 16384) grid = 16384;
    HW_CLEAR_ERR();
    const bool vec = (dim % 4 == 0) && ((((uintptr_t)shard | (uintptr_t)buf) & 15) == 0);
    hipStream_t st = (hipStream_t)stream;
#define ROWS(V, A) hipLaunchKernelGGL((shard_rows_kernel<V, A>), dim3((unsigned)grid), dim3(256), 0, st, shard, n_local, (int)dim, local, local_stride, counts, count_stride, (int)n_ranks, (int)cap, buf, peer_rows, row_offset, oob)
    if (acc) ROWS(false, true);
    else if (vec) ROWS(true, false);
    else ROWS(false, false);
#undef ROWS
    return HW_LAUNCH_RC();
}

extern "C" int hw_shard_gather(const float* shard, int64_t n_local_rows, int32_t dim, const int32_t* recv_local,
                               int64_t local_stride, const int32_t* recv_counts, int64_t count_stride, int32_t n_ranks,
                               int32_t cap, float* out, int64_t peer_rows, int64_t row_offset, int32_t* oob_counter,
                               void* stream) {
    return shard_rows_launch(const_cast<float*>(shard), n_local_rows, dim, recv_local, local_stride, recv_counts,
                             count_stride, n_ranks, cap, out, peer_rows, row_offset, oob_counter, stream, false);
}

extern "C" int hw_shard_scatter_add(float* grad_shard, int64_t n_local_rows, int32_t dim, const int32_t* recv_local,
                                    int64_t local_stride, const int32_t* recv_counts, int64_t count_stride, int32_t n_ranks,
                                    int32_t cap, const float* src, int64_t peer_rows, int64_t row_offset,
                                    int32_t* oob_counter, void* stream) {
    return shard_rows_launch(grad_shard, n_local_rows, dim, recv_local, local_stride, recv_counts, count_stride, n_ranks, cap,
                             const_cast<float*>(src), peer_rows, row_offset, oob_counter, stream, true);
}
