// TransR (model/knowledge_graph_embedding_recommender/transr.py:67-70): entities are mapped into the relation's space
// by a per-relation d x d matrix, p = M_r e, and scored like TransE there: -||M_r h + r - M_r t||.
//
// The matrices make this the one scorer whose per-triple cost is d^2, so it is split in two: these kernels produce /
// differentiate the projected rows, and the TransE gather kernels score them (the projected buffers are handed to
// hw_kge_score_grouped as its head / tail tables).
//   project_rows        out[i, c] = sum_k M[r_i][c][k] E[e_i][k]          wave per row, 4 matrix rows per reduction
//   project_rows_grad_e gE[e_i, k] += sum_c M[r_i][c][k] gP[i, c]          wave per row, no reductions
//   project_rows_grad_m gM[r][c][k] += sum_{i: r_i = r} gP[i, c] E[e_i][k]  block per (relation, 32-column tile,
//                       row split); accumulates in registers, one atomic flush per block -- deterministic per block,
//                       S-way contention instead of one atomic per (row, c, k)
#include "hw_common.h"

namespace hw {

template <int NE>
__global__ __launch_bounds__(256) void project_rows_kernel(const float* __restrict__ mats, int64_t n_rel,
                                                           const float* __restrict__ table, int64_t n_rows_table,
                                                           int dim, const int64_t* __restrict__ r_idx, int64_t r_fixed,
                                                           const int64_t* __restrict__ e_idx, int64_t n,
                                                           float* __restrict__ out) {
    using R = Row<64, NE, false>;
    const int gl = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int r = clamp_row(r_idx ? r_idx[i] : r_fixed, n_rel, nullptr);
    const int e = clamp_row(e_idx[i], n_rows_table, nullptr);
    R ev;
    ev.load(table + (size_t)e * dim, dim, gl);
    const float* __restrict__ M = mats + (size_t)r * dim * dim;
    for (int c0 = 0; c0 < dim; c0 += 4) {
        float p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = min(c0 + u, dim - 1);
            R m;
            m.load_raw(M + (size_t)c * dim, dim, gl);
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < NE; ++k) s = fmaf(m.v[k], ev.v[k], s);  // ev is zero padded: junk slots of m add 0
            p[u] = s;
        }
        int which;
        const float tot = group_reduce_multi<64, 4>(p, gl, which);
        if ((gl & 15) == 0 && c0 + which < dim) out[i * dim + c0 + which] = tot;
    }
}

template <int NE>
__global__ __launch_bounds__(256) void project_rows_grad_e_kernel(const float* __restrict__ mats, int64_t n_rel,
                                                                  int64_t n_rows_table, int dim,
                                                                  const int64_t* __restrict__ r_idx, int64_t r_fixed,
                                                                  const int64_t* __restrict__ e_idx, int64_t n,
                                                                  const float* __restrict__ gp, float* __restrict__ ge) {
    using R = Row<64, NE, false>;
    const int gl = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int r = clamp_row(r_idx ? r_idx[i] : r_fixed, n_rel, nullptr);
    const int e = clamp_row(e_idx[i], n_rows_table, nullptr);
    const float* __restrict__ M = mats + (size_t)r * dim * dim;
    const float* __restrict__ g = gp + i * dim;
    float acc[NE];
#pragma unroll
    for (int k = 0; k < NE; ++k) acc[k] = 0.f;
#pragma unroll 4
    for (int c = 0; c < dim; ++c) {
        const float gc = g[c];  // wave-uniform
        R m;
        m.load_raw(M + (size_t)c * dim, dim, gl);
#pragma unroll
        for (int k = 0; k < NE; ++k) acc[k] = fmaf(gc, m.v[k], acc[k]);
    }
#pragma unroll
    for (int k = 0; k < NE; ++k) {
        const int c = R::col(k, dim, gl);
        if (c >= 0 && acc[k] != 0.f) atomicAdd(ge + (size_t)e * dim + c, acc[k]);
    }
}

constexpr int GM_T = 32;     // columns (k) per block
constexpr int GM_MAXC = 32;  // register rows per thread: dim <= 8 * 32 = 256

__global__ __launch_bounds__(256) void project_rows_grad_m_kernel(const float* __restrict__ table, int64_t n_rows_table,
                                                                  int dim, const int64_t* __restrict__ r_idx,
                                                                  int64_t r_only, const int64_t* __restrict__ e_idx,
                                                                  int64_t n, int n_splits, const float* __restrict__ gp,
                                                                  float* __restrict__ gm) {
    __shared__ float gsh[HW_MAX_DIM];
    __shared__ int match[256];
    __shared__ int n_match;
    const int r = r_idx ? (int)blockIdx.x : (int)r_only;
    const int k0 = blockIdx.y * GM_T;
    const int tk = threadIdx.x % GM_T, tc = threadIdx.x / GM_T;  // 8 row groups
    const int k = k0 + tk;
    const int64_t per = (n + n_splits - 1) / n_splits;
    const int64_t lo = (int64_t)blockIdx.z * per, hi = min(n, lo + per);
    float acc[GM_MAXC];
#pragma unroll
    for (int j = 0; j < GM_MAXC; ++j) acc[j] = 0.f;
    bool any = false;
    for (int64_t base = lo; base < hi; base += 256) {
        const int64_t i = base + threadIdx.x;
        const bool m = i < hi && (!r_idx || r_idx[i] == r);
        __syncthreads();
        if (threadIdx.x == 0) n_match = 0;
        __syncthreads();
        if (m) match[atomicAdd(&n_match, 1)] = (int)(i - base);  // order inside a chunk does not matter: it is a sum...
        __syncthreads();
        const int nm = n_match;
        // ... but keep it deterministic: sort the (at most 256) offsets by value with a rank pass
        int mine = threadIdx.x < nm ? match[threadIdx.x] : 0x7fffffff, rank = 0;
        for (int q = 0; q < nm; ++q) rank += match[q] < mine;
        __syncthreads();
        if ((int)threadIdx.x < nm) match[rank] = mine;
        __syncthreads();
        for (int q = 0; q < nm; ++q) {
            const int64_t row = base + match[q];
            for (int c = threadIdx.x; c < dim; c += 256) gsh[c] = gp[row * dim + c];
            __syncthreads();
            const int e = clamp_row(e_idx[row], n_rows_table, nullptr);
            const float ek = k < dim ? table[(size_t)e * dim + k] : 0.f;
#pragma unroll
            for (int j = 0; j < GM_MAXC; ++j) {
                const int c = tc + 8 * j;
                if (c < dim) acc[j] = fmaf(gsh[c], ek, acc[j]);
            }
            any = true;
            __syncthreads();
        }
    }
    if (!any || k >= dim) return;
    float* __restrict__ dst = gm + (size_t)r * dim * dim;
#pragma unroll
    for (int j = 0; j < GM_MAXC; ++j) {
        const int c = tc + 8 * j;
        if (c < dim && acc[j] != 0.f) atomicAdd(dst + (size_t)c * dim + k, acc[j]);
    }
}

}  // namespace hw

using namespace hw;

extern "C" int hw_kge_project_rows(const float* mats, int64_t n_rel, const float* table, int64_t n_table_rows,
                                   int32_t dim, const int64_t* r_idx, int64_t r_fixed, const int64_t* e_idx, int64_t n,
                                   float* out, void* stream) {
    if (!mats || !table || !e_idx || !out || dim < 1 || dim > HW_MAX_DIM || n_rel < 1 || n_table_rows < 1 || n < 0)
        return HW_ERR_INVALID_ARG;
    if (n == 0) return HW_OK;
    HW_CLEAR_ERR();
    const unsigned grid = (unsigned)((n + 3) / 4);
    hipStream_t st = (hipStream_t)stream;
#define CALL(NE_)                                                                                                   \
    hipLaunchKernelGGL((project_rows_kernel<NE_>), dim3(grid), dim3(256), 0, st, mats, n_rel, table, n_table_rows, dim, \
                       r_idx, r_fixed, e_idx, n, out)
    if (dim <= 64) CALL(1);
    else if (dim <= 256) CALL(4);
    else CALL(16);
#undef CALL
    return HW_LAUNCH_RC();
}

extern "C" int hw_kge_project_rows_grad(const float* mats, int64_t n_rel, const float* table, int64_t n_table_rows,
                                        int32_t dim, const int64_t* r_idx, int64_t r_fixed, const int64_t* e_idx,
                                        int64_t n, const float* g_out, float* g_table, float* g_mats, void* stream) {
    if (!mats || !table || !e_idx || !g_out || dim < 1 || n_rel < 1 || n_table_rows < 1 || n < 0)
        return HW_ERR_INVALID_ARG;
    if (dim > 8 * GM_MAXC && g_mats) return HW_ERR_UNSUPPORTED;  // d^2 floats per relation: TransR is a small-d model
    if (dim > HW_MAX_DIM) return HW_ERR_INVALID_ARG;
    if (n == 0) return HW_OK;
    HW_CLEAR_ERR();
    hipStream_t st = (hipStream_t)stream;
    if (g_table) {
        const unsigned grid = (unsigned)((n + 3) / 4);
#define CALL(NE_)                                                                                                     \
    hipLaunchKernelGGL((project_rows_grad_e_kernel<NE_>), dim3(grid), dim3(256), 0, st, mats, n_rel, n_table_rows, dim, \
                       r_idx, r_fixed, e_idx, n, g_out, g_table)
        if (dim <= 64) CALL(1);
        else if (dim <= 256) CALL(4);
        else CALL(16);
#undef CALL
    }
    if (g_mats) {
        const int64_t rf = r_fixed < 0 ? 0 : (r_fixed >= n_rel ? n_rel - 1 : r_fixed);
        const unsigned n_r = r_idx ? (unsigned)n_rel : 1u;
        const unsigned tiles = (unsigned)((dim + GM_T - 1) / GM_T);
        // rows are split so that the launch has a few hundred blocks even when one relation owns every row
        int splits = (int)((512 + n_r * tiles - 1) / (n_r * tiles));
        const int max_splits = (int)((n + 255) / 256);
        splits = splits < 1 ? 1 : (splits > max_splits ? max_splits : splits);
        hipLaunchKernelGGL(project_rows_grad_m_kernel, dim3(n_r, tiles, (unsigned)splits), dim3(256), 0, st, table,
                           n_table_rows, dim, r_idx, rf, e_idx, n, splits, g_out, g_mats);
    }
    return HW_LAUNCH_RC();
}
