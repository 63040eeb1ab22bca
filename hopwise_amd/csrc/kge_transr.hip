// TransR (model/knowledge_graph_embedding_recommender/transr.py:67-70): entities are mapped into the relation's space
// by a per-relation d x d matrix, p = M_r e, and scored like TransE there: -||M_r h + r - M_r t||.
//
// The matrices make this the one scorer whose per-triple cost is d^2, so it is split in two: these kernels produce /
// differentiate the projected rows, and the TransE gather kernels score them (the projected buffers are handed to
// hw_kge_score_grouped as its head / tail tables).
//   project_rows        out[i, c] = sum_k M[r_i][c][k] E[e_i][k]          wave per row, 4 matrix rows per reduction
//   project_rows_grad_e gE[e_i, k] += sum_c M[r_i][c][k] gP[i, c]          wave per row, no reductions
//   gM[r][c][k] += sum_{i: r_i = r} gP[i, c] E[e_i][k]                      outer_by_relation (rel_matrix.h)
#include "rel_matrix.h"

namespace hw {

template <int NE>
__global__ __launch_bounds__(256) void project_rows_grad_e_kernel(const float* __restrict__ mats, int64_t n_rel,
                                                                  int64_t n_rows_table, int dim,
                                                                  const int64_t* __restrict__ r_idx, int64_t r_fixed,
                                                                  const int64_t* __restrict__ e_idx, int64_t n,
                                                                  const float* __restrict__ gp, float* __restrict__ ge) {
    using R = Row<64, NE, false>;
    const int gl = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + wave_in_block();
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
    hipLaunchKernelGGL((project_rows_kernel<NE_, false>), dim3(grid), dim3(256), 0, st, mats, n_rel, table,          \
                       n_table_rows, dim, r_idx, r_fixed, e_idx, n, out, nullptr, 0, 1)
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
    if (g_mats) {  // gM[r][c][k] += sum_{i: r_i = r} gP[i, c] E[e_i][k]
        const OuterSide A{g_out, dim, nullptr, n, dim};
        const OuterSide B{table, dim, e_idx, n_table_rows, dim};
        launch_outer_by_relation(A, B, r_idx, n_rel, r_fixed, 1, n, g_mats, st);
    }
    return HW_LAUNCH_RC();
}
