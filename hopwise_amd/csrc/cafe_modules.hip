// CAFE relation modules, evaluation mode, for all (level, relation) groups of a run_program call in ONE host call.
//
// SymbolicNetwork's modules (cafe.py:656-699) are three small Linear layers around two BatchNorm1d's and a residual; a
// run_program call applies one module to ~256 rows for each of the 13 (level, relation) pairs of the layout tree -- ~150
// launches of 5-30 us each, issued from Python: the call is bound by the host (3.75 ms per 256 users against 2.4 ms of device
// work; 3.5 ms with this file).  Here the same launches -- the SAME kernels, in the same order: hw_gemm_f32 with its bias / ReLU
// epilogue, hw_batchnorm1d_forward in evaluation mode -- are issued from C++; the glue that torch ran (cat of the gathered
// rows, relu(x + residual), the index_copy into the node table) is three trivial kernels below.  Results are bit-identical to
// the module-by-module path (tests/test_gpu_cafe.py).
#include "hw_common.h"

namespace hw {

// feature[r] = [ eh(r) | uv(r) ]: eh row from user_vec (index < n_users_rows) or vec_all (index - n_user_rows)
__global__ __launch_bounds__(256) void cafe_gather_cat_kernel(const float* __restrict__ user_vec,
                                                              const float* __restrict__ vec_all, int64_t n_user_rows, int d,
                                                              const int64_t* __restrict__ eh_idx,
                                                              const int64_t* __restrict__ uv_idx, int64_t n,
                                                              float* __restrict__ out) {
    const int64_t total = n * 2 * d;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / (2 * d);
        const int c = (int)(i - r * 2 * d);
        float v;
        if (c < d) {
            const int64_t e = eh_idx[r];
            v = e < n_user_rows ? user_vec[e * d + c] : vec_all[(e - n_user_rows) * d + c];
        } else {
            v = user_vec[uv_idx[r] * d + (c - d)];
        }
        out[i] = v;
    }
}

// out = relu(a + b)  (F.relu(fc2(x) + feature), cafe.py:694)
__global__ __launch_bounds__(256) void cafe_add_relu_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n,
                                                            float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = fmaxf(a[i] + b[i], 0.f);
}

// vec_all[rows[r]] = y[r] (+ feature[r, :d] when residual: RelationModule's "+ eh_vec", cafe.py:667)
__global__ __launch_bounds__(256) void cafe_scatter_rows_kernel(const float* __restrict__ y, const float* __restrict__ feature,
                                                                int d, const int64_t* __restrict__ rows, int64_t n,
                                                                float* __restrict__ vec_all) {
    const int64_t total = n * d;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / d;
        const int c = (int)(i - r * d);
        float v = y[i];
        if (feature) v += feature[r * 2 * d + c];
        vec_all[rows[r] * d + c] = v;
    }
}

// ---- the fused form: ONE launch per level ---------------------------------------------------------------------------------
// A module application is a 2d -> 256 -> (2d ->) d perceptron over ~256 rows: 60 MFLOP.  As GEMM / batch-norm / glue launches
// it is 8 launches of 5-25 us per (level, relation) group -- ~100 launches and ~1.1 ms of a 2.0 ms run_program call, none of
// them doing more than waiting for its own first loads.  Here a block takes CF_R rows of one group through the whole module
// with the activations in LDS (transposed: [column][row], so that a k-step reads its CF_R inputs as two broadcast
// ds_read_b128) and one output column per thread; the weights are read in the TRANSPOSED layout [in, out] (one coalesced
// 1 KB row per k-step, L2-resident: 0.5 MB per module), which the caller prepares once per weight version.  Groups of one
// level are independent (they read node vectors of earlier levels only): one launch covers up to CF_MAXG of them.
// Arithmetic: fp32 FMA chains in k order (the loop path runs the products as three-way bf16 splits on the matrix pipes; both
// are fp32-level, they differ in summation order), batch norm as fmaf(x, rs g, beta - mu rs g) like hw_batchnorm1d_forward.
constexpr int CF_R = 8, CF_MAXG = 8, CF_H = 256;

struct CfLevel {
    int32_t n_groups;
    int32_t blk0[CF_MAXG + 1];  // first block of each group
    int64_t r0[CF_MAXG], n[CF_MAXG];
    hw_cafe_module mod[CF_MAXG];
};

// acc[r] += sum_k act[k][r] * wt[k * ldw + c] over k < K (act: LDS [K][CF_R]), in k order.  The weight loads of U k-steps
// are issued together: the loop is paced by the latency of its L2-resident weight rows, not by their bandwidth (4 in flight:
// 65 us per level launch; 16: see DESIGN.md A16).
template <int U>
__device__ __forceinline__ void cf_dot_steps(const float (*act)[CF_R], int k, const float* __restrict__ wt, int ldw, int c,
                                             float (&acc)[CF_R]) {
    float w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) w[u] = wt[(size_t)(k + u) * ldw + c];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(&act[k + u][0]);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(&act[k + u][4]);
        acc[0] = fmaf(a0.x, w[u], acc[0]), acc[1] = fmaf(a0.y, w[u], acc[1]);
        acc[2] = fmaf(a0.z, w[u], acc[2]), acc[3] = fmaf(a0.w, w[u], acc[3]);
        acc[4] = fmaf(a1.x, w[u], acc[4]), acc[5] = fmaf(a1.y, w[u], acc[5]);
        acc[6] = fmaf(a1.z, w[u], acc[6]), acc[7] = fmaf(a1.w, w[u], acc[7]);
    }
}

__device__ __forceinline__ void cf_dot(const float (*act)[CF_R], int K, const float* __restrict__ wt, int ldw, int c,
                                       float (&acc)[CF_R]) {
    int k = 0;
    for (; k + 16 <= K; k += 16) cf_dot_steps<16>(act, k, wt, ldw, c, acc);
    for (; k + 4 <= K; k += 4) cf_dot_steps<4>(act, k, wt, ldw, c, acc);
    for (; k < K; ++k) cf_dot_steps<1>(act, k, wt, ldw, c, acc);
}

__global__ __launch_bounds__(256) void cafe_level_kernel(CfLevel L, const float* __restrict__ user_vec, int64_t n_user_rows, int d,
                                                         const int64_t* __restrict__ eh_idx, const int64_t* __restrict__ uv_idx,
                                                         const int64_t* __restrict__ out_rows, float* vec_all) {
    __shared__ __attribute__((aligned(16))) float xT[CF_H][CF_R], h1T[CF_H][CF_R], h2T[CF_H][CF_R];
    int g = 0;
    while (g + 1 < L.n_groups && (int)blockIdx.x >= L.blk0[g + 1]) ++g;  // block-uniform
    const hw_cafe_module& m = L.mod[g];
    const int64_t rb = L.r0[g] + (int64_t)((int)blockIdx.x - L.blk0[g]) * CF_R;
    const int64_t left = L.r0[g] + L.n[g] - rb;
    const int nr = left < CF_R ? (int)left : CF_R;
    const int tid = threadIdx.x, d2 = 2 * d;
    // feature[r] = [ eh(r) | uv(r) ], transposed into LDS; rows past the group's end are zero
    // (vec_all is not __restrict__: other blocks of this launch write rows of it -- rows no block of this level reads)
    for (int i = tid; i < CF_R * d2; i += 256) {
        const int r = i / d2, c = i - r * d2;
        float v = 0.f;
        if (r < nr) {
            if (c < d) {
                const int64_t e = eh_idx[rb + r];
                v = e < n_user_rows ? user_vec[e * d + c] : vec_all[(e - n_user_rows) * d + c];
            } else {
                v = user_vec[uv_idx[rb + r] * d + (c - d)];
            }
        }
        xT[c][r] = v;
    }
    __syncthreads();
    {  // fc1 + ReLU, bn1 (evaluation): 256 columns, one per thread
        const int c = tid;
        float acc[CF_R];
        const float b = m.b1[c];
#pragma unroll
        for (int r = 0; r < CF_R; ++r) acc[r] = b;
        cf_dot(xT, d2, m.w1, CF_H, c, acc);
        const float rs = 1.f / sqrtf(m.bn1_var[c] + m.bn1_eps), gm = m.bn1_weight ? m.bn1_weight[c] : 1.f;
        const float scale = rs * gm, shift = (m.bn1_bias ? m.bn1_bias[c] : 0.f) - m.bn1_mean[c] * rs * gm;
#pragma unroll
        for (int r = 0; r < CF_R; ++r) h1T[c][r] = fmaf(fmaxf(acc[r], 0.f), scale, shift);
    }
    __syncthreads();
    if (m.deep) {
        if (tid < d2) {  // relu(fc2(x) + feature), bn2
            const int c = tid;
            float acc[CF_R];
            const float b = m.b2[c];
#pragma unroll
            for (int r = 0; r < CF_R; ++r) acc[r] = b;
            cf_dot(h1T, CF_H, m.w2, d2, c, acc);
            const float rs = 1.f / sqrtf(m.bn2_var[c] + m.bn2_eps), gm = m.bn2_weight ? m.bn2_weight[c] : 1.f;
            const float scale = rs * gm, shift = (m.bn2_bias ? m.bn2_bias[c] : 0.f) - m.bn2_mean[c] * rs * gm;
#pragma unroll
            for (int r = 0; r < CF_R; ++r) h2T[c][r] = fmaf(fmaxf(acc[r] + xT[c][r], 0.f), scale, shift);
        }
        __syncthreads();
        if (tid < d) {  // fc3 -> the node's vector
            const int c = tid;
            float acc[CF_R];
            const float b = m.b3[c];
#pragma unroll
            for (int r = 0; r < CF_R; ++r) acc[r] = b;
            cf_dot(h2T, d2, m.w3, d, c, acc);
#pragma unroll
            for (int r = 0; r < CF_R; ++r)
                if (r < nr) vec_all[out_rows[rb + r] * d + c] = acc[r];
        }
    } else if (tid < d) {  // fc2(x) + eh
        const int c = tid;
        float acc[CF_R];
        const float b = m.b2[c];
#pragma unroll
        for (int r = 0; r < CF_R; ++r) acc[r] = b;
        cf_dot(h1T, CF_H, m.w2, d, c, acc);
#pragma unroll
        for (int r = 0; r < CF_R; ++r)
            if (r < nr) vec_all[out_rows[rb + r] * d + c] = acc[r] + xT[c][r];
    }
}

static unsigned grid_for(int64_t n) {
    int64_t g = (n + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace hw

using namespace hw;

extern "C" int64_t hw_cafe_modules_workspace_floats(int64_t max_rows, int32_t dim) {
    if (max_rows < 0 || dim < 1) return 0;
    // feature [n, 2d], h1 / h1n [n, 256] x 2, h2 / h2n [n, 2d] x 2, y [n, d]  (rows padded to 32 floats for the GEMM)
    const int64_t d2 = ((int64_t)2 * dim + 31) / 32 * 32, d1 = ((int64_t)dim + 31) / 32 * 32;
    return max_rows * (3 * d2 + 2 * 256 + d1);
}

extern "C" int hw_cafe_modules_forward(const hw_cafe_module* mods, const int32_t* group_mod, const int64_t* group_ptr,
                                       int32_t n_groups, const float* user_vec, int64_t n_user_rows, int32_t dim,
                                       const int64_t* eh_idx, const int64_t* uv_idx, const int64_t* out_rows, float* vec_all,
                                       float* workspace, int64_t workspace_floats, void* stream) {
    if (!mods || !group_mod || !group_ptr || n_groups < 0 || !user_vec || !eh_idx || !uv_idx || !out_rows || !vec_all ||
        !workspace || dim < 1 || (dim & 3))
        return HW_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int d = dim;
    const int64_t d2 = ((int64_t)2 * d + 31) / 32 * 32, d1 = ((int64_t)d + 31) / 32 * 32;
    int64_t max_rows = 0;
    for (int g = 0; g < n_groups; ++g) {
        const int64_t n = group_ptr[g + 1] - group_ptr[g];
        if (n < 0) return HW_ERR_INVALID_ARG;
        if (n > max_rows) max_rows = n;
    }
    if (workspace_floats < hw_cafe_modules_workspace_floats(max_rows, dim)) return HW_ERR_WORKSPACE;
    float* feature = workspace;
    float* h1 = feature + max_rows * d2;
    float* h1n = h1 + max_rows * 256;
    float* h2 = h1n + max_rows * 256;
    float* h2n = h2 + max_rows * d2;
    float* y = h2n + max_rows * d2;
    for (int g = 0; g < n_groups; ++g) {  // in order: a level's groups read node vectors the levels before it wrote
        const int64_t r0 = group_ptr[g], n = group_ptr[g + 1] - r0;
        if (n == 0) continue;
        const hw_cafe_module& m = mods[group_mod[g]];
        if (!m.w1 || !m.w2 || !m.bn1_mean || !m.bn1_var || (m.deep && (!m.w3 || !m.bn2_mean || !m.bn2_var)))
            return HW_ERR_INVALID_ARG;
        HW_CLEAR_ERR();
        hipLaunchKernelGGL(cafe_gather_cat_kernel, dim3(grid_for(n * 2 * d)), dim3(256), 0, st, user_vec, vec_all, n_user_rows, d,
                           eh_idx + r0, uv_idx + r0, n, feature);
        int rc = HW_LAUNCH_RC();
        if (rc) return rc;
        // fc1 + ReLU (dropout is the identity in evaluation), bn1
        rc = hw_gemm_f32(feature, 2 * d, 1, m.w1, 2 * d, 1, h1, 256, nullptr, n, 256, 2 * d, m.b1, 2, nullptr, 0, 0, 0, nullptr, 0,
                         stream);
        if (rc) return rc;
        rc = hw_batchnorm1d_forward(h1, n, 256, m.bn1_weight, m.bn1_bias, m.bn1_eps, 0.f, const_cast<float*>(m.bn1_mean),
                                    const_cast<float*>(m.bn1_var), nullptr, nullptr, h1n, nullptr, 0, stream);
        if (rc) return rc;
        if (m.deep) {
            // relu(fc2(x) + feature), bn2, fc3
            rc = hw_gemm_f32(h1n, 256, 1, m.w2, 256, 1, h2, 2 * d, nullptr, n, 2 * d, 256, m.b2, 0, nullptr, 0, 0, 0, nullptr, 0,
                             stream);
            if (rc) return rc;
            HW_CLEAR_ERR();
            hipLaunchKernelGGL(cafe_add_relu_kernel, dim3(grid_for(n * 2 * d)), dim3(256), 0, st, h2, feature, n * 2 * d, h2);
            rc = HW_LAUNCH_RC();
            if (rc) return rc;
            rc = hw_batchnorm1d_forward(h2, n, 2 * d, m.bn2_weight, m.bn2_bias, m.bn2_eps, 0.f, const_cast<float*>(m.bn2_mean),
                                        const_cast<float*>(m.bn2_var), nullptr, nullptr, h2n, nullptr, 0, stream);
            if (rc) return rc;
            rc = hw_gemm_f32(h2n, 2 * d, 1, m.w3, 2 * d, 1, y, d, nullptr, n, d, 2 * d, m.b3, 0, nullptr, 0, 0, 0, nullptr, 0,
                             stream);
            if (rc) return rc;
        } else {
            rc = hw_gemm_f32(h1n, 256, 1, m.w2, 256, 1, y, d, nullptr, n, d, 256, m.b2, 0, nullptr, 0, 0, 0, nullptr, 0, stream);
            if (rc) return rc;
        }
        HW_CLEAR_ERR();
        hipLaunchKernelGGL(cafe_scatter_rows_kernel, dim3(grid_for(n * d)), dim3(256), 0, st, y, m.deep ? nullptr : feature, d,
                           out_rows + r0, n, vec_all);
        rc = HW_LAUNCH_RC();
        if (rc) return rc;
    }
    (void)d1;
    return HW_OK;
}

extern "C" int hw_cafe_modules_forward_fused(const hw_cafe_module* mods, const int32_t* group_mod, const int64_t* group_ptr,
                                             const int32_t* group_level, int32_t n_groups, const float* user_vec,
                                             int64_t n_user_rows, int32_t dim, const int64_t* eh_idx, const int64_t* uv_idx,
                                             const int64_t* out_rows, float* vec_all, void* stream) {
    if (!mods || !group_mod || !group_ptr || !group_level || n_groups < 0 || !user_vec || !eh_idx || !uv_idx || !out_rows ||
        !vec_all || dim < 1)
        return HW_ERR_INVALID_ARG;
    if (2 * dim > CF_H) return HW_ERR_UNSUPPORTED;  // one output column per thread: the caller takes the launch-per-layer path
    hipStream_t st = (hipStream_t)stream;
    int g = 0;
    while (g < n_groups) {
        CfLevel L{};
        int blocks = 0;
        const int level = group_level[g];
        while (g < n_groups && group_level[g] == level && L.n_groups < CF_MAXG) {
            const int64_t n = group_ptr[g + 1] - group_ptr[g];
            if (n < 0) return HW_ERR_INVALID_ARG;
            const hw_cafe_module& m = mods[group_mod[g]];
            if (!m.w1 || !m.b1 || !m.w2 || !m.b2 || !m.bn1_mean || !m.bn1_var ||
                (m.deep && (!m.w3 || !m.b3 || !m.bn2_mean || !m.bn2_var)))
                return HW_ERR_INVALID_ARG;
            if (n > 0) {
                if (blocks + (n + CF_R - 1) / CF_R > INT32_MAX) return HW_ERR_INVALID_ARG;
                const int q = L.n_groups++;
                L.blk0[q] = blocks, L.r0[q] = group_ptr[g], L.n[q] = n, L.mod[q] = m;
                blocks += (int)((n + CF_R - 1) / CF_R);
                L.blk0[q + 1] = blocks;
            }
            ++g;
        }
        if (blocks == 0) continue;
        HW_CLEAR_ERR();
        hipLaunchKernelGGL(cafe_level_kernel, dim3((unsigned)blocks), dim3(256), 0, st, L, user_vec, n_user_rows, (int)dim, eh_idx,
                           uv_idx, out_rows, vec_all);
        const int rc = HW_LAUNCH_RC();
        if (rc) return rc;
    }
    return HW_OK;
}
