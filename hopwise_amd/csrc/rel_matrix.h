// Kernels over per-relation d x d matrices, shared by TransR (proj_mat_e, transr.py:50-51,67-70) and RESCAL's
// backward (relation_embedding viewed [d, d], rescal.py:43,51-54).
//   project_rows         out[i, c] = sum_k M[r_i][c][k] E[e_i][k]        wave per row, 4 matrix rows per reduction;
//                        SCATTER: the result row is added into out[o_i] instead (RESCAL's head gradient)
//   outer_by_relation    out[r][a][b] += sum_{i: r_i = r} A_i[a] B_i[b]  block per (relation, 64 x 64 tile, row split):
//                        the rows of the split that carry relation r are compacted in order, staged 16 at a time
//                        through LDS and accumulated in 4 x 4 register tiles; one atomic flush per block, so a
//                        relation sees (row splits)-way contention instead of one atomic per (row, a, b).
#pragma once
#include "hw_common.h"

namespace hw {

template <int NE, bool SCATTER>
__global__ __launch_bounds__(256) void project_rows_kernel(const float* __restrict__ mats, int64_t n_rel,
                                                           const float* __restrict__ table, int64_t n_rows_table,
                                                           int dim, const int64_t* __restrict__ r_idx, int64_t r_fixed,
                                                           const int64_t* __restrict__ e_idx, int64_t n,
                                                           float* __restrict__ out, const int64_t* __restrict__ o_idx,
                                                           int64_t n_rows_out, int idx_div) {
    using R = Row<64, NE, false>;
    const int gl = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int64_t ii = SCATTER ? i / idx_div : i;  // index arrays are per group, rows per (group, split) task
    const int r = clamp_row(r_idx ? r_idx[ii] : r_fixed, n_rel, nullptr);
    const int64_t e = e_idx ? (int64_t)clamp_row(e_idx[i], n_rows_table, nullptr) : i;
    R ev;
    ev.load(table + (size_t)e * dim, dim, gl);
    const float* __restrict__ M = mats + (size_t)r * dim * dim;
    float* __restrict__ orow = out + (size_t)(SCATTER ? (int64_t)clamp_row(o_idx[ii], n_rows_out, nullptr) : i) * dim;
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
        if ((gl & 15) == 0 && c0 + which < dim) {
            if constexpr (SCATTER) {
                if (tot != 0.f) atomicAdd(orow + c0 + which, tot);
            } else {
                orow[c0 + which] = tot;
            }
        }
    }
}

constexpr int RO_T = 64;   // tile edge: 256 threads x (4 x 4) accumulators
constexpr int RO_RB = 32;  // rows staged per batch (two per thread: independent load chains)

// Row i of one side: base + (idx ? clamp(idx[i / idx_div]) : i) * ld, `dim` floats.
struct OuterSide {
    const float* base;
    int64_t ld;
    const int64_t* idx;
    int64_t n_rows;
    int dim;
};

static __global__ __launch_bounds__(256) void outer_by_relation_kernel(OuterSide A, OuterSide B,
                                                                       const int64_t* __restrict__ rel_idx,
                                                                       int64_t n_rel, int64_t r_only, int idx_div,
                                                                       int64_t n, int n_splits, int tiles_b,
                                                                       float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float As[RO_RB][RO_T];
    __shared__ __attribute__((aligned(16))) float Bs[RO_RB][RO_T];
    __shared__ int row_a[256], row_b[256];  // source rows of the matched rows of the current 256-row chunk
    __shared__ int wave_cnt[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = rel_idx ? (int)blockIdx.x : (int)r_only;
    const int a0 = ((int)blockIdx.y / tiles_b) * RO_T, b0 = ((int)blockIdx.y % tiles_b) * RO_T;
    const int ty = tid >> 4, tx = tid & 15;
    const int c4 = tx * 4;
    const int64_t per = (n + n_splits - 1) / n_splits;
    const int64_t lo = (int64_t)blockIdx.z * per, hi = min(n, lo + per);
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    bool any = false;

    float av[2][4], bv[2][4];
    // this thread's share of one batch: rows (tid / 16) and (tid / 16) + 16, four columns of each side.  Branch-free
    // (clamped addresses, values selected afterwards) so that all 16 loads are in flight together.
    auto fetch = [&](int q0, int nm) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int rr = ty + 16 * h;
            const bool live = q0 + rr < nm;
            const float* __restrict__ pa = A.base + (int64_t)row_a[live ? q0 + rr : 0] * A.ld;
            const float* __restrict__ pb = B.base + (int64_t)row_b[live ? q0 + rr : 0] * B.ld;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ca = a0 + c4 + u, cb = b0 + c4 + u;
                av[h][u] = pa[ca < A.dim ? ca : 0];  // masked when staged: a select here would wait for the load
                bv[h][u] = pb[cb < B.dim ? cb : 0];
            }
        }
    };
    auto stage = [&](int q0, int nm) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const bool live = q0 + ty + 16 * h < nm;
            f32x4 xa, xb;
            xa.x = live && a0 + c4 + 0 < A.dim ? av[h][0] : 0.f;
            xa.y = live && a0 + c4 + 1 < A.dim ? av[h][1] : 0.f;
            xa.z = live && a0 + c4 + 2 < A.dim ? av[h][2] : 0.f;
            xa.w = live && a0 + c4 + 3 < A.dim ? av[h][3] : 0.f;
            xb.x = live && b0 + c4 + 0 < B.dim ? bv[h][0] : 0.f;
            xb.y = live && b0 + c4 + 1 < B.dim ? bv[h][1] : 0.f;
            xb.z = live && b0 + c4 + 2 < B.dim ? bv[h][2] : 0.f;
            xb.w = live && b0 + c4 + 3 < B.dim ? bv[h][3] : 0.f;
            *reinterpret_cast<f32x4*>(&As[ty + 16 * h][c4]) = xa;
            *reinterpret_cast<f32x4*>(&Bs[ty + 16 * h][c4]) = xb;
        }
    };

    for (int64_t base = lo; base < hi; base += 256) {
        const int64_t i = base + tid;
        const bool m = i < hi && (!rel_idx || clamp_row(rel_idx[i / idx_div], n_rel, nullptr) == r);
        const unsigned long long mask = __ballot(m);
        if (lane == 0) wave_cnt[w] = __popcll(mask);
        __syncthreads();
        int off = 0, nm = 0;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            off += v < w ? wave_cnt[v] : 0;
            nm += wave_cnt[v];
        }
        if (m) {  // row order kept
            const int slot = off + __popcll(mask & ((1ull << lane) - 1ull));
            row_a[slot] = A.idx ? clamp_row(A.idx[i / idx_div], A.n_rows, nullptr) : (int)i;
            row_b[slot] = B.idx ? clamp_row(B.idx[i / idx_div], B.n_rows, nullptr) : (int)i;
        }
        __syncthreads();
        if (nm == 0) continue;  // block-uniform
        fetch(0, nm);
        for (int q0 = 0; q0 < nm; q0 += RO_RB) {
            stage(q0, nm);
            __syncthreads();
            if (q0 + RO_RB < nm) fetch(q0 + RO_RB, nm);  // next batch's loads fly during this batch's FMAs
#pragma unroll
            for (int q = 0; q < RO_RB; ++q) {  // rows past nm were staged as zeros
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(&As[q][ty * 4]);
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(&Bs[q][tx * 4]);
                const float aa[4] = {a4.x, a4.y, a4.z, a4.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i2][j] = fmaf(aa[i2], bb[j], acc[i2][j]);
            }
            any = true;
            __syncthreads();
        }
    }
    if (!any) return;
    // flush, each row split starting at a different one of its 16 values so that the splits of one relation are not
    // all queued on the same cache lines at the same moment
    float* __restrict__ dst = out + (size_t)r * A.dim * B.dim;
#pragma unroll
    for (int e0 = 0; e0 < 16; ++e0) {
        const int e = (e0 + (int)blockIdx.z) & 15;
        const int i2 = e >> 2, j = e & 3;
        float v = 0.f;
#pragma unroll
        for (int x = 0; x < 16; ++x) v = x == e ? acc[x >> 2][x & 3] : v;  // registers cannot be indexed dynamically
        const int a = a0 + ty * 4 + i2, b = b0 + tx * 4 + j;
        if (a < A.dim && b < B.dim && v != 0.f) atomicAdd(dst + (size_t)a * B.dim + b, v);
    }
}

// out[r] += sum over the rows of relation r of A_i (x) B_i; rel_idx == nullptr: every row belongs to r_fixed.
static inline void launch_outer_by_relation(const OuterSide& A, const OuterSide& B, const int64_t* rel_idx,
                                            int64_t n_rel, int64_t r_fixed, int idx_div, int64_t n, float* out,
                                            hipStream_t st) {
    const int64_t rf = r_fixed < 0 ? 0 : (r_fixed >= n_rel ? n_rel - 1 : r_fixed);
    const unsigned n_r = rel_idx ? (unsigned)n_rel : 1u;
    const int tiles_a = (A.dim + RO_T - 1) / RO_T, tiles_b = (B.dim + RO_T - 1) / RO_T;
    const unsigned tiles = (unsigned)(tiles_a * tiles_b);
    // ~128 blocks of at least 64 rows: a 64 x 64 tile costs 16 wave-FMAs per row on one wave per SIMD, so long row
    // ranges are compute latency, while every extra split adds a 4096-atomic flush onto the same matrix (measured
    // ~45 G atomics/s here: d = 64, one relation, 4096 rows: 64 splits 12 us, 16 splits 18 us, 128 splits 19 us)
    int64_t splits = (128 + (int64_t)n_r * tiles - 1) / ((int64_t)n_r * tiles);
    const int64_t by_rows = (n + 63) / 64, max_splits = by_rows < 64 ? by_rows : 64;
    splits = splits < 1 ? 1 : (splits > max_splits ? max_splits : splits);
    hipLaunchKernelGGL(outer_by_relation_kernel, dim3(n_r, tiles, (unsigned)splits), dim3(256), 0, st, A, B, rel_idx,
                       n_rel, rf, idx_div, n, (int)splits, tiles_b, out);
}

}  // namespace hw
