// Backward of the grouped scorer: row-gradient scatter into dense grad tables with f32 atomics.
//
// The kernel recomputes the forward pieces from the gathered rows (cheaper than storing [B, G, d]
// intermediates: it re-reads rows that are L2/MALL resident) and emits
//   * one atomic row per tail table per (b, j)                       -- unavoidable traffic,
//   * one atomic row per head / relation table per WAVE TASK          -- the K+1 contributions of a
//     group are summed in registers first (Guideline 12: sum on chip what shares a destination).
// Rows use the dword-per-lane mapping (lane l owns columns l, l+64, ...) so that every atomic
// wave-instruction covers 256 contiguous bytes of one row, the shape global_atomic_add_f32 runs at
// full rate for (MI355X_MICROARCH.md 'Global float atomics').
//
// Gradient formulas (g = dL/dscore of one (b, j); u = residual / norm, 0 where norm == 0, which is
// torch's sub-gradient for vector_norm / norm at 0):
//   TransE   d = (Q - t) + eps, n = |d|          gQ = -g d/n      gt = +g d/n      gh = gr = gQ
//   DistMult                                      gQ =  g t        gt =  g Q        gh = gQ r, gr = gQ h
//   RotatE   a = Q0 - t_re, b = Q1 - t_im        gQ0 = -g a/n, gQ1 = -g b/n, gt_re = g a/n, gt_im = g b/n
//            gh_re = gQ0 c + gQ1 s, gh_im = -gQ0 s + gQ1 c, gtheta = gQ0 (-s h_re - c h_im) + gQ1 (-s h_im + c h_re)
//   ComplEx  gQ0 = g t_re, gQ1 = g t_im, gt_re = g Q0, gt_im = g Q1
//            gh_re = gQ0 r_re + gQ1 r_im, gh_im = gQ1 (r_re - r_im), gr_re = gQ0 h_re + gQ1 h_im, gr_im = gQ1 (h_re - h_im)
//   TransH   d = (Q0 - t c) + eps, u = g d/n      gQ0 = -u, gt = u c, gc = u t (+ gQ0 h at the end)
//            gh = gQ0 c, gr = gQ0, c = 1 - s w (s = sum w):  gw_k = -s gc_k - sum_i gc_i w_i
//   TransD   d = (Q0 - t - rv sigma) + eps, sigma = t.tv, u = g d/n, beta = sum_i u_i rv_i
//            gQ0 = -u, gt = u + beta tv, gtv = beta t, g(rv) += u sigma;  rho = sum_i gQ0_i rv_i
//            gh = gQ0 + rho hv, ghv = rho h, gr = gQ0, grv = gQ0 (h.hv) + g(rv)
//   Analogy  gQk = g t_k, gt_k = g Q_k;  gh = gQ0 r, gr = gQ0 h, gh_re = gQ1 r_re + gQ2 r_im, gh_im = gQ2 r_re - gQ1 r_im,
//            gr_re = gQ1 h_re + gQ2 h_im, gr_im = gQ2 h_re - gQ1 h_im
//   HolE     gQ = g t, gt = g Q;  gh_i = sum_j gQ_j r[(j - i) mod d], gr_m = sum_j gQ_j h[(j - m) mod d]
//   TorusE   not differentiated by the reference (its calculate_loss is TransE's) -> HW_ERR_UNSUPPORTED
#include "kge_model.h"
#include "rel_matrix.h"

namespace hw {

constexpr int REL_REPLICAS = 64;

// grad_rel[k][r_fixed, :] += sum over replicas (fixed order -> the replica stage adds no run-to-run noise of its own)
__global__ void rel_replica_reduce_kernel(const float* __restrict__ rep, int n_tables, int64_t row_len,
                                          float* __restrict__ g0, float* __restrict__ g1, float* __restrict__ g2,
                                          int64_t r_fixed) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= row_len * n_tables) return;
    const int k = (int)(i / row_len);
    const int64_t c = i % row_len;
    float s = 0.f;
    for (int q = 0; q < REL_REPLICAS; ++q) s += rep[((size_t)q * n_tables + k) * row_len + c];
    float* gdst = k == 0 ? g0 : (k == 1 ? g1 : g2);
    // an atomic like every other access to a gradient table between its clear and its consumer (the kg-side launch of
    // the same step adds to other rows of the same lines): one access class per buffer, no plain read-modify-write of a
    // line that memory-side atomics have just updated
    if (gdst) atomicAdd(gdst + (size_t)r_fixed * row_len + c, s);
}

template <int MODEL, int NE>
__global__ __launch_bounds__(256) void kge_grad_grouped_kernel(
    hw_kge_desc d, hw_kge_grads gr, const int64_t* __restrict__ h_idx, const int64_t* __restrict__ r_idx,
    int64_t r_fixed, const int64_t* __restrict__ t_idx, int64_t tsb, int64_t tsj, int64_t n_groups, int ntail, int S,
    const float* __restrict__ g, int64_t gsb, int64_t gsj, float* __restrict__ rel_replicas,
    float* __restrict__ gq_out) {
    constexpr int G = 64;
    using SC = Scorer<MODEL, G, NE, false>;
    using MT = ModelTraits<MODEL>;
    using R = Row<G, NE, false>;

    const int gl = threadIdx.x & (HW_WAVE - 1);
    const int64_t task = (int64_t)blockIdx.x * (blockDim.x / HW_WAVE) + wave_in_block();
    if (task >= n_groups * S) return;
    const int64_t b = task / S;
    const int s = (int)(task % S);

    const int h = clamp_row(h_idx[b], d.n_head, nullptr);
    const int r = clamp_row(r_idx ? r_idx[b] : r_fixed, d.n_rel, nullptr);
    SC sc;
    R hraw[HW_KGE_MAX_TABLES], rraw[HW_KGE_MAX_TABLES];
    sc.template build<true>(d, h, r, gl, hraw, rraw);
    // Fixed relation (rec side: every group uses the [UI-Relation] row): thousands of waves adding into ONE row run
    // an order of magnitude below the atomic rate (MI355X_MICROARCH.md 'Global float atomics', contention row), so the
    // relation gradient goes to one of REL_REPLICAS scratch rows picked by task id and is summed afterwards.
    const size_t rel_row_len = MODEL == HW_RESCAL ? (size_t)d.dim * d.dim : (size_t)d.dim;
    if (rel_replicas) {
        const size_t rep = (size_t)(task % REL_REPLICAS) * rel_row_len * MT::NR;
#pragma unroll
        for (int k = 0; k < MT::NR; ++k)  // so that "+ r * row_len" lands in the replica
            if (gr.rel[k]) gr.rel[k] = rel_replicas + rep + k * rel_row_len - (size_t)r * rel_row_len;
    }

    float GQ[MT::NQ][NE];
#pragma unroll
    for (int k = 0; k < MT::NQ; ++k)
#pragma unroll
        for (int i = 0; i < NE; ++i) GQ[k][i] = 0.f;

    const int64_t* __restrict__ tb = t_idx + b * tsb;
    const float* __restrict__ gb = g + b * gsb;

    for (int j = s; j < ntail; j += S) {
        const float gs = gb[(int64_t)j * gsj];
        if (gs == 0.f) continue;  // wave-uniform (one address per wave): inactive margin pairs add nothing
        const int tj = clamp_row(tb[(int64_t)j * tsj], d.n_tail, nullptr);
        R t[MT::NT];
#pragma unroll
        for (int k = 0; k < MT::NT; ++k) t[k].load(d.tail[k] + (size_t)tj * d.dim, d.dim, gl);
        float scale = gs;
        float sig = 0.f;
        if constexpr (MT::TWO_PASS) sig = group_sum<G>(sc.pre(t, d.dim, gl));
        if constexpr (MODEL == HW_TRANSE || MODEL == HW_ROTATE || MODEL == HW_TRANSH || MODEL == HW_TRANSD) {
            float p = group_sum<G>(sc.partial(t, d.eps, d.dim, gl, sig));
            const float n = sqrtf(p);
            scale = n > 0.f ? gs / n : 0.f;
        }
        const size_t to = (size_t)tj * d.dim;
        float beta = 0.f;
        if constexpr (MODEL == HW_TRANSD) {
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                if (R::col(i, d.dim, gl) < 0) continue;
                const float u = (((sc.q[0].v[i] - t[0].v[i]) - sc.q[1].v[i] * sig) + d.eps) * scale;
                beta = fmaf(u, sc.q[1].v[i], beta);
            }
            beta = group_sum<G>(beta);
        }
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int c = R::col(i, d.dim, gl);
            if (c < 0) continue;
            if constexpr (MODEL == HW_TRANSH) {
                const float u = ((sc.q[0].v[i] - t[0].v[i] * sc.q[1].v[i]) + d.eps) * scale;
                GQ[0][i] -= u;
                GQ[1][i] = fmaf(u, t[0].v[i], GQ[1][i]);
                if (gr.tail[0]) atomicAdd(gr.tail[0] + to + c, u * sc.q[1].v[i]);
            } else if constexpr (MODEL == HW_TRANSD) {
                const float u = (((sc.q[0].v[i] - t[0].v[i]) - sc.q[1].v[i] * sig) + d.eps) * scale;
                GQ[0][i] -= u;
                GQ[1][i] = fmaf(u, sig, GQ[1][i]);
                if (gr.tail[0]) atomicAdd(gr.tail[0] + to + c, fmaf(beta, t[1].v[i], u));
                if (gr.tail[1]) atomicAdd(gr.tail[1] + to + c, beta * t[0].v[i]);
            } else if constexpr (MODEL == HW_ANALOGY) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    GQ[k][i] = fmaf(scale, t[k].v[i], GQ[k][i]);
                    if (gr.tail[k]) atomicAdd(gr.tail[k] + to + c, scale * sc.q[k].v[i]);
                }
            } else if constexpr (MODEL == HW_TRANSE) {
                const float u = ((sc.q[0].v[i] - t[0].v[i]) + d.eps) * scale;
                GQ[0][i] -= u;
                if (gr.tail[0]) atomicAdd(gr.tail[0] + to + c, u);
            } else if constexpr (MODEL == HW_DISTMULT || MODEL == HW_RESCAL || MODEL == HW_HOLE) {
                GQ[0][i] = fmaf(scale, t[0].v[i], GQ[0][i]);
                if (gr.tail[0]) atomicAdd(gr.tail[0] + to + c, scale * sc.q[0].v[i]);
            } else if constexpr (MODEL == HW_ROTATE) {
                const float ua = (sc.q[0].v[i] - t[0].v[i]) * scale;
                const float ub = (sc.q[1].v[i] - t[1].v[i]) * scale;
                GQ[0][i] -= ua;
                GQ[1][i] -= ub;
                if (gr.tail[0]) atomicAdd(gr.tail[0] + to + c, ua);
                if (gr.tail[1]) atomicAdd(gr.tail[1] + to + c, ub);
            } else {
                GQ[0][i] = fmaf(scale, t[0].v[i], GQ[0][i]);
                GQ[1][i] = fmaf(scale, t[1].v[i], GQ[1][i]);
                if (gr.tail[0]) atomicAdd(gr.tail[0] + to + c, scale * sc.q[0].v[i]);
                if (gr.tail[1]) atomicAdd(gr.tail[1] + to + c, scale * sc.q[1].v[i]);
            }
        }
    }

    const size_t ho = (size_t)h * d.dim, ro = (size_t)r * d.dim;
    if constexpr (MODEL == HW_RESCAL) {
        if (gq_out) {  // dL/dQ of this task goes to scratch; the two d^2 products follow as their own launches
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                const int c = R::col(k, d.dim, gl);
                if (c >= 0) gq_out[(size_t)task * d.dim + c] = GQ[0][k];
            }
            return;
        }
        // Q = h^T M:  gh[i] = sum_j GQ[j] M[i][j]  (one wave reduction per matrix row),  gM[i][j] += h[i] GQ[j]
        const float* __restrict__ hrow = d.head[0] + ho;
        const float* __restrict__ mat = d.rel[0] + (size_t)r * d.dim * d.dim;
        float* gmat = gr.rel[0] ? gr.rel[0] + (size_t)r * d.dim * d.dim : nullptr;
        float gh[NE];
#pragma unroll
        for (int k = 0; k < NE; ++k) gh[k] = 0.f;
        for (int i = 0; i < d.dim; ++i) {
            R m;
            m.load(mat + (size_t)i * d.dim, d.dim, gl);
            float p = 0.f;
#pragma unroll
            for (int k = 0; k < NE; ++k) p = fmaf(GQ[0][k], m.v[k], p);
            p = group_sum<G>(p);
            const float hi = hrow[i];
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                const int c = R::col(k, d.dim, gl);
                if (c == i) gh[k] = p;
                if (gmat && c >= 0 && GQ[0][k] != 0.f) atomicAdd(gmat + (size_t)i * d.dim + c, hi * GQ[0][k]);
            }
        }
        if (gr.head[0]) {
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                const int c = R::col(k, d.dim, gl);
                if (c >= 0 && gh[k] != 0.f) atomicAdd(gr.head[0] + ho + c, gh[k]);
            }
        }
        return;
    }
    if constexpr (MODEL == HW_HOLE) {
        // gh_i = sum_j gQ_j r[(j - i) mod d], gr_m = sum_j gQ_j h[(j - m) mod d]: two circular correlations with dL/dQ.
        // The wave stages gQ and the (doubled, so that no index needs a modulo) relation and head rows in LDS; every
        // lane then walks j once with one broadcast read of gQ_j and conflict-free reads of the rolled rows -- 2 d^2 LDS
        // reads per task instead of 2 d^2 dword gathers through L1 (a HolE step was 3x the other models').
        constexpr int CAP = 64 * NE;
        __shared__ float hole_lds[4][5 * CAP];
        float* gq = hole_lds[threadIdx.x >> 6];
        float* r2 = gq + CAP;
        float* h2 = r2 + 2 * CAP;
        bool any = false;
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const int c = R::col(k, d.dim, gl);
            if (c >= 0) {
                gq[c] = GQ[0][k];
                r2[c] = r2[c + d.dim] = rraw[0].v[k];
                h2[c] = h2[c + d.dim] = hraw[0].v[k];
                any |= GQ[0][k] != 0.f;
            }
        }
        if (!__any(any)) return;  // every tail of the task had zero gradient (inactive margin pairs)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float gh[NE], gre[NE];
        int base[NE];
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            gh[k] = gre[k] = 0.f;
            const int c = R::col(k, d.dim, gl);
            base[k] = d.dim - (c < 0 ? 0 : c);  // j - c + dim
        }
#pragma unroll 4
        for (int j = 0; j < d.dim; ++j) {
            const float gqj = gq[j];
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                gh[k] = fmaf(gqj, r2[base[k] + j], gh[k]);
                gre[k] = fmaf(gqj, h2[base[k] + j], gre[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const int c = R::col(k, d.dim, gl);
            if (c < 0) continue;
            if (gr.head[0]) atomicAdd(gr.head[0] + ho + c, gh[k]);
            if (gr.rel[0]) atomicAdd(gr.rel[0] + ro + c, gre[k]);
        }
        return;
    }
    if constexpr (MODEL == HW_TRANSH || MODEL == HW_TRANSD) {
        float r0 = 0.f, r1 = 0.f;
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            if (R::col(i, d.dim, gl) < 0) continue;
            if constexpr (MODEL == HW_TRANSH) {
                const float gc = fmaf(GQ[0][i], hraw[0].v[i], GQ[1][i]);
                r0 += rraw[1].v[i];                    // s = sum w
                r1 = fmaf(gc, rraw[1].v[i], r1);       // sum gc w
            } else {
                r0 = fmaf(hraw[0].v[i], hraw[1].v[i], r0);  // h . hv
                r1 = fmaf(GQ[0][i], rraw[1].v[i], r1);      // rho = sum gQ0 rv
            }
        }
        r0 = group_sum<G>(r0);
        r1 = group_sum<G>(r1);
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int c = R::col(i, d.dim, gl);
            if (c < 0) continue;
            if constexpr (MODEL == HW_TRANSH) {
                const float gc = fmaf(GQ[0][i], hraw[0].v[i], GQ[1][i]);
                if (gr.head[0]) atomicAdd(gr.head[0] + ho + c, GQ[0][i] * sc.q[1].v[i]);
                if (gr.rel[0]) atomicAdd(gr.rel[0] + ro + c, GQ[0][i]);
                if (gr.rel[1]) atomicAdd(gr.rel[1] + ro + c, -r0 * gc - r1);
            } else {
                if (gr.head[0]) atomicAdd(gr.head[0] + ho + c, fmaf(r1, hraw[1].v[i], GQ[0][i]));
                if (gr.head[1]) atomicAdd(gr.head[1] + ho + c, r1 * hraw[0].v[i]);
                if (gr.rel[0]) atomicAdd(gr.rel[0] + ro + c, GQ[0][i]);
                if (gr.rel[1]) atomicAdd(gr.rel[1] + ro + c, fmaf(GQ[0][i], r0, GQ[1][i]));
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int c = R::col(i, d.dim, gl);
        if (c < 0) continue;
        if constexpr (MODEL == HW_ANALOGY) {
            const float g0 = GQ[0][i], g1 = GQ[1][i], g2 = GQ[2][i];
            if (gr.head[0]) atomicAdd(gr.head[0] + ho + c, g0 * rraw[0].v[i]);
            if (gr.rel[0]) atomicAdd(gr.rel[0] + ro + c, g0 * hraw[0].v[i]);
            if (gr.head[1]) atomicAdd(gr.head[1] + ho + c, g1 * rraw[1].v[i] + g2 * rraw[2].v[i]);
            if (gr.head[2]) atomicAdd(gr.head[2] + ho + c, g2 * rraw[1].v[i] - g1 * rraw[2].v[i]);
            if (gr.rel[1]) atomicAdd(gr.rel[1] + ro + c, g1 * hraw[1].v[i] + g2 * hraw[2].v[i]);
            if (gr.rel[2]) atomicAdd(gr.rel[2] + ro + c, g2 * hraw[1].v[i] - g1 * hraw[2].v[i]);
        } else if constexpr (MODEL == HW_TRANSE) {
            if (GQ[0][i] != 0.f) {
                if (gr.head[0]) atomicAdd(gr.head[0] + ho + c, GQ[0][i]);
                if (gr.rel[0]) atomicAdd(gr.rel[0] + ro + c, GQ[0][i]);
            }
        } else if constexpr (MODEL == HW_DISTMULT) {
            if (gr.head[0]) atomicAdd(gr.head[0] + ho + c, GQ[0][i] * rraw[0].v[i]);
            if (gr.rel[0]) atomicAdd(gr.rel[0] + ro + c, GQ[0][i] * hraw[0].v[i]);
        } else if constexpr (MODEL == HW_ROTATE) {
            const float cs = rraw[0].v[i], sn = rraw[1].v[i];
            const float hr = hraw[0].v[i], hi = hraw[1].v[i];
            const float g0 = GQ[0][i], g1 = GQ[1][i];
            if (gr.head[0]) atomicAdd(gr.head[0] + ho + c, g0 * cs + g1 * sn);
            if (gr.head[1]) atomicAdd(gr.head[1] + ho + c, g1 * cs - g0 * sn);
            if (gr.rel[0]) atomicAdd(gr.rel[0] + ro + c, g0 * (-sn * hr - cs * hi) + g1 * (cs * hr - sn * hi));
        } else {
            const float rr = rraw[0].v[i], ri = rraw[1].v[i];
            const float hr = hraw[0].v[i], hi = hraw[1].v[i];
            const float g0 = GQ[0][i], g1 = GQ[1][i];
            if (gr.head[0]) atomicAdd(gr.head[0] + ho + c, g0 * rr + g1 * ri);
            if (gr.head[1]) atomicAdd(gr.head[1] + ho + c, g1 * (rr - ri));
            if (gr.rel[0]) atomicAdd(gr.rel[0] + ro + c, g0 * hr + g1 * hi);
            if (gr.rel[1]) atomicAdd(gr.rel[1] + ro + c, g1 * (hr - hi));
        }
    }
}

}  // namespace hw

using namespace hw;

// splits: the head/relation rows are flushed once per wave task, so prefer few, long tasks; split
// only as far as needed to put ~8k waves on the chip.
static int grad_splits(int64_t n_groups, int ntail) {
    const int s_max = (ntail + 7) / 8;
    const int64_t want = (8192 + n_groups - 1) / n_groups;
    const int S = (int)(want < 1 ? 1 : (want > s_max ? s_max : want));
    return S < 1 ? 1 : S;
}

static int n_rel_tables(int model) {
    return model == HW_ANALOGY ? 3 : (model == HW_COMPLEX || model == HW_TRANSH || model == HW_TRANSD) ? 2 : 1;
}

extern "C" int64_t hw_kge_grad_grouped_workspace_floats(const hw_kge_desc* desc, int64_t n_groups,
                                                        int64_t n_tail_per_group, int32_t relation_is_fixed) {
    if (!desc || desc->dim < 1 || n_groups < 0 || n_tail_per_group < 0 || n_tail_per_group > INT32_MAX / 2) return 0;
    if (desc->model == HW_RESCAL)  // dL/dQ rows, one per (group, split) task
        return n_groups * grad_splits(n_groups > 0 ? n_groups : 1, (int)n_tail_per_group) * desc->dim;
    return relation_is_fixed ? (int64_t)REL_REPLICAS * n_rel_tables(desc->model) * desc->dim : 0;
}

extern "C" int hw_kge_grad_grouped(const hw_kge_desc* desc, const hw_kge_grads* grads, const int64_t* h_idx,
                                   const int64_t* r_idx, int64_t r_fixed, const int64_t* t_idx, int64_t t_stride_b,
                                   int64_t t_stride_j, int64_t n_groups, int64_t n_tail_per_group, const float* g,
                                   int64_t g_stride_b, int64_t g_stride_j, float* rel_workspace, void* stream) {
    if (!desc || !grads || !h_idx || !t_idx || !g) return HW_ERR_INVALID_ARG;
    if (desc->dim < 1 || desc->dim > HW_MAX_DIM || n_groups < 0 || n_tail_per_group < 0) return HW_ERR_INVALID_ARG;
    if (desc->n_head < 1 || desc->n_rel < 1 || desc->n_tail < 1) return HW_ERR_INVALID_ARG;
    if (n_tail_per_group > INT32_MAX / 2) return HW_ERR_INVALID_ARG;
    if (n_groups == 0 || n_tail_per_group == 0) return HW_OK;
    const int ntail = (int)n_tail_per_group;
    const int S = grad_splits(n_groups, ntail);
    const int64_t n_tasks = n_groups * S;
    const int64_t n_blocks = (n_tasks + 3) / 4;
    if (n_blocks > INT32_MAX) return HW_ERR_INVALID_ARG;
    const int NE = desc->dim <= 64 ? 1 : (desc->dim <= 256 ? 4 : 16);
    HW_CLEAR_ERR();
    hipStream_t st = (hipStream_t)stream;
    // replica scratch only when the relation is fixed, has a gradient, and enough tasks collide on it
    const int n_rel_tab = n_rel_tables(desc->model);
    const int64_t rel_row_len = desc->dim;
    // RESCAL with scratch: the kernel only leaves dL/dQ per task; the head and matrix gradients are two small
    // matrix kernels over those rows (rel_matrix.h) instead of d^2 atomics per task
    float* gq = desc->model == HW_RESCAL ? rel_workspace : nullptr;
    float* replicas =
        (!gq && desc->model != HW_RESCAL && !r_idx && rel_workspace && grads->rel[0] && n_tasks >= 256) ? rel_workspace
                                                                                                          : nullptr;
    if (replicas && hw::fill_bytes_async(replicas, 0, sizeof(float) * REL_REPLICAS * n_rel_tab * rel_row_len, st) != hipSuccess)
        return HW_ERR_LAUNCH;
#define CALL(MODEL, NE_)                                                                                           \
    hipLaunchKernelGGL((kge_grad_grouped_kernel<MODEL, NE_>), dim3((unsigned)n_blocks), dim3(256), 0, st, *desc,     \
                       *grads, h_idx, r_idx, r_fixed, t_idx, t_stride_b, t_stride_j, n_groups, ntail, S, g,          \
                       g_stride_b, g_stride_j, replicas, gq)
#define BYNE(MODEL)                        \
    do {                                   \
        if (NE == 1) CALL(MODEL, 1);       \
        else if (NE == 4) CALL(MODEL, 4);  \
        else CALL(MODEL, 16);              \
    } while (0)
    switch (desc->model) {
        case HW_TRANSE: BYNE(HW_TRANSE); break;
        case HW_ROTATE: BYNE(HW_ROTATE); break;
        case HW_COMPLEX: BYNE(HW_COMPLEX); break;
        case HW_DISTMULT: BYNE(HW_DISTMULT); break;
        case HW_RESCAL: BYNE(HW_RESCAL); break;
        case HW_TRANSH: BYNE(HW_TRANSH); break;
        case HW_TRANSD: BYNE(HW_TRANSD); break;
        case HW_ANALOGY: BYNE(HW_ANALOGY); break;
        case HW_HOLE: BYNE(HW_HOLE); break;
        default: return HW_ERR_UNSUPPORTED;  // TorusE: the reference never differentiates its scorer
    }
#undef BYNE
#undef CALL
    if (gq) {
        const int dim = desc->dim;
        if (grads->head[0]) {  // gh[h_b] += M[r_b] gQ
            const unsigned grid = (unsigned)((n_tasks + 3) / 4);
#define PCALL(NE_)                                                                                                    \
    hipLaunchKernelGGL((project_rows_kernel<NE_, true>), dim3(grid), dim3(256), 0, st, desc->rel[0], desc->n_rel, gq, \
                       n_tasks, dim, r_idx, r_fixed, nullptr, n_tasks, grads->head[0], h_idx, desc->n_head, S)
            if (NE == 1) PCALL(1);
            else if (NE == 4) PCALL(4);
            else PCALL(16);
#undef PCALL
        }
        if (grads->rel[0]) {  // gM[r][i][j] += h[i] gQ[j]
            const OuterSide A{desc->head[0], dim, h_idx, desc->n_head, dim};
            const OuterSide B{gq, dim, nullptr, n_tasks, dim};
            launch_outer_by_relation(A, B, r_idx, desc->n_rel, r_fixed, S, n_tasks, grads->rel[0], st);
        }
    }
    if (replicas) {
        const int64_t n = rel_row_len * n_rel_tab;
        const int64_t rf = r_fixed < 0 ? 0 : (r_fixed >= desc->n_rel ? desc->n_rel - 1 : r_fixed);
        hipLaunchKernelGGL(rel_replica_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, replicas,
                           n_rel_tab, rel_row_len, grads->rel[0], grads->rel[1], grads->rel[2], rf);
    }
    return HW_LAUNCH_RC();
}
