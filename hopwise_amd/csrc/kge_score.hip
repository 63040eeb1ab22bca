// KGE scoring kernels for gfx950: grouped (1 positive + K negatives per (h, r)) and full sort.
//
// Both are HBM/L2-gather bound: ~0.5 flop per byte.  The design goal is bytes in flight, not flops:
//   * one embedding row is spread over G lanes (G = 64 for d = 200/256: one 16-byte load per lane per
//     table covers the row; G = 16/32 for short rows so that a wave carries 4/2 rows per instruction);
//   * every sub-wave group keeps U tail rows in flight (U x NT x 16-byte loads per lane) before the
//     first one is consumed;
//   * the (h, r) query is built once per wave task and reused for all its tails (kge_model.h).
// No LDS is used: a tail row is consumed by exactly one group, so staging it would only add a hop
// (cdna_hip_programming.md, 'glds vs register staging', GEMV row).
#include "kge_model.h"

namespace hw {

// ------------------------------------------------------------------------------------------------
// grouped scorer.  wave task = (group b, split s of S); the NS = 64/G sub-groups of the wave and the
// S splits interleave over the group's tails:  j = (k*S + s)*NS + sub,  k = 0, 1, ...
// ------------------------------------------------------------------------------------------------
template <int MODEL, int G, int NE, bool VEC, int U>
__global__ __launch_bounds__(256) void kge_score_grouped_kernel(
    hw_kge_desc d, const int64_t* __restrict__ h_idx, const int64_t* __restrict__ r_idx, int64_t r_fixed,
    const int64_t* __restrict__ t_idx, int64_t tsb, int64_t tsj, int64_t n_groups, int ntail, int S,
    float* __restrict__ out, int64_t osb, int64_t osj, int32_t* oob) {
    using SC = Scorer<MODEL, G, NE, VEC>;
    using MT = ModelTraits<MODEL>;
    using R = Row<G, NE, VEC>;
    constexpr int NS = HW_WAVE / G;

    const int lane = threadIdx.x & (HW_WAVE - 1);
    const int sub = lane / G, gl = lane % G;
    const int64_t task = (int64_t)blockIdx.x * (blockDim.x / HW_WAVE) + (threadIdx.x / HW_WAVE);
    if (task >= n_groups * S) return;  // wave-uniform
    const int64_t b = task / S;
    const int s = (int)(task % S);

    const int h = clamp_row(h_idx[b], d.n_head, oob);
    const int r = clamp_row(r_idx ? r_idx[b] : r_fixed, d.n_rel, oob);
    SC sc;
    sc.template build<false>(d, h, r, gl, nullptr, nullptr);

    const int64_t* __restrict__ tb = t_idx + b * tsb;
    float* __restrict__ ob = out + b * osb;
    const int step = S * NS;

    for (int j0 = s * NS; j0 < ntail; j0 += step * U) {  // wave-uniform trip count
        int jj[U];
        int tj[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            jj[u] = j0 + u * step + sub;
            tj[u] = jj[u] < ntail ? clamp_row(tb[(int64_t)jj[u] * tsj], d.n_tail, oob) : 0;
        }
        R t[U][MT::NT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int k = 0; k < MT::NT; ++k) t[u][k].load(d.tail[k] + (size_t)tj[u] * d.dim, d.dim, gl);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float p = sc.partial(t[u], d.eps, d.dim, gl);
            p = group_sum<G>(p);
            if (gl == 0 && jj[u] < ntail) ob[(int64_t)jj[u] * osj] = SC::finalize(p, d.margin);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// full sort.  wave task = (tail chunk c, head block hb): HB queries live in registers, the chunk's
// tail rows stream through once and are scored against all HB of them.  Head block is the fastest
// varying task index so the waves that share a tail chunk run together (L1/L2 reuse).
// ------------------------------------------------------------------------------------------------
template <int MODEL, int G, int NE, bool VEC, int HB, int U>
__global__ __launch_bounds__(256) void kge_full_sort_kernel(
    hw_kge_desc d, const int64_t* __restrict__ h_idx, const int64_t* __restrict__ r_idx, int64_t r_fixed,
    int n_heads, int64_t lo, int64_t hi, int chunk, float* __restrict__ out, int64_t ld, int32_t* oob) {
    using SC = Scorer<MODEL, G, NE, VEC>;
    using MT = ModelTraits<MODEL>;
    using R = Row<G, NE, VEC>;
    constexpr int NS = HW_WAVE / G;

    const int lane = threadIdx.x & (HW_WAVE - 1);
    const int sub = lane / G, gl = lane % G;
    const int n_hb = (n_heads + HB - 1) / HB;
    const int64_t n_chunks = (hi - lo + chunk - 1) / chunk;
    const int64_t task = (int64_t)blockIdx.x * (blockDim.x / HW_WAVE) + (threadIdx.x / HW_WAVE);
    if (task >= n_chunks * n_hb) return;
    const int hb = (int)(task % n_hb);
    const int64_t c = task / n_hb;

    SC sc[HB];
#pragma unroll
    for (int q = 0; q < HB; ++q) {
        const int bi = min(hb * HB + q, n_heads - 1);
        const int h = clamp_row(h_idx[bi], d.n_head, oob);
        const int r = clamp_row(r_idx ? r_idx[bi] : r_fixed, d.n_rel, oob);
        sc[q].template build<false>(d, h, r, gl, nullptr, nullptr);
    }

    const int64_t n0 = lo + c * chunk;
    const int64_t n1 = min(n0 + chunk, hi);
    for (int64_t nb = n0; nb < n1; nb += NS * U) {
        int64_t n[U];
        R t[U][MT::NT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            n[u] = nb + u * NS + sub;
            const int64_t row = n[u] < n1 ? n[u] : n0;
#pragma unroll
            for (int k = 0; k < MT::NT; ++k) t[u][k].load(d.tail[k] + (size_t)row * d.dim, d.dim, gl);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int q = 0; q < HB; ++q) {
                float p = sc[q].partial(t[u], d.eps, d.dim, gl);
                p = group_sum<G>(p);
                const int bi = hb * HB + q;
                if (gl == 0 && n[u] < n1 && bi < n_heads) out[(int64_t)bi * ld + (n[u] - lo)] = SC::finalize(p, d.margin);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host-side dispatch
// ------------------------------------------------------------------------------------------------
static int check_desc(const hw_kge_desc* d) {
    if (!d) return HW_ERR_INVALID_ARG;
    if (d->dim < 1 || d->dim > HW_MAX_DIM) return HW_ERR_INVALID_ARG;
    if (d->n_head < 1 || d->n_rel < 1 || d->n_tail < 1) return HW_ERR_INVALID_ARG;
    int nh, nr, nt;
    switch (d->model) {
        case HW_TRANSE:
        case HW_DISTMULT: nh = 1, nr = 1, nt = 1; break;
        case HW_ROTATE: nh = 2, nr = 1, nt = 2; break;
        case HW_COMPLEX: nh = 2, nr = 2, nt = 2; break;
        default: return HW_ERR_UNSUPPORTED;
    }
    for (int k = 0; k < nh; ++k)
        if (!d->head[k]) return HW_ERR_INVALID_ARG;
    for (int k = 0; k < nr; ++k)
        if (!d->rel[k]) return HW_ERR_INVALID_ARG;
    for (int k = 0; k < nt; ++k)
        if (!d->tail[k]) return HW_ERR_INVALID_ARG;
    // 16-byte loads need 16-byte aligned rows: dim % 4 == 0 and aligned table bases
    return HW_OK;
}

static bool tables_aligned16(const hw_kge_desc* d) {
    uintptr_t m = 0;
    for (int k = 0; k < 2; ++k) m |= (uintptr_t)d->head[k] | (uintptr_t)d->rel[k] | (uintptr_t)d->tail[k];
    return (m & 15) == 0;
}

#define HW_DISPATCH_SHAPE(MODEL, shape, CALL)                                   \
    do {                                                                        \
        if (shape.vec) {                                                        \
            if (shape.G == 16) { CALL(MODEL, 16, 4, true); }                    \
            else if (shape.G == 32) { CALL(MODEL, 32, 4, true); }               \
            else if (shape.NE == 4) { CALL(MODEL, 64, 4, true); }               \
            else if (shape.NE == 8) { CALL(MODEL, 64, 8, true); }               \
            else { CALL(MODEL, 64, 16, true); }                                 \
        } else {                                                                \
            if (shape.NE == 1) { CALL(MODEL, 64, 1, false); }                   \
            else if (shape.NE == 4) { CALL(MODEL, 64, 4, false); }              \
            else { CALL(MODEL, 64, 16, false); }                                \
        }                                                                       \
    } while (0)

#define HW_DISPATCH_MODEL(model, shape, CALL)                                   \
    do {                                                                        \
        switch (model) {                                                        \
            case HW_TRANSE: HW_DISPATCH_SHAPE(HW_TRANSE, shape, CALL); break;   \
            case HW_ROTATE: HW_DISPATCH_SHAPE(HW_ROTATE, shape, CALL); break;   \
            case HW_COMPLEX: HW_DISPATCH_SHAPE(HW_COMPLEX, shape, CALL); break; \
            case HW_DISTMULT: HW_DISPATCH_SHAPE(HW_DISTMULT, shape, CALL); break; \
            default: return HW_ERR_UNSUPPORTED;                                 \
        }                                                                       \
    } while (0)

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

}  // namespace hw

using namespace hw;

extern "C" int hw_kge_score_grouped(const hw_kge_desc* desc, const int64_t* h_idx, const int64_t* r_idx,
                                    int64_t r_fixed, const int64_t* t_idx, int64_t t_stride_b, int64_t t_stride_j,
                                    int64_t n_groups, int64_t n_tail_per_group, float* out, int64_t o_stride_b,
                                    int64_t o_stride_j, int32_t* oob_counter, void* stream) {
    int rc = check_desc(desc);
    if (rc) return rc;
    if (!h_idx || !t_idx || !out || n_groups < 0 || n_tail_per_group < 0) return HW_ERR_INVALID_ARG;
    if (n_tail_per_group > INT32_MAX / 2) return HW_ERR_INVALID_ARG;
    if (n_groups == 0 || n_tail_per_group == 0) return HW_OK;
    const RowShape shape = row_shape(desc->dim, tables_aligned16(desc));
    const int ntail = (int)n_tail_per_group;
    constexpr int U = 4;
    const int NS = HW_WAVE / shape.G;
    // splits per group: enough waves to fill 256 CUs x 32 wave slots, but keep >= U*NS tails per wave
    // so the (h, r) query build is amortised.
    int S = env_int("HW_GROUPED_SPLITS", 0);
    if (S <= 0) {
        const int per_wave = U * NS * 2;
        const int s_max = (ntail + per_wave - 1) / per_wave;
        const int64_t want = (16384 + n_groups - 1) / n_groups;
        S = (int)(want < 1 ? 1 : (want > s_max ? s_max : want));
    }
    const int64_t n_tasks = n_groups * S;
    const int64_t n_blocks = (n_tasks + 3) / 4;
    if (n_blocks > INT32_MAX) return HW_ERR_INVALID_ARG;
    HW_CLEAR_ERR();
    hipStream_t st = (hipStream_t)stream;
#define CALL(MODEL, G_, NE_, VEC_)                                                                              \
    hipLaunchKernelGGL((kge_score_grouped_kernel<MODEL, G_, NE_, VEC_, U>), dim3((unsigned)n_blocks), dim3(256), 0, \
                       st, *desc, h_idx, r_idx, r_fixed, t_idx, t_stride_b, t_stride_j, n_groups, ntail, S, out,  \
                       o_stride_b, o_stride_j, oob_counter)
    HW_DISPATCH_MODEL(desc->model, shape, CALL);
#undef CALL
    return HW_LAUNCH_RC();
}

extern "C" int hw_kge_full_sort(const hw_kge_desc* desc, const int64_t* h_idx, const int64_t* r_idx, int64_t r_fixed,
                                int64_t n_heads, int64_t tail_lo, int64_t tail_hi, float* out, int64_t out_ld,
                                int32_t* oob_counter, void* stream) {
    int rc = check_desc(desc);
    if (rc) return rc;
    if (!h_idx || !out || n_heads < 0 || n_heads > INT32_MAX) return HW_ERR_INVALID_ARG;
    if (tail_lo < 0 || tail_hi > desc->n_tail || tail_lo > tail_hi || out_ld < tail_hi - tail_lo)
        return HW_ERR_INVALID_ARG;
    if (n_heads == 0 || tail_lo == tail_hi) return HW_OK;
    const RowShape shape = row_shape(desc->dim, tables_aligned16(desc));
    const int NS = HW_WAVE / shape.G;
    const int64_t n = tail_hi - tail_lo;
    constexpr int U = 2;
    const bool hb4 = n_heads > 1;
    const int n_hb = hb4 ? (int)((n_heads + 3) / 4) : 1;
    // tail rows per wave task: aim for >= 8192 tasks, at least one U-batch, at most 256 rows
    int chunk = env_int("HW_FULLSORT_CHUNK", 0);
    if (chunk <= 0) {
        int64_t c = (n * n_hb + 8191) / 8192;
        const int unit = NS * U;
        c = ((c + unit - 1) / unit) * unit;
        if (c < unit) c = unit;
        if (c > 256) c = 256;
        chunk = (int)c;
    }
    const int64_t n_chunks = (n + chunk - 1) / chunk;
    const int64_t n_tasks = n_chunks * n_hb;
    const int64_t n_blocks = (n_tasks + 3) / 4;
    if (n_blocks > INT32_MAX) return HW_ERR_INVALID_ARG;
    HW_CLEAR_ERR();
    hipStream_t st = (hipStream_t)stream;
    const int nh = (int)n_heads;
#define CALL(MODEL, G_, NE_, VEC_)                                                                                  \
    do {                                                                                                            \
        if (hb4)                                                                                                    \
            hipLaunchKernelGGL((kge_full_sort_kernel<MODEL, G_, NE_, VEC_, 4, U>), dim3((unsigned)n_blocks),         \
                               dim3(256), 0, st, *desc, h_idx, r_idx, r_fixed, nh, tail_lo, tail_hi, chunk, out,     \
                               out_ld, oob_counter);                                                                 \
        else                                                                                                        \
            hipLaunchKernelGGL((kge_full_sort_kernel<MODEL, G_, NE_, VEC_, 1, U>), dim3((unsigned)n_blocks),         \
                               dim3(256), 0, st, *desc, h_idx, r_idx, r_fixed, nh, tail_lo, tail_hi, chunk, out,     \
                               out_ld, oob_counter);                                                                 \
    } while (0)
    HW_DISPATCH_MODEL(desc->model, shape, CALL);
#undef CALL
    return HW_LAUNCH_RC();
}
