// Full sorts as ONE dense product on the MFMA GEMM (gemm.hip) -- the pieces around the product.
//
// A full sort scores B queries against every candidate row.  For the bilinear scorers the score IS <query, tail>
// (distmult.py:50-51, complex.py:53-62, analogy.py:66-70); for the distance scorers whose distance is a plain L2 norm over the
// row -- TransE (transe.py:55-57: -||h + r - t||) and hopwise's RotatE (rotate.py:61-69: margin - the Frobenius norm of the
// (re, im) differences, i.e. an L2 norm over 2 d columns) --
//     ||q - t||^2 = ||q||^2 + ||t||^2 - 2 <q, t>
// so B x N x K of the work is the same product.  What is not the product:
//   * kge_query_rows_kernel: the [B, K] query rows, written by the scorers' own fold (kge_model.h: Scorer::build) -- one launch
//     instead of a chain of torch gathers / products / cat -- and ||q||^2 for the distance form;
//   * rows_sqnorm_kernel: ||t||^2 per candidate row (cached by the caller per table version);
//   * dist_finish_kernel: dot -> score in place.  The expansion cancels: its error is ~1e-7 (||q||^2 + ||t||^2), so where
//     the squared distance is a small fraction of that sum the entry is recomputed directly as sum_k (q_k - t_k)^2 from
//     the rows (a wave per such entry, rare: the query's nearest neighbours) -- every score the pass leaves behind is
//     within a few 1e-6 relative of the fp32 reference formula;
//   * the evaluation top-k needs no finish pass at all: with -||t||^2 / 2 as the product's per-column bias (hw_gemm_f32's
//     bias epilogue) the product itself is p = <q, t> - ||t||^2 / 2, and ranking by p IS ranking by score
//     (score = base - sqrt(||q||^2 - 2 p), monotone);
//   * topk_rerank_kernel: the top-k of an evaluation stays EXACT.  The caller takes the top-c (c > k) of the finished
//     matrix, re-scores those c candidates with the gather scorer (the kernel `predict` runs), and this kernel ranks them
//     by (exact value desc, index asc).  It also proves the candidate set was large enough: a column outside the set has
//     an approximate score <= the set's smallest one, so if the exact k-th value clears that bound plus the error slack,
//     nothing outside can belong to the top-k.  A row that cannot prove it raises a flag, and the fused gather kernel
//     (hw_kge_full_sort_topk_if, launched right behind with the flag as its predicate) redoes the batch exactly.
#include "kge_score_kernels.h"

using namespace hw;

namespace {

// hw_topk_rows' key: (orderable image of the value) << 32 | ~column -- unique per column
__device__ __forceinline__ uint64_t topk_key_u(float v, uint32_t idx) {
    uint32_t u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((uint64_t)u << 32) | (uint32_t)(~idx);
}


template <int MODEL, int G, int NE, bool VEC>
__global__ __launch_bounds__(256) void kge_query_rows_kernel(hw_kge_desc d, const int64_t* __restrict__ h_idx,
                                                             const int64_t* __restrict__ r_idx, int64_t r_fixed,
                                                             int64_t n, float* __restrict__ out, int64_t ld,
                                                             float* __restrict__ sqnorm, int vec_store, int32_t* oob) {
    using SC = Scorer<MODEL, G, NE, VEC>;
    using MT = ModelTraits<MODEL>;
    using R = Row<G, NE, VEC>;
    constexpr int NS = HW_WAVE / G;
    const int lane = threadIdx.x & (HW_WAVE - 1);
    const int sub = lane / G, gl = lane % G;
    const int64_t i = ((int64_t)blockIdx.x * 4 + wave_in_block()) * NS + sub;
    const bool live = i < n;
    const int64_t bi = live ? i : n - 1;  // idle groups rebuild the last query and store nothing (whole waves stay active)
    const int h = clamp_row(h_idx[bi], d.n_head, live ? oob : nullptr);
    const int r = clamp_row(r_idx ? r_idx[bi] : r_fixed, d.n_rel, live ? oob : nullptr);
    SC sc;
    sc.template build<false>(d, h, r, gl, nullptr, nullptr);
    float s = 0.f;
    float* __restrict__ row = out + bi * ld;
#pragma unroll
    for (int k = 0; k < MT::NQ; ++k) {
        if constexpr (VEC) {
#pragma unroll
            for (int c = 0; c < NE / 4; ++c) {
                const int col = R::col(4 * c, d.dim, gl);
                const f32x4 v = {sc.q[k].v[4 * c], sc.q[k].v[4 * c + 1], sc.q[k].v[4 * c + 2], sc.q[k].v[4 * c + 3]};
                s = fmaf(v.x, v.x, s), s = fmaf(v.y, v.y, s), s = fmaf(v.z, v.z, s), s = fmaf(v.w, v.w, s);
                if (live && col >= 0) {
                    float* p = row + (size_t)k * d.dim + col;
                    if (vec_store) *reinterpret_cast<f32x4*>(p) = v;
                    else p[0] = v.x, p[1] = v.y, p[2] = v.z, p[3] = v.w;
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int col = R::col(e, d.dim, gl);
                const float v = sc.q[k].v[e];
                s = fmaf(v, v, s);
                if (live && col >= 0) row[(size_t)k * d.dim + col] = v;
            }
        }
    }
    s = group_sum<G>(s);
    if (sqnorm && live && gl == 0) sqnorm[i] = s;
}

// out[j] = sum_k x[j, k]^2, a wave per row
__global__ __launch_bounds__(256) void rows_sqnorm_kernel(const float* __restrict__ x, int64_t ld, int64_t n_rows, int k,
                                                          int vec, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t j = (int64_t)blockIdx.x * 4 + wave_in_block();
    if (j >= n_rows) return;
    const float* __restrict__ row = x + j * ld;
    float s = 0.f;
    if (vec) {
        const f32x4* __restrict__ r4 = reinterpret_cast<const f32x4*>(row);
        for (int c = lane; c < (k >> 2); c += 64) {
            const f32x4 v = r4[c];
            s = fmaf(v.x, v.x, s), s = fmaf(v.y, v.y, s), s = fmaf(v.z, v.z, s), s = fmaf(v.w, v.w, s);
        }
    } else {
        for (int c = lane; c < k; c += 64) s = fmaf(row[c], row[c], s);
    }
    s = group_sum<64>(s);
    if (lane == 0) out[j] = s;
}

// sum_k (q[k] - t[k])^2 by the whole wave (every lane gets the total)
__device__ __forceinline__ float wave_sqdist(const float* __restrict__ q, const float* __restrict__ t, int k, int lane) {
    float s = 0.f;
    for (int c = lane; c < k; c += 64) {
        const float df = q[c] - t[c];
        s = fmaf(df, df, s);
    }
    return group_sum<64>(s);
}

// c[i, j] = base - sqrt(max(qn[i] + tn[j] - 2 c[i, j], 0)), in place; block = (1024-column stretch, row i).
// VEC: 16-byte accesses (ldc % 4 == 0, aligned base).  Entries with d^2 < frac (qn + tn) are recomputed from the rows.
template <bool VEC>
__global__ __launch_bounds__(256) void dist_finish_kernel(float* __restrict__ c, int64_t ldc, int64_t n,
                                                          const float* __restrict__ q, int64_t ldq,
                                                          const float* __restrict__ t, int64_t ldt, int k,
                                                          const float* __restrict__ qn, const float* __restrict__ tn,
                                                          float base, float frac, int mask_col0,
                                                          int32_t* __restrict__ n_rescored) {
    const int64_t i = blockIdx.y;
    const int lane = threadIdx.x & 63;
    float* __restrict__ row = c + i * ldc;
    const float qi = qn[i];
    const float* __restrict__ qrow = q + i * ldq;
    int64_t col[4];
    float v[4], out[4];
    const int64_t j0 = (int64_t)blockIdx.x * 1024;
    if constexpr (VEC) {
        const int64_t j = j0 + 4 * (int64_t)threadIdx.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) col[e] = j + e;
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        if (j < n) x = *reinterpret_cast<const f32x4*>(row + j);  // ldc >= n rounded up to 4: whole quads are in the row
        v[0] = x.x, v[1] = x.y, v[2] = x.z, v[3] = x.w;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            col[e] = j0 + threadIdx.x + 256 * e;
            v[e] = col[e] < n ? row[col[e]] : 0.f;
        }
    }
    unsigned flags = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const bool in = col[e] < n;
        const float tj = in ? tn[col[e]] : 0.f;
        const float sum = qi + tj;
        const float d2 = fmaf(-2.f, v[e], sum);
        if (in && d2 < frac * sum) flags |= 1u << e;
        out[e] = base - __builtin_amdgcn_sqrtf(fmaxf(d2, 0.f));
    }
    // the rare entries: one at a time, the whole wave on its two rows
    uint64_t todo = __ballot(flags != 0);
    int n_fixed = 0;
    while (todo) {
        const int src = __builtin_ctzll(todo);
        todo &= todo - 1;
        const unsigned f = (unsigned)__builtin_amdgcn_readlane((int)flags, src);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (!((f >> e) & 1u)) continue;  // wave-uniform
            const int64_t cj = VEC ? j0 + 4 * (int64_t)((threadIdx.x & ~63) + src) + e
                                   : j0 + ((threadIdx.x & ~63) + src) + 256 * e;
            const float d2 = wave_sqdist(qrow, t + cj * ldt, k, lane);
            if (lane == src) out[e] = base - __builtin_amdgcn_sqrtf(d2);
            ++n_fixed;
        }
    }
    if (n_rescored && n_fixed && lane == 0) atomicAdd(n_rescored, n_fixed);
    if (mask_col0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] = col[e] == 0 ? -INFINITY : out[e];
    }
    if constexpr (VEC) {
        if (col[0] < n) {
            // the quad may reach past n inside the padded row: those columns are never read
            *reinterpret_cast<f32x4*>(row + col[0]) = f32x4{out[0], out[1], out[2], out[3]};
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (col[e] < n) row[col[e]] = out[e];
    }
}

__global__ void mask_scores_ld_kernel(float* __restrict__ s, int64_t n_rows, int64_t n_cols, int64_t ld,
                                      const int64_t* __restrict__ hr, const int64_t* __restrict__ hc, int64_t n_hist,
                                      int mask_column0) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (mask_column0 && i < n_rows) s[i * ld] = -INFINITY;
    if (i < n_hist) {
        const int64_t r = hr[i], c = hc[i];
        if (r >= 0 && r < n_rows && c >= 0 && c < n_cols) s[r * ld + c] = -INFINITY;
    }
}

// A wave per row: c <= 64 candidates (column, approximate score, exact score) -> the k best by (exact desc, column asc).
__global__ __launch_bounds__(256) void topk_rerank_kernel(const int64_t* __restrict__ cand_idx,
                                                          const float* __restrict__ cand_approx,
                                                          const float* __restrict__ cand_exact, int64_t m, int c, int k,
                                                          float rel_slack, float abs_slack, const float* __restrict__ qn,
                                                          float base, float tn_max, float* __restrict__ out_val,
                                                          int64_t* __restrict__ out_idx, int32_t* __restrict__ unverified,
                                                          int dot_form) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave_in_block();
    if (r >= m) return;
    const bool has = lane < c;
    const int64_t idx = has ? cand_idx[r * c + lane] : -1;
    const float ap = has ? cand_approx[r * c + lane] : INFINITY;
    float ex = has ? cand_exact[r * c + lane] : -INFINITY;
    const bool real = has && idx >= 0;
    if (real && ap == -INFINITY) ex = -INFINITY;  // a masked column stays masked
    const uint64_t key = real ? score_key(ex, (uint32_t)idx) : 0ull;
    int rank = 0;
    for (int j = 0; j < 64; ++j) {
        const uint64_t other = __shfl(key, j, 64);
        rank += other > key ? 1 : 0;
    }
    if (real && rank < k) {
        out_val[r * k + rank] = ex;
        out_idx[r * k + rank] = idx;
    }
    // fewer than k real candidates: the remaining ranks as the exact kernels write them
    const int n_real = __popcll(__ballot(real));
    if (lane >= n_real && lane < k) {
        out_val[r * k + lane] = -INFINITY;
        out_idx[r * k + lane] = -1;
    }
    // proof that no column outside the candidate set belongs to the top-k
    float lo = has ? (real ? ap : -INFINITY) : INFINITY;  // the set's smallest approximate score
#pragma unroll
    for (int mm = 32; mm >= 1; mm >>= 1) lo = fminf(lo, __shfl_xor(lo, mm, 64));
    const uint64_t kth = __ballot(real && rank == k - 1);
    bool ok;
    if (lo == -INFINITY || n_real < c) {
        ok = true;  // the set holds every unmasked column of the row
    } else if (!kth) {
        ok = false;
    } else {
        const float e_k = __shfl(ex, __builtin_ctzll(kth), 64);
        if (qn && dot_form) {
            // product form (the bilinear scorers: score = <q, t>): a column outside the set has an approximate product <= lo
            // and the approximation errs by at most rel_slack ||q|| max ||t|| (hw_gemm_rank: 1.2e-5 sum |q_k t_k|)
            ok = e_k > lo + rel_slack * __builtin_amdgcn_sqrtf(qn[r] * tn_max) + abs_slack;
        } else if (qn) {
            // distance form: the approximate values are p = <q, t> - ||t||^2 / 2 (the biased product), so a column outside
            // the set has p <= lo, i.e. d^2 = ||q||^2 - 2 p >= ||q||^2 - 2 lo, less the product's error (rel_slack of
            // ||q||^2 + max ||t||^2): its score is at most base - sqrt of that
            const float q2 = qn[r];
            const float d2 = q2 - 2.f * lo - rel_slack * (q2 + tn_max);
            ok = e_k > (base - __builtin_amdgcn_sqrtf(fmaxf(d2, 0.f))) + abs_slack;
        } else {
            ok = e_k > lo + (rel_slack * fabsf(lo) + abs_slack);  // false on NaN
        }
    }
    if (!ok && lane == 0) atomicOr(unverified, 1);
}


// ---- evaluation top-k from per-stretch maxima ----------------------------------------------------------------------------------
// The ranking product (hw_gemm_rank) leaves, next to the masked score matrix, the maximum of every 64-column stretch of every row.
// The k-th largest stretch maximum of a row, T, is a lower bound of the row's k-th largest score (k stretches hold a score >= T),
// so the row's k best columns all lie in the stretches whose maximum is >= T -- k of them plus ties.  One wave per row: T by a
// bitwise search over the images of the <= 16 maxima a lane holds (ballot counts, no cross-lane reduction), the qualifying
// stretches' 64 scores each, the scores >= T into a list in LDS, the k largest keys of the list in order.  A row of 45 539 columns
// is found from 712 maxima + ~16 x 256 bytes instead of its 182 KB (the two-stage top-k reads the whole matrix: 0.13-0.16 ms for
// 2 048 such rows).  Keys, order and the exhausted-row convention are hw_topk_rows' (value desc, column asc; -inf / -1).  Rows
// whose qualifying stretches or candidates outgrow the LDS lists (many equal scores) raise `overflow` instead of answering.
constexpr int TM_MAX_PER_LANE = 16;   // stretches per lane: n_cols <= 64 * 64 * 16 = 65 536
constexpr int TM_QUAL = 128;          // qualifying stretches kept per row
constexpr int TM_LIST = 512;          // candidate keys kept per row

__device__ __forceinline__ uint32_t tm_image(float v) {
    const uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(256) void topk_tile_max_kernel(const float* __restrict__ s, int64_t ld, int64_t n_rows,
                                                            int64_t n_cols, const float* __restrict__ tile_max,
                                                            int64_t tile_ld, int k, float* __restrict__ out_val,
                                                            int64_t* __restrict__ out_idx, int32_t* __restrict__ overflow) {
    __shared__ uint64_t list[4][TM_LIST];
    __shared__ int qual[4][TM_QUAL];
    const int lane = threadIdx.x & 63, w = wave_in_block();
    const int64_t row = (int64_t)blockIdx.x * 4 + w;
    if (row >= n_rows) return;  // no block-wide barrier below
    const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    const int n_tiles = (int)((n_cols + 63) / 64);
    const float* __restrict__ tm = tile_max + row * tile_ld;
    uint32_t img[TM_MAX_PER_LANE];
#pragma unroll
    for (int i = 0; i < TM_MAX_PER_LANE; ++i) {  // stretch t = 64 i + lane; 0 = no such stretch (a real image is never 0)
        const int t = 64 * i + lane;
        img[i] = t < n_tiles ? tm_image(tm[t < n_tiles ? t : 0]) : 0u;
    }
    // T = the k-th largest image, counted with multiplicity (0 when the row has fewer than k stretches)
    uint32_t T = 0u;
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t c = T | (1u << bit);
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < TM_MAX_PER_LANE; ++i) cnt += __popcll(__ballot(img[i] >= c));
        if (cnt >= k) T = c;  // wave-uniform
    }
    // the qualifying stretches, in stretch order
    int n_qual = 0;
#pragma unroll
    for (int i = 0; i < TM_MAX_PER_LANE; ++i) {
        const bool q = img[i] != 0u && img[i] >= T;
        const unsigned long long b = __ballot(q);
        const int pos = n_qual + __popcll(b & lt_mask);
        if (q && pos < TM_QUAL) qual[w][pos] = 64 * i + lane;
        n_qual += __popcll(b);
    }
    bool over = n_qual > TM_QUAL;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const float* __restrict__ srow = s + row * ld;
    int n_list = 0;
    const int nq = over ? 0 : n_qual;
    for (int q = 0; q < nq; ++q) {  // wave-uniform trips
        const int64_t col = (int64_t)qual[w][q] * 64 + lane;
        const bool in = col < n_cols;
        const float x = in ? srow[col] : -INFINITY;
        const bool pass = in && tm_image(x) >= T;
        const unsigned long long b = __ballot(pass);
        const int pos = n_list + __popcll(b & lt_mask);
        if (pass && pos < TM_LIST) list[w][pos] = topk_key_u(x, (uint32_t)col);
        n_list += __popcll(b);
    }
    over = over || n_list > TM_LIST;
    if (over) {  // no answer for this row: the flag says so, the outputs stay harmless (column 0, -inf)
        if (lane == 0) atomicOr(overflow, 1);
        for (int r = lane; r < k; r += 64) out_val[row * k + r] = -INFINITY, out_idx[row * k + r] = 0;
        return;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint64_t bound = ~0ull;
    for (int round = 0; round < k; ++round) {
        uint64_t best = 0;
        for (int j = lane; j < n_list; j += 64) {
            const uint64_t x = list[w][j];
            best = (x < bound && x > best) ? x : best;
        }
        best = wave_max_u64(best);
        if (lane == 0) {
            const int64_t o = row * k + round;
            if (best == 0) {
                out_val[o] = -INFINITY;
                out_idx[o] = -1;
            } else {
                const uint32_t idx = ~(uint32_t)(best & 0xffffffffu);
                out_val[o] = srow[idx];
                out_idx[o] = idx;
            }
        }
        bound = best;
    }
}

// history pairs -> the bit matrix hw_gemm_rank masks with: bit (row, col) of a zeroed [n_rows, words] array of 32-bit words
__global__ __launch_bounds__(256) void mask_bits_kernel(const int64_t* __restrict__ hr, const int64_t* __restrict__ hc,
                                                        int64_t n_hist, int64_t n_rows, int64_t n_cols, int64_t words,
                                                        int mask_column0, uint32_t* __restrict__ bits) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (mask_column0 && i < n_rows) atomicOr(bits + i * words, 1u);
    if (i < n_hist) {
        const int64_t r = hr[i], c = hc[i];
        if (r >= 0 && r < n_rows && c >= 0 && c < n_cols) atomicOr(bits + r * words + (c >> 5), 1u << (c & 31));
    }
}

}  // namespace

#define HW_QUERY_MODELS(model, shape, CALL)                                                 \
    do {                                                                                    \
        switch (model) {                                                                    \
            case HW_TRANSE: HW_DISPATCH_SHAPE(HW_TRANSE, shape, CALL); break;               \
            case HW_ROTATE: HW_DISPATCH_SHAPE(HW_ROTATE, shape, CALL); break;               \
            case HW_COMPLEX: HW_DISPATCH_SHAPE(HW_COMPLEX, shape, CALL); break;             \
            case HW_DISTMULT: HW_DISPATCH_SHAPE(HW_DISTMULT, shape, CALL); break;           \
            case HW_ANALOGY: HW_DISPATCH_SHAPE(HW_ANALOGY, shape, CALL); break;             \
            default: return HW_ERR_UNSUPPORTED;                                             \
        }                                                                                   \
    } while (0)

extern "C" int hw_kge_query_rows(const hw_kge_desc* desc, const int64_t* h_idx, const int64_t* r_idx, int64_t r_fixed,
                                 int64_t n, float* out, int64_t ld, float* sqnorm, int32_t* oob_counter, void* stream) {
    int rc = check_desc(desc);
    if (rc) return rc;
    if (!h_idx || !out || n < 0) return HW_ERR_INVALID_ARG;
    int nq;
    switch (desc->model) {
        case HW_TRANSE:
        case HW_DISTMULT: nq = 1; break;
        case HW_ROTATE:
        case HW_COMPLEX: nq = 2; break;
        case HW_ANALOGY: nq = 3; break;
        default: return HW_ERR_UNSUPPORTED;
    }
    if (ld < (int64_t)nq * desc->dim) return HW_ERR_INVALID_ARG;
    if (n == 0) return HW_OK;
    const RowShape shape = row_shape(desc->dim, tables_aligned16(desc));
    const int NS = HW_WAVE / shape.G;
    const int64_t n_blocks = (n + 4 * NS - 1) / (4 * NS);
    if (n_blocks > INT32_MAX) return HW_ERR_INVALID_ARG;
    const int vec_store = (((uintptr_t)out & 15) == 0 && (ld & 3) == 0) ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    HW_CLEAR_ERR();
#define CALL(MODEL, G_, NE_, VEC_)                                                                                    \
    hipLaunchKernelGGL((kge_query_rows_kernel<MODEL, G_, NE_, VEC_>), dim3((unsigned)n_blocks), dim3(256), 0, st, *desc, \
                       h_idx, r_idx, r_fixed, n, out, ld, sqnorm, vec_store, oob_counter)
    HW_QUERY_MODELS(desc->model, shape, CALL);
#undef CALL
    return HW_LAUNCH_RC();
}

extern "C" int hw_rows_sqnorm(const float* x, int64_t ld, int64_t n_rows, int32_t k, float* out, void* stream) {
    if (!x || !out || n_rows < 0 || k < 1 || ld < k) return HW_ERR_INVALID_ARG;
    if (n_rows == 0) return HW_OK;
    const int64_t n_blocks = (n_rows + 3) / 4;
    if (n_blocks > INT32_MAX) return HW_ERR_INVALID_ARG;
    const int vec = (((uintptr_t)x & 15) == 0 && (ld & 3) == 0 && (k & 3) == 0) ? 1 : 0;
    HW_CLEAR_ERR();
    hipLaunchKernelGGL(rows_sqnorm_kernel, dim3((unsigned)n_blocks), dim3(256), 0, (hipStream_t)stream, x, ld, n_rows, (int)k,
                       vec, out);
    return HW_LAUNCH_RC();
}

extern "C" int hw_kge_dist_finish(float* c, int64_t ldc, int64_t m, int64_t n, const float* q, int64_t ldq, const float* t,
                                  int64_t ldt, int32_t k, const float* qn, const float* tn, float base,
                                  float rescore_frac, int32_t mask_column0, int32_t* n_rescored, void* stream) {
    if (!c || !q || !t || !qn || !tn || m < 0 || n < 0 || k < 1 || ldc < n || ldq < k || ldt < k) return HW_ERR_INVALID_ARG;
    if (!(rescore_frac >= 0.f && rescore_frac <= 1.f)) return HW_ERR_INVALID_ARG;
    if (m == 0 || n == 0) return HW_OK;
    if (m > 65535) return HW_ERR_INVALID_ARG;  // rows ride in gridDim.y
    const int64_t gx = (n + 1023) / 1024;
    if (gx > INT32_MAX) return HW_ERR_INVALID_ARG;
    // 16-byte accesses when every row starts aligned and whole quads fit the row's storage
    const bool vec = ((uintptr_t)c & 15) == 0 && (ldc & 3) == 0 && ldc >= ((n + 3) & ~(int64_t)3);
    hipStream_t st = (hipStream_t)stream;
    HW_CLEAR_ERR();
    if (vec)
        hipLaunchKernelGGL((dist_finish_kernel<true>), dim3((unsigned)gx, (unsigned)m), dim3(256), 0, st, c, ldc, n, q, ldq, t,
                           ldt, (int)k, qn, tn, base, rescore_frac, (int)(mask_column0 != 0), n_rescored);
    else
        hipLaunchKernelGGL((dist_finish_kernel<false>), dim3((unsigned)gx, (unsigned)m), dim3(256), 0, st, c, ldc, n, q, ldq,
                           t, ldt, (int)k, qn, tn, base, rescore_frac, (int)(mask_column0 != 0), n_rescored);
    return HW_LAUNCH_RC();
}

extern "C" int hw_mask_scores_ld(float* scores, int64_t n_rows, int64_t n_cols, int64_t ld, const int64_t* hist_rows,
                                 const int64_t* hist_cols, int64_t n_hist, int32_t mask_column0, void* stream) {
    if (!scores || n_rows < 0 || n_cols < 1 || ld < n_cols || n_hist < 0 || (n_hist > 0 && (!hist_rows || !hist_cols)))
        return HW_ERR_INVALID_ARG;
    const int64_t n = (mask_column0 && n_rows > n_hist) ? n_rows : n_hist;
    if (n == 0 || n_rows == 0) return HW_OK;
    HW_CLEAR_ERR();
    hipLaunchKernelGGL(mask_scores_ld_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, scores,
                       n_rows, n_cols, ld, hist_rows, hist_cols, n_hist, (int)(mask_column0 != 0));
    return HW_LAUNCH_RC();
}

extern "C" int hw_topk_rerank(const int64_t* cand_idx, const float* cand_approx, const float* cand_exact, int64_t m,
                              int32_t c, int32_t k, float rel_slack, float abs_slack, const float* qn, float base,
                              float tn_max, float* out_val, int64_t* out_idx, int32_t* unverified, void* stream) {
    if (!cand_idx || !cand_approx || !cand_exact || !out_val || !out_idx || !unverified) return HW_ERR_INVALID_ARG;
    if (m < 0 || c < 1 || c > 64 || k < 1 || k > c || !(rel_slack >= 0.f) || !(abs_slack >= 0.f)) return HW_ERR_INVALID_ARG;
    if (qn && !(tn_max >= 0.f)) return HW_ERR_INVALID_ARG;
    if (m == 0) return HW_OK;
    const int64_t n_blocks = (m + 3) / 4;
    if (n_blocks > INT32_MAX) return HW_ERR_INVALID_ARG;
    HW_CLEAR_ERR();
    hipLaunchKernelGGL(topk_rerank_kernel, dim3((unsigned)n_blocks), dim3(256), 0, (hipStream_t)stream, cand_idx, cand_approx,
                       cand_exact, m, (int)c, (int)k, rel_slack, abs_slack, qn, base, tn_max, out_val, out_idx, unverified, 0);
    return HW_LAUNCH_RC();
}

extern "C" int hw_topk_rerank_dot(const int64_t* cand_idx, const float* cand_approx, const float* cand_exact, int64_t m,
                                  int32_t c, int32_t k, float rel_slack, float abs_slack, const float* qn, float tn_max,
                                  float* out_val, int64_t* out_idx, int32_t* unverified, void* stream) {
    if (!cand_idx || !cand_approx || !cand_exact || !out_val || !out_idx || !unverified || !qn) return HW_ERR_INVALID_ARG;
    if (m < 0 || c < 1 || c > 64 || k < 1 || k > c || !(rel_slack >= 0.f) || !(abs_slack >= 0.f) || !(tn_max >= 0.f))
        return HW_ERR_INVALID_ARG;
    if (m == 0) return HW_OK;
    const int64_t n_blocks = (m + 3) / 4;
    if (n_blocks > INT32_MAX) return HW_ERR_INVALID_ARG;
    HW_CLEAR_ERR();
    hipLaunchKernelGGL(topk_rerank_kernel, dim3((unsigned)n_blocks), dim3(256), 0, (hipStream_t)stream, cand_idx, cand_approx,
                       cand_exact, m, (int)c, (int)k, rel_slack, abs_slack, qn, 0.f, tn_max, out_val, out_idx, unverified, 1);
    return HW_LAUNCH_RC();
}

extern "C" int64_t hw_mask_bits_words(int64_t n_cols) { return n_cols < 1 ? 0 : (n_cols + 31) / 32; }

extern "C" int hw_mask_bits(const int64_t* hist_rows, const int64_t* hist_cols, int64_t n_hist, int64_t n_rows, int64_t n_cols,
                            int32_t mask_column0, uint32_t* bits, int64_t words_per_row, void* stream) {
    if (!bits || n_rows < 0 || n_cols < 1 || words_per_row < (n_cols + 31) / 32 || n_hist < 0 ||
        (n_hist > 0 && (!hist_rows || !hist_cols)))
        return HW_ERR_INVALID_ARG;
    if (n_rows == 0) return HW_OK;
    hipStream_t st = (hipStream_t)stream;
    if (hw::fill_bytes_async(bits, 0, sizeof(uint32_t) * (size_t)(n_rows * words_per_row), st) != hipSuccess) return HW_ERR_LAUNCH;
    const int64_t n = (mask_column0 && n_rows > n_hist) ? n_rows : n_hist;
    if (n == 0) return HW_OK;
    HW_CLEAR_ERR();
    hipLaunchKernelGGL(mask_bits_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, hist_rows, hist_cols, n_hist, n_rows,
                       n_cols, words_per_row, (int)(mask_column0 != 0), bits);
    return HW_LAUNCH_RC();
}

extern "C" int hw_topk_from_tile_max(const float* scores, int64_t ld, int64_t n_rows, int64_t n_cols, const float* tile_max,
                                     int64_t tile_ld, int32_t k, float* out_val, int64_t* out_idx, int32_t* overflow,
                                     void* stream) {
    if (!scores || !tile_max || !out_val || !out_idx || !overflow || n_rows < 0 || n_cols < 1 || ld < n_cols || k < 1 || k > 64 ||
        tile_ld < (n_cols + 63) / 64)
        return HW_ERR_INVALID_ARG;
    if (n_cols > (int64_t)64 * 64 * TM_MAX_PER_LANE) return HW_ERR_UNSUPPORTED;
    if (n_rows == 0) return HW_OK;
    const int64_t n_blocks = (n_rows + 3) / 4;
    if (n_blocks > INT32_MAX) return HW_ERR_INVALID_ARG;
    HW_CLEAR_ERR();
    hipLaunchKernelGGL(topk_tile_max_kernel, dim3((unsigned)n_blocks), dim3(256), 0, (hipStream_t)stream, scores, ld, n_rows, n_cols,
                       tile_max, tile_ld, (int)k, out_val, out_idx, overflow);
    return HW_LAUNCH_RC();
}
