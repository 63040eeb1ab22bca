/* hopwise_amd.h -- C ABI of the MI355X-native KGE-scoring / negative-sampling / PGPR beam-search engine.
 *
 * The shared library (hopwise_amd/csrc/libhopwise_amd.so, built by hipcc for gfx950) is the drop-in
 * boundary.  Every entry point
 *   - takes plain device pointers + sizes (caller owns every buffer; nothing is allocated or freed),
 *   - launches asynchronously on the given hipStream_t (passed as void*; NULL = the null stream),
 *   - returns HW_OK (0) or a negative hw_status code, never throws, never synchronises,
 *   - is safe to capture into a hipGraph.
 * All tables are row-major contiguous float32 [rows, dim]; all ids are int64 (what the reference's
 * Interaction carries, /root/reference/hopwise/data/interaction.py:44-118).  Out-of-range ids are
 * clamped into the table (never fault) and counted in the optional `oob_counter`.
 *
 * Each function names the reference interface it replaces (paths relative to /root/reference/hopwise).
 */
#ifndef HOPWISE_AMD_H
#define HOPWISE_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    HW_OK = 0,
    HW_ERR_INVALID_ARG = -1,  /* null pointer, negative size, unsupported dim */
    HW_ERR_UNSUPPORTED = -2,  /* model / option combination not built */
    HW_ERR_LAUNCH = -3,       /* hipGetLastError() != hipSuccess after the launch */
    HW_ERR_WORKSPACE = -4     /* caller workspace too small */
} hw_status;

typedef enum {
    HW_TRANSE = 0,   /* model/knowledge_graph_embedding_recommender/transe.py:55-57   */
    HW_ROTATE = 1,   /* .../rotate.py:61-69   */
    HW_COMPLEX = 2,  /* .../complex.py:53-62  */
    HW_DISTMULT = 3, /* .../distmult.py:50-51 */
    HW_RESCAL = 4,   /* .../rescal.py:51-54: h^T M_r t; rel[0] rows hold dim*dim floats (M_r row-major) */
    /* widened set */
    HW_TORUSE = 5,   /* .../toruse.py:68-79: -4 sum min(x^2, 1 - x^2), x = frac(h) + frac(r) - frac(t) (scoring only:
                        the reference trains TorusE with TransE's triplet loss on the raw rows, toruse.py:81-103) */
    HW_TRANSH = 6,   /* .../transh.py:51-55,74-75: -||(h - t) * c + r||, c = 1 - sum(w_r) * w_r (the reference's
                        project() multiplies by the SUM of the norm vector, element-wise -- reproduced) */
    HW_TRANSD = 7,   /* .../transd.py:89-94: -||h + rv (h . hv) + r - t - rv (t . tv)|| */
    HW_ANALOGY = 8,  /* .../analogy.py:66-70: DistMult on (e) + complex product on (re, im), all of width dim */
    HW_HOLE = 9      /* .../hole.py:53-61: (h circ-conv r) . t with R[i][j] = r[(j - i) mod d] */
} hw_kge_model;

#define HW_KGE_MAX_TABLES 3

/* Table set of one scorer call.  Unused slots are NULL.
 *   TransE/DistMult/RESCAL/TorusE/HolE: head[0], rel[0], tail[0]
 *   RotatE:          head = {re, im}, rel = {theta}, tail = {re, im}
 *   ComplEx:         head = {re, im}, rel = {re, im}, tail = {re, im}
 *   TransH:          head = {e}, rel = {r, norm_vec}, tail = {e}
 *   TransD:          head = {e, e_vec}, rel = {r, r_vec}, tail = {e, e_vec}
 *   Analogy:         head = {e, re, im}, rel = {r, re, im}, tail = {e, re, im}
 * "head" is the user table(s) on the rec side and the entity table(s) on the kg side. */
typedef struct {
    int32_t model;        /* hw_kge_model */
    int32_t dim;          /* embedding_size, 1..1024 */
    const float* head[HW_KGE_MAX_TABLES];
    const float* rel[HW_KGE_MAX_TABLES];
    const float* tail[HW_KGE_MAX_TABLES];
    int64_t n_head;       /* rows of head[*] */
    int64_t n_rel;
    int64_t n_tail;
    float margin;         /* RotatE: score = margin - ||.|| (rotate.py:69) */
    float eps;            /* TransE / TransH / TransD: distance of (h + r - t + eps); 0 for scoring, 1e-6 inside the
                             TripletMarginLoss path (torch pairwise_distance) */
} hw_kge_desc;

/* Gradient destination tables, same slot layout as hw_kge_desc (dense [rows, dim] float32,
 * accumulated into with atomic adds; the caller zeroes them).  NULL = that slot gets no gradient. */
typedef struct {
    float* head[HW_KGE_MAX_TABLES];
    float* rel[HW_KGE_MAX_TABLES];
    float* tail[HW_KGE_MAX_TABLES];
} hw_kge_grads;

/* ---- KGE scoring ---------------------------------------------------------------------------------
 * score[b, j] of group b = (h_idx[b], r) against tail t_idx[b*t_stride_b + j*t_stride_j], j in [0, n_tail_per_group).
 * r = r_idx[b] or, when r_idx == NULL, the fixed relation row r_fixed (the [UI-Relation] row used by the
 * rec side: transe.py:63, rotate.py:76-77).  out[b*o_stride_b + j*o_stride_j].
 * Replaces <Model>.forward on gathered rows as used by calculate_loss / predict / predict_kg
 * (rotate.py:98-159 and the same methods of transe.py / complex.py / distmult.py); with
 * n_tail_per_group = 1 + K it scores one positive and K negatives per (h, r) group in one pass. */
int hw_kge_score_grouped(const hw_kge_desc* desc,
                         const int64_t* h_idx, const int64_t* r_idx, int64_t r_fixed,
                         const int64_t* t_idx, int64_t t_stride_b, int64_t t_stride_j,
                         int64_t n_groups, int64_t n_tail_per_group,
                         float* out, int64_t o_stride_b, int64_t o_stride_j,
                         int32_t* oob_counter, void* stream);

/* The same with the negatives drawn inside the scorer: column 0 of t_idx (the positive) is read, columns
 * 1 .. n_tail_per_group-1 are DRAWN -- uniform over [1, value_num) minus used(keys[b]), the counters and rejection rule of
 * hw_neg_sample with num = n_tail_per_group - 1, so the ids are the ones that call would write -- scored, and written
 * back to t_idx for the gradient pass.  One launch instead of sampler + scorer.  The key's used set is an LDS bitmap for
 * value_num <= 65536 and is searched in place (binary search in its sorted CSR slice, as hw_neg_sample's generic kernel
 * does) beyond that.  HW_ERR_UNSUPPORTED for the widened scorer set: sample and score separately then. */
int hw_kge_score_grouped_sampled(const hw_kge_desc* desc, const int64_t* h_idx, const int64_t* r_idx, int64_t r_fixed,
                                 int64_t* t_idx, int64_t t_stride_b, int64_t t_stride_j, int64_t n_groups,
                                 int64_t n_tail_per_group, const int64_t* used_ptr, const int64_t* used_idx,
                                 int64_t n_key_space, const int64_t* keys, int64_t value_num, uint64_t seed,
                                 uint64_t offset, float* out, int64_t o_stride_b, int64_t o_stride_j,
                                 int32_t* oob_counter, void* stream);

/* hw_kge_score_grouped_sampled followed by hw_loss_self_adversarial(scores, alpha) (loss only, no gradient) with the loss's
 * per-group pass inside the scorer launch: the block that scores a group keeps its scores in LDS and reduces them; a
 * one-block finish kernel adds the per-group terms -- same arithmetic and order as the separate calls, so ids, scores and
 * loss are equal bit for bit.  group_loss_ws: n_groups floats; counter_ws: reserved (may be NULL).  HW_ERR_UNSUPPORTED when a group cannot be owned
 * by one workgroup (fewer than ~32 tails per split, or the cases hw_kge_score_grouped_sampled refuses): use the two calls. */
int hw_kge_score_grouped_sampled_selfadv(const hw_kge_desc* desc, const int64_t* h_idx, const int64_t* r_idx,
                                         int64_t r_fixed, int64_t* t_idx, int64_t t_stride_b, int64_t t_stride_j,
                                         int64_t n_groups, int64_t n_tail_per_group, const int64_t* used_ptr,
                                         const int64_t* used_idx, int64_t n_key_space, const int64_t* keys,
                                         int64_t value_num, uint64_t seed, uint64_t offset, float* out, int64_t o_stride_b,
                                         int64_t o_stride_j, float alpha, float* loss_out, float* group_loss_ws,
                                         int32_t* counter_ws, int32_t* oob_counter, void* stream);

/* Chained form of hw_kge_score_grouped_sampled_selfadv for a stream of batches (the training loop's forward leg; same
 * reference lines: sampler.py:226-257 + rotate.py:98-131 + the self-adversarial loss): this launch leaves its n_groups
 * group terms in group_loss_ws WITHOUT the one-block finish launch, and -- when prev_group_loss is not NULL -- its block 0
 * adds the prev_n_groups terms an earlier launch on the same stream left in prev_group_loss into prev_loss_out[0] (same
 * order as the finish kernel: equal bit for bit).  prev_group_loss must not alias group_loss_ws (alternate two buffers).
 * The last batch of a chain is finished by hw_selfadv_finish.  Saves the 4-5 us finish launch per batch. */
int hw_kge_score_grouped_sampled_selfadv_chained(const hw_kge_desc* desc, const int64_t* h_idx, const int64_t* r_idx,
                                                 int64_t r_fixed, int64_t* t_idx, int64_t t_stride_b, int64_t t_stride_j,
                                                 int64_t n_groups, int64_t n_tail_per_group, const int64_t* used_ptr,
                                                 const int64_t* used_idx, int64_t n_key_space, const int64_t* keys,
                                                 int64_t value_num, uint64_t seed, uint64_t offset, float* out,
                                                 int64_t o_stride_b, int64_t o_stride_j, float alpha, float* group_loss_ws,
                                                 const float* prev_group_loss, int64_t prev_n_groups, float* prev_loss_out,
                                                 int32_t* oob_counter, void* stream);

/* loss_out[0] = mean of group_loss[0 .. n_groups) in the finish order of hw_kge_score_grouped_sampled_selfadv. */
int hw_selfadv_finish(const float* group_loss, int64_t n_groups, float* loss_out, void* stream);

/* Full sort: out[b, n - tail_lo] = score(h_idx[b], r, n) for every tail row n in [tail_lo, tail_hi).
 * Replaces full_sort_predict / full_sort_predict_kg (rotate.py:161-220, transe.py:112-154,
 * complex.py:162-219, distmult.py:119-146) without the [B, N, d] temporaries. */
int hw_kge_full_sort(const hw_kge_desc* desc,
                     const int64_t* h_idx, const int64_t* r_idx, int64_t r_fixed,
                     int64_t n_heads, int64_t tail_lo, int64_t tail_hi,
                     float* out, int64_t out_ld,
                     int32_t* oob_counter, void* stream);

/* Full sort fused with the evaluation mask and top-k (KGTrainer._full_sort_batch_eval trainer.py:716-735 +
 * Collector.eval_batch_collect collector.py:173-183): scores every tail in [tail_lo, tail_hi), sets column 0 (when
 * mask_first_col) and every (hist_rows[i], hist_cols[i]) to -inf, and returns the k best (value desc, index asc)
 * columns per head without materialising the [n_heads, n_tails] matrix.  k <= 128.
 * workspace: hw_kge_full_sort_topk_workspace_bytes(n_heads, n_tails, k) bytes. */
int64_t hw_kge_full_sort_topk_workspace_bytes(int64_t n_heads, int64_t n_tails, int32_t k);
int hw_kge_full_sort_topk(const hw_kge_desc* desc, const int64_t* h_idx, const int64_t* r_idx, int64_t r_fixed,
                          int64_t n_heads, int64_t tail_lo, int64_t tail_hi, const int64_t* hist_rows,
                          const int64_t* hist_cols, int64_t n_hist, int32_t mask_first_col, int32_t k,
                          void* workspace, int64_t workspace_bytes, float* out_val, int64_t* out_idx, void* stream);

/* hw_kge_full_sort_topk under a device-side predicate: every workgroup leaves at once unless *run_if != 0 (run_if == NULL:
 * always run).  The exact fallback queued behind the re-rank of the GEMM full sort below; no host read decides it. */
int hw_kge_full_sort_topk_if(const hw_kge_desc* desc, const int64_t* h_idx, const int64_t* r_idx, int64_t r_fixed,
                             int64_t n_heads, int64_t tail_lo, int64_t tail_hi, const int64_t* hist_rows,
                             const int64_t* hist_cols, int64_t n_hist, int32_t mask_first_col, int32_t k,
                             void* workspace, int64_t workspace_bytes, float* out_val, int64_t* out_idx,
                             const int32_t* run_if, void* stream);

/* ---- full sorts as one dense product (csrc/kge_fullsort_gemm.hip) ------------------------------------------------------
 * The full_sort_predict / full_sort_predict_kg of the reference expand the query against every candidate row
 * (transe.py:112-154, rotate.py:161-220, distmult.py:100-146, complex.py:160-219, analogy.py:66-70 + its full sorts).
 * Bilinear scorers: score = <query(h, r), tail> -- one hw_gemm_f32 of the query rows with the candidate table.  TransE and
 * hopwise's RotatE (an L2 norm over the 2 d (re, im) differences, rotate.py:61-69): ||q - t||^2 = ||q||^2 + ||t||^2 - 2 <q, t>,
 * the same product followed by hw_kge_dist_finish.
 *
 * hw_kge_query_rows: out[i, :] = the folded query of (h_idx[i], r_idx[i] or r_fixed) -- the model's query slots side by side,
 * [n, NQ * dim] with NQ = 1 (TransE: h + r; DistMult: h * r), 2 (RotatE: the rotated head (re | im); ComplEx: complex.py:54-59
 * with the reference's t_im in the last term), 3 (Analogy); ld >= NQ * dim.  sqnorm (nullable) [n] receives ||query||^2.
 * Evaluation of the distance form: hw_gemm_f32 of these rows with the candidate table and bias[j] = -||t_j||^2 / 2 yields
 * p = <q, t> - ||t||^2 / 2, which orders the candidates exactly as the score does (score = base - sqrt(||q||^2 - 2 p)): an
 * evaluation's top-k needs no finish pass.
 * Replaces the torch gathers / products / cat in front of the product.  Other models: HW_ERR_UNSUPPORTED. */
int hw_kge_query_rows(const hw_kge_desc* desc, const int64_t* h_idx, const int64_t* r_idx, int64_t r_fixed, int64_t n,
                      float* out, int64_t ld, float* sqnorm, int32_t* oob_counter, void* stream);
/* out[j] = sum_k x[j, k]^2 for the n_rows rows of x [n_rows, k] (leading dimension ld): ||t||^2 of the candidate rows. */
int hw_rows_sqnorm(const float* x, int64_t ld, int64_t n_rows, int32_t k, float* out, void* stream);
/* In place on c [m, n] (leading dimension ldc) holding <q_i, t_j>:  c[i, j] = base - sqrt(max(qn[i] + tn[j] - 2 c[i, j], 0))
 * (base = margin for RotatE, 0 for TransE).  Where the expansion cancels -- d^2 < rescore_frac * (qn[i] + tn[j]) -- the entry
 * is recomputed as sum_k (q[i, k] - t[j, k])^2 from the rows, so that every score is fp32-accurate (a few 1e-6 relative).
 * mask_column0: column 0 = -inf (trainer.py:733).  n_rescored (nullable): counter of recomputed entries.  m <= 65535. */
int hw_kge_dist_finish(float* c, int64_t ldc, int64_t m, int64_t n, const float* q, int64_t ldq, const float* t,
                       int64_t ldt, int32_t k, const float* qn, const float* tn, float base, float rescore_frac,
                       int32_t mask_column0, int32_t* n_rescored, void* stream);
/* c[i, j] = <q_i, t_j> + bias[j] (bias may be NULL) from the three largest of the six split terms (hi.hi, hi.mid, mid.hi):
 * half the matrix work of hw_gemm_f32, |error| <= 1.2e-5 * sum_k |q_ik t_jk| per entry.  For candidate generation only -- the
 * evaluation top-k of the distance scorers ranks by it and re-scores the candidates exactly (hw_topk_rerank, whose slack covers
 * this bound; rotate.py:161-220, transe.py:112-154).  q [m, k] and t [n, k] row-major, 16-byte aligned rows (ld % 4 == 0);
 * HW_ERR_UNSUPPORTED otherwise or with HW_GEMM_SPLIT=0.  Optional, on the way out of the product's epilogue:
 *   mask_bits [m, mask_ld] 32-bit words (hw_mask_bits): entry (i, j) leaves as -inf where bit j of row i is set -- the
 *             evaluation's history mask (trainer.py:731-734) without a pass of its own;
 *   tile_max  [m, tile_ld] floats: the maximum of every 64-column stretch of every (masked) row, for hw_topk_from_tile_max. */
int hw_gemm_rank(const float* q, int64_t ldq, const float* t, int64_t ldt, const float* bias, float* c, int64_t ldc, int64_t m,
                 int64_t n, int64_t k, const uint32_t* mask_bits, int64_t mask_ld, float* tile_max, int64_t tile_ld,
                 void* stream);
/* The same by-products on the FULL six-term product (fp32-level error, hw_gemm_f32's values): the evaluation step of the bilinear
 * scorers (distmult.py:50-51, complex.py:53-62, analogy.py:66-70), whose product is the score itself. */
int hw_gemm_f32_masked(const float* q, int64_t ldq, const float* t, int64_t ldt, const float* bias, float* c, int64_t ldc,
                       int64_t m, int64_t n, int64_t k, const uint32_t* mask_bits, int64_t mask_ld, float* tile_max,
                       int64_t tile_ld, void* stream);

/* hw_gemm_f32's product (q [m, k], t [n, k] row-major, 16-byte aligned rows) that also leaves, per 64-column stretch of every row,
 * the pair (maximum, sum of exp(x - maximum)) in tile_max / tile_sum [m, tile_ld], tile_ld >= ceil(n / 64).  hw_lse_rows reduces a
 * row's pairs to row_lse[2 r] = max_j x_rj, row_lse[2 r + 1] = log sum_j exp(x_rj - max): F.log_softmax(x, dim=1)[r, j] is then
 * (x_rj - row_lse[2 r]) - row_lse[2 r + 1] -- the MetaProgramExecutor's `log_softmax(scores)` over the tail vocabulary
 * (cafe.py:820-822) without the pass that rewrites the [rows, vocabulary] matrix; hw_cafe_execute_logits applies it at the
 * entries it reads. */
int hw_gemm_f32_lse(const float* q, int64_t ldq, const float* t, int64_t ldt, const float* bias, float* c, int64_t ldc, int64_t m,
                    int64_t n, int64_t k, float* tile_max, float* tile_sum, int64_t tile_ld, void* stream);
int hw_lse_rows(const float* tile_max, const float* tile_sum, int64_t tile_ld, int64_t n_rows, int64_t n_stretches,
                float* row_lse, void* stream);

/* hw_gemm_rank on operands split ONCE (csrc/gemm_rank.hip): hw_rank_planes turns an fp32 matrix [rows, k] (row pitch ld) into the
 * (hi, mid) bf16 plane images the product's DMA loads straight into LDS -- the item table once per weight version, a batch's query
 * rows once per call; hw_rank_planes_bytes sizes the buffer (16-byte aligned).  hw_gemm_rank_planes: the same c, mask and
 * tile_max as hw_gemm_rank(q, t, ...) on the matrices the images were made from, bit for bit, on 256 x 256 tiles with no
 * conversion in the main loop (the evaluation's [2 048, 400] x [45 539, 400]^T: 0.38 -> see DESIGN.md section 4). */
int64_t hw_rank_planes_bytes(int64_t rows, int64_t k);
int hw_rank_planes(const float* x, int64_t ld, int64_t rows, int64_t k, void* planes, void* stream);
int hw_gemm_rank_planes(const void* q_planes, const void* t_planes, int64_t m, int64_t n, int64_t k, const float* bias, float* c,
                        int64_t ldc, const uint32_t* mask_bits, int64_t mask_ld, float* tile_max, int64_t tile_ld, void* stream);

/* History pairs -> the bit matrix hw_gemm_rank masks with: bits [n_rows, words_per_row] (words_per_row >= hw_mask_bits_words(n_cols))
 * is zeroed, then bit (hist_rows[i], hist_cols[i]) set for every pair inside the matrix (and column 0 of every row with
 * mask_column0: the [PAD] item, trainer.py:731). */
int64_t hw_mask_bits_words(int64_t n_cols);
int hw_mask_bits(const int64_t* hist_rows, const int64_t* hist_cols, int64_t n_hist, int64_t n_rows, int64_t n_cols,
                 int32_t mask_column0, uint32_t* bits, int64_t words_per_row, void* stream);
/* hw_topk_rows (same order, values and exhausted-row convention) of a score matrix whose 64-column stretch maxima are known
 * (hw_gemm_rank's tile_max): a row is answered from its maxima and the few stretches that can hold one of its k best columns
 * instead of from all its columns.  k <= 64, n_cols <= 65 536 (else HW_ERR_UNSUPPORTED).  Rows with so many equal scores that
 * the qualifying stretches / candidates outgrow the kernel's lists set *overflow (an int32 the caller zeroed) and leave
 * (-inf, column 0) behind: the caller falls back (hw_kge_full_sort_topk_if takes the flag as its predicate). */
int hw_topk_from_tile_max(const float* scores, int64_t ld, int64_t n_rows, int64_t n_cols, const float* tile_max, int64_t tile_ld,
                          int32_t k, float* out_val, int64_t* out_idx, int32_t* overflow, void* stream);

/* The product and the finish in ONE launch: c[i, j] = base - ||q_i - t_j|| for q [m, k], t [n, k] (K-contiguous rows), with the
 * same recomputation from the rows where the expansion cancels (the lane that owns the entry sums (q_k - t_k)^2 itself).  What
 * full_sort_predict / full_sort_predict_kg of the distance scorers run (transe.py:112-154, rotate.py:161-220). */
int hw_gemm_dist(const float* q, int64_t ldq, const float* t, int64_t ldt, float* c, int64_t ldc, int64_t m, int64_t n,
                 int64_t k, const float* qn, const float* tn, float base, float rescore_frac, void* stream);
/* hw_mask_scores_ex on rows with a leading dimension (a padded score matrix). */
int hw_mask_scores_ld(float* scores, int64_t n_rows, int64_t n_cols, int64_t ld, const int64_t* hist_rows,
                      const int64_t* hist_cols, int64_t n_hist, int32_t mask_column0, void* stream);
/* Exact top-k from c <= 64 candidates per row: cand_idx [m, c] columns (top-c of an approximate score matrix; -1 = none),
 * cand_approx [m, c] their approximate scores (-inf = masked column: stays -inf), cand_exact [m, c] the same candidates
 * re-scored exactly.  Writes the k best by (exact value desc, column asc) -- the order of hw_topk_rows -- and ORs 1 into
 * *unverified when a row cannot prove that no column outside its candidate set belongs to its top-k: proven when the exact
 * k-th value > min(cand_approx) + rel_slack * |min(cand_approx)| + abs_slack (a column outside the set has an approximate
 * score <= that minimum), or when the set holds every unmasked column.
 * qn != NULL: cand_approx are values p = <q, t> - ||t||^2 / 2 of the biased product (above), qn [m] = ||q||^2, and the
 * bound is taken in the distance domain: a column outside the set has d^2 >= qn - 2 min(p) - rel_slack * (qn + tn_max)
 * (tn_max = the largest ||t||^2 of the table), i.e. a score <= base - sqrt(that). */
int hw_topk_rerank(const int64_t* cand_idx, const float* cand_approx, const float* cand_exact, int64_t m, int32_t c,
                   int32_t k, float rel_slack, float abs_slack, const float* qn, float base, float tn_max, float* out_val,
                   int64_t* out_idx, int32_t* unverified, void* stream);
/* The same with the proof made for a PRODUCT score (the bilinear scorers: score = <q, t>, cand_approx from hw_gemm_rank): a
 * column outside the set has an approximate product <= the set's smallest one, and the approximation errs by at most
 * rel_slack * ||q|| * max ||t|| (qn[i] = ||q_i||^2, tn_max = max_j ||t_j||^2). */
int hw_topk_rerank_dot(const int64_t* cand_idx, const float* cand_approx, const float* cand_exact, int64_t m, int32_t c,
                       int32_t k, float rel_slack, float abs_slack, const float* qn, float tn_max, float* out_val,
                       int64_t* out_idx, int32_t* unverified, void* stream);

/* Backward of hw_kge_score_grouped: given g[b, j] = dLoss/dscore[b, j] (same strides as `out`),
 * atomically accumulates the row gradients into `grads`.  Replaces autograd through
 * nn.Embedding + <Model>.forward (dense embedding_dense_backward in the reference).
 * rel_workspace (nullable): hw_kge_grad_grouped_workspace_floats(...) floats of scratch.  With r_idx == NULL it keeps
 * the fixed relation row from being hammered by every wave (the adds go to 64 replica rows, summed afterwards); for
 * HW_RESCAL it holds dL/dQ per task so that the head and d x d matrix gradients are formed by two matrix kernels
 * instead of d^2 atomics per group.  NULL selects the plain atomic paths (same results up to float add order). */
int64_t hw_kge_grad_grouped_workspace_floats(const hw_kge_desc* desc, int64_t n_groups, int64_t n_tail_per_group,
                                             int32_t relation_is_fixed);
int hw_kge_grad_grouped(const hw_kge_desc* desc, const hw_kge_grads* grads,
                        const int64_t* h_idx, const int64_t* r_idx, int64_t r_fixed,
                        const int64_t* t_idx, int64_t t_stride_b, int64_t t_stride_j,
                        int64_t n_groups, int64_t n_tail_per_group,
                        const float* g, int64_t g_stride_b, int64_t g_stride_j,
                        float* rel_workspace, void* stream);

/* ---- TransR projections (transr.py:67-70) ---------------------------------------------------------------------------
 * out[i, :] = M[r_i] e_i with M[r] = mats[r] viewed [dim, dim] row-major (proj_mat_e rows), r_i = r_idx[i] or r_fixed,
 * e_i = table[e_idx[i]].  The projected rows are then scored with the HW_TRANSE kernels (out buffers as tables). */
int hw_kge_project_rows(const float* mats, int64_t n_rel, const float* table, int64_t n_table_rows, int32_t dim,
                        const int64_t* r_idx, int64_t r_fixed, const int64_t* e_idx, int64_t n, float* out,
                        void* stream);
/* g_table[e_idx[i], :] += M[r_i]^T g_out[i, :];  g_mats[r_i] += g_out[i, :] (x) e_i.  Either destination may be NULL.
 * g_mats needs dim <= 256. */
int hw_kge_project_rows_grad(const float* mats, int64_t n_rel, const float* table, int64_t n_table_rows, int32_t dim,
                             const int64_t* r_idx, int64_t r_fixed, const int64_t* e_idx, int64_t n, const float* g_out,
                             float* g_table, float* g_mats, void* stream);

/* ---- single-input-channel 2-D convolution (ConvE conve.py:63, ConvKB convkb.py:52) --------------------------------
 * out[b, c, i, j] = bias[c] + sum_{u, v} w[c, u, v] x[b, i + u, j + v]; stride 1, no padding, kh * kw <= 32.
 * x [batch, height, width], w [channels, kh, kw], out / dy [batch, channels, height - kh + 1, width - kw + 1].
 * backward: dx / dw / dbias nullable; workspace of hw_conv2d_1ch_workspace_floats() floats is needed for dw. */
int64_t hw_conv2d_1ch_workspace_floats(int32_t channels, int32_t kh, int32_t kw);
int hw_conv2d_1ch_forward(const float* x, const float* w, const float* bias, int64_t batch, int32_t channels,
                          int32_t height, int32_t width, int32_t kh, int32_t kw, float* out, void* stream);
int hw_conv2d_1ch_backward(const float* x, const float* w, const float* dy, int64_t batch, int32_t channels,
                           int32_t height, int32_t width, int32_t kh, int32_t kw, float* dx, float* dw, float* dbias,
                           float* workspace, void* stream);

/* ---- losses on score vectors (deterministic two-stage reductions) ---------------------------------
 * `workspace` must hold at least hw_loss_workspace_floats() floats. */
int64_t hw_loss_workspace_floats(void);

/* torch.nn.BCEWithLogitsLoss(mean) over x[0..n): label 1 for i < n_pos, else 0 (rotate.py:118-131,
 * complex.py:115-128).  loss_out[0] = loss * loss_scale...; gx (nullable) receives dLoss/dx. */
int hw_loss_bce_logits(const float* x, int64_t n, int64_t n_pos, float* loss_out, float* gx,
                       float* workspace, void* stream);
/* Same with every positive counted pos_weight times: loss = (pos_weight * sum_pos + sum_neg) / (pos_weight * n_pos +
 * n - n_pos).  The reference's rec batch tiles each positive sample_num times next to its negatives
 * (abstract_dataloader.py:192-198); scoring each positive once with pos_weight = sample_num gives the same mean. */
int hw_loss_bce_logits_weighted(const float* x, int64_t n, int64_t n_pos, float pos_weight, float* loss_out, float* gx,
                                float* workspace, void* stream);

/* mean(max(0, -(pos - neg) + margin)): torch MarginRankingLoss(target=+1) (distmult.py:86-93) and,
 * on scores = -distance, TripletMarginLoss (transe.py:48,96).  gpos/gneg nullable. */
int hw_loss_margin_ranking(const float* pos, const float* neg, int64_t n, float margin,
                           float* loss_out, float* gpos, float* gneg, float* workspace, void* stream);
/* The same on sigmoid(pos), sigmoid(neg): HolE's MarginRankingLoss(sigmoid(.), sigmoid(.), 1) (hole.py:100-104);
 * gpos / gneg include the sigmoid's derivative. */
int hw_loss_margin_ranking_sigmoid(const float* pos, const float* neg, int64_t n, float margin,
                           float* loss_out, float* gpos, float* gneg, float* workspace, void* stream);

/* Self-adversarial negative weighting (RotatE paper; NOT in the reference -- SURVEY.md §8 A9, opt-in):
 * scores [n_groups, 1+K] row-major, column 0 positive.  loss = mean_b( -logsig(s_pos)
 * - sum_i softmax(alpha*s_neg)_i * logsig(-s_neg_i) ), weights detached.  gscores nullable. */
int hw_loss_self_adversarial(const float* scores, int64_t n_groups, int64_t n_cols, float alpha,
                             float* loss_out, float* gscores, float* workspace, void* stream);

/* ---- evaluation: mask + top-k ---------------------------------------------------------------------
 * In-place masking of a [n_rows, n_cols] score matrix the way KGTrainer._full_sort_batch_eval does
 * (trainer/trainer.py:731-734): column 0 and every (hist_rows[i], hist_cols[i]) := -inf. */
int hw_mask_scores(float* scores, int64_t n_rows, int64_t n_cols,
                   const int64_t* hist_rows, const int64_t* hist_cols, int64_t n_hist, void* stream);
/* The same with the column-0 rule optional: a rank of the row-sharded layout that does not own entity 0 ([PAD]) masks its
 * history entries only (mask_column0 = 0).  History entries outside [0, n_rows) x [0, n_cols) are ignored in both. */
int hw_mask_scores_ex(float* scores, int64_t n_rows, int64_t n_cols,
                      const int64_t* hist_rows, const int64_t* hist_cols, int64_t n_hist,
                      int32_t mask_column0, void* stream);

/* Row-wise top-k ordered (value desc, index asc) -- replaces torch.topk in
 * evaluator/collector.py:173-183 and pgpr.py:454.  k <= 1024. */
int hw_topk_rows(const float* scores, int64_t n_rows, int64_t n_cols, int64_t ld, int32_t k,
                 float* out_val, int64_t* out_idx, void* stream);

/* The same result (bit for bit, ties included) for long rows in two stages: the row is cut into chunks of 4096 columns,
 * each read once and reduced to its k best, then the k best of those.  Taken when n_cols >= 1024 and k <= 64; otherwise the
 * call is hw_topk_rows and needs no workspace (hw_topk_rows_workspace_bytes returns 0).  workspace: 8-byte aligned,
 * n_rows * ceil(n_cols / 4096) * k * 8 bytes; HW_ERR_WORKSPACE when smaller. */
int64_t hw_topk_rows_workspace_bytes(int64_t n_rows, int64_t n_cols, int32_t k);
int hw_topk_rows_ws(const float* scores, int64_t n_rows, int64_t n_cols, int64_t ld, int32_t k,
                    float* out_val, int64_t* out_idx, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- negative sampling ----------------------------------------------------------------------------
 * Device restatement of AbstractSampler.sample_by_key_ids (sampler/sampler.py:140-183) for
 * KGSampler (:294-357) and Sampler (:186-291): for key i and draw j, the value written to
 * out[i*out_stride_key + j*out_stride_draw] is uniform over [1, value_num) minus used(keys[i]), by rejection.
 * (out_stride_key, out_stride_draw) = (1, n_keys) is the reference's layout value_ids[j*n_keys + i];
 * (num, 1) gives a row per key, which the grouped scorer reads with coalesced id loads.  used sets are a CSR (used_ptr[n_key_space+1],
 * used_idx sorted ascending inside each row).  Philox4x32-10 counter RNG keyed by (seed, offset).
 * alias_prob/alias_keys/alias_alias (nullable) switch to the popularity distribution (:68-116). */
int hw_neg_sample(const int64_t* used_ptr, const int64_t* used_idx, int64_t n_key_space,
                  const int64_t* keys, int64_t n_keys, int32_t num, int64_t value_num,
                  const float* alias_prob, const int64_t* alias_keys, const int64_t* alias_alias,
                  int64_t n_alias, uint64_t seed, uint64_t offset,
                  int64_t* out, int64_t out_stride_key, int64_t out_stride_draw,
                  int32_t* oob_counter, void* stream);

/* ---- PGPR beam-search traversal --------------------------------------------------------------------
 * Device restatement of model/knowledge_aware_recommender/pgpr.py beam_search (:440-484) and the
 * per-path primitives it calls.  Node types: 0 = user, 1 = entity; the self-loop relation is -1.
 * A beam row ("path") is HW_PGPR_ROW_INTS int32: [uid, n_nodes, (rel, type, id) x 3, batch index]; its hop
 * probabilities are 3 floats.  Row counts live in device memory (int32 *n_rows) so that a whole
 * multi-hop search is enqueued without a host round trip; `max_rows` bounds every launch. */
#define HW_PGPR_ROW_INTS 12
#define HW_PGPR_MAX_DIM 512

/* Collaborative KG as one CSR over node gid = (type == user ? id : n_users + id), rows sorted by
 * (relation, dst) with multiplicity -- what dataset.ckg_dict_graph() holds (kg_dataset.py:1044-1095). */
typedef struct {
    const int64_t* ptr;  /* [n_users + n_entities + 1] */
    const int32_t* rel;  /* [nnz] */
    const int32_t* dst;  /* [nnz] */
    int32_t n_users, n_entities, n_items, ui_relation;
    int32_t max_degree;  /* largest row length (sizes the trim scratch) */
} hw_pgpr_graph;

/* reset(): one single-node path per user (pgpr.py:215-221). */
int hw_pgpr_init_paths(const int64_t* users, int64_t n_users_in_batch, int32_t* paths, float* probs,
                       int32_t* n_rows, void* stream);

/* Float64 score tables for the top-max_acts trim of _get_actions (pgpr.py:263-284).  The reference scores a
 * candidate (r, n) of a path rooted at user u as u.U[n], (u + r_ui).E[n] or (u + r_ui + r).E[n]; these are
 *   score_user[b, n] = U[users[b]] . U[n],  score_entity[b, n] = (U[users[b]] + R[ui]) . E[n]    (per search batch)
 *   score_rel_entity[r, n] = R[r] . E[n]                                                          (once per model)
 * so that a candidate costs one or two lookups.  src_workspace: 2 * n_batch * dim doubles. */
int hw_pgpr_rel_entity_scores(const double* relation_emb64, const double* entity_emb64, int64_t n_relations,
                              int64_t n_entities, int32_t dim, double* out, void* stream);
int hw_pgpr_user_scores(const hw_pgpr_graph* graph, const double* user_emb64, const double* entity_emb64,
                        const double* relation_emb64, int32_t dim, const int64_t* users, int64_t n_batch,
                        double* src_workspace, double* score_user, double* score_entity, void* stream);
/* TPRec (tprec.py:454-471): the same tables with the [UI-Relation] row replaced per user by user_vec64[u]
 * ([n_users, dim] float64: sum_c uc_weight[u][c] * ui_clust_relation_embedding[c]). */
int hw_pgpr_user_scores_tp(const hw_pgpr_graph* graph, const double* user_emb64, const double* entity_emb64,
                           const double* relation_emb64, const double* user_vec64, int32_t dim, const int64_t* users,
                           int64_t n_batch, double* src_workspace, double* score_user, double* score_entity,
                           void* stream);

/* _get_actions (pgpr.py:226-290): acts[row] = [(self, cur)] + unvisited neighbours in (rel, dst) order; when
 * more than max_acts remain, the max_acts with the largest float64 score (ties: later CSR position wins),
 * re-sorted by (rel, dst) or, with fix_scores_sorting_bug, in descending score order.  The row's batch index
 * (path record int 11) selects its rows of score_user [n_batch, n_users] / score_entity [n_batch, n_entities].
 * acts: int32 [max_rows, max_acts + 1, 2]; n_acts: int32 [max_rows].
 * scratch: uint64 [scratch_blocks * scratch_stride], needed when graph->max_degree > 4096
 * (scratch_stride >= max_degree).
 * fix_scores_sorting_bug is a bit set: bit 0 = config fix_scores_sorting_bug; bit 1 = the start user is NOT in the
 * visited set -- the reference's training episodes keep the user id as a 0-dim tensor in path[0] (reset(), pgpr.py:215-216),
 * whose hash never matches an int (the tensor also rides along leading self-loop steps), so only calculate_loss lets
 * a path return to its own user; bit 2 = `acts` is written as uint16 EDGE OFFSETS into the current node's CSR slice
 * ([max_rows, HW_PGPR_ACT16_LD(max_acts + 1)] uint16, 0xFFFF = the self loop) instead of (relation, dst) pairs: an action
 * is a neighbour and the CSR already holds its relation and id, so the beam search moves 2 bytes per action instead of 8;
 * only hw_pgpr_topk_expand (acts_u16) reads this layout.  Not with bit 0, nor when graph->max_degree > 8192. */
#define HW_PGPR_FIX_SORTING 1
#define HW_PGPR_START_USER_REVISITABLE 2
#define HW_PGPR_ACTS_U16 4
#define HW_PGPR_ACT16_LD(act_dim) (((act_dim) + 7) & ~7)
/* work (nullable): hw_pgpr_actions_workspace_ints(max_rows) int32 of scratch in which the first pass queues the rows
 * that need the trim, by degree class, for the trim kernels; without it they scan every row. */
int64_t hw_pgpr_actions_workspace_ints(int64_t max_rows);
int hw_pgpr_actions(const hw_pgpr_graph* graph, const double* score_user, const double* score_entity,
                    const double* score_rel_entity, const int32_t* paths, const int32_t* n_rows,
                    int64_t max_rows, int32_t max_acts, int32_t fix_scores_sorting_bug,
                    int32_t* acts, int32_t* n_acts, uint64_t* scratch, int64_t scratch_blocks,
                    int64_t scratch_stride, int32_t* work, int64_t work_ints, void* stream);

/* _get_state (pgpr.py:296-333): state [max_rows, (2 + 2*state_history) * dim] float32. */
int hw_pgpr_state(const hw_pgpr_graph* graph, const float* user_emb32, const float* entity_emb32,
                  const float* relation_emb32, int32_t dim, int32_t state_history, const int32_t* paths,
                  const int32_t* n_rows, int64_t max_rows, float* state, void* stream);

/* One Linear (+ ELU) (+ dropout) layer of the policy network (pgpr.py:175-187) on fp32 MFMA:
 * c[m, n] = f(a[m, :] . w[n, :] + bias[n]).  m_rows (nullable) = device row count, m_max bounds the grid.
 * drop_mode 0: none; 1: multiply by drop_mask[m, n]; 2: Philox keep-prob 1/2, kept values x2. */
int hw_mlp_layer(const float* a, int64_t lda, const float* w, const float* bias, float* c, int64_t ldc,
                 const int32_t* m_rows, int64_t m_max, int64_t n, int64_t k, int32_t elu,
                 const float* drop_mask, int32_t drop_mode, uint64_t seed, uint64_t offset, void* stream);

/* First policy layer by table lookup (beam search, fixed weights): out[row, :] = dropout(ELU(bias + sum over the state's
 * slots of T_slot[node or relation of that slot, :])) with T_slot = table . W1[:, slot * dim : (slot + 1) * dim]^T
 * precomputed by the caller ([rows, hidden] each, 16-byte aligned).  slot_tables (HOST array of device pointers):
 * {user slot over users, current node over users, over entities, last node over users, over entities, last relation over
 * relations, older node over users, over entities, older relation}: 3 / 6 / 9 entries for state_history 0 / 1 / 2.
 * Slot semantics = hw_pgpr_state (pgpr.py:296-333); dropout modes / Philox words = hw_mlp_layer. */
int hw_pgpr_layer1_lookup(const float* const* slot_tables, const float* bias, int32_t hidden, int32_t state_history,
                          const int32_t* paths, const int32_t* n_rows, int64_t max_rows, const float* drop_mask,
                          int32_t drop_mode, uint64_t seed, uint64_t offset, float* out, int64_t ld, void* stream);

/* The whole policy network of one beam-search hop in one launch (pgpr.py:175-187 over the state of :296-333; beam search,
 * fixed weights): logits[row, :] = actor(dropout(ELU(l2(dropout(ELU(l1(state[row]))))))) with the first layer in
 * hw_pgpr_layer1_lookup's table form and both matrix products on the bf16 matrix pipes as three-way splits (fp32-level
 * error); x1 / x2 never leave the registers of the wave that owns the row.  hidden2 must be 256, hidden1 a multiple of 128
 * (<= 1024), act_dim <= 256 (hopwise's PGPR.yaml: [512, 256], max_acts 250); other shapes: HW_ERR_UNSUPPORTED, callers use
 * the three-launch form.
 * hw_pgpr_policy_prepare: l2.weight [256, hidden1] and actor.weight [act_dim, 256] -> `planes`
 * (hw_pgpr_policy_planes_bytes(hidden1) bytes, 16-byte aligned), once per weight version.
 * hw_pgpr_policy_fused: slot_tables as hw_pgpr_layer1_lookup; zero_row = hidden1 zeros; ba_padded = actor.bias padded to
 * 256 floats; logits [max_rows, ld], ld >= 256, ld % 4 == 0 (columns act_dim..255 receive 0).
 * drop_mode 0: none; 1: multiply by mask1[row, hidden1] / mask2[row, 256]; 3: Philox keep-prob 1/2, kept values x2, ONE
 * Philox4x32-10 block per 128 elements: element i of the [rows, H] activation keeps iff bit (i & 31) of word (i >> 5) & 3
 * of block i >> 7 of stream (seed, offset1 for the first layer's output, offset2 for the second's) is set
 * (oracle/philox.py:keep_bits_packed). */
int64_t hw_pgpr_policy_planes_bytes(int32_t hidden1);
int hw_pgpr_policy_prepare(const float* w2, int32_t hidden1, int32_t hidden2, const float* wa, int32_t act_dim, void* planes,
                           void* stream);
int hw_pgpr_policy_fused(const float* const* slot_tables, const float* zero_row, const float* b1, int32_t hidden1,
                         const void* planes, const float* b2, const float* ba_padded, int32_t act_dim,
                         int32_t state_history, const int32_t* paths, const int32_t* n_rows, int64_t max_rows,
                         int32_t drop_mode, const float* mask1, const float* mask2, uint64_t seed, uint64_t offset1,
                         uint64_t offset2, float* logits, int64_t ld, void* stream);

/* ---- Philox offsets that live on the device -----------------------------------------------------------------------------
 * Every `offset` of a (seed, offset) pair below may be INDIRECT: bit 63 set, bits 0..47 the address of a uint64 in device
 * memory, bits 48..62 a small addend; the stream offset is then *address + addend, read when the kernel runs.  A training
 * step captured in a HIP graph freezes its kernel arguments; with indirect offsets every replay still draws fresh dropout
 * masks / actions once the graph's last node has bumped the word (hopwise_amd/philox.py, driver.GraphedTrainStep) -- the
 * reference draws from torch's generator, whose Philox offset CUDA graphs advance the same way.  Accepted by hw_dropout,
 * hw_gemm_f32 / hw_pgpr_mlp_layer (drop_mode 2), hw_pgpr_layer1_lookup, hw_pgpr_action_dropout, hw_pgpr_policy_sample,
 * hw_pgpr_act_dropout_backward, the hw_tucker_contract_* family and hw_neg_sample.  HW_ERR_UNSUPPORTED from the entry
 * points that do not resolve it: hw_kge_score_grouped_sampled*, hw_pgpr_policy_fused (evaluation paths). */

/* General fp32 MFMA GEMM: c[m, n] = f(op(a)[m, :] . op(b)[:, n] + bias[n]).  a is [m, k] row-major when
 * a_k_contiguous, else stored transposed [k, m]; b is [n, k] row-major (nn.Linear weight) when b_k_contiguous, else
 * [k, n].  Replaces torch.mm / torch.bmm / nn.Linear on the dense paths (tucker.py:74-82, pgpr.py:175-187).
 * workspace (nullable, workspace_floats >= 2 * m_max * n to be useful) enables deterministic split-K for
 * long-K / small-output products (partial slabs summed in fixed order).  `elu` selects the activation:
 * 0 none, 1 ELU, 2 ReLU.
 * Numerics: fp32 in, fp32 accumulate.  With 16-byte aligned operands (base and leading dimension) each fp32 product is
 * evaluated as six bf16 x bf16 products of the exact three-way bf16 splits of its factors on the bf16 matrix pipes; the
 * error against a float64 product is no larger than that of a chain of fp32 FMAs (<= 3e-6 of sum_k |a_ik b_kj|,
 * tests/test_gpu_gemm_split.py).  The environment variable HW_GEMM_SPLIT=0 selects fp32-input MFMA FMA chains instead
 * (also what unaligned operands get).  An infinite operand yields NaN on the split path. */
int hw_gemm_f32(const float* a, int64_t lda, int32_t a_k_contiguous, const float* b, int64_t ldb,
                int32_t b_k_contiguous, float* c, int64_t ldc, const int32_t* m_rows, int64_t m_max,
                int64_t n, int64_t k, const float* bias, int32_t elu, const float* drop_mask,
                int32_t drop_mode, uint64_t seed, uint64_t offset, float* workspace,
                int64_t workspace_floats, void* stream);

/* ---- TuckER (model/knowledge_graph_embedding_recommender/tucker.py:69-127) --------------------------------
 * The dense products run on hw_gemm_f32; these are the two memory-bound pieces around them.
 * hw_tucker_contract: out[b, j] = sum_i x[b, i] * w[b, i, j] * mask[b, i, j]  (torch.bmm(x, hidden_dropout1(W_mat)),
 * tucker.py:76-79); mask nullable (eval), holding 0 or 1/(1-p).  *_grad: gx (nullable) and gw (nullable) of it. */
int hw_tucker_contract(const float* x, const float* w, const float* mask, int64_t batch, int32_t dim,
                       float* out, void* stream);
int hw_tucker_contract_grad(const float* x, const float* w, const float* mask, const float* gout,
                            int64_t batch, int32_t dim, float* gx, float* gw, void* stream);
/* The same with hidden_dropout1 drawn on the fly (no [B, d, d] mask tensor): element (b, i, j) is kept with probability
 * 1 - drop_p (scaled by 1 / (1 - drop_p)) from word (i & 3) of Philox block (b * ceil(d / 4) + (i >> 2)) * d + j of the
 * (seed, offset) stream; the gradient call regenerates the mask from the same (seed, offset). */
int hw_tucker_contract_dropout(const float* x, const float* w, int64_t batch, int32_t dim, float drop_p, uint64_t seed,
                               uint64_t offset, float* out, void* stream);
int hw_tucker_contract_dropout_grad(const float* x, const float* w, const float* gout, int64_t batch, int32_t dim,
                                    float drop_p, uint64_t seed, uint64_t offset, float* gx, float* gw, void* stream);

/* mm(r, W.view(d, d^2)) (tucker.py:74) has as many DISTINCT rows as the batch has distinct relations: one on the rec side
 * (every row carries [UI-Relation], tucker.py:100-101 / :133 / :158), a few dozen on the kg side.  These entry points
 * take the small set w [n_mats, d, d] (= mm(R_distinct, W)) and w_index [batch] (matrix of each row; NULL = every row
 * uses matrix 0) instead of the [B, d, d] tensor.  The dropout mask still differs per batch row: `mask` [batch, d, d]
 * (nullable; 0 or 1/(1-p)) or, when mask == NULL and drop_p > 0, the Philox words of hw_tucker_contract_dropout.
 * *_grad: gx [batch, d] (nullable) and gw [n_mats, d, d] (nullable) = sum over the rows of each matrix of
 * x[b, i] gout[b, j] m[b, i, j].  With w_index, `order` = the batch rows sorted by w_index (stable) and gw is accumulated
 * with float atomics; without it the reduction runs in fixed order through `workspace`
 * (hw_tucker_contract_shared_workspace_floats(batch, dim) floats). */
int hw_tucker_contract_shared(const float* x, const float* w, const int64_t* w_index, const float* mask, int64_t batch,
                              int32_t dim, float drop_p, uint64_t seed, uint64_t offset, float* out, void* stream);
int64_t hw_tucker_contract_shared_workspace_floats(int64_t batch, int32_t dim);
int hw_tucker_contract_shared_grad(const float* x, const float* w, const int64_t* w_index, const int64_t* order,
                                   int64_t n_mats, const float* mask, const float* gout, int64_t batch, int32_t dim,
                                   float drop_p, uint64_t seed, uint64_t offset, float* gx, float* gw, float* workspace,
                                   int64_t workspace_floats, void* stream);
/* The Philox form with the keep flags kept: the [batch, d, d] dropout of tucker.py:76 draws 4 x 10^7 Philox blocks per side
 * and call, and the gradient call would draw them three more times (gx, gw and, on the kg side, the per-relation gw).
 * *_keep writes the flags of the forward (`keep`: hw_tucker_keep_words(batch, dim) uint32 words -- 4 bits per (quad of rows,
 * column), 1.4 bytes per 16 mask elements) and *_grad_keep reads them.  Same masks, same results as the (seed, offset)
 * entry points bit for bit. */
int64_t hw_tucker_keep_words(int64_t batch, int32_t dim);
int hw_tucker_contract_shared_keep(const float* x, const float* w, const int64_t* w_index, int64_t batch, int32_t dim,
                                   float drop_p, uint64_t seed, uint64_t offset, float* out, uint32_t* keep, void* stream);
int hw_tucker_contract_shared_grad_keep(const float* x, const float* w, const int64_t* w_index, const int64_t* order,
                                        int64_t n_mats, const uint32_t* keep, const float* gout, int64_t batch, int32_t dim,
                                        float drop_p, float* gx, float* gw, float* workspace, int64_t workspace_floats,
                                        void* stream);

/* sigmoid + nn.BCELoss(mean) against target[b, c] = target_scale * pos_flag[c] + target_add (tucker.py:111-125:
 * the union-of-the-batch multi-hot with label smoothing).  pred / glogits nullable.  workspace:
 * hw_tucker_bce_workspace_floats() floats. */
int64_t hw_tucker_bce_workspace_floats(void);
int hw_tucker_sigmoid_bce(const float* logits, int64_t n_rows, int64_t n_cols, const uint8_t* pos_flag,
                          float target_scale, float target_add, float* loss_out, float* pred,
                          float* glogits, float* workspace, void* stream);
/* Same over a score matrix stored with leading dimension ld >= n_cols (pred / glogits share the layout; glogits may alias
 * logits).  The engine keeps the [B, N] 1-vs-all scores with ld rounded up to 32 floats: with N = 144 779 the unpadded
 * rows start at arbitrary 4-byte offsets, the GEMM's tile stores straddle cache lines and its K-contiguous reads of
 * the backward fall back to dword loads (75 -> 90, 59 -> 72, 66 -> 74 TFLOP/s on the three products, tools/gemm_ld_pad.py). */
int hw_tucker_sigmoid_bce_ld(const float* logits, int64_t ld, int64_t n_rows, int64_t n_cols, const uint8_t* pos_flag,
                             float target_scale, float target_add, float* loss_out, float* pred, float* glogits,
                             float* workspace, void* stream);
/* The same loss with the score product itself: g_out[b, c] = dLoss/dscore of score[b, c] = x[b, :] . table[c, :] + bias[c]
 * (x [m, k] and table [n, k] K-contiguous, bias nullable), loss_out[0] = the mean BCE.  The scores are never written: the
 * MFMA GEMM's epilogue turns each tile into its gradient and sums the tile's loss terms (deterministic two-stage sum).
 * workspace: hw_gemm_bce_workspace_floats(m, n) floats.  ldg >= n (the engine pads it to 32 floats). */
int64_t hw_gemm_bce_workspace_floats(int64_t m, int64_t n);
int hw_gemm_f32_bce(const float* x, int64_t ldx, const float* table, int64_t ldt, const float* bias, float* g_out,
                    int64_t ldg, int64_t m, int64_t n, int64_t k, const uint8_t* pos_flag, float target_scale,
                    float target_add, float* loss_out, float* workspace, int64_t workspace_floats, void* stream);
int hw_sigmoid(const float* x, int64_t n, float* y, void* stream);

/* logits -> masked softmax -> (+1 on valid actions, pgpr.py:453) -> top-k (value desc, index asc).
 * out_idx/out_val [max_rows, k]; entries past min(k, n_acts[row]) are idx = -1.  probs_out (nullable)
 * receives the plain softmax [max_rows, act_dim]. */
int hw_pgpr_policy_topk(const float* logits, int64_t ld, int32_t act_dim, const int32_t* n_acts,
                        const int32_t* n_rows, int64_t max_rows, int32_t k, int32_t* out_idx, float* out_val,
                        float* probs_out, void* stream);

/* Emit the children of every row in row order, top-k order inside a row (pgpr.py:460-479).
 * offsets: int32 [max_rows] workspace; block_sums_ws: int32 [ceil(max_rows / 1024)] workspace;
 * n_rows_out[0] = number of children (capped at out_capacity). */
int hw_pgpr_expand(const hw_pgpr_graph* graph, const int32_t* paths, const float* probs,
                   const int32_t* acts, int32_t act_dim, const int32_t* n_acts, const int32_t* n_rows,
                   int64_t max_rows, const int32_t* topk_idx, const float* topk_val, int32_t k,
                   int32_t* offsets, int32_t* block_sums_ws, int32_t* paths_out, float* probs_out,
                   int32_t* n_rows_out, int64_t out_capacity, void* stream);

/* The tail of a beam-search hop in ONE launch: masked softmax -> (+1 on valid actions, pgpr.py:453) -> top-k (value desc,
 * index asc, pgpr.py:454) -> children emitted in row order, top-k order inside a row (pgpr.py:460-479); what
 * hw_pgpr_policy_topk + hw_pgpr_expand do in two, with identical outputs, without the [max_rows, k] index / value arrays.
 * act_dim <= 256, k <= 64 (one wave ranks a row in registers); other shapes use the two calls above.
 * leaf_out == NULL: children as 12-int records + probability triples in paths_out / probs_out [out_capacity].
 * leaf_out != NULL (int32 [out_capacity, 4], 16-byte aligned; the LAST hop of a search): children as compact leaves
 * {parent row, relation, node id, bits of the hop's value} -- the 48 + 12 bytes of a full record repeat the parent's prefix
 * once per child; hw_pgpr_materialize_leaves rebuilds records, hw_pgpr_collect_leaves consumes the compact form.
 * acts_u16 != 0: `acts` holds uint16 edge offsets as hw_pgpr_actions writes them with flag bit 2 (see there).
 * Workspaces as hw_pgpr_expand. */
int hw_pgpr_topk_expand(const hw_pgpr_graph* graph, const int32_t* paths, const float* probs, const int32_t* acts,
                        int32_t act_dim, const int32_t* n_acts, const int32_t* n_rows, int64_t max_rows,
                        const float* logits, int64_t ld, int32_t k, int32_t* offsets, int32_t* block_sums_ws,
                        int32_t* paths_out, float* probs_out, int32_t* leaf_out, int32_t* n_rows_out,
                        int64_t out_capacity, int32_t acts_u16, void* stream);

/* Compact leaves -> full records (the path_pool / probs_pool entries of pgpr.py:464-478): row t of paths_out / probs_out =
 * leaf sel[t] (sel int32 [n_sel]; negative or out-of-range entries give zero rows), or every leaf in order when sel is NULL
 * (then n_leaf[0] rows are written, at most max_leaf). */
int hw_pgpr_materialize_leaves(const hw_pgpr_graph* graph, const int32_t* parent_paths, const float* parent_probs,
                               const int32_t* leaves, const int32_t* n_leaf, int64_t max_leaf, const int32_t* sel,
                               int64_t n_sel, int32_t* paths_out, float* probs_out, void* stream);

/* collect_scores (pgpr.py:486-552) on the leaf rows of a search over `users[n_batch]`:
 * results [n_batch, n_items] float32 (-inf where nothing is written), plus per user slot the written
 * (item, float64 score, leaf row of the attached path) triples in writing order: sel_* [n_batch, kmax],
 * sel_count [n_batch], sel_row [n_batch] (row of `results` the slot wrote to, -1 if none).
 * positives_bits: bitset [n_users, positives_words] of dataset.history_item_matrix()[0] rows.
 * Workspaces: best_ws uint64 [n_batch * n_items], score_ws double [n_batch * n_items], first_idx_ws int32 [n_batch]. */
int hw_pgpr_collect(const hw_pgpr_graph* graph, const double* user_emb64, const double* entity_emb64,
                    const double* relation_emb64, int32_t dim, const int64_t* users, int64_t n_batch,
                    const uint32_t* positives_bits, int64_t positives_words, const int32_t* leaf_paths,
                    const float* leaf_probs, const int32_t* n_leaf, int64_t max_leaf, int32_t n_nodes,
                    int32_t kmax, uint64_t* best_ws, double* score_ws, int32_t* first_idx_ws, float* results,
                    int32_t* sel_item, double* sel_score, int32_t* sel_leaf, int32_t* sel_count,
                    int32_t* sel_row, void* stream);
/* The same on compact leaves: parent_paths / parent_probs are the rows the last hop expanded, leaves int32 [max_leaf, 4]
 * as hw_pgpr_topk_expand wrote them; sel_leaf indexes `leaves`.  Bit-identical results to hw_pgpr_collect on the
 * materialised records. */
int hw_pgpr_collect_leaves(const hw_pgpr_graph* graph, const double* user_emb64, const double* entity_emb64,
                           const double* relation_emb64, int32_t dim, const int64_t* users, int64_t n_batch,
                           const uint32_t* positives_bits, int64_t positives_words, const int32_t* parent_paths,
                           const float* parent_probs, const int32_t* leaves, const int32_t* n_leaf, int64_t max_leaf,
                           int32_t n_nodes, int32_t kmax, uint64_t* best_ws, double* score_ws, int32_t* first_idx_ws,
                           float* results, int32_t* sel_item, double* sel_score, int32_t* sel_leaf, int32_t* sel_count,
                           int32_t* sel_row, void* stream);
/* The same with the parents' child ranges at hand: child_offsets int32 [max_parents] = first leaf of every parent row (the
 * `offsets` hw_pgpr_topk_expand filled for the last hop; at most 64 children per parent), n_parents = device count of
 * parent rows.  The best-path pass then takes a wave per parent (the parent through the scalar unit, a lane per child)
 * instead of a thread per leaf that re-reads its parent; same table bit for bit.  Either pointer NULL: the call above. */
int hw_pgpr_collect_leaves_ex(const hw_pgpr_graph* graph, const double* user_emb64, const double* entity_emb64,
                              const double* relation_emb64, int32_t dim, const int64_t* users, int64_t n_batch,
                              const uint32_t* positives_bits, int64_t positives_words, const int32_t* parent_paths,
                              const float* parent_probs, const int32_t* leaves, const int32_t* n_leaf, int64_t max_leaf,
                              int32_t n_nodes, int32_t kmax, uint64_t* best_ws, double* score_ws, int32_t* first_idx_ws,
                              float* results, int32_t* sel_item, double* sel_score, int32_t* sel_leaf,
                              int32_t* sel_count, int32_t* sel_row, const int32_t* child_offsets,
                              const int32_t* n_parents, int64_t max_parents, void* stream);

/* ---- PGPR policy-gradient training (pgpr.py:120-203, 335-396) --------------------------------------------------
 * One episode = max_path_len steps of: actions -> action dropout -> state -> policy MLP -> Categorical draw ->
 * step -> reward; then update().  The policy head is ONE matrix [act_dim + 1, hidden]: rows 0..act_dim-1 the actor,
 * the last row the critic, so `head` rows are [act_dim logits, value] with leading dimension ld >= act_dim + 1. */

/* batch_action_mask (pgpr.py:384-396): mask u8 [max_rows, act_dim]; rows with n_acts >= 5 keep action 0 and a uniform
 * random subset of int((n - 1) * (1 - dropout)) of the others (smallest Philox keys; key of (row, j) is word
 * row * act_dim + j of the (seed, offset) stream), other rows keep their n_acts actions. */
int hw_pgpr_action_dropout(const int32_t* n_acts, const int32_t* n_rows, int64_t max_rows, int32_t act_dim,
                           double dropout, uint64_t seed, uint64_t offset, uint8_t* mask, void* stream);

/* select_action (pgpr.py:120-135) after the MLP: probs = softmax(masked logits) [max_rows, act_dim]; acts =
 * Categorical draw by inverse CDF (u = top 24 bits of Philox word `row`) or forced_acts (nullable, replay);
 * logp / entropy with torch.distributions.Categorical's clamp_probs semantics; value = head[row, act_dim]. */
int hw_pgpr_policy_sample(const float* head, int64_t ld, int32_t act_dim, const uint8_t* mask, const int32_t* n_rows,
                          int64_t max_rows, const int32_t* forced_acts, uint64_t seed, uint64_t offset, float* probs,
                          int32_t* acts, float* logp, float* entropy, float* value, void* stream);

/* d loss / d head for loss = sum_rows(-coef_actor * logp - coef_entropy * entropy) + critic: dhead [n_rows, ld]. */
int hw_pgpr_policy_grad(const float* probs, int32_t act_dim, const int32_t* acts, const float* coef_actor,
                        float coef_entropy, const float* dvalue, int64_t n_rows, float* dhead, int64_t ld,
                        void* stream);
/* The same times upstream[0] (a device float: the gradient arriving at the scalar loss, pgpr.py:189-203 `loss.backward()`);
 * NULL = 1.  Everything behind dhead is linear in it, so the eight parameter gradients come out scaled. */
int hw_pgpr_policy_grad_scaled(const float* probs, int32_t act_dim, const int32_t* acts, const float* coef_actor,
                               float coef_entropy, const float* dvalue, int64_t n_rows, float* dhead, int64_t ld,
                               const float* upstream, void* stream);

/* dx = dy * d/dpre[dropout(act(pre))] recovered from y = dropout(act(pre)) (act 1 ELU, 2 ReLU, 0 none; drop modes as
 * hw_gemm_f32, mode 2 regenerates the Philox mask of the forward call with the same seed / offset). */
int hw_act_dropout_backward(const float* dy, const float* y, int64_t m_rows, int64_t n, int32_t act,
                            const float* drop_mask, int32_t drop_mode, uint64_t seed, uint64_t offset, float* dx,
                            void* stream);

/* out[n] = column sums of x[m_rows, n] (bias gradients), deterministic order. */
int hw_colsum_f32(const float* x, int64_t ld, int64_t m_rows, int64_t n, float* out, void* stream);
/* The same for tall matrices in two stages (row chunks of >= 32 rows, at most 256, then their partial sums in fixed order):
 * workspace of hw_colsum_workspace_floats(m_rows, n) floats; m_rows <= 32 takes the one-stage kernel. */
int64_t hw_colsum_workspace_floats(int64_t m_rows, int64_t n);
int hw_colsum_f32_ws(const float* x, int64_t ld, int64_t m_rows, int64_t n, float* out, float* workspace,
                     int64_t workspace_floats, void* stream);

/* _get_reward (pgpr.py:339-357) for paths [n_rows, HW_PGPR_ROW_INTS]: 0 unless the path has more than 2 nodes, its
 * relation sequence equals a pattern (patterns int32 [n_patterns, 4] = (n_nodes, rel, rel, rel), self loop = -1)
 * and it ends on an item; then max(score_entity[b, item] / scales[b], 0), float64.  scales_ws double [n_batch] =
 * u_p_scales (pgpr.py:85-88) = max over the n_items columns of score_entity, filled when compute_scales != 0. */
int hw_pgpr_reward(const int32_t* paths, int64_t n_rows, const int32_t* patterns, int32_t n_patterns,
                   const double* score_entity, int64_t n_entities, int64_t n_items, double* scales_ws,
                   int64_t n_batch, int32_t compute_scales, double* reward, void* stream);

/* update() (pgpr.py:137-173): rewards float64 [n_steps, n_batch] -> discounted returns in place; loss_out double[4] =
 * (loss, actor, critic, entropy); coef_actor = advantage / n_batch and dvalue = -2 advantage / n_batch feed
 * hw_pgpr_policy_grad. */
int hw_pgpr_reinforce_loss(double* rewards, const float* values, const float* logp, const float* entropy, int32_t n_steps,
                           int64_t n_batch, double gamma, double weight_factor, double* loss_out, float* coef_actor,
                           float* dvalue, void* stream);

/* ---- row exchange primitives for row-sharded tables (no reference counterpart -- the
 * reference replicates tables under DDP, trainer/trainer.py:83-84) -------------------------------------
 * out[i, :] = table[idx[i], :]  and  table[idx[i], :] += src[i, :] (float atomics; ids outside the shard are skipped). */
int hw_gather_rows(const float* table, int64_t n_table_rows, int32_t dim, const int64_t* idx, int64_t n,
                   float* out, int32_t* oob_counter, void* stream);
int hw_scatter_add_rows(float* table, int64_t n_table_rows, int32_t dim, const int64_t* idx, int64_t n,
                        const float* src, void* stream);

/* dst[0 .. n_bytes) = (uint8_t)byte_value, as KERNEL launches on `stream` (any size and alignment).  The stream-ordered fill
 * to use in code that may be captured in a HIP graph -- e.g. driver.GraphedTrainStep around Trainer._train_epoch
 * (trainer/trainer.py:208-268) -- instead of hipMemsetAsync: a memset NODE of an instantiated graph writes its captured value
 * on the first launch only and a stale 16-byte staging pattern afterwards on this stack (tools/memset_pattern_probe.py,
 * DESIGN.md section 2); every fill inside this library goes through the same kernels. */
int hw_fill_bytes(void* dst, int32_t byte_value, int64_t n_bytes, void* stream);

/* Device-side routing of one sharded lookup ("bucket indices by owner -> all_to_all(ids) -> local
 * gather -> all_to_all(rows)"; owner = id % n_ranks, local row = id / n_ranks).  No host read anywhere:
 *   inv[i]                               = owner(ids[i]) * inv_stride + rank of ids[i] among the owner's distinct ids (ascending)
 *   send_counts[o * count_stride]        = distinct ids asked of owner o (<= cap)
 *   send_local[o * local_stride + j]     = local row of the j-th of them
 *   *overflow                            = max distinct count seen when some owner is asked for more than cap rows (else
 *                                          untouched; zero it before the first call); the plan is then unusable.
 * The strides let ONE int32 message per peer carry the count and the list ([P, 1 + cap]: send_counts = msg, send_local =
 * msg + 1, both strides 1 + cap) and ONE float message per peer carry the rows of every table of the lookup ([P, T, cap, d]:
 * inv_stride = peer_rows = T * cap, table t at row_offset t * cap); the receive buffer of the row message is then the set
 * of compact tables (table t = the buffer from float t * cap * d on, indexed by inv).  The fixed-size all-to-alls are the
 * caller's (torch.distributed / RCCL).  workspace: hw_shard_plan_workspace_bytes(n_rows, n_ranks). */
int64_t hw_shard_plan_workspace_bytes(int64_t n_rows, int32_t n_ranks);
int hw_shard_plan(const int64_t* ids, int64_t n, int64_t n_rows, int32_t n_ranks, int32_t cap, void* workspace,
                  int64_t* inv, int64_t inv_stride, int32_t* send_counts, int64_t count_stride, int32_t* send_local,
                  int64_t local_stride, int32_t* overflow, int32_t* oob_counter, void* stream);
/* Owner side.  buf row of (peer p, slot j) = p * peer_rows + row_offset + j:
 * out[that row, :] = shard[recv_local[p * local_stride + j], :] for j < recv_counts[p * count_stride] (others untouched);
 * grad_shard[recv_local[...], :] += src[that row, :] (float atomics: two peers may send the same row). */
int hw_shard_gather(const float* shard, int64_t n_local_rows, int32_t dim, const int32_t* recv_local, int64_t local_stride,
                    const int32_t* recv_counts, int64_t count_stride, int32_t n_ranks, int32_t cap, float* out,
                    int64_t peer_rows, int64_t row_offset, int32_t* oob_counter, void* stream);
int hw_shard_scatter_add(float* grad_shard, int64_t n_local_rows, int32_t dim, const int32_t* recv_local,
                         int64_t local_stride, const int32_t* recv_counts, int64_t count_stride, int32_t n_ranks,
                         int32_t cap, const float* src, int64_t peer_rows, int64_t row_offset, int32_t* oob_counter,
                         void* stream);

/* ---- CAFE inference (model/knowledge_aware_recommender/cafe.py) ----------------------------------------------
 * y[r, :] = log_softmax(x[r, :]) (cafe.py:597-598, :819-820). */
int hw_log_softmax_rows(const float* x, int64_t ld, int64_t n_rows, int64_t n_cols, float* y, int64_t ldy,
                        void* stream);

/* One node of a user's reasoning program, in execution order (the chain of a metapath for
 * SymbolicNetwork.infer_with_path :587-642, the BFS order of the layout tree for MetaProgramExecutor.execute :790-857). */
typedef struct {
    int32_t parent;        /* index (within the task) of the node whose paths are extended; -1 = the user root */
    int32_t rel;           /* relation id followed */
    int32_t eh_is_entity;  /* type the parent's tail id is looked up as: the relation's head type (0 user, 1 entity) */
    int32_t k;             /* children kept per parent path */
    int32_t is_leaf;       /* drop excluded ids here (and add the picks to the exclusion set when grow_excluded) */
    int32_t lp_row;        /* row of `logprobs` holding this node's log-probabilities over its tail vocabulary */
    int32_t depth;         /* 1-based: number of ids after the user once this node has appended its pick */
    int32_t vocab;         /* size of the tail vocabulary (ids >= vocab are ignored) */
    int32_t cap;           /* capacity (paths) of this node's output slots */
    int32_t slot_off;      /* first output slot of this node, relative to the task's slot_ptr */
} hw_cafe_node;

/* One level of SymbolicNetwork.forward's loss (cafe.py:541-553): nn.CrossEntropyLoss(logits, target) (mean) and, when
 * pos / neg are given, mean sigmoid(log_softmax[neg] - log_softmax[pos]).  loss_out[0] = cross entropy, [1] = ranking term;
 * lse_out [n_rows] feeds the gradient call; row_ws: 2 * n_rows floats. */
int hw_cafe_level_loss(const float* logits, int64_t ld, int64_t n_rows, int64_t n_cols, const int64_t* target,
                       const int64_t* pos, const int64_t* neg, float* lse_out, float* loss_out, float* row_ws,
                       void* stream);
/* dlogits for upstream gradients (g_ce, g_rank) of the two losses. */
int hw_cafe_level_loss_grad(const float* logits, int64_t ld, int64_t n_rows, int64_t n_cols, const float* lse,
                            const int64_t* target, const int64_t* pos, const int64_t* neg, float g_ce, float g_rank,
                            float* dlogits, int64_t ldd, void* stream);
/* The same with the upstream gradients as device scalars (g_ce required; g_rank required when pos / neg are given): what
 * autograd hands the backward of the level loss already lives on the device -- no host read, and the launch can sit in a
 * captured training step (driver.GraphedStagedTrainStep). */
int hw_cafe_level_loss_grad_dev(const float* logits, int64_t ld, int64_t n_rows, int64_t n_cols, const float* lse,
                                const int64_t* target, const int64_t* pos, const int64_t* neg, const float* g_ce,
                                const float* g_rank, float* dlogits, int64_t ldd, void* stream);

/* Interpret every task's program: for each node in order and each parent path in order, candidates = distinct
 * neighbours of the path's tail under node.rel, minus ids already picked at this node, minus (leaves) the excluded
 * set; keep the top-k by logprobs[node.lp_row] (value desc, id asc); append.  Outputs per slot: 3 ids (after the
 * user) and 3 log-probs; out_count[node] paths per node.  excluded_init: bitset [n_tasks, bit_words]. */
int hw_cafe_execute(const hw_pgpr_graph* graph, const hw_cafe_node* nodes, const int32_t* node_ptr,
                    const int64_t* users, int64_t n_tasks, int32_t max_nodes_per_task,
                    const float* logprobs, int64_t logprob_ld, const uint32_t* excluded_init,
                    int64_t bit_words, int32_t grow_excluded, int32_t* out_ids, float* out_vals,
                    int32_t* out_count, const int64_t* slot_ptr, void* stream);
/* The same over LOGITS: `logits` is the vocabulary product as hw_gemm_f32_lse wrote it and row_lse [rows, 2] (hw_lse_rows) turns
 * an entry into its log-probability, (x - row_lse[2 r]) - row_lse[2 r + 1], where the interpreter reads one. */
int hw_cafe_execute_logits(const hw_pgpr_graph* graph, const hw_cafe_node* nodes, const int32_t* node_ptr,
                           const int64_t* users, int64_t n_tasks, int32_t max_nodes_per_task, const float* logits,
                           int64_t logit_ld, const float* row_lse, const uint32_t* excluded_init, int64_t bit_words,
                           int32_t grow_excluded, int32_t* out_ids, float* out_vals, int32_t* out_count,
                           const int64_t* slot_ptr, void* stream);

/* The inputs of hw_cafe_modules_forward(_fused) / hw_cafe_execute / hw_cafe_collect for a MetaProgramExecutor batch
 * (cafe.py:790-857), built on the device from each task's VISITING ORDER of the layout tree (children of the branching node
 * in a random order, cafe.py:807-808: the draw is the caller's).  Tables [n_orders, nodes_per_task] hold, per order, the
 * program's columns (parent index, relation CODE = index into mod_*, 1-based depth, output capacity, leaf flag) and
 * [n_orders, n_leaves] the program row of every leaf chain; mod_* [n_mods]: relation id, head type, tail vocabulary.
 * key = depth * n_mods + relation code; key_count[key] = nodes of that key in ONE program (the same for every order),
 * key_first[key] = sum of key_count below it.  Outputs: nodes / node_ptr / slot_ptr / leaf_nodes as the interpreter and the
 * collector take them (slot_ptr[b] = b * slots_per_task), and the three index columns [n_tasks * nodes_per_task] of the
 * module stage -- rows grouped by key, tasks ascending inside a group, a task's nodes in program order: eh_idx (the row of
 * the parent's vector: task b's user row for parent < 0, else n_tasks + the parent's node row), uv_idx (b), row_idx (the
 * node row).  Equal to the host construction element for element (tests/test_gpu_cafe.py). */
int hw_cafe_plan(const int64_t* order_of_task, int64_t n_tasks, int32_t nodes_per_task, int32_t n_leaves,
                 const int32_t* t_parent, const int32_t* t_rel_code, const int32_t* t_depth, const int32_t* t_cap,
                 const int32_t* t_is_leaf, const int32_t* t_leaf_row, int64_t n_orders, const int32_t* mod_rid,
                 const int32_t* mod_eh_is_entity, const int32_t* mod_vocab, int32_t n_mods, const int32_t* key_first,
                 const int32_t* key_count, int32_t n_keys, int32_t k, int64_t slots_per_task, hw_cafe_node* nodes,
                 int32_t* node_ptr, int64_t* slot_ptr, int64_t* eh_idx, int64_t* uv_idx, int64_t* row_idx,
                 int32_t* leaf_nodes, void* stream);

/* One relation module of SymbolicNetwork in evaluation mode (cafe.py:645-699): device pointers to the parameters of
 * RelationModule (deep = 0: fc1 [256, 2d], bn1, fc2 [d, 256], residual + eh_vec) or DeepRelationModule (deep = 1: fc1
 * [256, 2d], bn1, fc2 [2d, 256], residual + feature, bn2 [2d], fc3 [d, 2d]).  Weights are nn.Linear layouts [out, in]. */
typedef struct {
    const float *w1, *b1, *bn1_weight, *bn1_bias, *bn1_mean, *bn1_var;
    const float *w2, *b2, *bn2_weight, *bn2_bias, *bn2_mean, *bn2_var;
    const float *w3, *b3;
    float bn1_eps, bn2_eps;
    int32_t deep, reserved;
} hw_cafe_module;

/* The module applications of one run_program call (the per-(level, relation) loop around `module((eh_vec, user_vec))` of
 * MetaProgramExecutor, cafe.py:807-835) from one host call.  Group g applies module mods[group_mod[g]] to rows
 * group_ptr[g] .. group_ptr[g + 1] of the index columns: input = [ eh | user_vec[uv_idx] ] with eh = user_vec[eh_idx] when
 * eh_idx < n_user_rows, else vec_all[eh_idx - n_user_rows]; the result row goes to vec_all[out_rows].  Groups run in order
 * (a level reads what the levels before it wrote).  mods / group_mod / group_ptr are HOST arrays, the rest device memory;
 * dim % 4 == 0; workspace: hw_cafe_modules_workspace_floats(largest group, dim) floats.  The launches are hw_gemm_f32 and
 * hw_batchnorm1d_forward(training = 0) themselves: results equal the module-by-module path bit for bit. */
int64_t hw_cafe_modules_workspace_floats(int64_t max_rows, int32_t dim);
int hw_cafe_modules_forward(const hw_cafe_module* mods, const int32_t* group_mod, const int64_t* group_ptr, int32_t n_groups,
                            const float* user_vec, int64_t n_user_rows, int32_t dim, const int64_t* eh_idx,
                            const int64_t* uv_idx, const int64_t* out_rows, float* vec_all, float* workspace,
                            int64_t workspace_floats, void* stream);

/* The same module applications (MetaProgramExecutor's `module((eh_vec, user_vec))` per node, cafe.py:807-835, over the
 * RelationModule / DeepRelationModule forwards of cafe.py:645-699) with ONE launch per level (csrc/cafe_modules.hip,
 * cafe_level_kernel): a block takes eight rows
 * of a group through the whole module with the activations in LDS.  Differences to hw_cafe_modules_forward: the weight
 * pointers of `mods` (w1, w2, w3) address TRANSPOSED copies [in, out] (prepared by the caller once per weight version);
 * group_level[g] (host, non-decreasing) is the depth of group g's nodes -- groups of one level only read node vectors of
 * earlier levels and share a launch; no workspace.  2 * dim <= 256, else HW_ERR_UNSUPPORTED.  Products are fp32 FMA chains
 * in k order (the launch-per-layer form runs them as three-way bf16 splits): equal to it to fp32 rounding, not bit for bit. */
int hw_cafe_modules_forward_fused(const hw_cafe_module* mods, const int32_t* group_mod, const int64_t* group_ptr,
                                  const int32_t* group_level, int32_t n_groups, const float* user_vec, int64_t n_user_rows,
                                  int32_t dim, const int64_t* eh_idx, const int64_t* uv_idx, const int64_t* out_rows,
                                  float* vec_all, void* stream);

/* collect_results + the ranking loop of run_program (cafe.py:417-441, :859-867) on hw_cafe_execute's output, one user
 * per task: entries = the paths of the task's leaf nodes in `leaf_nodes` order (int32 [n_tasks, n_leaves], node index
 * inside the task, -1 = absent), score = product of the path's float32 log-probs left to right.  Per item the entry with
 * the highest score (ties: the earlier entry) is emitted with the path of the LAST entry that reached the item (the
 * reference's inst[item]); emission order = score descending, entry ascending.  results [n_tasks, n_items] (-inf where
 * nothing is written); records rec_* [n_tasks, max_rec] (item, score, leaf ordinal of the attached path, its 3 ids) and
 * rec_count [n_tasks].  best_ws uint64 / last_ws int32 [n_tasks, n_items] scratch.  *overflow (zero it first) = the
 * largest count of distinct items reached by one task when that exceeds the 8 192 winners one workgroup ranks in LDS (the
 * task's scores then leave as NaN); with the reference's 5 paths per node and the 64 leaves accepted here a task has at most
 * 8 000 entries, so the limit is out of reach under the reference's settings. */
int hw_cafe_collect(const hw_cafe_node* nodes, const int32_t* node_ptr, const int64_t* slot_ptr, const int32_t* ids,
                    const float* vals, const int32_t* counts, const int32_t* leaf_nodes, int64_t n_tasks, int32_t n_leaves,
                    int32_t n_items, uint64_t* best_ws, int32_t* last_ws, float* results, int32_t max_rec,
                    int32_t* rec_item, float* rec_score, int32_t* rec_leaf, int32_t* rec_ids, int32_t* rec_count,
                    int32_t* overflow, void* stream);

/* ---- nn.BatchNorm2d, training mode, NCHW with few channels (conve.py:62-66, convkb.py:51-56) ------------------------
 * y = (x - mean_c) rstd_c gamma_c + beta_c with the batch statistics over N x HW (biased variance), running_mean / _var
 * updated in place with `momentum` (running_var from the unbiased variance, as torch does); save_mean / save_rstd [C] feed
 * the backward.  gamma / beta / running_* nullable.  workspace: hw_batchnorm2d_workspace_doubles(C) float64. */
int64_t hw_batchnorm2d_workspace_doubles(int32_t channels);
int hw_batchnorm2d_forward(const float* x, int64_t batch, int32_t channels, int32_t hw, const float* gamma,
                           const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                           float* save_mean, float* save_rstd, float* y, double* workspace, void* stream);
int hw_batchnorm2d_backward(const float* x, const float* dy, int64_t batch, int32_t channels, int32_t hw,
                            const float* gamma, const float* save_mean, const float* save_rstd, float* dx, float* dgamma,
                            float* dbeta, double* workspace, void* stream);

/* nn.BatchNorm1d over [batch, channels] rows (tucker.py:58-59 bn0 / bn1, conve.py:67 bn2, cafe.py:667-690): training != 0
 * uses the batch statistics (biased variance; running stats updated in place with `momentum`, running_var from the
 * unbiased variance; save_mean / save_rstd feed the backward), training == 0 the running statistics.  gamma / beta nullable.
 * workspace: hw_batchnorm1d_workspace_doubles(channels) float64 (training and backward). */
int64_t hw_batchnorm1d_workspace_doubles(int32_t channels);
int hw_batchnorm1d_forward(const float* x, int64_t batch, int32_t channels, const float* gamma, const float* beta, float eps,
                           float momentum, float* running_mean, float* running_var, float* save_mean, float* save_rstd,
                           float* y, double* workspace, int32_t training, void* stream);
int hw_batchnorm1d_backward(const float* x, const float* dy, int64_t batch, int32_t channels, const float* gamma,
                            const float* save_mean, const float* save_rstd, float* dx, float* dgamma, float* dbeta,
                            double* workspace, void* stream);
/* nn.Dropout on n floats: element i kept with probability 1 - drop_p and scaled by 1 / (1 - drop_p), from Philox word i of
 * the (seed, offset) stream; the backward is the same call on the upstream gradient. */
int hw_dropout(const float* x, int64_t n, float drop_p, uint64_t seed, uint64_t offset, float* y, void* stream);

/* ---- dense Adam over all tables of a model in one launch (trainer/trainer.py:189-190: optim.Adam(params, lr,
 * weight_decay), the reference's default learner; torch/optim/adam.py update rule, no amsgrad / maximize).
 * *step is a DEVICE counter, incremented by the call (so the launch replays inside a HIP graph); every segment is
 * updated with the incremented value.  flags: HW_ADAM_ZERO_GRADS also clears each g (stands in for the next zero_grad);
 * HW_ADAM_DECOUPLED_WD applies the weight decay to p directly (optim.AdamW, trainer.py:191-192) instead of to g. */
#define HW_ADAM_ZERO_GRADS 1
#define HW_ADAM_DECOUPLED_WD 2
#define HW_ADAM_MAX_SEGS 16
typedef struct {
    float* p;  /* parameter */
    float* g;  /* gradient */
    float* m;  /* exp_avg */
    float* v;  /* exp_avg_sq */
    int64_t n; /* elements */
} hw_adam_seg;
typedef struct {
    hw_adam_seg seg[HW_ADAM_MAX_SEGS];
    int32_t n_segs;
} hw_adam_args;
int hw_adam_step(const hw_adam_args* args, double lr, double beta1, double beta2, double eps, double weight_decay,
                 int64_t* step, int32_t flags, void* stream);

const char* hw_version(void);

#ifdef __cplusplus
}
#endif
#endif /* HOPWISE_AMD_H */
