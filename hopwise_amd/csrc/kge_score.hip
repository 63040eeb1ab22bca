// C-ABI entry points of the KGE scorers; kernels and launch logic live in kge_score_kernels.h.  This translation unit
// instantiates the five hot-path models, kge_score_ext.hip the widened set.
#include "kge_score_kernels.h"

using namespace hw;

// HW_SCORER_STAGGER=0: the fused sampler + scorer without its issue-priority stagger (read once)
static int scorer_stagger() {
    static const int on = [] { const char* e = getenv("HW_SCORER_STAGGER"); return (e && e[0] == '0') ? 0 : 1; }();
    return on;
}

extern "C" int hw_kge_score_grouped(const hw_kge_desc* desc, const int64_t* h_idx, const int64_t* r_idx,
                                    int64_t r_fixed, const int64_t* t_idx, int64_t t_stride_b, int64_t t_stride_j,
                                    int64_t n_groups, int64_t n_tail_per_group, float* out, int64_t o_stride_b,
                                    int64_t o_stride_j, int32_t* oob_counter, void* stream) {
    if (desc && HW_FAMILY_OF(desc->model) == 1)
        return ext_score_grouped(desc, h_idx, r_idx, r_fixed, t_idx, t_stride_b, t_stride_j, n_groups, n_tail_per_group,
                                 out, o_stride_b, o_stride_j, oob_counter, stream);
    return score_grouped_impl<0>(desc, h_idx, r_idx, r_fixed, t_idx, t_stride_b, t_stride_j, n_groups, n_tail_per_group,
                                 out, o_stride_b, o_stride_j, oob_counter, stream);
}

extern "C" int hw_kge_score_grouped_sampled(const hw_kge_desc* desc, const int64_t* h_idx, const int64_t* r_idx,
                                            int64_t r_fixed, int64_t* t_idx, int64_t t_stride_b, int64_t t_stride_j,
                                            int64_t n_groups, int64_t n_tail_per_group, const int64_t* used_ptr,
                                            const int64_t* used_idx, int64_t n_key_space, const int64_t* keys,
                                            int64_t value_num, uint64_t seed, uint64_t offset, float* out,
                                            int64_t o_stride_b, int64_t o_stride_j, int32_t* oob_counter, void* stream) {
    if (!desc || !used_ptr || !keys || !t_idx || n_key_space < 1 || value_num < 2) return HW_ERR_INVALID_ARG;
    if (offset >> 63) return HW_ERR_UNSUPPORTED;  // device-indirect Philox offsets (hw_common.h): hw_neg_sample, then the scorer
    if (HW_FAMILY_OF(desc->model) != 0) return HW_ERR_UNSUPPORTED;  // caller: sample, then score
    SamplerArgs sa;
    sa.used_ptr = used_ptr, sa.used_idx = used_idx, sa.keys = keys, sa.n_key_space = n_key_space;
    sa.value_num = value_num, sa.seed = seed, sa.offset = offset, sa.words = value_num <= 65536 ? (int)((value_num + 31) / 32) : 0, sa.stagger = scorer_stagger();  // 0: no LDS bitmap, binary search in the sorted used slice
    return score_grouped_impl<0>(desc, h_idx, r_idx, r_fixed, t_idx, t_stride_b, t_stride_j, n_groups, n_tail_per_group,
                                 out, o_stride_b, o_stride_j, oob_counter, stream, &sa);
}

static int selfadv_launch(const hw_kge_desc* desc, const int64_t* h_idx, const int64_t* r_idx, int64_t r_fixed, int64_t* t_idx,
                          int64_t t_stride_b, int64_t t_stride_j, int64_t n_groups, int64_t n_tail_per_group,
                          const int64_t* used_ptr, const int64_t* used_idx, int64_t n_key_space, const int64_t* keys,
                          int64_t value_num, uint64_t seed, uint64_t offset, float* out, int64_t o_stride_b,
                          int64_t o_stride_j, float alpha, float* loss_out, float* group_loss_ws,
                          const float* prev_group_loss, int64_t prev_n_groups, float* prev_loss_out, int32_t* oob_counter,
                          void* stream) {
    if (!desc || !used_ptr || !keys || !t_idx || !group_loss_ws || n_key_space < 1 || value_num < 2 || n_groups < 1 ||
        n_tail_per_group < 2)
        return HW_ERR_INVALID_ARG;
    if (prev_group_loss && (!prev_loss_out || prev_n_groups < 1 || prev_n_groups > INT32_MAX || prev_group_loss == group_loss_ws))
        return HW_ERR_INVALID_ARG;
    if (HW_FAMILY_OF(desc->model) != 0 || (offset >> 63)) return HW_ERR_UNSUPPORTED;
    SamplerArgs sa;
    sa.used_ptr = used_ptr, sa.used_idx = used_idx, sa.keys = keys, sa.n_key_space = n_key_space;
    sa.value_num = value_num, sa.seed = seed, sa.offset = offset, sa.words = value_num <= 65536 ? (int)((value_num + 31) / 32) : 0, sa.stagger = scorer_stagger();  // 0: no LDS bitmap, binary search in the sorted used slice
    SelfAdvArgs la;
    la.alpha = alpha, la.inv_b = 1.0f / (float)n_groups, la.group_loss = group_loss_ws, la.loss_out = loss_out;
    if (prev_group_loss) la.prev_group_loss = prev_group_loss, la.prev_loss_out = prev_loss_out, la.prev_n = (int)prev_n_groups;
    return score_grouped_impl<0>(desc, h_idx, r_idx, r_fixed, t_idx, t_stride_b, t_stride_j, n_groups, n_tail_per_group,
                                 out, o_stride_b, o_stride_j, oob_counter, stream, &sa, &la);
}

extern "C" int hw_kge_score_grouped_sampled_selfadv(const hw_kge_desc* desc, const int64_t* h_idx, const int64_t* r_idx,
                                                    int64_t r_fixed, int64_t* t_idx, int64_t t_stride_b, int64_t t_stride_j,
                                                    int64_t n_groups, int64_t n_tail_per_group, const int64_t* used_ptr,
                                                    const int64_t* used_idx, int64_t n_key_space, const int64_t* keys,
                                                    int64_t value_num, uint64_t seed, uint64_t offset, float* out,
                                                    int64_t o_stride_b, int64_t o_stride_j, float alpha, float* loss_out,
                                                    float* group_loss_ws, int32_t* counter_ws, int32_t* oob_counter,
                                                    void* stream) {
    (void)counter_ws;
    if (!loss_out) return HW_ERR_INVALID_ARG;
    return selfadv_launch(desc, h_idx, r_idx, r_fixed, t_idx, t_stride_b, t_stride_j, n_groups, n_tail_per_group, used_ptr,
                          used_idx, n_key_space, keys, value_num, seed, offset, out, o_stride_b, o_stride_j, alpha, loss_out,
                          group_loss_ws, nullptr, 0, nullptr, oob_counter, stream);
}

extern "C" int hw_kge_score_grouped_sampled_selfadv_chained(
    const hw_kge_desc* desc, const int64_t* h_idx, const int64_t* r_idx, int64_t r_fixed, int64_t* t_idx, int64_t t_stride_b,
    int64_t t_stride_j, int64_t n_groups, int64_t n_tail_per_group, const int64_t* used_ptr, const int64_t* used_idx,
    int64_t n_key_space, const int64_t* keys, int64_t value_num, uint64_t seed, uint64_t offset, float* out,
    int64_t o_stride_b, int64_t o_stride_j, float alpha, float* group_loss_ws, const float* prev_group_loss,
    int64_t prev_n_groups, float* prev_loss_out, int32_t* oob_counter, void* stream) {
    return selfadv_launch(desc, h_idx, r_idx, r_fixed, t_idx, t_stride_b, t_stride_j, n_groups, n_tail_per_group, used_ptr,
                          used_idx, n_key_space, keys, value_num, seed, offset, out, o_stride_b, o_stride_j, alpha, nullptr,
                          group_loss_ws, prev_group_loss, prev_n_groups, prev_loss_out, oob_counter, stream);
}

extern "C" int hw_selfadv_finish(const float* group_loss, int64_t n_groups, float* loss_out, void* stream) {
    if (!group_loss || !loss_out || n_groups < 1 || n_groups > INT32_MAX) return HW_ERR_INVALID_ARG;
    HW_CLEAR_ERR();
    hipLaunchKernelGGL(selfadv_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, group_loss, (int)n_groups,
                       1.0f / (float)n_groups, loss_out);
    return HW_LAUNCH_RC();
}

extern "C" int hw_kge_full_sort(const hw_kge_desc* desc, const int64_t* h_idx, const int64_t* r_idx, int64_t r_fixed,
                                int64_t n_heads, int64_t tail_lo, int64_t tail_hi, float* out, int64_t out_ld,
                                int32_t* oob_counter, void* stream) {
    if (desc && HW_FAMILY_OF(desc->model) == 1)
        return ext_full_sort(desc, h_idx, r_idx, r_fixed, n_heads, tail_lo, tail_hi, out, out_ld, oob_counter, stream);
    return full_sort_impl<0>(desc, h_idx, r_idx, r_fixed, n_heads, tail_lo, tail_hi, out, out_ld, oob_counter, stream);
}

extern "C" int64_t hw_kge_full_sort_topk_workspace_bytes(int64_t n_heads, int64_t n_tails, int32_t k) {
    return full_sort_topk_workspace_bytes(n_heads, n_tails, k);
}

extern "C" int hw_kge_full_sort_topk(const hw_kge_desc* desc, const int64_t* h_idx, const int64_t* r_idx, int64_t r_fixed,
                                     int64_t n_heads, int64_t tail_lo, int64_t tail_hi, const int64_t* hist_rows,
                                     const int64_t* hist_cols, int64_t n_hist, int32_t mask_first_col, int32_t k,
                                     void* workspace, int64_t workspace_bytes, float* out_val, int64_t* out_idx,
                                     void* stream) {
    if (desc && HW_FAMILY_OF(desc->model) == 1)
        return ext_full_sort_topk(desc, h_idx, r_idx, r_fixed, n_heads, tail_lo, tail_hi, hist_rows, hist_cols, n_hist,
                                  mask_first_col, k, workspace, workspace_bytes, out_val, out_idx, stream);
    return full_sort_topk_impl<0>(desc, h_idx, r_idx, r_fixed, n_heads, tail_lo, tail_hi, hist_rows, hist_cols, n_hist,
                                  mask_first_col, k, workspace, workspace_bytes, out_val, out_idx, stream);
}

// The same launches under a device-side predicate: every workgroup leaves at once unless *run_if != 0 (nullptr: always run).
// What the GEMM full sort of the distance scorers queues behind its re-rank (kge_fullsort_gemm.hip): the exact kernel redoes
// the batch when a row could not prove its candidate set, and costs two empty launches otherwise -- no host read either way.
extern "C" int hw_kge_full_sort_topk_if(const hw_kge_desc* desc, const int64_t* h_idx, const int64_t* r_idx, int64_t r_fixed,
                                        int64_t n_heads, int64_t tail_lo, int64_t tail_hi, const int64_t* hist_rows,
                                        const int64_t* hist_cols, int64_t n_hist, int32_t mask_first_col, int32_t k,
                                        void* workspace, int64_t workspace_bytes, float* out_val, int64_t* out_idx,
                                        const int32_t* run_if, void* stream) {
    if (desc && HW_FAMILY_OF(desc->model) == 1)
        return ext_full_sort_topk(desc, h_idx, r_idx, r_fixed, n_heads, tail_lo, tail_hi, hist_rows, hist_cols, n_hist,
                                  mask_first_col, k, workspace, workspace_bytes, out_val, out_idx, stream, run_if);
    return full_sort_topk_impl<0>(desc, h_idx, r_idx, r_fixed, n_heads, tail_lo, tail_hi, hist_rows, hist_cols, n_hist,
                                  mask_first_col, k, workspace, workspace_bytes, out_val, out_idx, stream, run_if);
}
