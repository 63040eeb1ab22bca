// Widened scorer family: TorusE, TransH, TransD, Analogy, HolE on the same gather kernels.
#include "kge_score_kernels.h"

namespace hw {

int ext_score_grouped(const hw_kge_desc* desc, const int64_t* h_idx, const int64_t* r_idx, int64_t r_fixed,
                      const int64_t* t_idx, int64_t t_stride_b, int64_t t_stride_j, int64_t n_groups,
                      int64_t n_tail_per_group, float* out, int64_t o_stride_b, int64_t o_stride_j, int32_t* oob_counter,
                      void* stream) {
    return score_grouped_impl<1>(desc, h_idx, r_idx, r_fixed, t_idx, t_stride_b, t_stride_j, n_groups, n_tail_per_group,
                                 out, o_stride_b, o_stride_j, oob_counter, stream);
}

int ext_full_sort(const hw_kge_desc* desc, const int64_t* h_idx, const int64_t* r_idx, int64_t r_fixed, int64_t n_heads,
                  int64_t tail_lo, int64_t tail_hi, float* out, int64_t out_ld, int32_t* oob_counter, void* stream) {
    return full_sort_impl<1>(desc, h_idx, r_idx, r_fixed, n_heads, tail_lo, tail_hi, out, out_ld, oob_counter, stream);
}

int ext_full_sort_topk(const hw_kge_desc* desc, const int64_t* h_idx, const int64_t* r_idx, int64_t r_fixed,
                       int64_t n_heads, int64_t tail_lo, int64_t tail_hi, const int64_t* hist_rows,
                       const int64_t* hist_cols, int64_t n_hist, int32_t mask_first_col, int32_t k, void* workspace,
                       int64_t workspace_bytes, float* out_val, int64_t* out_idx, void* stream, const int32_t* run_if) {
    return full_sort_topk_impl<1>(desc, h_idx, r_idx, r_fixed, n_heads, tail_lo, tail_hi, hist_rows, hist_cols, n_hist,
                                  mask_first_col, k, workspace, workspace_bytes, out_val, out_idx, stream, run_if);
}

}  // namespace hw
