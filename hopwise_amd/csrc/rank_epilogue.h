// Row maxima of a 32-row x 64-column accumulator slab (two 32x32 MFMA tiles side by side): the per-stretch maxima the ranking
// products leave for hw_topk_from_tile_max.  A lane holds, for each of its 16 rows rg (row = (rg & 3) + 8 (rg >> 2) + 4 (lane >> 5)),
// the larger of its two columns; the 32 lanes of a half-wave hold the row's 64 columns.  Reducing every row across the 32 lanes
// separately is 16 x 5 cross-lane exchanges (55 us of a 330 us product); here the exchange with the lane 16 / 8 / 4 / 2 away hands
// over HALF of the rows still held each time -- 8 + 4 + 2 + 1 exchanges -- after which a lane holds one row's maximum over 16
// lanes, and one more exchange with its neighbour completes it.
#pragma once

namespace hw {

// v[rg] -> (the maximum of row `rg_out` over the half-wave, in every lane; lanes 2 j and 2 j + 1 hold the same row)
__device__ __forceinline__ float half_wave_row_max(const float (&v)[16], int lane, int& rg_out) {
    const bool h4 = (lane >> 4) & 1, h3 = (lane >> 3) & 1, h2 = (lane >> 2) & 1, h1 = (lane >> 1) & 1;
    float a[8], b[4], c[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) {  // keep rows with bit 3 == h4, hand the others to the lane 16 away
        const float mine = h4 ? v[8 + i] : v[i], send = h4 ? v[i] : v[8 + i];
        a[i] = fmaxf(mine, __shfl_xor(send, 16, 64));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float mine = h3 ? a[4 + i] : a[i], send = h3 ? a[i] : a[4 + i];
        b[i] = fmaxf(mine, __shfl_xor(send, 8, 64));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float mine = h2 ? b[2 + i] : b[i], send = h2 ? b[i] : b[2 + i];
        c[i] = fmaxf(mine, __shfl_xor(send, 4, 64));
    }
    const float mine = h1 ? c[1] : c[0], send = h1 ? c[0] : c[1];
    float d = fmaxf(mine, __shfl_xor(send, 2, 64));
    d = fmaxf(d, __shfl_xor(d, 1, 64));
    rg_out = (h4 ? 8 : 0) | (h3 ? 4 : 0) | (h2 ? 2 : 0) | (h1 ? 1 : 0);
    return d;
}

// The same exchange carrying (maximum, sum of exp(x - maximum)) pairs: the per-stretch pieces of a row's log-sum-exp (the product
// that leaves them lets a row log-softmax over a vocabulary skip its pass over the matrix: csrc/cafe.hip, hw_lse_rows).
// merge: M = max(m1, m2), s = s1 exp(m1 - M) + s2 exp(m2 - M) with ONE exponential, d = exp(-|m1 - m2|), applied to the smaller
// side; a side that holds nothing is (-inf, 0) and stays out (d = 0; both empty: d irrelevant, s = 0).
__device__ __forceinline__ void lse_merge(float& m, float& s, float m2, float s2) {
    const float d = m == m2 ? 1.f : __expf(-fabsf(m - m2));  // m == m2 covers (-inf, -inf), whose difference is not a number
    s = m >= m2 ? s + s2 * d : s * d + s2;
    m = fmaxf(m, m2);
}

__device__ __forceinline__ void half_wave_row_lse(const float (&vm)[16], const float (&vs)[16], int lane, int& rg_out,
                                                  float& m_out, float& s_out) {
    const bool h4 = (lane >> 4) & 1, h3 = (lane >> 3) & 1, h2 = (lane >> 2) & 1, h1 = (lane >> 1) & 1;
    float am[8], as[8], bm[4], bs[4], cm[2], cs[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        am[i] = h4 ? vm[8 + i] : vm[i], as[i] = h4 ? vs[8 + i] : vs[i];
        lse_merge(am[i], as[i], __shfl_xor(h4 ? vm[i] : vm[8 + i], 16, 64), __shfl_xor(h4 ? vs[i] : vs[8 + i], 16, 64));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        bm[i] = h3 ? am[4 + i] : am[i], bs[i] = h3 ? as[4 + i] : as[i];
        lse_merge(bm[i], bs[i], __shfl_xor(h3 ? am[i] : am[4 + i], 8, 64), __shfl_xor(h3 ? as[i] : as[4 + i], 8, 64));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        cm[i] = h2 ? bm[2 + i] : bm[i], cs[i] = h2 ? bs[2 + i] : bs[i];
        lse_merge(cm[i], cs[i], __shfl_xor(h2 ? bm[i] : bm[2 + i], 4, 64), __shfl_xor(h2 ? bs[i] : bs[2 + i], 4, 64));
    }
    float m = h1 ? cm[1] : cm[0], s = h1 ? cs[1] : cs[0];
    lse_merge(m, s, __shfl_xor(h1 ? cm[0] : cm[1], 2, 64), __shfl_xor(h1 ? cs[0] : cs[1], 2, 64));
    lse_merge(m, s, __shfl_xor(m, 1, 64), __shfl_xor(s, 1, 64));
    rg_out = (h4 ? 8 : 0) | (h3 ? 4 : 0) | (h2 ? 2 : 0) | (h1 ? 1 : 0);
    m_out = m, s_out = s;
}

}  // namespace hw
