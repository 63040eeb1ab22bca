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

}  // namespace hw
