// TuckER pieces that are not plain GEMMs (tucker.py:69-84, :97-127):
//   * the batched vector-matrix contraction x[b, :] . W_b[:, :] with hidden_dropout1 fused (the [B, d, d] tensor
//     is read exactly once, forward and backward) -- HBM-bound, d*d*4 bytes per row of the batch;
//   * sigmoid + BCELoss against the reference's smoothed "union of the batch's positives" target, fused with its
//     gradient (one pass over the [B, N] logits instead of five in the reference).
// The dense products (r . W, x . E^T and their backward) run on gemm.hip.
#include "hw_common.h"

namespace hw {

// Dropout mask of element (b, i, j) in Philox mode: word (i & 3) of block (b * ceil(d / 4) + (i >> 2)) * d + j of the
// (seed, offset) stream, so that a thread walking i for a fixed j (forward) or a wave handling four consecutive rows
// (backward) pays for one Philox block per four elements.
__device__ __forceinline__ void philox_block(uint64_t seed, uint64_t offset, uint64_t blk, uint32_t (&w)[4]) {
    uint32_t c0 = (uint32_t)blk, c1 = (uint32_t)(blk >> 32), c2 = (uint32_t)offset, c3 = (uint32_t)(offset >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0, c1 = n1, c2 = n2, c3 = n3;
        k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
    }
    w[0] = c0, w[1] = c1, w[2] = c2, w[3] = c3;
}
__device__ __forceinline__ float keep_scale(uint32_t word, float p, float inv_keep) {
    return (float)(word >> 8) * (1.0f / 16777216.0f) >= p ? inv_keep : 0.f;
}

// MODE 3: the keep flags written by the forward (see tucker_contract_fwd_kernel) -- nibble of (b, q, j)
__device__ __forceinline__ uint32_t keep_nibble(const float* __restrict__ mask, int b, int nq, int q, int d, int j) {
    const uint32_t* __restrict__ kb = reinterpret_cast<const uint32_t*>(mask);
    return (kb[((size_t)b * ((nq + 7) >> 3) + (q >> 3)) * d + j] >> (4 * (q & 7))) & 15u;
}

// out[b, j] = sum_i x[b, i] * w[b, i, j] * m[b, i, j]; MODE 0: no mask, 1: mask tensor (0 or 1/(1-p)), 2: Philox
template <int MODE>
__global__ __launch_bounds__(256) void tucker_contract_fwd_kernel(const float* __restrict__ x,
                                                                  const float* __restrict__ w,
                                                                  const float* __restrict__ mask, int d, float drop_p,
                                                                  uint64_t seed, uint64_t offset,
                                                                  float* __restrict__ out, int64_t w_stride,
                                                                  const int64_t* __restrict__ w_index,
                                                                  uint32_t* __restrict__ keep_out, int64_t batch) {
    if constexpr (MODE == 2) offset = philox_offset(offset);
    // one thread per (batch row, column), numbered flat over the batch: with a block per row, d = 200 filled three waves and an
    // eighth of the fourth, and the kernel is bound by the Philox rounds its lanes execute (2 x 10^7 blocks per call)
    const int64_t item = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (item >= batch * d) return;
    const int b = (int)(item / d);
    // keep_out (MODE 2, nullable): the keep flags of this call, 4 bits per (quad of rows q, column j) -- bit k = row 4 q + k --
    // eight quads per word, word ((b * ceil(nq / 8)) + (q >> 3)) * d + j.  The three backward kernels read them (MODE 3)
    // instead of evaluating the same Philox blocks again: the [B, d, d] dropout is Philox-bound (82 M draws per side).
    // w_stride = d * d with w_index == NULL: the reference's [B, d, d] tensor; with w_index: matrix w_index[b] of a small
    // set (one per relation); w_stride = 0: one [d, d] matrix shared by the whole batch
    const float* __restrict__ wb = w + (size_t)(w_index ? w_index[b] : b) * w_stride;
    const float* __restrict__ mb = MODE == 1 ? mask + (size_t)b * d * d : nullptr;
    const float* __restrict__ xb = x + (size_t)b * d;
    const int nq = (d + 3) >> 2;
    const float inv_keep = 1.f / (1.f - drop_p);
    {
        const int j = (int)(item - (int64_t)b * d);
        float acc = 0.f;
        if constexpr (MODE == 2) {
            uint32_t kw = 0u;
            const int nq8 = (nq + 7) >> 3;
            for (int q = 0; q < nq; ++q) {
                uint32_t rw[4];
                philox_block(seed, offset, ((uint64_t)b * nq + q) * d + j, rw);
                if (keep_out) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        kw |= (keep_scale(rw[k], drop_p, inv_keep) != 0.f ? 1u : 0u) << (4 * (q & 7) + k);
                    if ((q & 7) == 7 || q == nq - 1) {
                        keep_out[((size_t)b * nq8 + (q >> 3)) * d + j] = kw;
                        kw = 0u;
                    }
                }
                if (4 * q + 3 < d) {  // full quad: the four row loads are issued together (no test between load and use)
                    float wv[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) wv[k] = wb[(size_t)(4 * q + k) * d + j];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        acc = fmaf(xb[4 * q + k], wv[k] * keep_scale(rw[k], drop_p, inv_keep), acc);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int i = 4 * q + k;
                        if (i < d) acc = fmaf(xb[i], wb[(size_t)i * d + j] * keep_scale(rw[k], drop_p, inv_keep), acc);
                    }
                }
            }
        } else if constexpr (MODE == 1) {
#pragma unroll 4
            for (int i = 0; i < d; ++i) acc = fmaf(xb[i], wb[(size_t)i * d + j] * mb[(size_t)i * d + j], acc);
        } else {
#pragma unroll 4
            for (int i = 0; i < d; ++i) acc = fmaf(xb[i], wb[(size_t)i * d + j], acc);
        }
        out[(size_t)b * d + j] = acc;
    }
}

// gx[b, i] = sum_j go[b, j] w[b, i, j] m[b, i, j];   gw[b, i, j] = x[b, i] go[b, j] m[b, i, j]
template <int MODE>
__global__ __launch_bounds__(256) void tucker_contract_bwd_kernel(const float* __restrict__ x,
                                                                  const float* __restrict__ w,
                                                                  const float* __restrict__ mask,
                                                                  const float* __restrict__ go, int d, float drop_p,
                                                                  uint64_t seed, uint64_t offset,
                                                                  float* __restrict__ gx, float* __restrict__ gw,
                                                                  int64_t w_stride, const int64_t* __restrict__ w_index) {
    if constexpr (MODE == 2) offset = philox_offset(offset);
    const int b = blockIdx.x;
    const int wave = wave_in_block(), lane = threadIdx.x & 63;
    const float* __restrict__ wb = w + (size_t)(w_index ? w_index[b] : b) * w_stride;
    const float* __restrict__ mb = MODE == 1 ? mask + (size_t)b * d * d : nullptr;
    const float* __restrict__ gob = go + (size_t)b * d;
    const int nq = (d + 3) >> 2;
    const float inv_keep = 1.f / (1.f - drop_p);
    if constexpr (MODE == 3) {
        // gx only, d <= 256 (the training step's call): the four column passes of a quad are unrolled with clamped columns, so
        // that their 24 loads (flag word, go, four matrix rows) are in flight together -- as a loop over j the wave paid one
        // memory round trip per pass (105 us per call for 164 MFLOP).  Same sums in the same order: a column past d adds 0.
        if (!gw && d <= 256) {
            const uint32_t* __restrict__ kb = reinterpret_cast<const uint32_t*>(mask);
            const int nq8 = (nq + 7) >> 3;
            for (int q = wave; q < nq; q += 4) {
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
                uint32_t wd[4];
                float gj[4], wv[4][4];
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int j = min(lane + 64 * it, d - 1);
                    wd[it] = kb[((size_t)b * nq8 + (q >> 3)) * d + j];
                    gj[it] = gob[j];
#pragma unroll
                    for (int k = 0; k < 4; ++k) wv[it][k] = wb[(size_t)min(4 * q + k, d - 1) * d + j];
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const bool live = lane + 64 * it < d;
                    const uint32_t nib = (wd[it] >> (4 * (q & 7))) & 15u;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float g = live && ((nib >> k) & 1u) ? gj[it] * inv_keep : (live ? gj[it] * 0.f : 0.f);
                        acc[k] = fmaf(g, wv[it][k], acc[k]);
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float v = group_sum<64>(acc[k]);
                    if (gx && lane == 0 && 4 * q + k < d) gx[(size_t)b * d + 4 * q + k] = v;
                }
            }
            return;
        }
    }
    for (int q = wave; q < nq; q += 4) {  // four consecutive rows i = 4q .. 4q + 3 per wave pass
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        float xi[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) xi[k] = 4 * q + k < d ? x[(size_t)b * d + 4 * q + k] : 0.f;
        for (int j = lane; j < d; j += 64) {
            uint32_t rw[4] = {0u, 0u, 0u, 0u};
            uint32_t nib = 0u;
            if constexpr (MODE == 2) philox_block(seed, offset, ((uint64_t)b * nq + q) * d + j, rw);
            if constexpr (MODE == 3) nib = keep_nibble(mask, b, nq, q, d, j);
            const float gj = gob[j];
            if (4 * q + 3 < d) {  // full quad: loads first, no test between a load and its use
                float wv[4], mv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    wv[k] = wb[(size_t)(4 * q + k) * d + j];
                    mv[k] = 1.f;
                    if constexpr (MODE == 1) mv[k] = mb[(size_t)(4 * q + k) * d + j];
                    if constexpr (MODE == 2) mv[k] = keep_scale(rw[k], drop_p, inv_keep);
                    if constexpr (MODE == 3) mv[k] = (nib >> k) & 1u ? inv_keep : 0.f;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float g = gj * mv[k];
                    acc[k] = fmaf(g, wv[k], acc[k]);
                    if (gw) gw[(size_t)b * d * d + (size_t)(4 * q + k) * d + j] = xi[k] * g;
                }
                continue;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = 4 * q + k;
                if (i >= d) continue;
                float m = 1.f;
                if constexpr (MODE == 1) m = mb[(size_t)i * d + j];
                if constexpr (MODE == 2) m = keep_scale(rw[k], drop_p, inv_keep);
                if constexpr (MODE == 3) m = (nib >> k) & 1u ? inv_keep : 0.f;
                const float g = gj * m;
                acc[k] = fmaf(g, wb[(size_t)i * d + j], acc[k]);
                if (gw) gw[(size_t)b * d * d + (size_t)i * d + j] = xi[k] * g;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float v = group_sum<64>(acc[k]);
            if (gx && lane == 0 && 4 * q + k < d) gx[(size_t)b * d + 4 * q + k] = v;
        }
    }
}

// Gradient of a [d, d] matrix SHARED by the whole batch (the rec side of TuckER: every row uses the [UI-Relation] row, so
// r . W is one matrix): gw[i, j] = sum_b x[b, i] go[b, j] m[b, i, j].  Block (q, c) owns rows i = 4q .. 4q + 3 and the
// batch rows of chunk c; thread j accumulates in registers; the chunk partials are summed in fixed order afterwards
// (deterministic).  Same Philox words as the forward.
constexpr int TGW_CHUNK = 64;
template <int MODE>
__global__ __launch_bounds__(256) void tucker_gw_shared_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                                                               const float* __restrict__ go, int64_t batch, int d,
                                                               float drop_p, uint64_t seed, uint64_t offset,
                                                               float* __restrict__ partial) {
    if constexpr (MODE == 2) offset = philox_offset(offset);
    const int q = blockIdx.x, c = blockIdx.y;
    const int nq = (d + 3) >> 2;
    const float inv_keep = 1.f / (1.f - drop_p);
    const int64_t b0 = (int64_t)c * TGW_CHUNK, b1 = min(batch, b0 + TGW_CHUNK);
    for (int j = threadIdx.x; j < d; j += 256) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int64_t b = b0; b < b1; ++b) {  // unrolled: the loads of four batch rows in flight, the sums in the same order
            uint32_t rw[4] = {0u, 0u, 0u, 0u};
            uint32_t nib = 0u;
            if constexpr (MODE == 2) philox_block(seed, offset, ((uint64_t)b * nq + q) * d + j, rw);
            if constexpr (MODE == 3) nib = keep_nibble(mask, (int)b, nq, q, d, j);
            const float gj = go[(size_t)b * d + j];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = min(4 * q + k, d - 1);  // clamped: the out-of-range rows of the last quad are not stored
                float m = 1.f;
                if constexpr (MODE == 1) m = mask[((size_t)b * d + i) * d + j];
                if constexpr (MODE == 2) m = keep_scale(rw[k], drop_p, inv_keep);
                if constexpr (MODE == 3) m = (nib >> k) & 1u ? inv_keep : 0.f;
                acc[k] = fmaf(x[(size_t)b * d + i] * m, gj, acc[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (4 * q + k < d) partial[((size_t)c * d + 4 * q + k) * d + j] = acc[k];
    }
}

// The same for a SET of matrices selected per row (kg side: one matrix per relation, a few dozen of them): `order` lists the
// batch rows sorted by matrix index, block (q, c) walks 64 consecutive sorted rows and flushes its register sums with
// float atomics whenever the matrix changes (a matrix receives a handful of flushes per address).  gw is zeroed by the caller.
template <int MODE>
__global__ __launch_bounds__(256) void tucker_gw_indexed_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                                                                const float* __restrict__ go,
                                                                const int64_t* __restrict__ order,
                                                                const int64_t* __restrict__ w_index, int64_t n_mats,
                                                                int64_t batch, int d, float drop_p, uint64_t seed,
                                                                uint64_t offset, float* __restrict__ gw) {
    if constexpr (MODE == 2) offset = philox_offset(offset);
    const int q = blockIdx.x, c = blockIdx.y;
    const int nq = (d + 3) >> 2;
    const float inv_keep = 1.f / (1.f - drop_p);
    const int64_t s0 = (int64_t)c * TGW_CHUNK, s1 = min(batch, s0 + TGW_CHUNK);
    const int n = (int)(s1 - s0);
    // the chunk's (batch row, matrix) pairs once, in LDS: read per row inside the loop they were a chain of three dependent
    // global loads (order -> w_index -> data) in front of every row's arithmetic
    __shared__ int sb[TGW_CHUNK], sm[TGW_CHUNK];
    if ((int)threadIdx.x < n) {
        const int64_t bb = min(max(order[s0 + threadIdx.x], (int64_t)0), batch - 1);
        sb[threadIdx.x] = (int)bb;
        sm[threadIdx.x] = (int)min(max(w_index[bb], (int64_t)0), n_mats - 1);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < d; j += 256) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        int cur = -1;
        auto flush = [&]() {
            if (cur >= 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (4 * q + k < d) atomicAdd(gw + ((size_t)cur * d + 4 * q + k) * d + j, acc[k]);
            }
            acc[0] = acc[1] = acc[2] = acc[3] = 0.f;
        };
        for (int base = 0; base < n; base += 4) {  // four rows' loads in flight, their sums in row order
            float gj[4], xm[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int bb = sb[min(base + u, n - 1)];
                uint32_t rw[4] = {0u, 0u, 0u, 0u};
                uint32_t nib = 0u;
                if constexpr (MODE == 2) philox_block(seed, offset, ((uint64_t)bb * nq + q) * d + j, rw);
                if constexpr (MODE == 3) nib = keep_nibble(mask, bb, nq, q, d, j);
                gj[u] = go[(size_t)bb * d + j];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = min(4 * q + k, d - 1);
                    float m = 1.f;
                    if constexpr (MODE == 1) m = mask[((size_t)bb * d + i) * d + j];
                    if constexpr (MODE == 2) m = keep_scale(rw[k], drop_p, inv_keep);
                    if constexpr (MODE == 3) m = (nib >> k) & 1u ? inv_keep : 0.f;
                    xm[u][k] = x[(size_t)bb * d + i] * m;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (base + u < n) {  // uniform across the block: every thread walks the same rows
                    const int mat = sm[base + u];
                    if (mat != cur) {
                        flush();
                        cur = mat;
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[k] = fmaf(xm[u][k], gj[u], acc[k]);
                }
            }
        }
        flush();
    }
}

__global__ __launch_bounds__(256) void tucker_gw_reduce_kernel(const float* __restrict__ partial, int chunks, int64_t n,
                                                               float* __restrict__ gw) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float v = 0.f;
        for (int c = 0; c < chunks; ++c) v += partial[(size_t)c * n + i];
        gw[i] = v;
    }
}

// ---- sigmoid + BCELoss(mean) against t[b, c] = scale * flag[c] + add  ------------------------------------------
// Arithmetic order of the reference: p = sigmoid(x) in fp32, loss terms max(log(p), -100) and max(log(1 - p), -100)
// (torch BCELoss), gradient (p - t) / max((1 - p) p, 1e-12) * p (1 - p) / numel (binary_cross_entropy_backward
// followed by sigmoid_backward).  pred (nullable) receives p; glogits (nullable) the gradient.
constexpr int TB_BLOCKS = 4096;

// ld > n_cols (padded rows, see hw_tucker_sigmoid_bce_ld): the sweep runs over the flat [n_rows, ld] buffer and the
// ld - n_cols trailing entries of each row count for nothing (no loss, gradient 0; their content is never interpreted).
__global__ __launch_bounds__(256) void tucker_bce_kernel(const float* __restrict__ logits, int64_t n_rows, int64_t n_valid,
                                                         int64_t n_cols, const uint8_t* __restrict__ flag, float scale,
                                                         float add, float inv_numel, float* __restrict__ pred,
                                                         float* __restrict__ glogits, float* __restrict__ partial) {
    __shared__ float red[4];
    const int64_t total = n_rows * n_cols;  // n_cols = leading dimension, n_valid = real columns
    const int64_t n4 = total >> 2;
    const int64_t stride = (int64_t)gridDim.x * 256;
    const int64_t stride_mod = (stride * 4) % n_cols;
    float acc = 0.f;
    int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t c = (i4 * 4) % n_cols;
    const f32x4* __restrict__ in4 = reinterpret_cast<const f32x4*>(logits);
    for (; i4 < n4; i4 += stride) {
        const f32x4 x = in4[i4];
        const float xs[4] = {x.x, x.y, x.z, x.w};
        float ps[4], gs[4];
        int64_t cc = c;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool live = cc < n_valid;
            const float t = (flag && live) ? fmaf(scale, (float)flag[cc], add) : add;
            const float l = bce_elem(live ? xs[k] : 0.f, t, inv_numel, &ps[k], &gs[k]);
            acc += live ? l : 0.f;
            gs[k] = live ? gs[k] : 0.f;
            cc = cc + 1 == n_cols ? 0 : cc + 1;
        }
        if (pred) reinterpret_cast<f32x4*>(pred)[i4] = f32x4{ps[0], ps[1], ps[2], ps[3]};
        if (glogits) reinterpret_cast<f32x4*>(glogits)[i4] = f32x4{gs[0], gs[1], gs[2], gs[3]};
        c += stride_mod;
        c = c >= n_cols ? c - n_cols : c;
    }
    if (blockIdx.x == 0 && threadIdx.x < (total & 3)) {  // up to three tail elements
        const int64_t i = (n4 << 2) + threadIdx.x;
        const bool live = i % n_cols < n_valid;
        const float t = (flag && live) ? fmaf(scale, (float)flag[i % n_cols], add) : add;
        float pv, gv;
        const float l = bce_elem(live ? logits[i] : 0.f, t, inv_numel, &pv, &gv);
        acc += live ? l : 0.f;
        if (pred) pred[i] = pv;
        if (glogits) glogits[i] = live ? gv : 0.f;
    }
    acc = group_sum<64>(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void tucker_bce_finish_kernel(const float* __restrict__ partial, int n, float inv_numel,
                                                                float* __restrict__ loss) {
    __shared__ float red[4];
    float v = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) v += partial[i];
    v = group_sum<64>(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = ((red[0] + red[1]) + (red[2] + red[3])) * inv_numel;
}

__global__ void sigmoid_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = 1.f / (1.f + expf(-x[i]));
}

}  // namespace hw

using namespace hw;

static int contract_launch(int mode, const float* x, const float* w, const float* mask, const float* gout, int64_t batch,
                           int32_t dim, float drop_p, uint64_t seed, uint64_t offset, float* out, float* gx, float* gw,
                           bool backward, void* stream, bool shared_w = false, const int64_t* w_index = nullptr,
                           uint32_t* keep_out = nullptr) {
    const int64_t w_stride = (shared_w && !w_index) ? 0 : (int64_t)dim * dim;
    if (!x || !w || batch < 0 || dim < 1 || batch > INT32_MAX) return HW_ERR_INVALID_ARG;
    if ((mode == 2 || mode == 3) && !(drop_p >= 0.f && drop_p < 1.f)) return HW_ERR_INVALID_ARG;
    if (mode == 3 && (!backward || !mask)) return HW_ERR_INVALID_ARG;
    if (backward ? !gout : !out) return HW_ERR_INVALID_ARG;
    if (batch == 0) return HW_OK;
    HW_CLEAR_ERR();
    hipStream_t st = (hipStream_t)stream;
#define FWD(M) hipLaunchKernelGGL((tucker_contract_fwd_kernel<M>), dim3((unsigned)((batch * dim + 255) / 256)), dim3(256), 0, st, x, w, mask, dim, drop_p, seed, offset, out, w_stride, w_index, keep_out, batch)
#define BWD(M) hipLaunchKernelGGL((tucker_contract_bwd_kernel<M>), dim3((unsigned)batch), dim3(256), 0, st, x, w, mask, gout, dim, drop_p, seed, offset, gx, gw, w_stride, w_index)
    if (!backward) {
        if (mode == 2) FWD(2);
        else if (mode == 1) FWD(1);
        else FWD(0);
    } else {
        if (mode == 3) BWD(3);
        else if (mode == 2) BWD(2);
        else if (mode == 1) BWD(1);
        else BWD(0);
    }
#undef FWD
#undef BWD
    return HW_LAUNCH_RC();
}

extern "C" int hw_tucker_contract(const float* x, const float* w, const float* mask, int64_t batch, int32_t dim,
                                  float* out, void* stream) {
    return contract_launch(mask ? 1 : 0, x, w, mask, nullptr, batch, dim, 0.f, 0, 0, out, nullptr, nullptr, false, stream);
}

extern "C" int hw_tucker_contract_grad(const float* x, const float* w, const float* mask, const float* gout,
                                       int64_t batch, int32_t dim, float* gx, float* gw, void* stream) {
    return contract_launch(mask ? 1 : 0, x, w, mask, gout, batch, dim, 0.f, 0, 0, nullptr, gx, gw, true, stream);
}

extern "C" int hw_tucker_contract_dropout(const float* x, const float* w, int64_t batch, int32_t dim, float drop_p,
                                          uint64_t seed, uint64_t offset, float* out, void* stream) {
    return contract_launch(2, x, w, nullptr, nullptr, batch, dim, drop_p, seed, offset, out, nullptr, nullptr, false, stream);
}

extern "C" int hw_tucker_contract_dropout_grad(const float* x, const float* w, const float* gout, int64_t batch,
                                               int32_t dim, float drop_p, uint64_t seed, uint64_t offset, float* gx,
                                               float* gw, void* stream) {
    return contract_launch(2, x, w, nullptr, gout, batch, dim, drop_p, seed, offset, nullptr, gx, gw, true, stream);
}

extern "C" int hw_tucker_contract_shared(const float* x, const float* w, const int64_t* w_index, const float* mask,
                                         int64_t batch, int32_t dim, float drop_p, uint64_t seed, uint64_t offset,
                                         float* out, void* stream) {
    const int mode = mask ? 1 : (drop_p > 0.f ? 2 : 0);
    return contract_launch(mode, x, w, mask, nullptr, batch, dim, mode == 2 ? drop_p : 0.f, seed, offset, out, nullptr,
                           nullptr, false, stream, true, w_index);
}

extern "C" int64_t hw_tucker_contract_shared_workspace_floats(int64_t batch, int32_t dim) {
    return batch < 0 || dim < 1 ? 0 : ((batch + TGW_CHUNK - 1) / TGW_CHUNK) * (int64_t)dim * dim;
}

static int shared_grad(int mode, const float* x, const float* w, const int64_t* w_index, const int64_t* order,
                       int64_t n_mats, const float* mask, const float* gout, int64_t batch, int32_t dim, float drop_p,
                       uint64_t seed, uint64_t offset, float* gx, float* gw, float* workspace, int64_t workspace_floats,
                       void* stream) {
    if (!x || !w || !gout || batch < 0 || dim < 1 || batch > INT32_MAX || n_mats < 1) return HW_ERR_INVALID_ARG;
    if (w_index && gw && !order) return HW_ERR_INVALID_ARG;
    if (batch == 0) return HW_OK;
    if (gx) {
        const int rc = contract_launch(mode, x, w, mask, gout, batch, dim, mode >= 2 ? drop_p : 0.f, seed, offset, nullptr,
                                       gx, nullptr, true, stream, true, w_index);
        if (rc) return rc;
    }
    if (!gw) return HW_OK;
    const int64_t chunks = (batch + TGW_CHUNK - 1) / TGW_CHUNK, n = (int64_t)dim * dim;
    HW_CLEAR_ERR();
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((dim + 3) / 4), (unsigned)chunks);
    const float p = mode >= 2 ? drop_p : 0.f;
    if (w_index) {  // one matrix per relation: atomic flushes into gw (zeroed here)
        if (hw::fill_bytes_async(gw, 0, (size_t)n_mats * n * sizeof(float), st) != hipSuccess) return HW_ERR_LAUNCH;
#define GWI(M) hipLaunchKernelGGL((tucker_gw_indexed_kernel<M>), grid, dim3(256), 0, st, x, mask, gout, order, w_index, n_mats, batch, dim, p, seed, offset, gw)
        if (mode == 3) GWI(3);
        else if (mode == 2) GWI(2);
        else if (mode == 1) GWI(1);
        else GWI(0);
#undef GWI
        return HW_LAUNCH_RC();
    }
    if (!workspace || workspace_floats < chunks * n) return HW_ERR_WORKSPACE;
#define GWS(M) hipLaunchKernelGGL((tucker_gw_shared_kernel<M>), grid, dim3(256), 0, st, x, mask, gout, batch, dim, p, seed, offset, workspace)
    if (mode == 3) GWS(3);
    else if (mode == 2) GWS(2);
    else if (mode == 1) GWS(1);
    else GWS(0);
#undef GWS
    int64_t rg = (n + 255) / 256;
    if (rg > 2048) rg = 2048;
    hipLaunchKernelGGL(tucker_gw_reduce_kernel, dim3((unsigned)rg), dim3(256), 0, st, workspace, (int)chunks, n, gw);
    return HW_LAUNCH_RC();
}

extern "C" int hw_tucker_contract_shared_grad(const float* x, const float* w, const int64_t* w_index,
                                              const int64_t* order, int64_t n_mats, const float* mask, const float* gout,
                                              int64_t batch, int32_t dim, float drop_p, uint64_t seed, uint64_t offset,
                                              float* gx, float* gw, float* workspace, int64_t workspace_floats,
                                              void* stream) {
    return shared_grad(mask ? 1 : (drop_p > 0.f ? 2 : 0), x, w, w_index, order, n_mats, mask, gout, batch, dim, drop_p, seed,
                       offset, gx, gw, workspace, workspace_floats, stream);
}

extern "C" int64_t hw_tucker_keep_words(int64_t batch, int32_t dim) {
    return batch < 0 || dim < 1 ? 0 : batch * ((((int64_t)dim + 3) / 4 + 7) / 8) * dim;
}

extern "C" int hw_tucker_contract_shared_keep(const float* x, const float* w, const int64_t* w_index, int64_t batch,
                                              int32_t dim, float drop_p, uint64_t seed, uint64_t offset, float* out,
                                              uint32_t* keep, void* stream) {
    if (!keep || !(drop_p > 0.f)) return HW_ERR_INVALID_ARG;
    return contract_launch(2, x, w, nullptr, nullptr, batch, dim, drop_p, seed, offset, out, nullptr, nullptr, false, stream,
                           true, w_index, keep);
}

extern "C" int hw_tucker_contract_shared_grad_keep(const float* x, const float* w, const int64_t* w_index,
                                                   const int64_t* order, int64_t n_mats, const uint32_t* keep,
                                                   const float* gout, int64_t batch, int32_t dim, float drop_p, float* gx,
                                                   float* gw, float* workspace, int64_t workspace_floats, void* stream) {
    if (!keep || !(drop_p > 0.f)) return HW_ERR_INVALID_ARG;
    return shared_grad(3, x, w, w_index, order, n_mats, reinterpret_cast<const float*>(keep), gout, batch, dim, drop_p, 0, 0,
                       gx, gw, workspace, workspace_floats, stream);
}

extern "C" int64_t hw_tucker_bce_workspace_floats(void) { return TB_BLOCKS; }

extern "C" int hw_tucker_sigmoid_bce_ld(const float* logits, int64_t ld, int64_t n_rows, int64_t n_cols,
                                        const uint8_t* pos_flag, float target_scale, float target_add, float* loss_out,
                                        float* pred, float* glogits, float* workspace, void* stream) {
    if (!logits || !loss_out || !workspace || n_rows < 1 || n_cols < 1 || ld < n_cols) return HW_ERR_INVALID_ARG;
    HW_CLEAR_ERR();
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = n_rows * ld;
    int64_t grid = (total / 4 + 255) / 256;
    if (grid > TB_BLOCKS) grid = TB_BLOCKS;
    if (grid < 1) grid = 1;
    if ((((uintptr_t)logits | (uintptr_t)pred | (uintptr_t)glogits) & 15) != 0) return HW_ERR_INVALID_ARG;  // float4 sweep
    const float inv = (float)(1.0 / ((double)n_rows * (double)n_cols));
    hipLaunchKernelGGL(tucker_bce_kernel, dim3((unsigned)grid), dim3(256), 0, st, logits, n_rows, n_cols, ld, pos_flag,
                       target_scale, target_add, inv, pred, glogits, workspace);
    hipLaunchKernelGGL(tucker_bce_finish_kernel, dim3(1), dim3(256), 0, st, workspace, (int)grid, inv, loss_out);
    return HW_LAUNCH_RC();
}

extern "C" int hw_tucker_sigmoid_bce(const float* logits, int64_t n_rows, int64_t n_cols, const uint8_t* pos_flag,
                                     float target_scale, float target_add, float* loss_out, float* pred,
                                     float* glogits, float* workspace, void* stream) {
    return hw_tucker_sigmoid_bce_ld(logits, n_cols, n_rows, n_cols, pos_flag, target_scale, target_add, loss_out, pred,
                                    glogits, workspace, stream);
}

extern "C" int hw_sigmoid(const float* x, int64_t n, float* y, void* stream) {
    if (!x || !y || n < 0) return HW_ERR_INVALID_ARG;
    if (n == 0) return HW_OK;
    HW_CLEAR_ERR();
    int64_t grid = (n + 255) / 256;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(sigmoid_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, n, y);
    return HW_LAUNCH_RC();
}
