// Loss reductions over score vectors + the evaluation mask / top-k.
//
// Reductions are two-stage and deterministic (fixed partition, fixed order): stage 1 writes one
// partial per block into the caller's workspace, stage 2 (one block) adds them in index order.
#include "hw_common.h"

namespace hw {

constexpr int LOSS_BLOCKS = 256;
constexpr int LOSS_THREADS = 256;

__device__ __forceinline__ float block_sum_256(float v) {
    __shared__ float red[LOSS_THREADS / HW_WAVE];
    v = group_sum<64>(v);
    const int w = threadIdx.x / HW_WAVE;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < LOSS_THREADS / HW_WAVE; ++i) t += red[i];
    return t;
}

__global__ __launch_bounds__(LOSS_THREADS) void finish_mean_kernel(const float* __restrict__ partial, int n_partial,
                                                                   float inv_n, float* __restrict__ loss_out) {
    // single block; every thread takes a fixed subset, then an ordered tree
    float v = 0.f;
    for (int i = threadIdx.x; i < n_partial; i += LOSS_THREADS) v += partial[i];
    v = block_sum_256(v);
    if (threadIdx.x == 0) loss_out[0] = v * inv_n;
}

// BCEWithLogits: l = max(x,0) - x*y + log1p(exp(-|x|)); dl/dx = sigmoid(x) - y
// pos_weight: every positive stands for pos_weight identical rows of the reference's tiled batch (1 = plain mean)
__global__ __launch_bounds__(LOSS_THREADS) void bce_logits_kernel(const float* __restrict__ x, int64_t n, int64_t n_pos,
                                                                  float pos_weight, float inv_n, float* __restrict__ gx,
                                                                  float* __restrict__ partial) {
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * LOSS_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * LOSS_THREADS) {
        const float xi = x[i];
        const float y = i < n_pos ? 1.f : 0.f;
        const float w = i < n_pos ? pos_weight : 1.f;
        const float e = expf(-fabsf(xi));
        acc = fmaf(w, fmaxf(xi, 0.f) - xi * y + log1pf(e), acc);
        if (gx) {
            const float sg = xi >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
            gx[i] = (sg - y) * (w * inv_n);
        }
    }
    acc = block_sum_256(acc);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

__device__ __forceinline__ float sigmoidf(float x) {
    const float e = expf(-fabsf(x));
    return x >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
}

// SIG: the scores go through a sigmoid first (HolE, hole.py:103); the chain rule factor is folded into gpos / gneg.
template <bool SIG>
__global__ __launch_bounds__(LOSS_THREADS) void margin_ranking_kernel(const float* __restrict__ pos,
                                                                      const float* __restrict__ neg, int64_t n,
                                                                      float margin, float inv_n, float* __restrict__ gpos,
                                                                      float* __restrict__ gneg,
                                                                      float* __restrict__ partial) {
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * LOSS_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * LOSS_THREADS) {
        float p = pos[i], q = neg[i], dp = 1.f, dq = 1.f;
        if constexpr (SIG) {
            p = sigmoidf(p), q = sigmoidf(q);
            dp = p * (1.f - p), dq = q * (1.f - q);
        }
        const float l = -(p - q) + margin;
        const bool act = l > 0.f;
        acc += act ? l : 0.f;
        if (gpos) gpos[i] = act ? -inv_n * dp : 0.f;
        if (gneg) gneg[i] = act ? inv_n * dq : 0.f;
    }
    acc = block_sum_256(acc);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

// one wave per group row [1 + K]
__global__ __launch_bounds__(256) void self_adv_kernel(const float* __restrict__ sc, int64_t n_groups, int n_cols,
                                                       float alpha, float inv_b, float* __restrict__ gsc,
                                                       float* __restrict__ group_loss) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + wave_in_block();
    if (b >= n_groups) return;
    const float* __restrict__ row = sc + b * n_cols;
    float mx = -INFINITY;
    for (int j = 1 + lane; j < n_cols; j += 64) mx = fmaxf(mx, alpha * row[j]);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
    float z = 0.f;
    for (int j = 1 + lane; j < n_cols; j += 64) z += expf(alpha * row[j] - mx);
    z = group_sum<64>(z);
    const float inv_z = z > 0.f ? 1.f / z : 0.f;
    float acc = 0.f;
    for (int j = 1 + lane; j < n_cols; j += 64) {
        const float s = row[j];
        const float p = expf(alpha * s - mx) * inv_z;
        acc -= p * log_sigmoid(-s);
        if (gsc) gsc[b * n_cols + j] = p * sigmoidf(s) * inv_b;
    }
    acc = group_sum<64>(acc);
    if (lane == 0) {
        const float sp = row[0];
        group_loss[b] = acc - log_sigmoid(sp);
        if (gsc) gsc[b * n_cols] = -(1.f - sigmoidf(sp)) * inv_b;
    }
}

__global__ __launch_bounds__(LOSS_THREADS) void sum_partial_kernel(const float* __restrict__ v, int64_t n,
                                                                   float* __restrict__ partial) {
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * LOSS_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * LOSS_THREADS)
        acc += v[i];
    acc = block_sum_256(acc);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

// ---- evaluation mask (trainer.py:731-734) ----
__global__ void mask_scores_kernel(float* __restrict__ s, int64_t n_rows, int64_t n_cols,
                                   const int64_t* __restrict__ hr, const int64_t* __restrict__ hc, int64_t n_hist,
                                   int mask_column0) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (mask_column0 && i < n_rows) s[i * n_cols] = -INFINITY;
    if (i < n_hist) {
        const int64_t r = hr[i], c = hc[i];
        if (r >= 0 && r < n_rows && c >= 0 && c < n_cols) s[r * n_cols + c] = -INFINITY;
    }
}

// ---- row-wise top-k, ordered (value desc, index asc) ----
// key = (orderable(value) << 32) | ~index : unique per column, so "next best" is simply the largest
// key strictly below the previous winner -- no marking pass.  One block per row, k rounds.
__device__ __forceinline__ uint64_t topk_key(float v, uint32_t idx) {
    uint32_t u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((uint64_t)u << 32) | (uint32_t)(~idx);
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void topk_rows_kernel(const float* __restrict__ s, int64_t n_cols, int64_t ld,
                                                            int k, float* __restrict__ out_val,
                                                            int64_t* __restrict__ out_idx) {
    __shared__ uint64_t red[THREADS / HW_WAVE];
    __shared__ uint64_t winner;
    const float* __restrict__ row = s + (int64_t)blockIdx.x * ld;
    uint64_t bound = ~0ull;  // exclusive upper bound
    for (int round = 0; round < k; ++round) {
        uint64_t best = 0;
        for (int64_t c = threadIdx.x; c < n_cols; c += THREADS) {
            const uint64_t key = topk_key(row[c], (uint32_t)c);
            if (key < bound && key > best) best = key;
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const uint64_t o = __shfl_xor(best, m, 64);
            best = o > best ? o : best;
        }
        if ((threadIdx.x & 63) == 0) red[threadIdx.x / HW_WAVE] = best;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint64_t w = red[0];
#pragma unroll
            for (int i = 1; i < THREADS / HW_WAVE; ++i) w = red[i] > w ? red[i] : w;
            winner = w;
            const int64_t o = (int64_t)blockIdx.x * k + round;
            if (w == 0) {  // fewer than k columns
                out_val[o] = -INFINITY;
                out_idx[o] = -1;
            } else {
                const uint32_t idx = ~(uint32_t)(w & 0xffffffffu);
                out_val[o] = row[idx];
                out_idx[o] = idx;
            }
        }
        __syncthreads();
        bound = winner;
        if (bound == 0) bound = 0;  // stays exhausted
    }
}


// ---- long rows: two stages ----
// The one-block-per-row form above walks the whole row k times with two barriers per walk: fine for a few thousand columns,
// 0.5 ms for 256 rows of 45 539 (too few blocks, every block ten passes deep).  For long rows the row is cut into chunks of
// TOPK_CHUNK columns; stage 1 (one block per (row, chunk)) holds its chunk in registers, reads it ONCE, and emits the chunk's
// k best keys; stage 2 (one block per row) takes the k best of the n_chunks * k keys.  max over the unique keys is
// associative, so the result is the one-stage kernel's bit for bit, ties included.
constexpr int TOPK_CHUNK_THREADS = 256;
constexpr int TOPK_CHUNK_PER = 16;
constexpr int TOPK_CHUNK = TOPK_CHUNK_THREADS * TOPK_CHUNK_PER;  // 4096 columns
// from 1 024 columns on (round 5; 8 192 before): the one-block-per-row kernel walks its row k times from memory with two barriers
// per walk, the chunk kernel reads it once into registers -- 2 048 rows x 2 985 columns: 0.047 -> 0.021 ms at k = 10, 0.143 -> 0.048 ms at k = 32
static int topk_two_stage_min_cols() {
    static const int v = [] { const char* e = getenv("HW_TOPK_TWO_STAGE_MIN"); return e ? atoi(e) : 1024; }();
    return v;
}
constexpr int TOPK_TWO_STAGE_MAX_K = 64;

// Stage 1, one block per (row, chunk): the chunk's k largest keys, descending, zeros when it has fewer.
// Not k rounds over the whole chunk (each round a block-wide arg-max and a rescan by the winner's lane: ~4 600 wave
// instructions per block, which made the first form of this kernel issue-bound at 1.5 TB/s).  Instead:
//   1. every lane's largest key (its "lane maximum"), 256 of them in LDS;
//   2. tau = the k-th largest lane maximum (one wave, k rounds over 4 keys per lane).  The chunk's k-th largest key is >= tau,
//      and the keys >= tau all sit in the k lanes whose maximum is >= tau: at most 16 k of them, typically k + a few;
//   3. every lane appends its keys >= tau to a candidate list in LDS;
//   4. one wave takes the k largest candidates in order.
// Keys are unique (value image, ~column), so "largest below the previous winner" needs no marking, and the result does not
// depend on the order the candidates were appended in.
// VEC: rows that start 16-byte aligned (a padded score matrix) are read as float4 -- lane t holds columns 4 t .. 4 t + 3 of each
// 1 024-column stretch instead of columns t, t + 256, ...: which lane holds which column changes nothing in the argument above.
template <bool VEC>
__global__ __launch_bounds__(TOPK_CHUNK_THREADS) void topk_chunk_kernel(const float* __restrict__ s, int64_t n_cols,
                                                                        int64_t ld, int k, int n_chunks,
                                                                        uint64_t* __restrict__ cand) {
    __shared__ uint64_t lane_max[TOPK_CHUNK_THREADS];
    __shared__ uint64_t list[TOPK_CHUNK_PER * TOPK_TWO_STAGE_MAX_K];
    __shared__ uint64_t tau_sh;
    __shared__ int n_list;
    const int chunk = blockIdx.x % n_chunks;
    const int64_t r = blockIdx.x / n_chunks;
    const float* __restrict__ row = s + r * ld;
    const int64_t base = (int64_t)chunk * TOPK_CHUNK + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t key[TOPK_CHUNK_PER];  // 0 = no column here (a real key has a non-zero low word unless its column is 2^32 - 1)
    uint64_t mine = 0;
    if constexpr (VEC) {
        static_assert(TOPK_CHUNK_PER % 4 == 0, "quads");
#pragma unroll
        for (int i = 0; i < TOPK_CHUNK_PER / 4; ++i) {
            const int64_t c0 = (int64_t)chunk * TOPK_CHUNK + (int64_t)i * (4 * TOPK_CHUNK_THREADS) + 4 * threadIdx.x;
            hw::f32x4 x = {0.f, 0.f, 0.f, 0.f};
            if (c0 < n_cols) x = *reinterpret_cast<const hw::f32x4*>(row + c0);  // ld >= n_cols rounded up to 4 (launcher)
            const float xv[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                key[4 * i + e] = c0 + e < n_cols ? topk_key(xv[e], (uint32_t)(c0 + e)) : 0ull;
                mine = key[4 * i + e] > mine ? key[4 * i + e] : mine;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < TOPK_CHUNK_PER; ++i) {
            const int64_t c = base + (int64_t)i * TOPK_CHUNK_THREADS;
            key[i] = c < n_cols ? topk_key(row[c], (uint32_t)c) : 0ull;
            mine = key[i] > mine ? key[i] : mine;
        }
    }
    lane_max[threadIdx.x] = mine;
    if (threadIdx.x == 0) n_list = 0;
    __syncthreads();
    if (wave == 0) {  // tau: k rounds of "largest lane maximum below the previous one"
        uint64_t v[TOPK_CHUNK_THREADS / 64];
#pragma unroll
        for (int j = 0; j < TOPK_CHUNK_THREADS / 64; ++j) v[j] = lane_max[lane + 64 * j];
        uint64_t bound = ~0ull;
        for (int round = 0; round < k && bound != 0; ++round) {
            uint64_t best = 0;
#pragma unroll
            for (int j = 0; j < TOPK_CHUNK_THREADS / 64; ++j) best = (v[j] < bound && v[j] > best) ? v[j] : best;
            bound = wave_max_u64(best);  // 0 once the lane maxima are used up: every key of the chunk is a candidate then
        }
        if (lane == 0) tau_sh = bound;
    }
    __syncthreads();
    const uint64_t tau = tau_sh;
#pragma unroll
    for (int i = 0; i < TOPK_CHUNK_PER; ++i)
        if (key[i] != 0 && key[i] >= tau) list[atomicAdd(&n_list, 1)] = key[i];
    __syncthreads();
    if (wave == 0) {
        const int n = n_list;  // <= 16 k when tau != 0, <= the chunk's columns (< k lanes had any) otherwise
        uint64_t* __restrict__ out = cand + ((int64_t)blockIdx.x) * k;
        uint64_t bound = ~0ull;
        for (int round = 0; round < k; ++round) {
            uint64_t best = 0;
            for (int j = lane; j < n; j += 64) {
                const uint64_t x = list[j];
                best = (x < bound && x > best) ? x : best;
            }
            best = wave_max_u64(best);
            if (lane == 0) out[round] = best;
            bound = best;  // 0 = used up: the remaining rounds write zeros
            if (best == 0) {
                for (int q = round + 1 + lane; q < k; q += 64) out[q] = 0;
                break;
            }
        }
    }
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void topk_merge_kernel(const float* __restrict__ s, int64_t ld,
                                                             const uint64_t* __restrict__ cand, int n_cand, int k,
                                                             float* __restrict__ out_val, int64_t* __restrict__ out_idx) {
    __shared__ uint64_t red[THREADS / HW_WAVE];
    __shared__ uint64_t winner;
    const float* __restrict__ row = s + (int64_t)blockIdx.x * ld;
    const uint64_t* __restrict__ keys = cand + (int64_t)blockIdx.x * n_cand;
    uint64_t bound = ~0ull;
    for (int round = 0; round < k; ++round) {
        uint64_t best = 0;
        for (int c = threadIdx.x; c < n_cand; c += THREADS) {
            const uint64_t key = keys[c];
            if (key < bound && key > best) best = key;
        }
        best = wave_max_u64(best);
        if (THREADS > HW_WAVE) {
            if ((threadIdx.x & 63) == 0) red[threadIdx.x / HW_WAVE] = best;
            __syncthreads();
            if (threadIdx.x == 0) {
                uint64_t w = red[0];
#pragma unroll
                for (int i = 1; i < THREADS / HW_WAVE; ++i) w = red[i] > w ? red[i] : w;
                winner = w;
            }
            __syncthreads();
            best = winner;
        }
        if (threadIdx.x == 0) {
            const int64_t o = (int64_t)blockIdx.x * k + round;
            if (best == 0) {  // fewer than k columns
                out_val[o] = -INFINITY;
                out_idx[o] = -1;
            } else {
                const uint32_t idx = ~(uint32_t)(best & 0xffffffffu);
                out_val[o] = row[idx];
                out_idx[o] = idx;
            }
        }
        bound = best;
    }
}

}  // namespace hw

using namespace hw;

extern "C" int64_t hw_loss_workspace_floats(void) { return LOSS_BLOCKS; }

static int loss_grid(int64_t n) {
    int64_t b = (n + LOSS_THREADS - 1) / LOSS_THREADS;
    return (int)(b < 1 ? 1 : (b > LOSS_BLOCKS ? LOSS_BLOCKS : b));
}

extern "C" int hw_loss_bce_logits_weighted(const float* x, int64_t n, int64_t n_pos, float pos_weight, float* loss_out,
                                           float* gx, float* workspace, void* stream) {
    if (!x || !loss_out || !workspace || n <= 0 || n_pos < 0 || n_pos > n || !(pos_weight > 0.f))
        return HW_ERR_INVALID_ARG;
    HW_CLEAR_ERR();
    hipStream_t st = (hipStream_t)stream;
    const int grid = loss_grid(n);
    const float inv_n = (float)(1.0 / ((double)pos_weight * (double)n_pos + (double)(n - n_pos)));
    hipLaunchKernelGGL(bce_logits_kernel, dim3(grid), dim3(LOSS_THREADS), 0, st, x, n, n_pos, pos_weight, inv_n, gx,
                       workspace);
    hipLaunchKernelGGL(finish_mean_kernel, dim3(1), dim3(LOSS_THREADS), 0, st, workspace, grid, inv_n, loss_out);
    return HW_LAUNCH_RC();
}

extern "C" int hw_loss_bce_logits(const float* x, int64_t n, int64_t n_pos, float* loss_out, float* gx,
                                  float* workspace, void* stream) {
    return hw_loss_bce_logits_weighted(x, n, n_pos, 1.0f, loss_out, gx, workspace, stream);
}

static int margin_ranking_launch(bool sig, const float* pos, const float* neg, int64_t n, float margin, float* loss_out,
                                 float* gpos, float* gneg, float* workspace, void* stream) {
    if (!pos || !neg || !loss_out || !workspace || n <= 0) return HW_ERR_INVALID_ARG;
    HW_CLEAR_ERR();
    hipStream_t st = (hipStream_t)stream;
    const int grid = loss_grid(n);
    const float inv_n = 1.0f / (float)n;
    if (sig)
        hipLaunchKernelGGL(margin_ranking_kernel<true>, dim3(grid), dim3(LOSS_THREADS), 0, st, pos, neg, n, margin, inv_n,
                           gpos, gneg, workspace);
    else
        hipLaunchKernelGGL(margin_ranking_kernel<false>, dim3(grid), dim3(LOSS_THREADS), 0, st, pos, neg, n, margin, inv_n,
                           gpos, gneg, workspace);
    hipLaunchKernelGGL(finish_mean_kernel, dim3(1), dim3(LOSS_THREADS), 0, st, workspace, grid, inv_n, loss_out);
    return HW_LAUNCH_RC();
}

extern "C" int hw_loss_margin_ranking(const float* pos, const float* neg, int64_t n, float margin, float* loss_out,
                                      float* gpos, float* gneg, float* workspace, void* stream) {
    return margin_ranking_launch(false, pos, neg, n, margin, loss_out, gpos, gneg, workspace, stream);
}

extern "C" int hw_loss_margin_ranking_sigmoid(const float* pos, const float* neg, int64_t n, float margin,
                                              float* loss_out, float* gpos, float* gneg, float* workspace,
                                              void* stream) {
    return margin_ranking_launch(true, pos, neg, n, margin, loss_out, gpos, gneg, workspace, stream);
}

extern "C" int hw_loss_self_adversarial(const float* scores, int64_t n_groups, int64_t n_cols, float alpha,
                                        float* loss_out, float* gscores, float* workspace, void* stream) {
    // workspace: LOSS_BLOCKS partials followed by n_groups per-group losses
    if (!scores || !loss_out || !workspace || n_groups <= 0 || n_cols < 2 || n_cols > INT32_MAX) return HW_ERR_INVALID_ARG;
    HW_CLEAR_ERR();
    hipStream_t st = (hipStream_t)stream;
    float* group_loss = workspace + LOSS_BLOCKS;
    const float inv_b = 1.0f / (float)n_groups;
    hipLaunchKernelGGL(self_adv_kernel, dim3((unsigned)((n_groups + 3) / 4)), dim3(256), 0, st, scores, n_groups,
                       (int)n_cols, alpha, inv_b, gscores, group_loss);
    if (n_groups <= 65536) {
        // few groups: one block adds the per-group losses in a fixed order (saves a launch on the step's critical path)
        hipLaunchKernelGGL(finish_mean_kernel, dim3(1), dim3(LOSS_THREADS), 0, st, group_loss, (int)n_groups, inv_b,
                           loss_out);
    } else {
        const int grid = loss_grid(n_groups);
        hipLaunchKernelGGL(sum_partial_kernel, dim3(grid), dim3(LOSS_THREADS), 0, st, group_loss, n_groups, workspace);
        hipLaunchKernelGGL(finish_mean_kernel, dim3(1), dim3(LOSS_THREADS), 0, st, workspace, grid, inv_b, loss_out);
    }
    return HW_LAUNCH_RC();
}

extern "C" int hw_mask_scores_ex(float* scores, int64_t n_rows, int64_t n_cols, const int64_t* hist_rows,
                                 const int64_t* hist_cols, int64_t n_hist, int32_t mask_column0, void* stream) {
    if (!scores || n_rows < 0 || n_cols < 1 || n_hist < 0 || (n_hist > 0 && (!hist_rows || !hist_cols)))
        return HW_ERR_INVALID_ARG;
    const int64_t n = (mask_column0 && n_rows > n_hist) ? n_rows : n_hist;
    if (n == 0 || n_rows == 0) return HW_OK;
    HW_CLEAR_ERR();
    hipLaunchKernelGGL(mask_scores_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, scores,
                       n_rows, n_cols, hist_rows, hist_cols, n_hist, (int)(mask_column0 != 0));
    return HW_LAUNCH_RC();
}

extern "C" int hw_mask_scores(float* scores, int64_t n_rows, int64_t n_cols, const int64_t* hist_rows,
                              const int64_t* hist_cols, int64_t n_hist, void* stream) {
    return hw_mask_scores_ex(scores, n_rows, n_cols, hist_rows, hist_cols, n_hist, 1, stream);
}

extern "C" int hw_topk_rows(const float* scores, int64_t n_rows, int64_t n_cols, int64_t ld, int32_t k, float* out_val,
                            int64_t* out_idx, void* stream) {
    if (!scores || !out_val || !out_idx || n_rows < 0 || n_cols < 1 || ld < n_cols || k < 1 || k > 1024)
        return HW_ERR_INVALID_ARG;
    if (n_cols > 0xffffffffll || n_rows > INT32_MAX) return HW_ERR_INVALID_ARG;
    if (n_rows == 0) return HW_OK;
    HW_CLEAR_ERR();
    hipStream_t st = (hipStream_t)stream;
    if (n_cols <= 1024)
        hipLaunchKernelGGL((topk_rows_kernel<64>), dim3((unsigned)n_rows), dim3(64), 0, st, scores, n_cols, ld, k,
                           out_val, out_idx);
    else
        hipLaunchKernelGGL((topk_rows_kernel<256>), dim3((unsigned)n_rows), dim3(256), 0, st, scores, n_cols, ld, k,
                           out_val, out_idx);
    return HW_LAUNCH_RC();
}

static bool topk_two_stage(int64_t n_cols, int32_t k) {
    return n_cols >= topk_two_stage_min_cols() && k <= TOPK_TWO_STAGE_MAX_K;
}

extern "C" int64_t hw_topk_rows_workspace_bytes(int64_t n_rows, int64_t n_cols, int32_t k) {
    if (n_rows <= 0 || n_cols < 1 || k < 1 || !topk_two_stage(n_cols, k)) return 0;
    const int64_t n_chunks = (n_cols + TOPK_CHUNK - 1) / TOPK_CHUNK;
    return n_rows * n_chunks * k * (int64_t)sizeof(uint64_t);
}

extern "C" int hw_topk_rows_ws(const float* scores, int64_t n_rows, int64_t n_cols, int64_t ld, int32_t k, float* out_val,
                               int64_t* out_idx, void* workspace, int64_t workspace_bytes, void* stream) {
    if (n_rows >= 0 && n_cols >= 1 && k >= 1 && !topk_two_stage(n_cols, k))
        return hw_topk_rows(scores, n_rows, n_cols, ld, k, out_val, out_idx, stream);
    if (!scores || !out_val || !out_idx || n_rows < 0 || n_cols < 1 || ld < n_cols || k < 1) return HW_ERR_INVALID_ARG;
    if (n_cols > 0xffffffffll || n_rows > INT32_MAX) return HW_ERR_INVALID_ARG;
    if (n_rows == 0) return HW_OK;
    const int64_t n_chunks = (n_cols + TOPK_CHUNK - 1) / TOPK_CHUNK;
    if (n_rows * n_chunks > INT32_MAX || n_chunks * k > INT32_MAX) return HW_ERR_INVALID_ARG;
    if (!workspace || ((uintptr_t)workspace & 7) || workspace_bytes < hw_topk_rows_workspace_bytes(n_rows, n_cols, k))
        return HW_ERR_WORKSPACE;
    HW_CLEAR_ERR();
    hipStream_t st = (hipStream_t)stream;
    uint64_t* cand = (uint64_t*)workspace;
    if (((uintptr_t)scores & 15) == 0 && (ld & 3) == 0 && ld >= ((n_cols + 3) & ~(int64_t)3))
        hipLaunchKernelGGL((topk_chunk_kernel<true>), dim3((unsigned)(n_rows * n_chunks)), dim3(TOPK_CHUNK_THREADS), 0, st,
                           scores, n_cols, ld, (int)k, (int)n_chunks, cand);
    else
        hipLaunchKernelGGL((topk_chunk_kernel<false>), dim3((unsigned)(n_rows * n_chunks)), dim3(TOPK_CHUNK_THREADS), 0, st,
                           scores, n_cols, ld, (int)k, (int)n_chunks, cand);
    const int n_cand = (int)(n_chunks * k);
    if (n_cand <= 512)
        hipLaunchKernelGGL((topk_merge_kernel<64>), dim3((unsigned)n_rows), dim3(64), 0, st, scores, ld, cand, n_cand,
                           (int)k, out_val, out_idx);
    else
        hipLaunchKernelGGL((topk_merge_kernel<256>), dim3((unsigned)n_rows), dim3(256), 0, st, scores, ld, cand, n_cand,
                           (int)k, out_val, out_idx);
    return HW_LAUNCH_RC();
}
