// PGPR policy-gradient training (REINFORCE with a critic baseline) on the device.
//
// Restates /root/reference/hopwise/model/knowledge_aware_recommender/pgpr.py:
//   batch_action_mask :384-396   action dropout (keep int((n-1)(1-p)) random non-self-loop actions when n >= 5)
//   select_action     :120-135   masked softmax -> Categorical sample, log-prob, entropy
//   _get_reward       :339-357   soft reward max((U[u]+R_ui).E[item] / max_i(...), 0) on pattern-matching paths
//   update            :137-173   discounted returns, actor / critic / entropy losses (float64 like the reference)
// plus the hand-written backward of the policy head and of ELU/ReLU (+dropout), so that no autograd graph is
// kept across the episode: every step's activations are stacked and the weight gradients come from one GEMM
// pass over all steps (hw_gemm_f32).
//
// One wave per row everywhere: act_dim is a few hundred (max_acts + 1 = 251), so a row is 4 elements per lane.
#include "hw_common.h"

namespace hw {

constexpr int TRAIN_MAX_ACT_DIM = 4096;
constexpr float PROB_EPS = 1.1920928955078125e-07f;  // torch.finfo(float32).eps: clamp_probs in Categorical

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, HW_WAVE));
    return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, HW_WAVE);
    return v;
}

// ---- batch_action_mask ---------------------------------------------------------------------------------------
// Uniform random subset without replacement = the keep_size smallest of i.i.d. keys.  key(row, j) =
// philox(seed, offset, row * act_dim + j); ties broken by index.  mask[row, 0] (self loop) is always kept.
__global__ __launch_bounds__(64) void action_dropout_kernel(const int32_t* __restrict__ n_acts,
                                                            const int32_t* __restrict__ n_rows, int32_t act_dim,
                                                            double dropout, uint64_t seed, uint64_t offset,
                                                            uint8_t* __restrict__ mask) {
    offset = philox_offset(offset);
    __shared__ uint32_t keys[TRAIN_MAX_ACT_DIM];
    const int64_t row = blockIdx.x;
    if (n_rows && row >= n_rows[0]) return;
    const int lane = threadIdx.x;
    const int n = min(n_acts[row], act_dim);
    uint8_t* out = mask + row * act_dim;
    if (!(dropout > 0.0) || n < 5) {
        for (int j = lane; j < act_dim; j += HW_WAVE) out[j] = j < n ? 1 : 0;
        return;
    }
    const int keep = (int)((double)(n - 1) * (1.0 - dropout));
    for (int j = lane; j < n; j += HW_WAVE) keys[j] = philox_bits(seed, offset, (uint64_t)row * act_dim + j);
    __syncthreads();
    for (int j = lane; j < act_dim; j += HW_WAVE) {
        uint8_t m = 0;
        if (j == 0) m = 1;
        else if (j < n) {
            const uint32_t kj = keys[j];
            int rank = 0;
            for (int i = 1; i < n; ++i) {
                const uint32_t ki = keys[i];
                rank += (ki < kj) || (ki == kj && i < j);
            }
            m = rank < keep;
        }
        out[j] = m;
    }
}

// ---- select_action --------------------------------------------------------------------------------------------
// head[row, 0:act_dim] = actor logits, head[row, act_dim] = critic value (one fused GEMM writes both).
// probs = softmax(where(mask, logits, -inf)); act = inverse-CDF draw (float64 running sum, fixed order) with
// u = philox(seed, offset, row) >> 8) / 2^24, or forced_acts[row] when given (replay); logp = log(clamp(p[act]));
// entropy = -sum p log(clamp(p)) -- torch.distributions.Categorical(probs=...) semantics.
__global__ __launch_bounds__(64) void policy_sample_kernel(const float* __restrict__ head, int64_t ld, int32_t act_dim,
                                                           const uint8_t* __restrict__ mask,
                                                           const int32_t* __restrict__ n_rows,
                                                           const int32_t* __restrict__ forced_acts, uint64_t seed,
                                                           uint64_t offset, float* __restrict__ probs,
                                                           int32_t* __restrict__ acts, float* __restrict__ logp,
                                                           float* __restrict__ entropy, float* __restrict__ value) {
    if (!forced_acts) offset = philox_offset(offset);
    __shared__ float e[TRAIN_MAX_ACT_DIM];
    const int64_t row = blockIdx.x;
    if (n_rows && row >= n_rows[0]) return;
    const int lane = threadIdx.x;
    const float* z = head + row * ld;
    const uint8_t* mk = mask + row * act_dim;
    float mx = -INFINITY;
    for (int j = lane; j < act_dim; j += HW_WAVE) {
        const float v = mk[j] ? z[j] : -INFINITY;
        e[j] = v;
        mx = fmaxf(mx, v);
    }
    mx = wave_max(mx);
    float s = 0.f;
    for (int j = lane; j < act_dim; j += HW_WAVE) {
        const float v = e[j] == -INFINITY ? 0.f : expf(e[j] - mx);
        e[j] = v;
        s += v;
    }
    s = group_sum<64>(s);
    float ent = 0.f;
    for (int j = lane; j < act_dim; j += HW_WAVE) {
        const float p = e[j] / s;
        e[j] = p;
        probs[row * act_dim + j] = p;
        ent -= p * logf(fminf(fmaxf(p, PROB_EPS), 1.f - PROB_EPS));
    }
    ent = group_sum<64>(ent);
    __syncthreads();
    // inverse CDF: chunks of 64 in index order, float64 prefix inside the chunk by shuffles
    int act;
    if (forced_acts) {
        act = forced_acts[row];
    } else {
        double total = 0.0;
        for (int j = lane; j < act_dim; j += HW_WAVE) total += (double)e[j];
        total = wave_sum_f64(total);
        const double u = (double)(philox_bits(seed, offset, (uint64_t)row) >> 8) * (1.0 / 16777216.0);
        const double thr = u * total;
        double base = 0.0;
        act = -1;
        int last_valid = 0;
        for (int j0 = 0; j0 < act_dim; j0 += HW_WAVE) {
            const int j = j0 + lane;
            const double p = j < act_dim ? (double)e[j] : 0.0;
            double incl = p;
#pragma unroll
            for (int m = 1; m < HW_WAVE; m <<= 1) {
                const double o = __shfl_up(incl, m, HW_WAVE);
                if (lane >= m) incl += o;
            }
            incl += base;
            const unsigned long long hit = __ballot(p > 0.0 && incl > thr);
            const unsigned long long pos = __ballot(p > 0.0);
            if (pos) last_valid = j0 + 63 - __builtin_clzll(pos);
            if (hit) {
                act = j0 + __builtin_ctzll(hit);
                break;
            }
            base = __shfl(incl, HW_WAVE - 1, HW_WAVE);
        }
        if (act < 0) act = last_valid;  // rounding at u -> 1
    }
    if (lane == 0) {
        // [CAVEAT] of the reference (pgpr.py:129-131): an action outside the mask falls back to action 0
        if (act < 0 || act >= act_dim || !mk[act]) act = 0;
        acts[row] = act;
        logp[row] = logf(fminf(fmaxf(e[act], PROB_EPS), 1.f - PROB_EPS));
        entropy[row] = ent;
        value[row] = z[act_dim];
    }
}

// ---- d loss / d head ------------------------------------------------------------------------------------------
// loss = sum_rows [ -coef_actor[row] * logp + coef_entropy * sum_j p_j log(clamp p_j) ] + critic term;
// dhead[row, j] = p_j (g_j - sum_i p_i g_i), g_j = coef_entropy (log clamp p_j + inrange_j) - [j == act] coef_actor
// inrange_act / p_act;  dhead[row, act_dim] = dvalue[row].
__global__ __launch_bounds__(64) void policy_grad_kernel(const float* __restrict__ probs, int32_t act_dim,
                                                         const int32_t* __restrict__ acts,
                                                         const float* __restrict__ coef_actor, float coef_entropy,
                                                         const float* __restrict__ dvalue, int64_t n_rows,
                                                         float* __restrict__ dhead, int64_t ld,
                                                         const float* __restrict__ upstream) {
    const int64_t row = blockIdx.x;
    if (row >= n_rows) return;
    const int lane = threadIdx.x;
    const float* p = probs + row * act_dim;
    const int a = acts[row];
    const float ca = coef_actor[row];
    // d loss / d head is linear in the upstream gradient of the scalar loss: scaled HERE, every gradient behind it (three
    // GEMM passes, three column sums) comes out scaled -- instead of eight `grad * upstream` launches at the end
    const float up = upstream ? upstream[0] : 1.f;
    float gbar = 0.f;
    for (int j = lane; j < act_dim; j += HW_WAVE) {
        const float pj = p[j];
        const bool inr = pj >= PROB_EPS && pj <= 1.f - PROB_EPS;
        float g = coef_entropy * (logf(fminf(fmaxf(pj, PROB_EPS), 1.f - PROB_EPS)) + (inr ? 1.f : 0.f));
        if (j == a && inr) g -= ca / pj;
        gbar += pj * g;
    }
    gbar = group_sum<64>(gbar);
    for (int j = lane; j < act_dim; j += HW_WAVE) {
        const float pj = p[j];
        const bool inr = pj >= PROB_EPS && pj <= 1.f - PROB_EPS;
        float g = coef_entropy * (logf(fminf(fmaxf(pj, PROB_EPS), 1.f - PROB_EPS)) + (inr ? 1.f : 0.f));
        if (j == a && inr) g -= ca / pj;
        dhead[row * ld + j] = up * (pj * (g - gbar));
    }
    if (lane == 0) dhead[row * ld + act_dim] = up * dvalue[row];
}

// ---- backward of y = dropout(act(pre)) from y alone ---------------------------------------------------------------
// y = m * f(pre), m in {0, 1/(1-p)} (or 1): f(pre) = y / m; ELU'(pre) = f > 0 ? 1 : f + 1; ReLU' = f > 0.
__global__ __launch_bounds__(256) void act_dropout_backward_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                                  int64_t m_rows, int64_t n, int32_t act,
                                                                  const float* __restrict__ drop_mask, int32_t drop_mode,
                                                                  uint64_t seed, uint64_t offset, float* __restrict__ dx) {
    if (drop_mode == 2) offset = philox_offset(offset);
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m_rows * n) return;
    float m = 1.f;
    if (drop_mode == 1) m = drop_mask[i];
    else if (drop_mode == 2) m = (philox_bits(seed, offset, (uint64_t)i) & 1u) ? 2.f : 0.f;
    float g = 0.f;
    if (m != 0.f) {
        const float f = y[i] / m;
        float d = 1.f;
        if (act == 1) d = f > 0.f ? 1.f : f + 1.f;
        else if (act == 2) d = f > 0.f ? 1.f : 0.f;
        g = dy[i] * m * d;
    }
    dx[i] = g;
}

// column sums of a [m, n] matrix (bias gradients): block per 64 columns, fixed-order tree -> deterministic.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int64_t ld, int64_t m, int64_t n,
                                                     float* __restrict__ out) {
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int r0 = threadIdx.x >> 6;
    float s = 0.f;
    if (c < n)
        for (int64_t r = r0; r < m; r += 4) s += x[r * ld + c];
    part[r0][threadIdx.x & 63] = s;
    __syncthreads();
    if (r0 == 0 && c < n) out[c] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// The same in two stages for tall matrices: the one-stage kernel has n / 64 blocks (8 for a 512-wide layer) each walking all
// m rows -- 360 us for the [6 144, 512] gradient of a 2 048-user PGPR episode, a third of the training step.  Stage 1: block
// (column block, row chunk) sums its chunk's rows; stage 2: the chunks' partial sums in fixed order.  Deterministic.
constexpr int COLSUM_MAX_CHUNKS = 256;
__host__ __device__ __forceinline__ int64_t colsum_chunks(int64_t m) {
    const int64_t c = (m + 31) / 32;  // at least 32 rows per chunk
    return c < 1 ? 1 : (c > COLSUM_MAX_CHUNKS ? COLSUM_MAX_CHUNKS : c);
}

__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, int64_t ld, int64_t m, int64_t n,
                                                             int64_t rows_per_chunk, float* __restrict__ part_out) {
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int r0 = threadIdx.x >> 6;
    const int64_t lo = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t hi = lo + rows_per_chunk < m ? lo + rows_per_chunk : m;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;  // four loads in flight per thread; combined in a fixed order
    if (c < n) {
        int64_t r = lo + r0;
        for (; r + 12 < hi; r += 16) {
            s0 += x[r * ld + c];
            s1 += x[(r + 4) * ld + c];
            s2 += x[(r + 8) * ld + c];
            s3 += x[(r + 12) * ld + c];
        }
        for (; r < hi; r += 4) s0 += x[r * ld + c];
    }
    part[r0][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (r0 == 0 && c < n)
        part_out[(int64_t)blockIdx.y * n + c] =
            (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// ---- _get_reward ----------------------------------------------------------------------------------------------------
// patterns: int32 [n_patterns, 4] = (length incl. the leading self loop, rel, rel, rel); a path matches when its
// relation sequence equals one pattern of the same length.  score_entity [n_batch, n_entities] is the float64 table
// hw_pgpr_user_scores builds: its first n_items columns are (U[u] + R_ui) . E[item] (pgpr.py:85-88, 351-355).
__global__ __launch_bounds__(64) void reward_kernel(const int32_t* __restrict__ paths, int64_t n_rows,
                                                    const int32_t* __restrict__ patterns, int32_t n_patterns,
                                                    const double* __restrict__ score_entity, int64_t n_entities,
                                                    int64_t n_items, const double* __restrict__ scales,
                                                    double* __restrict__ reward) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    const int32_t* p = paths + row * HW_PGPR_ROW_INTS;
    const int n_nodes = p[1];
    double r = 0.0;
    if (n_nodes > 2) {
        bool ok = false;
        for (int q = 0; q < n_patterns && !ok; ++q) {
            const int32_t* pat = patterns + q * 4;
            if (pat[0] != n_nodes) continue;
            bool same = true;
            for (int h = 1; h < n_nodes; ++h) same = same && pat[h] == p[2 + 3 * (h - 1)];
            ok = same;
        }
        const int ctype = p[2 + 3 * (n_nodes - 2) + 1], cid = p[2 + 3 * (n_nodes - 2) + 2];
        if (ok && ctype == 1 && cid < n_items) {
            const int b = p[HW_PGPR_ROW_INTS - 1];
            const double s = score_entity[(int64_t)b * n_entities + cid] / scales[b];
            r = s > 0.0 ? s : 0.0;
        }
    }
    reward[row] = r;
}

// u_p_scales (pgpr.py:85-88) for the batch users: max over the item columns of the float64 score table.
__global__ __launch_bounds__(256) void row_max_f64_kernel(const double* __restrict__ x, int64_t ld, int64_t n_cols,
                                                          double* __restrict__ out) {
    __shared__ double part[256];
    const double* r = x + (int64_t)blockIdx.x * ld;
    double m = -INFINITY;
    for (int64_t j = threadIdx.x; j < n_cols; j += 256) m = fmax(m, r[j]);
    part[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) part[threadIdx.x] = fmax(part[threadIdx.x], part[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = part[0];
}

// ---- update() -------------------------------------------------------------------------------------------------------
// rewards float64 [T, B]; values / logp / entropy float32 [T, B].  Discounted returns in place of `rewards`
// (pgpr.py:150-151), then loss parts out[0..3] = (loss, actor, critic, entropy) as float64 and the backward
// coefficients coef_actor = adv / B, dvalue = -2 adv / B (float32 [T, B]).  One block; fixed-order reduction.
__global__ __launch_bounds__(256) void reinforce_loss_kernel(double* __restrict__ rewards, const float* __restrict__ values,
                                                             const float* __restrict__ logp,
                                                             const float* __restrict__ entropy, int32_t T, int64_t B,
                                                             double gamma, double weight, double* __restrict__ out,
                                                             float* __restrict__ coef_actor, float* __restrict__ dvalue) {
    __shared__ double part[3][256];
    double sa = 0.0, sc = 0.0, se = 0.0;
    for (int64_t b = threadIdx.x; b < B; b += 256) {
        for (int t = T - 2; t >= 0; --t) rewards[(int64_t)t * B + b] += gamma * rewards[(int64_t)(t + 1) * B + b];
        for (int t = 0; t < T; ++t) {
            const int64_t i = (int64_t)t * B + b;
            const double adv = rewards[i] - (double)values[i];
            sa += -(double)logp[i] * adv;
            sc += adv * adv;
            se += -(double)entropy[i];
            coef_actor[i] = (float)(adv / (double)B);
            dvalue[i] = (float)(-2.0 * adv / (double)B);
        }
    }
    part[0][threadIdx.x] = sa, part[1][threadIdx.x] = sc, part[2][threadIdx.x] = se;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
            for (int q = 0; q < 3; ++q) part[q][threadIdx.x] += part[q][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double a = part[0][0] / (double)B, c = part[1][0] / (double)B, e = part[2][0] / (double)B;
        out[0] = a + c + weight * e, out[1] = a, out[2] = c, out[3] = e;
    }
}

}  // namespace hw

extern "C" {

int hw_pgpr_action_dropout(const int32_t* n_acts, const int32_t* n_rows, int64_t max_rows, int32_t act_dim,
                           double dropout, uint64_t seed, uint64_t offset, uint8_t* mask, void* stream) {
    if (!n_acts || !mask || act_dim < 1 || act_dim > hw::TRAIN_MAX_ACT_DIM) return HW_ERR_INVALID_ARG;
    if (max_rows == 0) return HW_OK;
    HW_CLEAR_ERR();
    hipLaunchKernelGGL(hw::action_dropout_kernel, dim3((unsigned)max_rows), dim3(64), 0, (hipStream_t)stream, n_acts,
                       n_rows, act_dim, dropout, seed, offset, mask);
    return HW_LAUNCH_RC();
}

int hw_pgpr_policy_sample(const float* head, int64_t ld, int32_t act_dim, const uint8_t* mask, const int32_t* n_rows,
                          int64_t max_rows, const int32_t* forced_acts, uint64_t seed, uint64_t offset, float* probs,
                          int32_t* acts, float* logp, float* entropy, float* value, void* stream) {
    if (!head || !mask || !probs || !acts || !logp || !entropy || !value) return HW_ERR_INVALID_ARG;
    if (act_dim < 1 || act_dim > hw::TRAIN_MAX_ACT_DIM || ld < act_dim + 1) return HW_ERR_INVALID_ARG;
    if (max_rows == 0) return HW_OK;
    HW_CLEAR_ERR();
    hipLaunchKernelGGL(hw::policy_sample_kernel, dim3((unsigned)max_rows), dim3(64), 0, (hipStream_t)stream, head, ld,
                       act_dim, mask, n_rows, forced_acts, seed, offset, probs, acts, logp, entropy, value);
    return HW_LAUNCH_RC();
}

int hw_pgpr_policy_grad_scaled(const float* probs, int32_t act_dim, const int32_t* acts, const float* coef_actor,
                               float coef_entropy, const float* dvalue, int64_t n_rows, float* dhead, int64_t ld,
                               const float* upstream, void* stream) {
    if (!probs || !acts || !coef_actor || !dvalue || !dhead || act_dim < 1 || ld < act_dim + 1) return HW_ERR_INVALID_ARG;
    if (n_rows == 0) return HW_OK;
    HW_CLEAR_ERR();
    hipLaunchKernelGGL(hw::policy_grad_kernel, dim3((unsigned)n_rows), dim3(64), 0, (hipStream_t)stream, probs, act_dim,
                       acts, coef_actor, coef_entropy, dvalue, n_rows, dhead, ld, upstream);
    return HW_LAUNCH_RC();
}

int hw_pgpr_policy_grad(const float* probs, int32_t act_dim, const int32_t* acts, const float* coef_actor,
                        float coef_entropy, const float* dvalue, int64_t n_rows, float* dhead, int64_t ld,
                        void* stream) {
    return hw_pgpr_policy_grad_scaled(probs, act_dim, acts, coef_actor, coef_entropy, dvalue, n_rows, dhead, ld, nullptr, stream);
}

int hw_act_dropout_backward(const float* dy, const float* y, int64_t m_rows, int64_t n, int32_t act,
                            const float* drop_mask, int32_t drop_mode, uint64_t seed, uint64_t offset, float* dx,
                            void* stream) {
    if (!dy || !y || !dx || act < 0 || act > 2 || drop_mode < 0 || drop_mode > 2) return HW_ERR_INVALID_ARG;
    if (drop_mode == 1 && !drop_mask) return HW_ERR_INVALID_ARG;
    const int64_t total = m_rows * n;
    if (total == 0) return HW_OK;
    HW_CLEAR_ERR();
    hipLaunchKernelGGL(hw::act_dropout_backward_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, dy, y, m_rows, n, act, drop_mask, drop_mode, seed, offset, dx);
    return HW_LAUNCH_RC();
}

int hw_colsum_f32(const float* x, int64_t ld, int64_t m_rows, int64_t n, float* out, void* stream) {
    if (!x || !out || ld < n) return HW_ERR_INVALID_ARG;
    if (n == 0) return HW_OK;
    HW_CLEAR_ERR();
    hipLaunchKernelGGL(hw::colsum_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, (hipStream_t)stream, x, ld,
                       m_rows, n, out);
    return HW_LAUNCH_RC();
}

int64_t hw_colsum_workspace_floats(int64_t m_rows, int64_t n) {
    return (m_rows <= 0 || n <= 0) ? 0 : hw::colsum_chunks(m_rows) * n;
}

int hw_colsum_f32_ws(const float* x, int64_t ld, int64_t m_rows, int64_t n, float* out, float* workspace,
                     int64_t workspace_floats, void* stream) {
    if (!x || !out || ld < n || m_rows < 0) return HW_ERR_INVALID_ARG;
    if (n == 0) return HW_OK;
    const int64_t chunks = hw::colsum_chunks(m_rows);
    if (chunks <= 1) return hw_colsum_f32(x, ld, m_rows, n, out, stream);
    if (!workspace || workspace_floats < chunks * n) return HW_ERR_WORKSPACE;
    HW_CLEAR_ERR();
    const int64_t rows_per_chunk = (m_rows + chunks - 1) / chunks;
    hipLaunchKernelGGL(hw::colsum_partial_kernel, dim3((unsigned)((n + 63) / 64), (unsigned)chunks), dim3(256), 0,
                       (hipStream_t)stream, x, ld, m_rows, n, rows_per_chunk, workspace);
    hipLaunchKernelGGL(hw::colsum_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, (hipStream_t)stream, workspace, n,
                       chunks, n, out);
    return HW_LAUNCH_RC();
}

int hw_pgpr_reward(const int32_t* paths, int64_t n_rows, const int32_t* patterns, int32_t n_patterns,
                   const double* score_entity, int64_t n_entities, int64_t n_items, double* scales_ws,
                   int64_t n_batch, int32_t compute_scales, double* reward, void* stream) {
    if (!paths || !score_entity || !scales_ws || !reward || (n_patterns > 0 && !patterns)) return HW_ERR_INVALID_ARG;
    if (n_rows == 0) return HW_OK;
    HW_CLEAR_ERR();
    if (compute_scales)
        hipLaunchKernelGGL(hw::row_max_f64_kernel, dim3((unsigned)n_batch), dim3(256), 0, (hipStream_t)stream,
                           score_entity, n_entities, n_items, scales_ws);
    hipLaunchKernelGGL(hw::reward_kernel, dim3((unsigned)((n_rows + 63) / 64)), dim3(64), 0, (hipStream_t)stream, paths,
                       n_rows, patterns, n_patterns, score_entity, n_entities, n_items, scales_ws, reward);
    return HW_LAUNCH_RC();
}

int hw_pgpr_reinforce_loss(double* rewards, const float* values, const float* logp, const float* entropy, int32_t n_steps,
                           int64_t n_batch, double gamma, double weight_factor, double* loss_out, float* coef_actor,
                           float* dvalue, void* stream) {
    if (!rewards || !values || !logp || !entropy || !loss_out || !coef_actor || !dvalue || n_steps < 1 || n_batch < 1)
        return HW_ERR_INVALID_ARG;
    HW_CLEAR_ERR();
    hipLaunchKernelGGL(hw::reinforce_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, rewards, values, logp,
                       entropy, n_steps, n_batch, gamma, weight_factor, loss_out, coef_actor, dvalue);
    return HW_LAUNCH_RC();
}

}  // extern "C"
