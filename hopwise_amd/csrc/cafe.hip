// CAFE inference primitives (model/knowledge_aware_recommender/cafe.py):
//   * row-wise log_softmax over an entity vocabulary (cafe.py:597-598, :819-820);
//   * the program interpreter: for every tree node, in order, and for every parent path, in order, take the
//     neighbours of the path's tail under the node's relation (KGMask.get_ids :734-750), drop the ids already
//     picked at this node (visited_ids) and -- at leaves -- the excluded items, keep the top-k by the node's
//     log-probabilities, append (SymbolicNetwork.infer_with_path :587-642 and MetaProgramExecutor.execute
//     :790-857 are this same loop with different k and exclusion rules).
// One workgroup interprets one user's whole program; the sequential dependence (visited / excluded sets grow while
// parent paths are consumed in order) stays inside the workgroup, users run in parallel.
#include "hw_common.h"
#include "rank_epilogue.h"

namespace hw {

__global__ __launch_bounds__(256) void log_softmax_rows_kernel(const float* __restrict__ x, int64_t ld, int n_cols,
                                                               float* __restrict__ y, int64_t ldy) {
    __shared__ float red[4];
    const float* __restrict__ r = x + (size_t)blockIdx.x * ld;
    float* __restrict__ o = y + (size_t)blockIdx.x * ldy;
    float m = -INFINITY;
    for (int c = threadIdx.x; c < n_cols; c += 256) m = fmaxf(m, r[c]);
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) m = fmaxf(m, __shfl_xor(m, s, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int c = threadIdx.x; c < n_cols; c += 256) s += expf(r[c] - m);
    s = group_sum<64>(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float ls = logf((red[0] + red[1]) + (red[2] + red[3]));
    for (int c = threadIdx.x; c < n_cols; c += 256) o[c] = (r[c] - m) - ls;
}

// The same row with its values held in registers between the three passes (NPT per thread, column c = tid + 256 i as above):
// the row is read once instead of three times.  Per-thread order of the maximum and of the sum, the wave butterflies and the
// four-way combination are those of log_softmax_rows_kernel: results are equal bit for bit.  n_cols <= 256 NPT.
template <int NPT>
__global__ __launch_bounds__(256) void log_softmax_rows_reg_kernel(const float* __restrict__ x, int64_t ld, int n_cols,
                                                                   float* __restrict__ y, int64_t ldy) {
    __shared__ float red[4];
    const float* __restrict__ r = x + (size_t)blockIdx.x * ld;
    float* __restrict__ o = y + (size_t)blockIdx.x * ldy;
    float v[NPT];
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const int c = threadIdx.x + 256 * i;
        v[i] = r[c < n_cols ? c : n_cols - 1];  // unconditional: all loads of the row in flight together
    }
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < NPT; ++i)
        if ((int)threadIdx.x + 256 * i < n_cols) m = fmaxf(m, v[i]);
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) m = fmaxf(m, __shfl_xor(m, s, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NPT; ++i)
        if ((int)threadIdx.x + 256 * i < n_cols) s += expf(v[i] - m);
    s = group_sum<64>(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float ls = logf((red[0] + red[1]) + (red[2] + red[3]));
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const int c = threadIdx.x + 256 * i;
        if (c < n_cols) o[c] = (v[i] - m) - ls;
    }
}

// ---- training: cross entropy over one level's vocabulary + the ranking term (cafe.py:541-553) -------------------------
// Per row: lse = logsumexp(logits), ce = lse - logits[target]; rank = sigmoid(logits[neg] - logits[pos]) (the two
// log-softmax values differ from the logits by the same lse).  Row results go to row_ws [2, n_rows]; a second
// one-block kernel sums them in a fixed order.
__global__ __launch_bounds__(256) void level_loss_kernel(const float* __restrict__ x, int64_t ld, int n_cols,
                                                         const int64_t* __restrict__ target,
                                                         const int64_t* __restrict__ pos, const int64_t* __restrict__ neg,
                                                         int64_t n_rows, float* __restrict__ lse_out,
                                                         float* __restrict__ row_ws) {
    __shared__ float red[4];
    const int64_t row = blockIdx.x;
    const float* __restrict__ r = x + (size_t)row * ld;
    float m = -INFINITY;
    for (int c = threadIdx.x; c < n_cols; c += 256) m = fmaxf(m, r[c]);
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) m = fmaxf(m, __shfl_xor(m, s, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int c = threadIdx.x; c < n_cols; c += 256) s += expf(r[c] - m);
    s = group_sum<64>(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float lse = m + logf((red[0] + red[1]) + (red[2] + red[3]));
        lse_out[row] = lse;
        int64_t t = target[row];
        t = t < 0 ? 0 : (t >= n_cols ? n_cols - 1 : t);
        row_ws[row] = lse - r[t];
        float rk = 0.f;
        if (pos && neg) {
            int64_t a = pos[row], b = neg[row];
            a = a < 0 ? 0 : (a >= n_cols ? n_cols - 1 : a);
            b = b < 0 ? 0 : (b >= n_cols ? n_cols - 1 : b);
            const float z = r[b] - r[a];
            const float e = expf(-fabsf(z));
            rk = z >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
        }
        row_ws[n_rows + row] = rk;
    }
}

__global__ __launch_bounds__(256) void level_loss_finish_kernel(const float* __restrict__ row_ws, int64_t n_rows,
                                                                float* __restrict__ out) {
    __shared__ float part[2][256];
    float a = 0.f, b = 0.f;
    for (int64_t i = threadIdx.x; i < n_rows; i += 256) a += row_ws[i], b += row_ws[n_rows + i];
    part[0][threadIdx.x] = a, part[1][threadIdx.x] = b;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            part[0][threadIdx.x] += part[0][threadIdx.x + s];
            part[1][threadIdx.x] += part[1][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = part[0][0] / (float)n_rows, out[1] = part[1][0] / (float)n_rows;
}

// dlogits = g_ce / n (softmax - onehot(target)) + g_rank / n r (1 - r) (onehot(neg) - onehot(pos))
__global__ __launch_bounds__(256) void level_loss_grad_kernel(const float* __restrict__ x, int64_t ld, int n_cols,
                                                              const float* __restrict__ lse,
                                                              const int64_t* __restrict__ target,
                                                              const int64_t* __restrict__ pos,
                                                              const int64_t* __restrict__ neg, int64_t n_rows,
                                                              float g_ce, float g_rank,
                                                              const float* __restrict__ g_ce_dev,
                                                              const float* __restrict__ g_rank_dev,
                                                              float* __restrict__ dx, int64_t ldd) {
    // upstream gradients as device scalars when given (a captured step cannot read them on the host, and the eager step
    // saves two host reads per level)
    if (g_ce_dev) g_ce = g_ce_dev[0];
    if (g_rank_dev) g_rank = g_rank_dev[0];
    const int64_t row = blockIdx.x;
    const float* __restrict__ r = x + (size_t)row * ld;
    float* __restrict__ o = dx + (size_t)row * ldd;
    const float l = lse[row];
    const float inv_n = 1.f / (float)n_rows;
    int64_t t = target[row];
    t = t < 0 ? 0 : (t >= n_cols ? n_cols - 1 : t);
    int64_t a = -1, b = -1;
    float w = 0.f;
    if (pos && neg) {
        a = pos[row], b = neg[row];
        a = a < 0 ? 0 : (a >= n_cols ? n_cols - 1 : a);
        b = b < 0 ? 0 : (b >= n_cols ? n_cols - 1 : b);
        const float z = r[b] - r[a];
        const float e = expf(-fabsf(z));
        const float sg = z >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
        w = g_rank * inv_n * sg * (1.f - sg);
    }
    const float gc = g_ce * inv_n;
    for (int c = threadIdx.x; c < n_cols; c += 256) {
        float v = gc * expf(r[c] - l);
        if (c == t) v -= gc;
        if (c == b) v += w;
        if (c == a) v -= w;
        o[c] = v;
    }
}

__device__ __forceinline__ uint64_t lp_key(float v, uint32_t id) {
    uint32_t u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((uint64_t)u << 32) | (uint32_t)(~id);
}

__device__ __forceinline__ bool bit_get(const uint32_t* b, int i) { return (b[i >> 5] >> (i & 31)) & 1u; }

constexpr int CAFE_BIT_WORDS = 8192;  // 262 144 ids per LDS bitmap (2 x 32 KiB)
constexpr int CAFE_MAX_NODES = 128;
constexpr int CAFE_PCHUNK = 128;  // parent paths whose adjacency bounds are looked up together

__global__ __launch_bounds__(256) void cafe_execute_kernel(
    hw_pgpr_graph g, const hw_cafe_node* __restrict__ nodes, const int32_t* __restrict__ node_ptr,
    const int64_t* __restrict__ users, const float* __restrict__ logprobs, int64_t lp_ld, const float* __restrict__ row_lse,
    const uint32_t* __restrict__ excluded_init, int64_t bit_words, int grow_excluded,
    int32_t* __restrict__ out_ids, float* __restrict__ out_vals, int32_t* __restrict__ out_count,
    const int64_t* __restrict__ slot_ptr) {
    // the two id bitmaps, sized by the launch to the graph's entities (dynamic LDS: 2 x bit_words words).  As static arrays of
    // CAFE_BIT_WORDS they took 64 KB whatever the graph -- two blocks per CU -- while ml-1m's 13 805 entities need 3.4 KB: a
    // block is a long chain of dependent steps, so how many of them a CU holds is the throughput.
    extern __shared__ uint32_t cafe_bits[];
    uint32_t* const vis = cafe_bits;
    uint32_t* const excl = cafe_bits + bit_words;
    __shared__ int counts[CAFE_MAX_NODES];
    __shared__ uint64_t red2[2][4];
    __shared__ long long pt_base[CAFE_PCHUNK];  // parent table (see the node loop)
    __shared__ int pt_deg[CAFE_PCHUNK], pt_lo[CAFE_PCHUNK], pt_hi[CAFE_PCHUNK], pt_id[CAFE_PCHUNK][2];
    __shared__ float pt_v[CAFE_PCHUNK][2];
    const int task = blockIdx.x;
    const int tid = threadIdx.x;
    const int uid = (int)users[task];
    const int n0 = node_ptr[task], n1 = node_ptr[task + 1];
    const int64_t slot0 = slot_ptr[task];
    for (int w = tid; w < bit_words; w += 256) excl[w] = excluded_init[(size_t)task * bit_words + w];
    __syncthreads();

    for (int ni = n0; ni < n1; ++ni) {
        const hw_cafe_node nd = nodes[ni];
        for (int w = tid; w < bit_words; w += 256) vis[w] = 0u;  // visited_ids = [] at every node
        __syncthreads();
        const int depth = nd.depth;  // this node appends element depth - 1 of the id list after the user
        const int n_parent = nd.parent < 0 ? 1 : counts[nd.parent];
        const int p_off = nd.parent < 0 ? 0 : nodes[n0 + nd.parent].slot_off;
        const int32_t* p_ids = out_ids + (slot0 + p_off) * 3;
        const float* p_vals = out_vals + (slot0 + p_off) * 3;
        int32_t* c_ids = out_ids + (slot0 + nd.slot_off) * 3;
        float* c_vals = out_vals + (slot0 + nd.slot_off) * 3;
        const float* __restrict__ lp = logprobs + (size_t)nd.lp_row * lp_ld;
        // row_lse: `logprobs` holds the LOGITS and row r's log-softmax is (x - row_lse[2 r]) - row_lse[2 r + 1] (its maximum and
        // the log of its sum of exponentials, hw_lse_rows): the row pass that rewrote the matrix, taken at the few entries read
        const float lp_max = row_lse ? row_lse[2 * (size_t)nd.lp_row] : 0.f;
        const float lp_ls = row_lse ? row_lse[2 * (size_t)nd.lp_row + 1] : 0.f;
        int n_out = 0;
        int pfx_id[2] = {0, 0};  // this parent path's ids / values (depth - 1 <= 2 of them)
        float pfx_v[2] = {0.f, 0.f};
        for (int pp = 0; pp < n_parent && n_out < nd.cap; ++pp) {
            const int ps = pp % CAFE_PCHUNK;
            if (ps == 0) {
                // Parent table of the next CAFE_PCHUNK parent paths, built by the whole block in two sweeps of independent
                // loads: (A) thread t reads parent pp + t's ids / values and the bounds of its tail's adjacency; (B) wave w
                // counts, for parents w, w + 4, ..., the edges below / up to nd.rel = the relation's segment [lo, hi) of the
                // (relation, destination)-sorted adjacency.  Per parent path this used to be two dependent trips by thread 0
                // plus a block-wide sweep and three barriers: ~1.7 of the ~2.7 us a parent path cost, ~180 paths per user.
                // The parent paths were written by thread 0 of THIS block before the barrier that ended their node: visible.
                __syncthreads();  // the previous chunk's table is no longer read
                const int chunk_n = min(CAFE_PCHUNK, n_parent - pp);
                if (tid < chunk_n) {
                    const int q = pp + tid;
                    int ids[2] = {0, 0};
                    float vs[2] = {0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        if (j < depth - 1) ids[j] = p_ids[(size_t)q * 3 + j], vs[j] = p_vals[(size_t)q * 3 + j];
                    const int last = nd.parent < 0 ? uid : ids[depth == 3 ? 1 : 0];
                    // adjacency of (the RELATION's head type, last) under nd.rel: the lookup type comes from the relation,
                    // not from the node (cafe.py:603 kg_mask.get_ids(module.eh_name, path[-1], ...))
                    long long base = 0;
                    int deg = 0;
                    const int64_t n_type = nd.eh_is_entity ? g.n_entities : g.n_users;
                    if (last >= 0 && last < n_type) {
                        const int64_t gid = nd.eh_is_entity ? (int64_t)g.n_users + last : last;
                        base = g.ptr[gid];
                        deg = (int)(g.ptr[gid + 1] - base);
                    }
                    pt_base[tid] = base, pt_deg[tid] = deg;
                    pt_id[tid][0] = ids[0], pt_id[tid][1] = ids[1], pt_v[tid][0] = vs[0], pt_v[tid][1] = vs[1];
                }
                __syncthreads();
                for (int t = tid >> 6; t < chunk_n; t += 4) {  // wave-uniform
                    const int deg = pt_deg[t];
                    const int32_t* __restrict__ rl = g.rel + pt_base[t];
                    int lt = 0, le = 0;
                    for (int e = tid & 63; e < deg; e += 64) {
                        const int r = rl[e];
                        lt += r < nd.rel;
                        le += r <= nd.rel;
                    }
                    lt = wave_sum_i32(lt), le = wave_sum_i32(le);
                    if ((tid & 63) == 0) pt_lo[t] = lt, pt_hi[t] = le;
                }
                __syncthreads();
            }
            pfx_id[0] = pt_id[ps][0], pfx_id[1] = pt_id[ps][1], pfx_v[0] = pt_v[ps][0], pfx_v[1] = pt_v[ps][1];
            const long long p_base = pt_base[ps];
            const int lo = pt_lo[ps], hi = pt_hi[ps];
            const int32_t* __restrict__ dst = g.dst + p_base;
            uint64_t bound = ~0ull;
            // candidate key of adjacency slot e (0: not a candidate).  During a parent's k rounds the visited / excluded sets
            // only gain that parent's own picks, which the rounds exclude through `bound` anyway: a slot's key can be taken once.
            auto slot_key = [&](int e) -> uint64_t {
                const int id = dst[e];
                if (e > lo && dst[e - 1] == id) return 0ull;  // set(): the adjacency keeps multiplicity
                if (id < 0 || id >= nd.vocab) return 0ull;
                if ((vis[id >> 5] >> (id & 31)) & 1u) return 0ull;
                if (nd.is_leaf && ((excl[id >> 5] >> (id & 31)) & 1u)) return 0ull;
                return lp_key(row_lse ? (lp[id] - lp_max) - lp_ls : lp[id], (uint32_t)id);
            };
            // this thread's first four slots are gathered ONCE and kept for all rounds (segments of up to 1 024 edges: almost
            // all of them); the rounds used to gather every slot again, k times per parent path
            constexpr int CK = 4;
            uint64_t ck[CK];
#pragma unroll
            for (int i = 0; i < CK; ++i) {
                const int e = lo + tid + 256 * i;
                ck[i] = e < hi ? slot_key(e) : 0ull;
            }
            for (int r = 0; r < nd.k && n_out < nd.cap; ++r) {
                uint64_t best = 0;
#pragma unroll
                for (int i = 0; i < CK; ++i)
                    if (ck[i] < bound && ck[i] > best) best = ck[i];
                for (int e = lo + tid + 256 * CK; e < hi; e += 256) {
                    const uint64_t key = slot_key(e);
                    if (key < bound && key > best) best = key;
                }
                best = wave_max_u64(best);
                // ONE barrier per round: the four wave maxima go to the round's half of a double buffer and every thread takes
                // their maximum itself (the other half was last read before the barrier of the round before this one)
                if ((tid & 63) == 0) red2[r & 1][tid >> 6] = best;
                __syncthreads();
                uint64_t w = red2[r & 1][0];
#pragma unroll
                for (int i = 1; i < 4; ++i) w = red2[r & 1][i] > w ? red2[r & 1][i] : w;
                if (w == 0) break;
                if (tid == 0) {
                    const int id = (int)(~(uint32_t)(w & 0xffffffffu));
                    int32_t* q = c_ids + (size_t)n_out * 3;
                    float* v = c_vals + (size_t)n_out * 3;
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        if (j < depth - 1) q[j] = pfx_id[j], v[j] = pfx_v[j];
                    q[depth - 1] = id;
                    // the log-probability back out of the key (lp_key's map is a bijection on float bits): no load here
                    const uint32_t uu = (uint32_t)(w >> 32);
                    v[depth - 1] = __uint_as_float((uu & 0x80000000u) ? (uu & 0x7fffffffu) : ~uu);
                    // the pick becomes visited (and, at executor leaves, excluded) for the parents after this one.  Entering it
                    // now is equivalent: this parent's later rounds exclude it through `bound` already
                    vis[id >> 5] |= 1u << (id & 31);
                    if (nd.is_leaf && grow_excluded) excl[id >> 5] |= 1u << (id & 31);
                }
                bound = w;
                ++n_out;
            }
            __syncthreads();  // the picks' bits are in place before the next parent's candidates are gathered
        }
        if (tid == 0) {
            counts[ni - n0] = n_out;
            out_count[ni] = n_out;
        }
        __syncthreads();
    }
}

// Row r's (maximum, log sum exp(x - maximum)) from the per-stretch pairs hw_gemm_f32_lse left: one wave per row.
__global__ __launch_bounds__(256) void lse_rows_kernel(const float* __restrict__ tile_max, const float* __restrict__ tile_sum,
                                                       int64_t tile_ld, int64_t n_rows, int n_stretches,
                                                       float* __restrict__ row_lse) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int lane = threadIdx.x & 63;
    float m = -INFINITY, s = 0.f;
    for (int t = lane; t < n_stretches; t += 64)
        lse_merge(m, s, tile_max[(size_t)row * tile_ld + t], tile_sum[(size_t)row * tile_ld + t]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) lse_merge(m, s, __shfl_xor(m, d, 64), __shfl_xor(s, d, 64));
    if (lane == 0) row_lse[2 * row] = m, row_lse[2 * row + 1] = logf(s);
}

}  // namespace hw

using namespace hw;

extern "C" int hw_lse_rows(const float* tile_max, const float* tile_sum, int64_t tile_ld, int64_t n_rows, int64_t n_stretches,
                           float* row_lse, void* stream) {
    if (!tile_max || !tile_sum || !row_lse || n_rows < 0 || n_stretches < 1 || tile_ld < n_stretches || n_stretches > INT32_MAX)
        return HW_ERR_INVALID_ARG;
    if (n_rows == 0) return HW_OK;
    HW_CLEAR_ERR();
    hipLaunchKernelGGL(lse_rows_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, tile_max, tile_sum,
                       tile_ld, n_rows, (int)n_stretches, row_lse);
    return HW_LAUNCH_RC();
}

extern "C" int hw_log_softmax_rows(const float* x, int64_t ld, int64_t n_rows, int64_t n_cols, float* y, int64_t ldy,
                                   void* stream) {
    if (!x || !y || n_rows < 0 || n_cols < 1 || ld < n_cols || ldy < n_cols || n_cols > INT32_MAX || n_rows > INT32_MAX)
        return HW_ERR_INVALID_ARG;
    if (n_rows == 0) return HW_OK;
    HW_CLEAR_ERR();
    if (n_cols <= 256 * 16)
        hipLaunchKernelGGL(log_softmax_rows_reg_kernel<16>, dim3((unsigned)n_rows), dim3(256), 0, (hipStream_t)stream, x, ld,
                           (int)n_cols, y, ldy);
    else if (n_cols <= 256 * 64)
        hipLaunchKernelGGL(log_softmax_rows_reg_kernel<64>, dim3((unsigned)n_rows), dim3(256), 0, (hipStream_t)stream, x, ld,
                           (int)n_cols, y, ldy);
    else
        hipLaunchKernelGGL(log_softmax_rows_kernel, dim3((unsigned)n_rows), dim3(256), 0, (hipStream_t)stream, x, ld,
                           (int)n_cols, y, ldy);
    return HW_LAUNCH_RC();
}

extern "C" int hw_cafe_level_loss(const float* logits, int64_t ld, int64_t n_rows, int64_t n_cols, const int64_t* target,
                                  const int64_t* pos, const int64_t* neg, float* lse_out, float* loss_out, float* row_ws,
                                  void* stream) {
    if (!logits || !target || !lse_out || !loss_out || !row_ws || n_rows < 1 || n_cols < 1 || n_cols > INT32_MAX ||
        ld < n_cols || (!pos) != (!neg))
        return HW_ERR_INVALID_ARG;
    HW_CLEAR_ERR();
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(level_loss_kernel, dim3((unsigned)n_rows), dim3(256), 0, st, logits, ld, (int)n_cols, target, pos,
                       neg, n_rows, lse_out, row_ws);
    hipLaunchKernelGGL(level_loss_finish_kernel, dim3(1), dim3(256), 0, st, row_ws, n_rows, loss_out);
    return HW_LAUNCH_RC();
}

extern "C" int hw_cafe_level_loss_grad(const float* logits, int64_t ld, int64_t n_rows, int64_t n_cols, const float* lse,
                                       const int64_t* target, const int64_t* pos, const int64_t* neg, float g_ce,
                                       float g_rank, float* dlogits, int64_t ldd, void* stream) {
    if (!logits || !lse || !target || !dlogits || n_rows < 1 || n_cols < 1 || n_cols > INT32_MAX || ld < n_cols ||
        ldd < n_cols || (!pos) != (!neg))
        return HW_ERR_INVALID_ARG;
    HW_CLEAR_ERR();
    hipLaunchKernelGGL(level_loss_grad_kernel, dim3((unsigned)n_rows), dim3(256), 0, (hipStream_t)stream, logits, ld,
                       (int)n_cols, lse, target, pos, neg, n_rows, g_ce, g_rank, (const float*)nullptr,
                       (const float*)nullptr, dlogits, ldd);
    return HW_LAUNCH_RC();
}

extern "C" int hw_cafe_level_loss_grad_dev(const float* logits, int64_t ld, int64_t n_rows, int64_t n_cols, const float* lse,
                                           const int64_t* target, const int64_t* pos, const int64_t* neg,
                                           const float* g_ce, const float* g_rank, float* dlogits, int64_t ldd,
                                           void* stream) {
    if (!logits || !lse || !target || !dlogits || !g_ce || n_rows < 1 || n_cols < 1 || n_cols > INT32_MAX || ld < n_cols ||
        ldd < n_cols || (!pos) != (!neg) || (pos && !g_rank))
        return HW_ERR_INVALID_ARG;
    HW_CLEAR_ERR();
    hipLaunchKernelGGL(level_loss_grad_kernel, dim3((unsigned)n_rows), dim3(256), 0, (hipStream_t)stream, logits, ld,
                       (int)n_cols, lse, target, pos, neg, n_rows, 0.f, 0.f, g_ce, pos ? g_rank : (const float*)nullptr,
                       dlogits, ldd);
    return HW_LAUNCH_RC();
}

// ---- collect_results + the ranking loop of run_program (cafe.py:417-441, :859-867), one workgroup per user ---------------
// The reference appends every leaf path to `tmp` in leaf order with score = reduce(mul) over its np.float32 log-probs,
// inst[item] keeps the LAST path that reached an item, tmp is sorted by score (stable, descending) and an entry is emitted
// when it beats results[item] -- i.e. per item the first entry in sorted order, carrying inst[item]'s path (not necessarily
// its own).  Here: entry e = (leaf ordinal, row) in leaf order; pass 1 takes per item the max of key(score, e) = (orderable
// score, ~e) and the max of e (global atomics on this user's [n_items] rows); pass 2 keeps the winners' keys in LDS and sorts
// them (bitonic, descending: score desc, entry asc = the reference's emission order); pass 3 writes one record per winner.
// The LDS sort holds the WINNERS only (one entry per distinct item reached, compacted): 8 192 keys = 64 KB of dynamic LDS.
// The reference's executor keeps 5 per node (manual_topk, cafe.py:790), i.e. at most 125 paths per leaf: with the 64 leaves
// this kernel takes, a user has at most 8 000 entries -- the winner limit cannot be reached under the reference's settings.
constexpr int CAFE_MAX_WINNERS = 8192;
constexpr int CAFE_MAX_LEAVES = 64;

__device__ __forceinline__ uint32_t orderable(float s) {
    const uint32_t u = __float_as_uint(s);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(256) void cafe_collect_kernel(const hw_cafe_node* __restrict__ nodes,
                                                           const int32_t* __restrict__ node_ptr,
                                                           const int64_t* __restrict__ slot_ptr,
                                                           const int32_t* __restrict__ ids, const float* __restrict__ vals,
                                                           const int32_t* __restrict__ counts,
                                                           const int32_t* __restrict__ leaf_nodes, int n_leaves, int n_items,
                                                           unsigned long long* __restrict__ best_ws,
                                                           int32_t* __restrict__ last_ws, float* __restrict__ results,
                                                           int max_rec, int32_t* __restrict__ rec_item,
                                                           float* __restrict__ rec_score, int32_t* __restrict__ rec_leaf,
                                                           int32_t* __restrict__ rec_ids, int32_t* __restrict__ rec_count,
                                                           int32_t* __restrict__ overflow) {
    extern __shared__ unsigned long long keys[];  // CAFE_MAX_WINNERS
    __shared__ int pre[CAFE_MAX_LEAVES + 1];
    __shared__ int l_off[CAFE_MAX_LEAVES], l_dep[CAFE_MAX_LEAVES];
    __shared__ int n_win;
    const int task = blockIdx.x, tid = threadIdx.x;
    const int n0 = node_ptr[task];
    const int64_t slot0 = slot_ptr[task];
    unsigned long long* best = best_ws + (size_t)task * n_items;
    int32_t* last = last_ws + (size_t)task * n_items;
    float* res = results + (size_t)task * n_items;
    for (int i = tid; i < n_items; i += 256) {
        res[i] = -INFINITY;
        best[i] = 0ull;
        last[i] = -1;
    }
    if (tid == 0) {
        int acc = 0;
        for (int l = 0; l < n_leaves; ++l) {
            const int ni = leaf_nodes[(size_t)task * n_leaves + l];
            pre[l] = acc;
            if (ni >= 0) {
                const hw_cafe_node nd = nodes[n0 + ni];
                l_off[l] = nd.slot_off;
                l_dep[l] = nd.depth;
                acc += counts[n0 + ni];
            } else {
                l_off[l] = 0, l_dep[l] = 1;
            }
        }
        pre[n_leaves] = acc;
        n_win = 0;
    }
    __syncthreads();
    const int n = pre[n_leaves];
    auto entry = [&](int e, int& item, float& score, int& leaf, int& row) {
        int l = 0;
        while (l + 1 < n_leaves && pre[l + 1] <= e) ++l;
        leaf = l, row = e - pre[l];
        const size_t s = (size_t)(slot0 + l_off[l] + row) * 3;
        const int dep = l_dep[l];
        float sc = vals[s];
        for (int d = 1; d < dep; ++d) sc = __fmul_rn(sc, vals[s + d]);  // reduce(mul), float32, left to right
        score = sc;
        item = ids[s + dep - 1];
    };
    for (int e = tid; e < n; e += 256) {
        int item, leaf, row;
        float score;
        entry(e, item, score, leaf, row);
        if (item >= 0 && item < n_items) {
            atomicMax(last + item, e);  // inst[item]: the last path to the item, whatever its score
            if (score > -INFINITY)
                atomicMax(best + item, ((unsigned long long)orderable(score) << 32) | (0xFFFFFFFFu - (uint32_t)e));
        }
    }
    __syncthreads();
    // the winners, compacted (any order: the keys are distinct, the sort below fixes the order)
    for (int e = tid; e < n; e += 256) {
        int item, leaf, row;
        float score;
        entry(e, item, score, leaf, row);
        if (item >= 0 && item < n_items && score > -INFINITY) {
            const unsigned long long mine = ((unsigned long long)orderable(score) << 32) | (0xFFFFFFFFu - (uint32_t)e);
            if (__hip_atomic_load(best + item, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == mine) {
                const int pos = atomicAdd(&n_win, 1);
                if (pos < CAFE_MAX_WINNERS) keys[pos] = mine;
            }
        }
    }
    __syncthreads();
    const int n_win_all = n_win;
    const bool too_many = n_win_all > CAFE_MAX_WINNERS;  // block-uniform
    if (too_many && tid == 0) atomicMax(overflow, n_win_all);
    const int m = too_many ? CAFE_MAX_WINNERS : n_win_all;
    int np2 = 1;
    while (np2 < m) np2 <<= 1;
    for (int i = m + tid; i < np2; i += 256) keys[i] = 0ull;
    __syncthreads();
    for (int size = 2; size <= np2; size <<= 1) {  // bitonic sort, descending
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < np2; i += 256) {
                const int j = i ^ stride;
                if (j > i) {
                    const bool desc = (i & size) == 0;
                    const unsigned long long a = keys[i], b = keys[j];
                    if (desc ? a < b : a > b) keys[i] = b, keys[j] = a;
                }
            }
            __syncthreads();
        }
    }
    const int nw = min(m, max_rec);
    if (tid == 0) rec_count[task] = nw;
    for (int j = tid; j < nw; j += 256) {
        const int e = (int)(0xFFFFFFFFu - (uint32_t)(keys[j] & 0xFFFFFFFFull));
        int item, leaf, row;
        float score;
        entry(e, item, score, leaf, row);
        res[item] = score;
        int it2, sl, sr;
        float s2;
        entry(__hip_atomic_load(last + item, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), it2, s2, sl, sr);  // inst[item]'s path
        const size_t o = (size_t)task * max_rec + j;
        rec_item[o] = item;
        rec_score[o] = score;
        rec_leaf[o] = sl;
        const size_t s = (size_t)(slot0 + l_off[sl] + sr) * 3;
        rec_ids[o * 3 + 0] = ids[s], rec_ids[o * 3 + 1] = ids[s + 1], rec_ids[o * 3 + 2] = ids[s + 2];
    }
    // a user with more leaf paths than the workgroup ranks (counted into `overflow`): its scores leave as NaN, so that a caller
    // that keeps everything on the device and does not read the counter cannot consume a truncated ranking unnoticed
    if (too_many) {
        __syncthreads();
        for (int i = tid; i < n_items; i += 256) res[i] = __builtin_nanf("");
    }
}

extern "C" int hw_cafe_collect(const hw_cafe_node* nodes, const int32_t* node_ptr, const int64_t* slot_ptr,
                               const int32_t* ids, const float* vals, const int32_t* counts, const int32_t* leaf_nodes,
                               int64_t n_tasks, int32_t n_leaves, int32_t n_items, uint64_t* best_ws, int32_t* last_ws,
                               float* results, int32_t max_rec, int32_t* rec_item, float* rec_score, int32_t* rec_leaf,
                               int32_t* rec_ids, int32_t* rec_count, int32_t* overflow, void* stream) {
    if (!nodes || !node_ptr || !slot_ptr || !ids || !vals || !counts || !leaf_nodes || !best_ws || !last_ws || !results ||
        !rec_item || !rec_score || !rec_leaf || !rec_ids || !rec_count || !overflow)
        return HW_ERR_INVALID_ARG;
    if (n_tasks < 0 || n_tasks > INT32_MAX || n_leaves < 1 || n_items < 1 || max_rec < 1) return HW_ERR_INVALID_ARG;
    if (n_leaves > CAFE_MAX_LEAVES) return HW_ERR_UNSUPPORTED;
    if (n_tasks == 0) return HW_OK;
    HW_CLEAR_ERR();
    constexpr size_t key_bytes = (size_t)CAFE_MAX_WINNERS * sizeof(unsigned long long);
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(cafe_collect_kernel),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)key_bytes);
    if (attr != hipSuccess) return HW_ERR_LAUNCH;
    hipLaunchKernelGGL(cafe_collect_kernel, dim3((unsigned)n_tasks), dim3(256), key_bytes, (hipStream_t)stream, nodes, node_ptr,
                       slot_ptr, ids, vals, counts, leaf_nodes, (int)n_leaves, (int)n_items,
                       (unsigned long long*)best_ws, last_ws, results, (int)max_rec, rec_item, rec_score, rec_leaf, rec_ids,
                       rec_count, overflow);
    return HW_LAUNCH_RC();
}

static int cafe_execute_go(const hw_pgpr_graph* graph, const hw_cafe_node* nodes, const int32_t* node_ptr,
                           const int64_t* users, int64_t n_tasks, int32_t max_nodes_per_task, const float* logprobs,
                           int64_t logprob_ld, const float* row_lse, const uint32_t* excluded_init, int64_t bit_words,
                           int32_t grow_excluded, int32_t* out_ids, float* out_vals, int32_t* out_count,
                           const int64_t* slot_ptr, void* stream) {
    if (!graph || !graph->ptr || !graph->rel || !graph->dst || !nodes || !node_ptr || !users || !logprobs ||
        !excluded_init || !out_ids || !out_vals || !out_count || !slot_ptr)
        return HW_ERR_INVALID_ARG;
    if (n_tasks < 0 || n_tasks > INT32_MAX || bit_words < 1) return HW_ERR_INVALID_ARG;
    if (bit_words > CAFE_BIT_WORDS || max_nodes_per_task > CAFE_MAX_NODES) return HW_ERR_UNSUPPORTED;
    if (n_tasks == 0) return HW_OK;
    HW_CLEAR_ERR();
    const size_t bits_bytes = 2 * (size_t)bit_words * sizeof(uint32_t);  // <= 64 KB (bit_words <= CAFE_BIT_WORDS)
    hipLaunchKernelGGL(cafe_execute_kernel, dim3((unsigned)n_tasks), dim3(256), bits_bytes, (hipStream_t)stream, *graph, nodes,
                       node_ptr, users, logprobs, logprob_ld, row_lse, excluded_init, bit_words, grow_excluded, out_ids,
                       out_vals, out_count, slot_ptr);
    return HW_LAUNCH_RC();
}

extern "C" int hw_cafe_execute(const hw_pgpr_graph* graph, const hw_cafe_node* nodes, const int32_t* node_ptr,
                               const int64_t* users, int64_t n_tasks, int32_t max_nodes_per_task,
                               const float* logprobs, int64_t logprob_ld, const uint32_t* excluded_init,
                               int64_t bit_words, int32_t grow_excluded, int32_t* out_ids, float* out_vals,
                               int32_t* out_count, const int64_t* slot_ptr, void* stream) {
    return cafe_execute_go(graph, nodes, node_ptr, users, n_tasks, max_nodes_per_task, logprobs, logprob_ld, nullptr,
                           excluded_init, bit_words, grow_excluded, out_ids, out_vals, out_count, slot_ptr, stream);
}

// hw_cafe_execute over LOGITS: `logits` is the product's output as it stands and row_lse [rows, 2] (hw_lse_rows) what turns an
// entry into its log-probability where the interpreter reads one.
extern "C" int hw_cafe_execute_logits(const hw_pgpr_graph* graph, const hw_cafe_node* nodes, const int32_t* node_ptr,
                                      const int64_t* users, int64_t n_tasks, int32_t max_nodes_per_task, const float* logits,
                                      int64_t logit_ld, const float* row_lse, const uint32_t* excluded_init,
                                      int64_t bit_words, int32_t grow_excluded, int32_t* out_ids, float* out_vals,
                                      int32_t* out_count, const int64_t* slot_ptr, void* stream) {
    if (!row_lse) return HW_ERR_INVALID_ARG;
    return cafe_execute_go(graph, nodes, node_ptr, users, n_tasks, max_nodes_per_task, logits, logit_ld, row_lse,
                           excluded_init, bit_words, grow_excluded, out_ids, out_vals, out_count, slot_ptr, stream);
}
