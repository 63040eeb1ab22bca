// PGPR.collect_scores on the device (pgpr.py:486-552): filter leaf paths, keep the most probable path per
// (user, item), rank each user's items by (score, prob), and fill the [batch, n_items] result matrix --
// including the reference's quirks (row numbering by order of users that have a surviving item; when fewer
// than max(topk) items survive, the list is padded with the best unseen items, reversed, and zip() with the
// shorter path list drops the tail -- so the padded items displace the best path items).
//
// Work split: (1) mark: one thread per leaf path -> atomicMax of (prob bits, ~leaf index) into best[slot, item]
//             (2) select: one block per user slot -> score row in float64, top-kmax of candidates, pads
//             (3) rows + fill: row numbering scan, scatter into results.
#include "hw_common.h"
#include <cstdlib>

namespace hw {

constexpr int CPATH_INTS = HW_PGPR_ROW_INTS;

// first_idx[i] = the first slot that holds users[i].  One wave per slot, the earlier slots compared 64 at a time: a thread per
// slot walking all earlier ones (the first form) took 0.39 ms for the 4 096 users of one evaluation batch.
__global__ __launch_bounds__(256) void collect_slots_kernel(const int64_t* __restrict__ users, int n,
                                                            int32_t* __restrict__ first_idx) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    const int64_t u = users[i];
    int f = i;
    for (int j0 = 0; j0 < i; j0 += 64) {
        const int j = j0 + lane;
        const unsigned long long hit = __ballot(j < i && users[j] == u);
        if (hit) {
            f = j0 + __builtin_ctzll(hit);
            break;
        }
    }
    if (lane == 0) first_idx[i] = f;
}

__device__ __forceinline__ bool bit_test(const uint32_t* __restrict__ bits, int64_t words, int uid, int item) {
    return (bits[(size_t)uid * words + (item >> 5)] >> (item & 31)) & 1u;
}

// One leaf -> (user slot, item, key) of collect_scores' "best path per (user, item)" (pgpr.py:494-524), or false when the
// path is dropped (a self loop, not ending on an unseen item).  key = (bits of the float32 product of the hop values, ~leaf
// index): the maximum over a cell's leaves is the most probable path, the earlier leaf among equals.
// COMPACT: the last hop's children as {parent row, relation, node, value bits} (pgpr.hip: pgpr_topk_expand_kernel<true>) next
// to the parent rows they extend, instead of one 12-int record + probability triple per leaf.  Same tests, same key.
template <bool COMPACT>
__device__ __forceinline__ bool leaf_key(const hw_pgpr_graph& g, const int32_t* __restrict__ leaf,
                                         const float* __restrict__ probs, const int4* __restrict__ cleaf, int64_t i,
                                         int n_nodes, const int32_t* __restrict__ first_idx,
                                         const uint32_t* __restrict__ pos_bits, int64_t pos_words, int& slot, int& pid,
                                         unsigned long long& key) {
    int uid, batch;
    float pp;
    if constexpr (COMPACT) {
        const int4 lf = cleaf[i];
        // the parent (n_nodes - 1 nodes): its 48-byte record as three 16-byte loads -- field by field this kernel issued ten
        // vector-memory instructions per leaf, and with 66 M leaves that is what paced it
        int p[CPATH_INTS];
        {
            const int4* pr = reinterpret_cast<const int4*>(leaf + (size_t)lf.x * CPATH_INTS);
            const int4 a = pr[0], b = pr[1], c = pr[2];
            p[0] = a.x, p[1] = a.y, p[2] = a.z, p[3] = a.w, p[4] = b.x, p[5] = b.y, p[6] = b.z, p[7] = b.w;
            p[8] = c.x, p[9] = c.y, p[10] = c.z, p[11] = c.w;
        }
        if (p[1] != n_nodes - 1) return false;
        bool ok = lf.y != -1;                                                         // no self loop after the root
        for (int j = 1; j < n_nodes - 1; ++j) ok &= p[2 + 3 * (j - 1)] != -1;
        const int cur_t = n_nodes == 2 ? 0 : p[2 + 3 * (n_nodes - 3) + 1];
        const int lt = (cur_t == 1 && lf.y == g.ui_relation) ? 0 : 1;                 // pgpr.py:209-213
        pid = lf.z;
        ok &= (lt == 1) && (pid < g.n_items);                                         // ends on an item
        if (!ok) return false;
        uid = p[0], batch = p[11];
        const float* q = probs + (size_t)lf.x * 3;
        const float v = __uint_as_float((uint32_t)lf.w);
        pp = n_nodes == 2 ? v : q[0];
        for (int j = 1; j < n_nodes - 1; ++j) pp = pp * (j == n_nodes - 2 ? v : q[j]);  // reduce(mul) in float32
    } else {
        int p[CPATH_INTS];
        {
            const int4* pr = reinterpret_cast<const int4*>(leaf + (size_t)i * CPATH_INTS);
            const int4 a = pr[0], b = pr[1], c = pr[2];
            p[0] = a.x, p[1] = a.y, p[2] = a.z, p[3] = a.w, p[4] = b.x, p[5] = b.y, p[6] = b.z, p[7] = b.w;
            p[8] = c.x, p[9] = c.y, p[10] = c.z, p[11] = c.w;
        }
        if (p[1] != n_nodes) return false;
        bool ok = true;
        for (int j = 1; j < n_nodes; ++j) ok &= p[2 + 3 * (j - 1)] != -1;            // no self loop after the root
        const int lt = p[2 + 3 * (n_nodes - 2) + 1];
        pid = p[2 + 3 * (n_nodes - 2) + 2];
        ok &= (lt == 1) && (pid < g.n_items);                                         // ends on an item
        if (!ok) return false;
        uid = p[0], batch = p[11];
        const float* q = probs + (size_t)i * 3;
        pp = q[0];
        for (int j = 1; j < n_nodes - 1; ++j) pp = pp * q[j];                          // reduce(mul) in float32
    }
    if (bit_test(pos_bits, pos_words, uid, pid)) return false;                         // already a train positive
    slot = first_idx[batch];
    key = ((unsigned long long)__float_as_uint(pp) << 32) | (uint32_t)(~(uint32_t)i);
    return true;
}

template <bool COMPACT>
__global__ __launch_bounds__(256) void collect_mark_kernel(hw_pgpr_graph g, const int32_t* __restrict__ leaf,
                                                           const float* __restrict__ probs,
                                                           const int4* __restrict__ cleaf,
                                                           const int32_t* __restrict__ n_leaf_ptr, int n_nodes,
                                                           const int32_t* __restrict__ first_idx,
                                                           const uint32_t* __restrict__ pos_bits, int64_t pos_words,
                                                           unsigned long long* __restrict__ best) {
    const int n_leaf = n_leaf_ptr[0];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_leaf; i += (int64_t)gridDim.x * blockDim.x) {
        int slot, pid;
        unsigned long long key;
        if (leaf_key<COMPACT>(g, leaf, probs, cleaf, i, n_nodes, first_idx, pos_bits, pos_words, slot, pid, key))
            atomicMax(best + (size_t)slot * g.n_items + pid, key);
    }
}

// The same with the maxima of a block's leaves taken in LDS first.  A device-scope atomic on this part is carried out at the
// memory side (nothing stays in an XCD's L2), ~25 G of them per second: the 66 M leaves of a 256-user 64/64/64 search cost
// 2.65 ms that way.  Leaves are emitted user by user, so a block that takes MARK_LEAVES consecutive leaves sees one user (two
// at a boundary): it keeps that user's [n_items] maxima in LDS (ds_max_u64) and only flushes the cells it touched -- a few
// thousand memory-side atomics per block instead of 32 768.  Leaves of another user go straight to memory.  max() is
// associative and commutative: the table ends up bit-identical.
constexpr int MARK_LEAVES = 32768;
template <bool COMPACT>
__global__ __launch_bounds__(256) void collect_mark_lds_kernel(hw_pgpr_graph g, const int32_t* __restrict__ leaf,
                                                               const float* __restrict__ probs,
                                                               const int4* __restrict__ cleaf,
                                                               const int32_t* __restrict__ n_leaf_ptr, int n_nodes,
                                                               const int32_t* __restrict__ first_idx,
                                                               const uint32_t* __restrict__ pos_bits, int64_t pos_words,
                                                               unsigned long long* __restrict__ best) {
    extern __shared__ unsigned long long lds_best[];  // [n_items]
    __shared__ int slot0_sh;
    const int n_leaf = n_leaf_ptr[0];
    const int64_t lo = (int64_t)blockIdx.x * MARK_LEAVES;
    if (lo >= n_leaf) return;
    const int64_t hi = lo + MARK_LEAVES < n_leaf ? lo + MARK_LEAVES : n_leaf;
    for (int c = threadIdx.x; c < g.n_items; c += 256) lds_best[c] = 0ull;
    if (threadIdx.x == 0) {  // the user slot of the block's first leaf: the one its LDS table stands for
        int batch;
        if constexpr (COMPACT) batch = leaf[(size_t)cleaf[lo].x * CPATH_INTS + 11];
        else batch = leaf[(size_t)lo * CPATH_INTS + 11];
        slot0_sh = first_idx[batch];
    }
    __syncthreads();
    const int slot0 = slot0_sh;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
        int slot, pid;
        unsigned long long key;
        if (!leaf_key<COMPACT>(g, leaf, probs, cleaf, i, n_nodes, first_idx, pos_bits, pos_words, slot, pid, key)) continue;
        if (slot == slot0) atomicMax(&lds_best[pid], key);
        else atomicMax(best + (size_t)slot * g.n_items + pid, key);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < g.n_items; c += 256) {
        const unsigned long long k = lds_best[c];
        if (k) atomicMax(best + (size_t)slot0 * g.n_items + c, k);
    }
}

// The compact last level, parent by parent: a wave takes a parent row -- its record, probability prefix, user slot and the
// validity tests that do not depend on the child arrive through the scalar unit, once per 64 children -- and a lane takes
// one child (16 bytes, a positives bit, one LDS max).  The leaf-per-thread form above re-reads the parent per leaf: ten, then
// six vector-memory instructions per leaf, which is what paced it (0.65 ms for 66 M leaves).  child_offsets[p] = first
// leaf of parent p (hw_pgpr_topk_expand's `offsets`), at most 64 children per parent.  Same keys, same cells: max() is
// order-free, the table is bit-identical.
constexpr int MARK_PARENTS = 512;
__global__ __launch_bounds__(256) void collect_mark_parents_kernel(hw_pgpr_graph g, const int32_t* __restrict__ parents,
                                                                   const float* __restrict__ parent_probs,
                                                                   const int4* __restrict__ cleaf,
                                                                   const int32_t* __restrict__ n_leaf_ptr,
                                                                   const int32_t* __restrict__ child_offsets,
                                                                   const int32_t* __restrict__ n_parents_ptr, int n_nodes,
                                                                   const int32_t* __restrict__ first_idx,
                                                                   const uint32_t* __restrict__ pos_bits, int64_t pos_words,
                                                                   unsigned long long* __restrict__ best) {
    extern __shared__ unsigned long long lds_best[];  // [n_items]
    __shared__ int slot0_sh;
    const int n_leaf = n_leaf_ptr[0], n_par = n_parents_ptr[0];
    const int p0 = blockIdx.x * MARK_PARENTS;
    if (p0 >= n_par) return;
    const int p1 = p0 + MARK_PARENTS < n_par ? p0 + MARK_PARENTS : n_par;
    for (int c = threadIdx.x; c < g.n_items; c += 256) lds_best[c] = 0ull;
    if (threadIdx.x == 0) slot0_sh = first_idx[parents[(size_t)p0 * CPATH_INTS + 11]];
    __syncthreads();
    const int slot0 = slot0_sh;
    const int wave = wave_in_block(), lane = threadIdx.x & 63;
    // A parent is a chain of dependent reads -- record, probabilities, child range (scalar unit) -> user slot, the children
    // (vector) -> positives bit -> LDS max -- and a wave walks 128 parents one after the other.  The reads are therefore issued
    // AHEAD of the parent that needs them, branch-free (clamped indices; a wave's last parent is simply requested again):
    // the scalar reads two parents ahead, the user slot and the children one ahead (as in pgpr.hip: pgpr_topk_expand_kernel).
    static_assert(CPATH_INTS == 12, "three 16-byte pieces");
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    struct Par {
        int p;
        i32x4 q0, q1, q2;   // the record
        float f0, f1, f2;   // its probability prefix
        int c0, c1;         // child range
    };
    auto read_scalars = [&](int p) {  // p is wave-uniform: scalar loads
        Par r;
        r.p = p;
        const i32x4* pr = reinterpret_cast<const i32x4*>(parents + (size_t)p * CPATH_INTS);
        r.q0 = pr[0], r.q1 = pr[1], r.q2 = pr[2];
        const float* __restrict__ q = parent_probs + (size_t)p * 3;
        r.f0 = q[0], r.f1 = q[1], r.f2 = q[2];
        r.c0 = child_offsets[p];
        r.c1 = child_offsets[p + 1 < n_par ? p + 1 : p];  // the last parent's range ends at n_leaf (below)
        return r;
    };
    auto clamp_next = [&](int p) { return p + 4 < p1 ? p + 4 : p; };
    auto request = [&](const Par& r, int& slot, int4& lf) {
        slot = first_idx[r.q2.w];  // record word 11: the user's index in the batch
        const int i = r.c0 + lane;
        lf = cleaf[i < n_leaf ? i : (n_leaf > 0 ? n_leaf - 1 : 0)];
    };
    const int pw = p0 + wave;
    if (pw < p1 && n_leaf > 0) {
        Par s1 = read_scalars(__builtin_amdgcn_readfirstlane(pw));
        Par s2 = read_scalars(clamp_next(s1.p));
        int slot_n;
        int4 lf_n;
        request(s1, slot_n, lf_n);
        for (bool more = true; more;) {
            const Par cur = s1;
            const int slot = slot_n;
            const int4 lf = lf_n;
            more = cur.p + 4 < p1;  // wave-uniform
            s1 = s2;
            s2 = read_scalars(clamp_next(s1.p));
            request(s1, slot_n, lf_n);
            __builtin_amdgcn_sched_barrier(0);
            // the tests that do not depend on the child, on values (selects, no loads)
            const int len = cur.q0.y;
            const int rel1 = cur.q0.z, rel2 = cur.q1.y, rel3 = cur.q2.x;  // relations of hops 1..3
            bool okp = len == n_nodes - 1;
            if (n_nodes - 1 > 1) okp = okp && rel1 != -1;  // no self loop after the root
            if (n_nodes - 1 > 2) okp = okp && rel2 != -1;
            if (n_nodes - 1 > 3) okp = okp && rel3 != -1;
            if (!okp) continue;  // wave-uniform
            // type of the parent's last node (hop n_nodes - 2)
            const int cur_t = n_nodes == 2 ? 0 : (n_nodes == 3 ? cur.q0.w : (n_nodes == 4 ? cur.q1.z : cur.q2.y));
            const int uid = cur.q0.x;
            float prefix = n_nodes == 2 ? 1.f : cur.f0;  // reduce(mul) in float32, the leaf's own value last (leaf_key's order)
            if (n_nodes - 2 > 1) prefix = prefix * cur.f1;
            if (n_nodes - 2 > 2) prefix = prefix * cur.f2;
            int c1 = cur.p + 1 < n_par ? cur.c1 : n_leaf;
            c1 = c1 < n_leaf ? c1 : n_leaf;
            const int i = cur.c0 + lane;
            if (i >= c1) continue;
            const int lt = (cur_t == 1 && lf.y == g.ui_relation) ? 0 : 1;  // pgpr.py:209-213
            const int pid = lf.z;
            if (lf.y == -1 || lt != 1 || pid >= g.n_items) continue;       // a self loop / not an item
            if (bit_test(pos_bits, pos_words, uid, pid)) continue;         // already a train positive
            const float v = __uint_as_float((uint32_t)lf.w);
            const float pp = n_nodes == 2 ? v : prefix * v;
            const unsigned long long key = ((unsigned long long)__float_as_uint(pp) << 32) | (uint32_t)(~(uint32_t)i);
            if (slot == slot0) atomicMax(&lds_best[pid], key);
            else atomicMax(best + (size_t)slot * g.n_items + pid, key);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < g.n_items; c += 256) {
        const unsigned long long k = lds_best[c];
        if (k) atomicMax(best + (size_t)slot0 * g.n_items + c, k);
    }
}

__device__ __forceinline__ uint64_t ord64(double x) {
    uint64_t u = (uint64_t)__double_as_longlong(x);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

// 128-bit max over the block of (hi, lo); returns winner in (hi, lo) for every thread.  4 waves.
__device__ __forceinline__ void block_max128(uint64_t& hi, uint64_t& lo, uint64_t* sh) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const uint64_t ohi = __shfl_xor(hi, m, 64), olo = __shfl_xor(lo, m, 64);
        if (ohi > hi || (ohi == hi && olo > lo)) hi = ohi, lo = olo;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[2 * (threadIdx.x >> 6)] = hi, sh[2 * (threadIdx.x >> 6) + 1] = lo;
    __syncthreads();
    hi = sh[0], lo = sh[1];
#pragma unroll
    for (int w = 1; w < 4; ++w)
        if (sh[2 * w] > hi || (sh[2 * w] == hi && sh[2 * w + 1] > lo)) hi = sh[2 * w], lo = sh[2 * w + 1];
    __syncthreads();
}

// path_scores[uid, :] of many slots as a register-tiled product.  The select kernel below scores its own slot, one block per
// slot, and every block re-reads the whole item table from L2 -- 2.4 MB x 4 096 slots = 1.1 ms for one evaluation batch of
// hopwise's default size; its sixteen-lanes-per-item sum spends more on the shuffle tree than on the products.  Here a
// block owns 32 slots x 32 items (both tiles k-major in LDS, coalesced loads), a thread 2 x 2 outputs, and per output the
// SAME sixteen partial sums (k = l, l + 16, ... ascending, one fma chain each) combined in the shuffle tree's order
// (8, 4, 2, 1): bit-identical scores.
constexpr int CIS_T = 32, CIS_KC = 16;
__global__ __launch_bounds__(256) void collect_item_scores_kernel(hw_pgpr_graph g, const double* __restrict__ U64,
                                                                  const double* __restrict__ E64,
                                                                  const double* __restrict__ R64, int d,
                                                                  const int64_t* __restrict__ users, int nb,
                                                                  double* __restrict__ score_ws) {
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    __shared__ __attribute__((aligned(16))) double sS[CIS_KC][CIS_T + 2];
    __shared__ __attribute__((aligned(16))) double sT[CIS_KC][CIS_T + 2];
    __shared__ int64_t uid_sh[CIS_T];
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    const int s0 = blockIdx.y * CIS_T, it0 = blockIdx.x * CIS_T;
    if (tid < CIS_T) uid_sh[tid] = users[s0 + tid < nb ? s0 + tid : nb - 1];
    __syncthreads();
    double p[2][2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int l = 0; l < 16; ++l) p[i][j][l] = 0.0;
    for (int k0 = 0; k0 < d; k0 += CIS_KC) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {  // 32 x 16 values per tile: two per thread
            const int idx = j * 256 + tid, r = idx >> 4, kk = idx & 15, k = k0 + kk;
            sS[kk][r] = k < d ? U64[(size_t)uid_sh[r] * d + k] + R64[(size_t)g.ui_relation * d + k] : 0.0;
            const int it = it0 + r;
            sT[kk][r] = (k < d && it < g.n_items) ? E64[(size_t)it * d + k] : 0.0;
        }
        __syncthreads();
        const int kn = d - k0 < CIS_KC ? d - k0 : CIS_KC;
#pragma unroll
        for (int kk = 0; kk < CIS_KC; ++kk) {
            if (kk < kn) {  // block-uniform
                const f64x2 a = *reinterpret_cast<const f64x2*>(&sS[kk][2 * ty]);
                const f64x2 b = *reinterpret_cast<const f64x2*>(&sT[kk][2 * tx]);
                p[0][0][kk] += a.x * b.x;
                p[0][1][kk] += a.x * b.y;
                p[1][0][kk] += a.y * b.x;
                p[1][1][kk] += a.y * b.y;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            double* q = p[i][j];
            // the xor-shuffle tree of the sixteen lanes, as lane 0 sees it: l ^ 8, then ^ 4, ^ 2, ^ 1
            const double v0 = (q[0] + q[8]) + (q[4] + q[12]), v2 = (q[2] + q[10]) + (q[6] + q[14]);
            const double v1 = (q[1] + q[9]) + (q[5] + q[13]), v3 = (q[3] + q[11]) + (q[7] + q[15]);
            const double r = (v0 + v2) + (v1 + v3);
            const int slot = s0 + 2 * ty + i, it = it0 + 2 * tx + j;
            if (slot < nb && it < g.n_items) score_ws[(size_t)slot * g.n_items + it] = r;
        }
}

template <bool SCORES_READY>
__global__ __launch_bounds__(256) void collect_select_kernel(
    hw_pgpr_graph g, const double* __restrict__ U64, const double* __restrict__ E64, const double* __restrict__ R64, int d,
    const int64_t* __restrict__ users, const int32_t* __restrict__ first_idx, const uint32_t* __restrict__ pos_bits,
    int64_t pos_words, const unsigned long long* __restrict__ best, double* __restrict__ score_ws, int kmax,
    int32_t* __restrict__ sel_item, double* __restrict__ sel_score, int32_t* __restrict__ sel_leaf,
    int32_t* __restrict__ sel_count) {
    __shared__ double src[HW_PGPR_MAX_DIM];
    __shared__ uint64_t sh[8];
    __shared__ int c_item[1024];
    __shared__ int p_item[1024];
    const int slot = blockIdx.x;
    const int tid = threadIdx.x;
    if (first_idx[slot] != slot) {  // duplicate of an earlier user: no row of its own
        if (tid == 0) sel_count[slot] = 0;
        return;
    }
    const int uid = (int)users[slot];
    if constexpr (!SCORES_READY) {
        for (int k = tid; k < d; k += 256) src[k] = U64[(size_t)uid * d + k] + R64[(size_t)g.ui_relation * d + k];
        __syncthreads();
    }
    double* __restrict__ sc = score_ws + (size_t)slot * g.n_items;
    const unsigned long long* __restrict__ bst = best + (size_t)slot * g.n_items;
    if constexpr (!SCORES_READY) {  // path_scores[uid, :] in float64 (pgpr.py:490-492), 16 lanes per item
        const int grp = tid >> 4, gl = tid & 15;
        for (int base = 0; base < g.n_items; base += 16) {
            const int it = base + grp;
            double acc = 0.0;
            if (it < g.n_items) {
                const double* __restrict__ em = E64 + (size_t)it * d;
                for (int k = gl; k < d; k += 16) acc += src[k] * em[k];
            }
#pragma unroll
            for (int m = 8; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
            if (it < g.n_items && gl == 0) sc[it] = acc;
        }
    }
    __syncthreads();
    // candidates ranked by (score, prob) descending; equal keys fall back to the earlier leaf
    int m = 0;
    uint64_t bhi = ~0ull, blo = ~0ull;  // exclusive bound of the previous winner
    for (int r = 0; r < kmax; ++r) {
        uint64_t hi = 0, lo = 0;
        for (int it = tid; it < g.n_items; it += 256) {
            const unsigned long long k = bst[it];
            if (k == 0) continue;
            const uint64_t khi = ord64(sc[it]), klo = k;  // lo = (prob bits, ~leaf)
            const bool below = khi < bhi || (khi == bhi && klo < blo);
            if (below && (khi > hi || (khi == hi && klo > lo))) hi = khi, lo = klo;
        }
        // the item index is recovered from the leaf row afterwards
        block_max128(hi, lo, sh);
        if (hi == 0 && lo == 0) break;
        if (tid == 0) c_item[r] = (int)(~(uint32_t)(lo & 0xffffffffu));  // leaf index for now
        bhi = hi, blo = lo;
        ++m;
    }
    // pads: best-scoring items that are not train positives (may repeat candidates; item 0 only if unpadded history)
    const int n_pad = kmax - m;
    if (m > 0 && n_pad > 0) {
        uint64_t phi = ~0ull, plo = ~0ull;
        for (int r = 0; r < n_pad; ++r) {
            uint64_t hi = 0, lo = 0;
            for (int it = tid; it < g.n_items; it += 256) {
                if (bit_test(pos_bits, pos_words, uid, it)) continue;
                // np.argsort ascending then reversed: ties resolve to the larger index first
                const uint64_t khi = ord64(sc[it]), klo = (uint64_t)it + 1;
                const bool below = khi < phi || (khi == phi && klo < plo);
                if (below && (khi > hi || (khi == hi && klo > lo))) hi = khi, lo = klo;
            }
            block_max128(hi, lo, sh);
            if (tid == 0) p_item[r] = (hi == 0 && lo == 0) ? -1 : (int)(lo - 1);
            phi = hi, plo = lo;
        }
    }
    __syncthreads();
    // final pairs in the order the reference writes them: zip(reversed(top_items), reversed(top_paths)) -> m pairs
    if (tid == 0) {
        sel_count[slot] = m;
        for (int j = 0; j < m; ++j) {
            // reversed(top_items)[j]: pads first (last pad first), then candidates from the worst up
            int item, leaf_for_item = -1;
            if (j < n_pad) {
                item = p_item[n_pad - 1 - j];
            } else {
                leaf_for_item = c_item[m - 1 - (j - n_pad)];
                item = -2;  // resolved by the fill kernel from the leaf row
            }
            const int path_leaf = c_item[m - 1 - j];  // reversed(top_paths)[j]
            sel_item[(size_t)slot * kmax + j] = item == -2 ? -2 - leaf_for_item : item;
            sel_leaf[(size_t)slot * kmax + j] = path_leaf;
        }
    }
}

// sel_row[i] = number of earlier slots with a result row, or -1.  One block: a contiguous piece per thread, the pieces'
// counts scanned in LDS (one thread walking all n took 0.27 ms for 4 096 users).
__global__ __launch_bounds__(256) void collect_rows_kernel(const int32_t* __restrict__ sel_count, int n,
                                                           int32_t* __restrict__ sel_row) {
    __shared__ int cnt[256];
    const int per = (n + 255) / 256, lo = threadIdx.x * per, hi = lo + per < n ? lo + per : n;
    int c = 0;
    for (int i = lo; i < hi; ++i) c += sel_count[i] > 0;
    cnt[threadIdx.x] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int t = 0; t < 256; ++t) {
            const int v = cnt[t];
            cnt[t] = run;
            run += v;
        }
    }
    __syncthreads();
    int r = cnt[threadIdx.x];
    for (int i = lo; i < hi; ++i) sel_row[i] = sel_count[i] > 0 ? r++ : -1;
}

__global__ void collect_fill_kernel(hw_pgpr_graph g, const int32_t* __restrict__ leaf, const int4* __restrict__ cleaf,
                                    const double* __restrict__ score_ws,
                                    int kmax, int n_nodes, int32_t* __restrict__ sel_item, double* __restrict__ sel_score,
                                    const int32_t* __restrict__ sel_count, const int32_t* __restrict__ sel_row,
                                    float* __restrict__ results, int n_slots) {
    const int slot = blockIdx.x;
    if (slot >= n_slots || threadIdx.x != 0) return;
    const int m = sel_count[slot], row = sel_row[slot];
    for (int j = 0; j < m; ++j) {  // sequential: later writes win, as in the reference's loop
        int item = sel_item[(size_t)slot * kmax + j];
        if (item <= -2) {
            const int lf = -2 - item;
            item = cleaf ? cleaf[lf].z : leaf[(size_t)lf * CPATH_INTS + 2 + 3 * (n_nodes - 2) + 2];
            sel_item[(size_t)slot * kmax + j] = item;
        }
        if (item < 0) continue;
        const double s = score_ws[(size_t)slot * g.n_items + item];
        sel_score[(size_t)slot * kmax + j] = s;
        results[(size_t)row * g.n_items + item] = (float)s;
    }
}

__global__ void fill_f32_kernel(float* __restrict__ p, int64_t n, float v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

}  // namespace hw

using namespace hw;

static int collect_impl(const hw_pgpr_graph* graph, const double* user_emb64, const double* entity_emb64,
                        const double* relation_emb64, int32_t dim, const int64_t* users, int64_t n_batch,
                        const uint32_t* positives_bits, int64_t positives_words, const int32_t* leaf_paths,
                        const float* leaf_probs, const int32_t* compact_leaves, const int32_t* n_leaf, int64_t max_leaf,
                        int32_t n_nodes, int32_t kmax, uint64_t* best_ws, double* score_ws, int32_t* first_idx_ws,
                        float* results, int32_t* sel_item, double* sel_score, int32_t* sel_leaf, int32_t* sel_count,
                        int32_t* sel_row, void* stream, const int32_t* child_offsets = nullptr,
                        const int32_t* n_parents = nullptr, int64_t max_parents = 0) {
    if (!graph || !user_emb64 || !entity_emb64 || !relation_emb64 || !users || !positives_bits || !leaf_paths ||
        !leaf_probs || !n_leaf || !best_ws || !score_ws || !first_idx_ws || !results || !sel_item || !sel_score ||
        !sel_leaf || !sel_count || !sel_row)
        return HW_ERR_INVALID_ARG;
    if (dim < 1 || dim > HW_PGPR_MAX_DIM || n_batch < 1 || n_batch > 65535 || kmax < 1 || kmax > 1024 || n_nodes < 2 ||
        n_nodes > 4 || max_leaf < 0 || positives_words * 32 < graph->n_items || (((uintptr_t)compact_leaves) & 15) != 0)
        return HW_ERR_INVALID_ARG;
    HW_CLEAR_ERR();
    hipStream_t st = (hipStream_t)stream;
    const int nb = (int)n_batch;
    const int64_t cells = n_batch * (int64_t)graph->n_items;
    const int4* cleaf = reinterpret_cast<const int4*>(compact_leaves);
    if (hw::fill_bytes_async(best_ws, 0, sizeof(uint64_t) * (size_t)cells, st) != hipSuccess ||
        hw::fill_bytes_async(sel_leaf, 0xff, sizeof(int32_t) * (size_t)n_batch * kmax, st) != hipSuccess ||
        hw::fill_bytes_async(sel_item, 0xff, sizeof(int32_t) * (size_t)n_batch * kmax, st) != hipSuccess ||
        hw::fill_bytes_async(sel_score, 0, sizeof(double) * (size_t)n_batch * kmax, st) != hipSuccess)
        return HW_ERR_LAUNCH;
    hipLaunchKernelGGL(fill_f32_kernel, dim3((unsigned)((cells + 255) / 256 > 4096 ? 4096 : (cells + 255) / 256)),
                       dim3(256), 0, st, results, cells, -INFINITY);
    hipLaunchKernelGGL(collect_slots_kernel, dim3((nb + 3) / 4), dim3(256), 0, st, users, nb, first_idx_ws);
    if (max_leaf > 0) {
        int64_t grid = (max_leaf + 255) / 256;
        if (grid > 16384) grid = 16384;
        const size_t lds_bytes = (size_t)graph->n_items * sizeof(unsigned long long);
        // an item table per block fits in LDS, and there are enough leaves for blocks of MARK_LEAVES to fill the chip;
        // HW_PGPR_MARK_LDS=0 keeps the direct form (what the tests compare the LDS form with)
        static const bool lds_off = [] { const char* e = getenv("HW_PGPR_MARK_LDS"); return e && e[0] == '0'; }();
        const bool in_lds = !lds_off && lds_bytes <= 48 * 1024 && max_leaf >= 2 * MARK_LEAVES;
        const int64_t grid_lds = (max_leaf + MARK_LEAVES - 1) / MARK_LEAVES;
#define MARK(KERNEL, GRID, LDS)                                                                                          \
    hipLaunchKernelGGL(KERNEL, dim3((unsigned)(GRID)), dim3(256), LDS, st, *graph, leaf_paths, leaf_probs, cleaf, n_leaf, \
                       n_nodes, first_idx_ws, positives_bits, positives_words, (unsigned long long*)best_ws)
        static const bool by_leaf = [] { const char* e = getenv("HW_PGPR_MARK_PARENTS"); return e && e[0] == '0'; }();
        if (in_lds && cleaf && child_offsets && n_parents && max_parents > 0 && !by_leaf)
            hipLaunchKernelGGL(collect_mark_parents_kernel, dim3((unsigned)((max_parents + MARK_PARENTS - 1) / MARK_PARENTS)),
                               dim3(256), lds_bytes, st, *graph, leaf_paths, leaf_probs, cleaf, n_leaf, child_offsets,
                               n_parents, n_nodes, first_idx_ws, positives_bits, positives_words,
                               (unsigned long long*)best_ws);
        else if (in_lds && cleaf) MARK(collect_mark_lds_kernel<true>, grid_lds, lds_bytes);
        else if (in_lds) MARK(collect_mark_lds_kernel<false>, grid_lds, lds_bytes);
        else if (cleaf) MARK(collect_mark_kernel<true>, grid, 0);
        else MARK(collect_mark_kernel<false>, grid, 0);
#undef MARK
    }
    if (nb >= 64) {  // many slots: their item scores as a tiled product first (same sums), then the ranking alone
        hipLaunchKernelGGL(collect_item_scores_kernel, dim3((unsigned)((graph->n_items + CIS_T - 1) / CIS_T),
                                                            (unsigned)((nb + CIS_T - 1) / CIS_T)),
                           dim3(256), 0, st, *graph, user_emb64, entity_emb64, relation_emb64, dim, users, nb, score_ws);
        hipLaunchKernelGGL((collect_select_kernel<true>), dim3(nb), dim3(256), 0, st, *graph, user_emb64, entity_emb64,
                           relation_emb64, dim, users, first_idx_ws, positives_bits, positives_words,
                           (const unsigned long long*)best_ws, score_ws, kmax, sel_item, sel_score, sel_leaf, sel_count);
    } else {
        hipLaunchKernelGGL((collect_select_kernel<false>), dim3(nb), dim3(256), 0, st, *graph, user_emb64, entity_emb64,
                           relation_emb64, dim, users, first_idx_ws, positives_bits, positives_words,
                           (const unsigned long long*)best_ws, score_ws, kmax, sel_item, sel_score, sel_leaf, sel_count);
    }
    hipLaunchKernelGGL(collect_rows_kernel, dim3(1), dim3(256), 0, st, sel_count, nb, sel_row);
    hipLaunchKernelGGL(collect_fill_kernel, dim3(nb), dim3(64), 0, st, *graph, leaf_paths, cleaf, score_ws, kmax, n_nodes,
                       sel_item, sel_score, sel_count, sel_row, results, nb);
    return HW_LAUNCH_RC();
}

extern "C" int hw_pgpr_collect(const hw_pgpr_graph* graph, const double* user_emb64, const double* entity_emb64,
                               const double* relation_emb64, int32_t dim, const int64_t* users, int64_t n_batch,
                               const uint32_t* positives_bits, int64_t positives_words, const int32_t* leaf_paths,
                               const float* leaf_probs, const int32_t* n_leaf, int64_t max_leaf, int32_t n_nodes,
                               int32_t kmax, uint64_t* best_ws, double* score_ws, int32_t* first_idx_ws, float* results,
                               int32_t* sel_item, double* sel_score, int32_t* sel_leaf, int32_t* sel_count,
                               int32_t* sel_row, void* stream) {
    return collect_impl(graph, user_emb64, entity_emb64, relation_emb64, dim, users, n_batch, positives_bits,
                        positives_words, leaf_paths, leaf_probs, nullptr, n_leaf, max_leaf, n_nodes, kmax, best_ws, score_ws,
                        first_idx_ws, results, sel_item, sel_score, sel_leaf, sel_count, sel_row, stream);
}

// The same on the compact output of the last hop (hw_pgpr_topk_expand with leaf_out): parent_paths / parent_probs are the
// rows the last hop expanded, leaves[i] = {parent row, relation, node, bits of the hop value}; sel_leaf indexes leaves.
extern "C" int hw_pgpr_collect_leaves_ex(const hw_pgpr_graph* graph, const double* user_emb64,
                                         const double* entity_emb64, const double* relation_emb64, int32_t dim,
                                         const int64_t* users, int64_t n_batch, const uint32_t* positives_bits,
                                         int64_t positives_words, const int32_t* parent_paths, const float* parent_probs,
                                         const int32_t* leaves, const int32_t* n_leaf, int64_t max_leaf, int32_t n_nodes,
                                         int32_t kmax, uint64_t* best_ws, double* score_ws, int32_t* first_idx_ws,
                                         float* results, int32_t* sel_item, double* sel_score, int32_t* sel_leaf,
                                         int32_t* sel_count, int32_t* sel_row, const int32_t* child_offsets,
                                         const int32_t* n_parents, int64_t max_parents, void* stream) {
    if (!leaves || max_parents < 0 || max_parents > INT32_MAX) return HW_ERR_INVALID_ARG;
    return collect_impl(graph, user_emb64, entity_emb64, relation_emb64, dim, users, n_batch, positives_bits,
                        positives_words, parent_paths, parent_probs, leaves, n_leaf, max_leaf, n_nodes, kmax, best_ws,
                        score_ws, first_idx_ws, results, sel_item, sel_score, sel_leaf, sel_count, sel_row, stream,
                        child_offsets, n_parents, max_parents);
}

extern "C" int hw_pgpr_collect_leaves(const hw_pgpr_graph* graph, const double* user_emb64, const double* entity_emb64,
                                      const double* relation_emb64, int32_t dim, const int64_t* users, int64_t n_batch,
                                      const uint32_t* positives_bits, int64_t positives_words, const int32_t* parent_paths,
                                      const float* parent_probs, const int32_t* leaves, const int32_t* n_leaf,
                                      int64_t max_leaf, int32_t n_nodes, int32_t kmax, uint64_t* best_ws, double* score_ws,
                                      int32_t* first_idx_ws, float* results, int32_t* sel_item, double* sel_score,
                                      int32_t* sel_leaf, int32_t* sel_count, int32_t* sel_row, void* stream) {
    if (!leaves) return HW_ERR_INVALID_ARG;
    return collect_impl(graph, user_emb64, entity_emb64, relation_emb64, dim, users, n_batch, positives_bits,
                        positives_words, parent_paths, parent_probs, leaves, n_leaf, max_leaf, n_nodes, kmax, best_ws,
                        score_ws, first_idx_ws, results, sel_item, sel_score, sel_leaf, sel_count, sel_row, stream);
}
