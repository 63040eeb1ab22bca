// PGPR beam-search traversal on gfx950: one hop = actions -> state -> policy MLP -> masked softmax + top-k
// -> expand, all on device with no host round trip (row counts live in device memory).
//
// Reference: /root/reference/hopwise/model/knowledge_aware_recommender/pgpr.py
//   _get_actions :226-290, _get_state :296-333, forward :175-187, beam_search :440-484.
//
// Data layout (all resident in HBM for the whole search):
//   graph   one CSR over node gid = (type == USER ? id : n_users + id); rows sorted by (relation, dst)
//           with multiplicity (kg_dataset.py:1044-1095); 4-byte relation and dst arrays.
//   tables  float64 copies for the top-max_acts trim (the reference scores in float64, pgpr.py:276) and
//           float32 copies for the state gather (torch.FloatTensor(state), pgpr.py:447).
//   paths   12 x int32 per beam row: [uid, n_nodes, (rel, type, id) x 3, batch index]; node 0 is (SELF, USER, uid).
//   acts    [rows, act_dim] (rel, dst) pairs, action 0 = self loop; n_acts[rows].  On the beam-search path the lists are
//           uint16 offsets into the current node's CSR slice instead (ActOut<true>), and the last hop's children leave as
//           16-byte leaves {parent row, rel, node, value bits} (pgpr_topk_expand_kernel<true, .>).
#include "hw_common.h"
#include <type_traits>

namespace hw {

constexpr int PATH_INTS = HW_PGPR_ROW_INTS;
constexpr int SELF_LOOP = -1;
constexpr int T_USER = 0, T_ENTITY = 1;

__device__ __forceinline__ int next_type(const hw_pgpr_graph& g, int cur_type, int rel) {
    return (cur_type == T_ENTITY && rel == g.ui_relation) ? T_USER : T_ENTITY;  // pgpr.py:209-213
}

// ---- block-level helpers (256 threads) ------------------------------------------------------------
__device__ __forceinline__ int wave_incl_scan(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

// exclusive scan over the block; `total` = block sum.  `wsum` is 4 ints of LDS.
__device__ __forceinline__ int block_excl_scan(int v, int* wsum, int& total) {
    const int w = wave_in_block(), lane = threadIdx.x & 63;
    const int inc = wave_incl_scan(v);
    __syncthreads();
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) base += (i < w) ? wsum[i] : 0;
    total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    return base + inc - v;
}

__device__ __forceinline__ uint64_t orderable(double x) {
    uint64_t u = (uint64_t)__double_as_longlong(x);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

// ---- init ------------------------------------------------------------------------------------------
__global__ void pgpr_init_paths_kernel(const int64_t* __restrict__ users, int n, int32_t* __restrict__ paths,
                                       int32_t* __restrict__ n_rows) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) n_rows[0] = n;
    if (i >= n) return;
    int32_t* p = paths + (size_t)i * PATH_INTS;
    p[0] = (int32_t)users[i];
    p[1] = 1;
#pragma unroll
    for (int k = 2; k < PATH_INTS; ++k) p[k] = 0;
    p[PATH_INTS - 1] = i;  // batch index of the root user; copied to every descendant (used by collect_scores)
}

// ---- A10: actions ------------------------------------------------------------------------------------
constexpr int ACT_LDS_KEYS = 4096;  // candidates whose select keys fit in LDS (32 KiB); larger rows use scratch
constexpr int TRIM_WAVE_MAX_DEG = 2048;   // rows trimmed by one wave with the keys in registers
constexpr int TRIM_BLOCK_MAX_DEG = 8192;  // ... by one block, same scheme; beyond: radix select over LDS / scratch keys

// A path record read whole (three 16-byte loads issued together) before any field is used: indexing the record by the length
// read from it chains memory round trips (len -> hop fields -> CSR pointers -> edges).
struct PathRec {
    int uid, len, batch;      // batch = index of the root user in the search (last int of the record)
    int hr[3], ht[3], hn[3];  // hop j = 1..3 at [j - 1]: relation, node type, node id
    __device__ __forceinline__ void load(const int32_t* __restrict__ paths, int row) {
        static_assert(PATH_INTS == 12, "three 16-byte pieces");
        const int4* pr = reinterpret_cast<const int4*>(paths + (size_t)row * PATH_INTS);
        const int4 q0 = pr[0], q1 = pr[1], q2 = pr[2];
        uid = q0.x, len = q0.y;
        hr[0] = q0.z, ht[0] = q0.w, hn[0] = q1.x;
        hr[1] = q1.y, ht[1] = q1.z, hn[1] = q1.w;
        hr[2] = q2.x, ht[2] = q2.y, hn[2] = q2.z;
        batch = q2.w;
    }
    __device__ __forceinline__ static int pick(const int (&a)[3], int j) { return j == 1 ? a[0] : (j == 2 ? a[1] : a[2]); }
    __device__ __forceinline__ int rel(int j) const { return pick(hr, j); }
    __device__ __forceinline__ int type(int j) const { return pick(ht, j); }
    __device__ __forceinline__ int node(int j) const { return pick(hn, j); }
};

// Where a row's action list goes.  A16 = false: [act_dim] (relation, dst) int32 pairs, action 0 = (SELF_LOOP, current node) -- the
// layout the training kernels, the traces and hw_pgpr_expand read.  A16 = true (beam search with the one-launch tail, rows of
// at most TRIM_BLOCK_MAX_DEG neighbours): [act16_ld(act_dim)] uint16 EDGE OFFSETS into the current node's CSR slice, 0xFFFF for
// the self loop -- an action is a neighbour, and the CSR (12 MB here, L2 / Infinity-Cache resident) already holds its relation
// and id; 2 bytes instead of 8 per action written by the action kernels and gathered by the tail of the hop.
__host__ __device__ __forceinline__ int act16_ld(int act_dim) { return (act_dim + 7) & ~7; }  // 16-byte aligned rows
constexpr unsigned ACT16_SELF = 0xFFFFu;
template <bool A16>
struct ActOut {
    int32_t* p32;
    uint16_t* p16;
    __device__ __forceinline__ ActOut(int32_t* acts, int row, int act_dim) {
        p32 = A16 ? nullptr : acts + (size_t)row * act_dim * 2;
        p16 = A16 ? reinterpret_cast<uint16_t*>(acts) + (size_t)row * act16_ld(act_dim) : nullptr;
    }
    __device__ __forceinline__ void put(int pos, int rel, int dst, int edge) const {
        if constexpr (A16) p16[pos] = (uint16_t)edge;
        else *reinterpret_cast<int2*>(&p32[2 * pos]) = make_int2(rel, dst);  // (relation, dst): one 8-byte store
    }
    __device__ __forceinline__ void self_loop(int cur_n) const {
        if constexpr (A16) p16[0] = (uint16_t)ACT16_SELF;
        else p32[0] = SELF_LOOP, p32[1] = cur_n;
    }
};

// First pass, one WAVE per row and no block barrier: rows whose unvisited neighbours fit max_acts -- 98 % of the rows of
// a 64/64/64 beam on the ml-1m-shaped graph, median degree 99 -- are finished here in ONE sweep over their edges with
// ballot / popcount compaction in CSR (= relation, dst) order; rows that overflow max_acts get n_acts = -1 and are queued
// for the trim kernels below.
template <bool A16>
__global__ __launch_bounds__(256) void pgpr_actions_small_kernel(hw_pgpr_graph g, const int32_t* __restrict__ paths,
                                                                 const int32_t* __restrict__ n_rows_ptr, int max_acts,
                                                                 int flags, int32_t* __restrict__ acts,
                                                                 int32_t* __restrict__ n_acts,
                                                                 int32_t* __restrict__ work, int64_t list_cap) {
    const int n_rows = n_rows_ptr[0];
    const int act_dim = max_acts + 1;
    const int lane = threadIdx.x & 63;
    const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int row = blockIdx.x * 4 + wave_in_block(); row < n_rows; row += gridDim.x * 4) {
        PathRec rec;
        rec.load(paths, row);
        const int uid = rec.uid, len = rec.len;
        int cur_t = T_USER, cur_n = uid;
        if (len > 1) cur_t = rec.type(len - 1), cur_n = rec.node(len - 1);
        int vt[4], vn[4];  // visited (type, id); wave-uniform
        bool lead = true;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int t = -1, n = -1;
            if (k == 0) t = T_USER, n = uid;
            else if (k < len) t = rec.ht[k - 1], n = rec.hn[k - 1];
            if (k >= 1 && k < len) lead = lead && rec.hr[k - 1] == SELF_LOOP;
            if ((flags & 2) && k < len && lead) t = -1;  // training: the start user and its self-loop copies (see below)
            vt[k] = t, vn[k] = n;
        }
        const int64_t gid = cur_t == T_USER ? cur_n : (int64_t)g.n_users + cur_n;
        const int64_t lo = g.ptr[gid];
        const int deg = (int)(g.ptr[gid + 1] - lo);
        const int32_t* __restrict__ erel = g.rel + lo;
        const int32_t* __restrict__ edst = g.dst + lo;
        // One sweep: entries are written as they are found (CSR = (relation, dst) order) and the row is handed to the trim
        // kernels as soon as more than max_acts have been seen -- what was written so far is overwritten there.  Four
        // 64-edge chunks per trip with their loads issued together (clamped index, selects afterwards).
        const ActOut<A16> arow(acts, row, act_dim);
        int running = 0;
        // NCH 64-edge chunks per trip, their loads issued together.  The kernel is bound by instruction issue on the scalar
        // unit (~170 scalar instructions per row: ballots, popcounts, the wave-uniform record fields; counters in
        // profiles/r4/pmc_pgpr_r4.txt), and the median row of a beam has ~100 neighbours: rows of at most 128 take two
        // chunks per trip instead of four clamped ones.
        auto sweep = [&](auto nch_c) {
            constexpr int NCH = decltype(nch_c)::value;
            for (int base = 0; base < deg && running <= max_acts; base += 64 * NCH) {
                int rr[NCH], nn[NCH];
#pragma unroll
                for (int u = 0; u < NCH; ++u) {
                    const int e = base + 64 * u + lane;
                    const int ec = e < deg ? e : deg - 1;
                    rr[u] = erel[ec], nn[u] = edst[ec];
                }
#pragma unroll
                for (int u = 0; u < NCH; ++u) {
                    const int e = base + 64 * u + lane;
                    const int nt = next_type(g, cur_t, rr[u]);
                    bool seen = false;
#pragma unroll
                    for (int k = 0; k < 4; ++k) seen |= (vt[k] == nt && vn[k] == nn[u]);
                    const bool f = e < deg && !seen;
                    const unsigned long long m = __ballot(f);
                    const int pos = 1 + running + __popcll(m & lt_mask);
                    if (f && pos <= max_acts) arow.put(pos, rr[u], nn[u], e);
                    running += __popcll(m);
                }
            }
        };
        if (deg <= 128) sweep(std::integral_constant<int, 2>{});
        else sweep(std::integral_constant<int, 4>{});
        if (running > max_acts) {
            // the trim kernels score and cut this row; with a work buffer it is queued by degree class so that they
            // walk short lists instead of scanning every row for n_acts < 0
            if (lane == 0) {
                n_acts[row] = -1;
                if (work) {
                    const int cls = (flags & 1) ? 2 : (deg <= TRIM_WAVE_MAX_DEG ? 0 : (deg <= TRIM_BLOCK_MAX_DEG ? 1 : 2));
                    const int at = atomicAdd(&work[cls], 1);
                    work[4 + (int64_t)cls * list_cap + at] = row;
                }
            }
            continue;
        }
        if (lane == 0) arow.self_loop(cur_n), n_acts[row] = 1 + running;
    }
}

// Select keys of the edges e = i * STRIDE + t (i < KPL) of one row: orderable float64 score of an unvisited neighbour,
// 0 for visited ones and past the end.  Straight-line loads (clamped edge index, clamped table indices, selects
// afterwards), eight edges per lane in flight at a time.
template <int KPL, int STRIDE>
__device__ __forceinline__ void load_select_keys(const hw_pgpr_graph& g, const double* __restrict__ su,
                                                 const double* __restrict__ se,
                                                 const double* __restrict__ score_rel_entity, int cur_t,
                                                 const int (&vt)[4], const int (&vn)[4],
                                                 const int32_t* __restrict__ erel, const int32_t* __restrict__ edst,
                                                 int deg, int t, uint64_t (&key)[KPL]) {
#pragma unroll
    for (int i0 = 0; i0 < KPL; i0 += 8) {
#pragma unroll
        for (int i = i0; i < i0 + 8; ++i) {
            const int e = i * STRIDE + t;
            const int ec = e < deg ? e : deg - 1;
            const int r = erel[ec], n = edst[ec];
            const int nt = next_type(g, cur_t, r);
            bool seen = false;
#pragma unroll
            for (int k = 0; k < 4; ++k) seen |= (vt[k] == nt && vn[k] == n);
            const bool to_user = nt == T_USER;
            const double s1 = (to_user ? su : se)[n];
            const double s2 = score_rel_entity[(size_t)r * g.n_entities + (to_user ? 0 : n)];
            const double sc = (to_user || n < g.n_items) ? s1 : s1 + s2;  // pgpr.py:268-276 via the per-search tables
            key[i] = (e < deg && !seen) ? orderable(sc) : 0ull;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Second pass, still one WAVE per row: rows that need the float64 trim but have at most 64 * KPL neighbours keep their
// select keys in registers (KPL per lane) and find the max_acts-th largest key by a bitwise binary search whose counts are
// ballots + scalar popcounts -- no LDS, no block barrier.  The block-per-row radix select below costs 40-80 us per row in
// barriers and LDS atomics (8 digit passes x 5 syncs); a hop over item nodes (every row trimmed, degree ~10^3) spent
// 160-180 us there, this pass needs under 10.  Same keys, same threshold, same tie rule (ties kept from the END of the
// row) and the same CSR output order as the block kernel; the descending order of fix_scores_sorting_bug stays there.
template <int KPL, bool A16>
__device__ __forceinline__ void trim_row_in_registers(const hw_pgpr_graph& g, const double* __restrict__ su,
                                                      const double* __restrict__ se,
                                                      const double* __restrict__ score_rel_entity, int cur_t,
                                                      const int (&vt)[4], const int (&vn)[4],
                                                      const int32_t* __restrict__ erel,
                                                      const int32_t* __restrict__ edst, int deg, int max_acts,
                                                      const ActOut<A16>& arow, int lane,
                                                      unsigned long long lt_mask) {
    uint64_t key[KPL];
    load_select_keys<KPL, 64>(g, su, se, score_rel_entity, cur_t, vt, vn, erel, edst, deg, lane, key);
    // T = max_acts-th largest key: the largest v with count(key >= v) >= max_acts, built bit by bit
    // ... and abandoned as soon as a prefix is met by exactly max_acts keys: keeping every key >= that prefix is then the
    // answer with no tie to break (distinct float64 scores separate after ~log2(deg) bits below their common exponent)
    uint64_t T = 0;
    for (int bit = 63; bit >= 0; --bit) {
        const uint64_t c = T | (1ull << bit);
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < KPL; ++i) cnt += __popcll(__ballot(key[i] >= c));
        if (cnt >= max_acts) T = c;  // wave-uniform
        if (cnt == max_acts) break;
    }
    int c_gt = 0, c_eq = 0;
#pragma unroll
    for (int i = 0; i < KPL; ++i) {
        c_gt += __popcll(__ballot(key[i] > T));
        c_eq += __popcll(__ballot(key[i] == T));
    }
    const int tie_skip = c_eq - (max_acts - c_gt);  // ties are kept from the END of the row (stable argsort, [-max_acts:])
    int run_keep = 0, run_tie = 0;
#pragma unroll
    for (int i = 0; i < KPL; ++i) {
        const int e = i * 64 + lane;
        const int ec = e < deg ? e : deg - 1;
        const int r = erel[ec], n = edst[ec];  // L1 hits; unconditional so that they are issued together
        const bool tie = key[i] == T;
        const unsigned long long tb = __ballot(tie);
        const bool keep = key[i] > T || (tie && run_tie + __popcll(tb & lt_mask) >= tie_skip);
        const unsigned long long kb = __ballot(keep);
        if (keep) {
            const int pos = 1 + run_keep + __popcll(kb & lt_mask);
            arow.put(pos, r, n, e);
        }
        run_keep += __popcll(kb);
        run_tie += __popcll(tb);
    }
}

// (bx, nb): this block's index among the nb blocks that share the rows -- blockIdx.x / gridDim.x of the kernel of its own, a slice
// of the grid in pgpr_actions_trim_kernel
template <bool A16>
__device__ __forceinline__ void trim_wave_body(
    const hw_pgpr_graph& g, const double* __restrict__ score_user, const double* __restrict__ score_entity,
    const double* __restrict__ score_rel_entity, const int32_t* __restrict__ paths,
    const int32_t* __restrict__ n_rows_ptr, int max_acts, int flags, int32_t* __restrict__ acts,
    int32_t* __restrict__ n_acts, const int32_t* __restrict__ list, const int32_t* __restrict__ list_len, int bx, int nb) {
    const int n_rows = list ? list_len[0] : n_rows_ptr[0];
    const int act_dim = max_acts + 1;
    const int lane = threadIdx.x & 63;
    const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int it = bx * 4 + wave_in_block(); it < n_rows; it += nb * 4) {
        const int row = list ? list[it] : it;
        if (n_acts[row] >= 0) continue;  // wave-uniform: finished by the first pass
        PathRec rec;
        rec.load(paths, row);
        const int uid = rec.uid, len = rec.len;
        int cur_t = T_USER, cur_n = uid;
        if (len > 1) cur_t = rec.type(len - 1), cur_n = rec.node(len - 1);
        const int64_t gid = cur_t == T_USER ? cur_n : (int64_t)g.n_users + cur_n;
        const int64_t lo = g.ptr[gid];
        const int deg = (int)(g.ptr[gid + 1] - lo);
        if (deg > TRIM_WAVE_MAX_DEG) continue;  // left to the block kernel
        int vt[4], vn[4];  // visited (type, id), as in the first pass
        bool lead = true;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int t = -1, n = -1;
            if (k == 0) t = T_USER, n = uid;
            else if (k < len) t = rec.ht[k - 1], n = rec.hn[k - 1];
            if (k >= 1 && k < len) lead = lead && rec.hr[k - 1] == SELF_LOOP;
            if ((flags & 2) && k < len && lead) t = -1;
            vt[k] = t, vn[k] = n;
        }
        const int b = rec.batch;
        const double* __restrict__ su = score_user + (size_t)b * g.n_users;
        const double* __restrict__ se = score_entity + (size_t)b * g.n_entities;
        const int32_t* __restrict__ erel = g.rel + lo;
        const int32_t* __restrict__ edst = g.dst + lo;
        const ActOut<A16> arow(acts, row, act_dim);
        if (deg <= 512) trim_row_in_registers<8, A16>(g, su, se, score_rel_entity, cur_t, vt, vn, erel, edst, deg, max_acts, arow, lane, lt_mask);
        else if (deg <= 1024) trim_row_in_registers<16, A16>(g, su, se, score_rel_entity, cur_t, vt, vn, erel, edst, deg, max_acts, arow, lane, lt_mask);
        else trim_row_in_registers<32, A16>(g, su, se, score_rel_entity, cur_t, vt, vn, erel, edst, deg, max_acts, arow, lane, lt_mask);
        if (lane == 0) arow.self_loop(cur_n), n_acts[row] = 1 + max_acts;
    }
}

// Third pass, one BLOCK per row for 2048 < neighbours <= 8192: the same register-resident search with the counts of the
// four waves summed through LDS -- one barrier per search step and three for the output, against the ~150 barriers and the
// same-bin LDS atomics of the radix select below (a 6 000-neighbour item row took ~120 us there).
template <int KPL, bool A16>
__device__ __forceinline__ void trim_row_in_block(const hw_pgpr_graph& g, const double* __restrict__ su,
                                                  const double* __restrict__ se,
                                                  const double* __restrict__ score_rel_entity, int cur_t,
                                                  const int (&vt)[4], const int (&vn)[4],
                                                  const int32_t* __restrict__ erel, const int32_t* __restrict__ edst,
                                                  int deg, int max_acts, const ActOut<A16>& arow,
                                                  unsigned long long lt_mask, int (*sh_cnt)[2][4], int (*sh_slot)[32][4]) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    uint64_t key[KPL];
    load_select_keys<KPL, 256>(g, su, se, score_rel_entity, cur_t, vt, vn, erel, edst, deg, tid, key);
    // block-wide count of a predicate over all keys: per-wave ballots, one LDS exchange, one barrier (stage `par`)
    int par = 0;
    auto block_sum2 = [&](int a, int b, int& sa, int& sb) {
        if (lane == 0) sh_cnt[par][0][w] = a, sh_cnt[par][1][w] = b;
        __syncthreads();
        sa = sh_cnt[par][0][0] + sh_cnt[par][0][1] + sh_cnt[par][0][2] + sh_cnt[par][0][3];
        sb = sh_cnt[par][1][0] + sh_cnt[par][1][1] + sh_cnt[par][1][2] + sh_cnt[par][1][3];
        par ^= 1;  // the other stage is only rewritten after the next barrier, when everyone has read this one
    };
    uint64_t T = 0;
    for (int bit = 63; bit >= 0; --bit) {
        const uint64_t c = T | (1ull << bit);
        int mine = 0;
#pragma unroll
        for (int i = 0; i < KPL; ++i) mine += __popcll(__ballot(key[i] >= c));
        int cnt, unused;
        block_sum2(mine, 0, cnt, unused);
        if (cnt >= max_acts) T = c;  // block-uniform
        if (cnt == max_acts) break;
    }
    int gt_w = 0, eq_w = 0;
#pragma unroll
    for (int i = 0; i < KPL; ++i) {
        gt_w += __popcll(__ballot(key[i] > T));
        eq_w += __popcll(__ballot(key[i] == T));
    }
    int c_gt, c_eq;
    block_sum2(gt_w, eq_w, c_gt, c_eq);
    const int tie_skip = c_eq - (max_acts - c_gt);  // ties are kept from the END of the row
    // edge order = (slot i, wave w, lane).  Per-(slot, wave) tie and keep counts go through LDS once each; the exclusive
    // prefixes over that order are then rebuilt on the fly where they are used (no per-slot arrays in registers).
    const bool ties = tie_skip > 0;  // block-uniform; rare: several keys equal to the threshold, not all of them wanted
    if (ties) {
#pragma unroll
        for (int i = 0; i < KPL; ++i) {
            const int c = __popcll(__ballot(key[i] == T));
            if (lane == 0) sh_slot[0][i][w] = c;
        }
    }
    __syncthreads();
    auto keep_of = [&](int i, int tie_before) -> bool {
        const bool tie = key[i] == T;
        const unsigned long long tb = __ballot(tie);
        return key[i] > T || (tie && tie_before + __popcll(tb & lt_mask) >= tie_skip);
    };
    auto step_prefix = [&](int (*tab)[4], int i, int& run) -> int {  // exclusive prefix of (slot i, this wave); run += slot
        int here = run;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int x = tab[i][v];
            here += v < w ? x : 0;
            run += x;
        }
        return here;
    };
    {
        int run_t = 0;
#pragma unroll
        for (int i = 0; i < KPL; ++i) {
            const int tb4 = ties ? step_prefix(sh_slot[0], i, run_t) : 0;
            const int c = __popcll(__ballot(keep_of(i, tb4)));
            if (lane == 0) sh_slot[1][i][w] = c;
        }
    }
    __syncthreads();
    int run_t = 0, run_k = 0;
#pragma unroll
    for (int i = 0; i < KPL; ++i) {
        const int e = i * 256 + tid;
        const int ec = e < deg ? e : deg - 1;
        const int r = erel[ec], n = edst[ec];
        const int tb4 = ties ? step_prefix(sh_slot[0], i, run_t) : 0;
        const bool keep = keep_of(i, tb4);
        const unsigned long long kb = __ballot(keep);
        const int before = step_prefix(sh_slot[1], i, run_k);
        if (keep) {
            const int pos = 1 + before + __popcll(kb & lt_mask);
            arow.put(pos, r, n, e);
        }
    }
}

template <bool A16>
__device__ __forceinline__ void trim_block_body(
    const hw_pgpr_graph& g, const double* __restrict__ score_user, const double* __restrict__ score_entity,
    const double* __restrict__ score_rel_entity, const int32_t* __restrict__ paths,
    const int32_t* __restrict__ n_rows_ptr, int max_acts, int flags, int32_t* __restrict__ acts,
    int32_t* __restrict__ n_acts, const int32_t* __restrict__ list, const int32_t* __restrict__ list_len, int bx, int nb) {
    __shared__ int sh_cnt[2][2][4];
    __shared__ int sh_slot[2][32][4];
    const int n_rows = list ? list_len[0] : n_rows_ptr[0];
    const int act_dim = max_acts + 1;
    const int lane = threadIdx.x & 63;
    const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int it = bx; it < n_rows; it += nb) {
        const int row = list ? list[it] : it;
        if (n_acts[row] >= 0) continue;  // block-uniform
        PathRec rec;
        rec.load(paths, row);
        const int uid = rec.uid, len = rec.len;
        int cur_t = T_USER, cur_n = uid;
        if (len > 1) cur_t = rec.type(len - 1), cur_n = rec.node(len - 1);
        const int64_t gid = cur_t == T_USER ? cur_n : (int64_t)g.n_users + cur_n;
        const int64_t lo = g.ptr[gid];
        const int deg = (int)(g.ptr[gid + 1] - lo);
        if (deg > TRIM_BLOCK_MAX_DEG) continue;  // left to the radix-select kernel
        int vt[4], vn[4];
        bool lead = true;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int t = -1, n = -1;
            if (k == 0) t = T_USER, n = uid;
            else if (k < len) t = rec.ht[k - 1], n = rec.hn[k - 1];
            if (k >= 1 && k < len) lead = lead && rec.hr[k - 1] == SELF_LOOP;
            if ((flags & 2) && k < len && lead) t = -1;
            vt[k] = t, vn[k] = n;
        }
        const int b = rec.batch;
        const double* __restrict__ su = score_user + (size_t)b * g.n_users;
        const double* __restrict__ se = score_entity + (size_t)b * g.n_entities;
        const int32_t* __restrict__ erel = g.rel + lo;
        const int32_t* __restrict__ edst = g.dst + lo;
        const ActOut<A16> arow(acts, row, act_dim);
        __syncthreads();  // LDS of the previous row is free
        if (deg <= 4096) trim_row_in_block<16, A16>(g, su, se, score_rel_entity, cur_t, vt, vn, erel, edst, deg, max_acts, arow, lt_mask, sh_cnt, sh_slot);
        else trim_row_in_block<32, A16>(g, su, se, score_rel_entity, cur_t, vt, vn, erel, edst, deg, max_acts, arow, lt_mask, sh_cnt, sh_slot);
        __syncthreads();  // every wave has finished the row before it is marked done
        if (threadIdx.x == 0) arow.self_loop(cur_n), n_acts[row] = 1 + max_acts;
    }
}

// The two trims of a hop as kernels of their own (the rows of a degree class queued by the first pass) ...
template <bool A16>
__global__ __launch_bounds__(256, 3) void pgpr_actions_trim_wave_kernel(
    hw_pgpr_graph g, const double* __restrict__ score_user, const double* __restrict__ score_entity,
    const double* __restrict__ score_rel_entity, const int32_t* __restrict__ paths,
    const int32_t* __restrict__ n_rows_ptr, int max_acts, int flags, int32_t* __restrict__ acts,
    int32_t* __restrict__ n_acts, const int32_t* __restrict__ list, const int32_t* __restrict__ list_len) {
    trim_wave_body<A16>(g, score_user, score_entity, score_rel_entity, paths, n_rows_ptr, max_acts, flags, acts, n_acts, list,
                        list_len, (int)blockIdx.x, (int)gridDim.x);
}

template <bool A16>
__global__ __launch_bounds__(256, 3) void pgpr_actions_trim_block_kernel(
    hw_pgpr_graph g, const double* __restrict__ score_user, const double* __restrict__ score_entity,
    const double* __restrict__ score_rel_entity, const int32_t* __restrict__ paths,
    const int32_t* __restrict__ n_rows_ptr, int max_acts, int flags, int32_t* __restrict__ acts,
    int32_t* __restrict__ n_acts, const int32_t* __restrict__ list, const int32_t* __restrict__ list_len) {
    trim_block_body<A16>(g, score_user, score_entity, score_rel_entity, paths, n_rows_ptr, max_acts, flags, acts, n_acts, list,
                         list_len, (int)blockIdx.x, (int)gridDim.x);
}

// ... and as ONE launch: blocks [0, nb_wave) take the wave trim's queue, the rest the block trim's.  The two are independent
// (different rows); for the few hundred over-full rows of a training batch each is a latency chain of ~30 us that fills a
// fraction of the chip, so one after the other they cost twice that (a sixth of the replayed 1 024-user step).
template <bool A16>
__global__ __launch_bounds__(256, 3) void pgpr_actions_trim_kernel(
    hw_pgpr_graph g, const double* __restrict__ score_user, const double* __restrict__ score_entity,
    const double* __restrict__ score_rel_entity, const int32_t* __restrict__ paths,
    const int32_t* __restrict__ n_rows_ptr, int max_acts, int flags, int32_t* __restrict__ acts,
    int32_t* __restrict__ n_acts, const int32_t* __restrict__ list_w, const int32_t* __restrict__ len_w,
    const int32_t* __restrict__ list_b, const int32_t* __restrict__ len_b, int nb_wave, int nb_block) {
    if ((int)blockIdx.x < nb_wave)
        trim_wave_body<A16>(g, score_user, score_entity, score_rel_entity, paths, n_rows_ptr, max_acts, flags, acts, n_acts,
                            list_w, len_w, (int)blockIdx.x, nb_wave);
    else
        trim_block_body<A16>(g, score_user, score_entity, score_rel_entity, paths, n_rows_ptr, max_acts, flags, acts, n_acts,
                             list_b, len_b, (int)blockIdx.x - nb_wave, nb_block);
}

__global__ __launch_bounds__(256) void pgpr_actions_kernel(
    hw_pgpr_graph g, const double* __restrict__ score_user, const double* __restrict__ score_entity,
    const double* __restrict__ score_rel_entity, const int32_t* __restrict__ paths,
    const int32_t* __restrict__ n_rows_ptr, int max_acts, int fix_bug, int32_t* __restrict__ acts,
    int32_t* __restrict__ n_acts, uint64_t* __restrict__ scratch, int64_t scratch_stride,
    const int32_t* __restrict__ list, const int32_t* __restrict__ list_len) {
    __shared__ uint64_t lds_keys[ACT_LDS_KEYS];
    __shared__ unsigned hist[256];
    __shared__ int wsum[4];
    __shared__ int sh_bin, sh_need, sh_done;
    __shared__ int vis_t[4], vis_n[4];

    const int n_rows = list ? list_len[0] : n_rows_ptr[0];
    const int act_dim = max_acts + 1;
    const int tid = threadIdx.x;

    for (int it = blockIdx.x; it < n_rows; it += gridDim.x) {
        const int row = list ? list[it] : it;
        if (n_acts[row] >= 0) continue;  // finished by an earlier pass (block-uniform: one value per row)
        const int32_t* p = paths + (size_t)row * PATH_INTS;
        const int uid = p[0], len = p[1];
        int cur_t, cur_n;
        if (len == 1) {
            cur_t = T_USER, cur_n = uid;
        } else {
            cur_t = p[2 + 3 * (len - 2) + 1], cur_n = p[2 + 3 * (len - 2) + 2];
        }
        __syncthreads();  // previous row's LDS use is over
        if (tid < 4) {
            int t = -1, n = -1;
            if (tid == 0) t = T_USER, n = uid;
            else if (tid < len) t = p[2 + 3 * (tid - 1) + 1], n = p[2 + 3 * (tid - 1) + 2];
            if ((fix_bug & 2) && tid < len) {
                // bit 1 (training): the start user and its self-loop copies carry a tensor id in the reference and
                // never compare equal to a graph id -> they are not "visited"
                bool lead = true;
                for (int j = 1; j <= tid; ++j) lead = lead && p[2 + 3 * (j - 1)] == SELF_LOOP;
                if (lead) t = -1;
            }
            vis_t[tid] = t, vis_n[tid] = n;
        }
        __syncthreads();
        int32_t* arow = acts + (size_t)row * act_dim * 2;
        if (tid == 0) arow[0] = SELF_LOOP, arow[1] = cur_n;

        const int64_t gid = cur_t == T_USER ? cur_n : (int64_t)g.n_users + cur_n;
        const int64_t lo = g.ptr[gid];
        const int deg = (int)(g.ptr[gid + 1] - lo);
        const int32_t* __restrict__ erel = g.rel + lo;
        const int32_t* __restrict__ edst = g.dst + lo;


        // Only rows the first pass marked n_acts = -1 arrive here: more than max_acts unvisited neighbours, so
        // (pgpr.py:258-288) float64 scores, keep the top max_acts by (score, position)
        if (deg > ACT_LDS_KEYS && (!scratch || scratch_stride < deg)) {  // host validation makes this unreachable
            if (tid == 0) n_acts[row] = 1;
            continue;
        }
        uint64_t* keys = deg <= ACT_LDS_KEYS ? lds_keys : scratch + (size_t)blockIdx.x * scratch_stride;
        // float64 score of every candidate (pgpr.py:266-281) from the per-search tables built by hw_pgpr_user_scores:
        //   next is a user            u . U[n]                      = score_user[b, n]
        //   next is an item           (u + r_ui) . E[n]             = score_entity[b, n]
        //   next is another entity    (u + r_ui + r) . E[n]         = score_entity[b, n] + score_rel_entity[r, n]
        // (b = batch index of the path's root user).  One or two 8-byte lookups per edge instead of a d-long
        // dot product over gathered float64 rows.
        {
            const int b = p[PATH_INTS - 1];
            const double* __restrict__ su = score_user + (size_t)b * g.n_users;
            const double* __restrict__ se = score_entity + (size_t)b * g.n_entities;
            // four edges per thread per trip, loads on clamped indices with the selects afterwards: the trip count of this
            // loop (deg / 1024) times two dependent load latencies is most of what a long row costs
            for (int e0 = tid; e0 < deg; e0 += 1024) {
                uint64_t k4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = e0 + 256 * u;
                    const int ec = e < deg ? e : deg - 1;
                    const int r = erel[ec], n = edst[ec];
                    const int nt = next_type(g, cur_t, r);
                    bool seen = false;
#pragma unroll
                    for (int k = 0; k < 4; ++k) seen |= (vis_t[k] == nt && vis_n[k] == n);
                    const bool to_user = nt == T_USER;
                    const double s1 = (to_user ? su : se)[n];
                    const double s2 = score_rel_entity[(size_t)r * g.n_entities + (to_user ? 0 : n)];
                    const double sc = (to_user || n < g.n_items) ? s1 : s1 + s2;
                    k4[u] = seen ? 0ull : orderable(sc);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (e0 + 256 * u < deg) keys[e0 + 256 * u] = k4[u];
            }
        }
        __syncthreads();

        // radix select: T = max_acts-th largest key; `need` = how many keys equal to T are kept
        uint64_t prefix = 0, mask = 0;
        int need = max_acts;
        for (int pass = 7; pass >= 0; --pass) {
            const int shift = pass * 8;
            hist[tid] = 0;
            __syncthreads();
            for (int e = tid; e < deg; e += 256) {
                const uint64_t k = keys[e];
                if ((k & mask) == prefix) atomicAdd(&hist[(unsigned)(k >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (tid < 64) {
                // lane l owns bins 4l..4l+3; suffix sums from the top bin down
                const unsigned h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
                const int mine = (int)(h0 + h1 + h2 + h3);
                int suf = mine;  // inclusive suffix over lanes >= tid
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int t = __shfl_down(suf, o, 64);
                    if (tid + o < 64) suf += t;
                }
                const int above = suf - mine;  // keys in higher lanes' bins
                if (above < need && need <= suf) {
                    int run = above, bin = 4 * tid + 3;
                    const unsigned hh[4] = {h0, h1, h2, h3};
#pragma unroll
                    for (int b = 3; b >= 0; --b) {
                        if (run < need && need <= run + (int)hh[b]) {
                            bin = 4 * tid + b;
                            break;
                        }
                        run += (int)hh[b];
                    }
                    sh_bin = bin;
                    sh_need = need - run;
                    // every key of the selected bin is wanted: nothing below this digit needs resolving
                    sh_done = (int)hh[bin & 3] == need - run ? 1 : 0;
                }
            }
            __syncthreads();
            prefix |= (uint64_t)sh_bin << shift;
            mask |= 0xFFull << shift;
            need = sh_need;
            const int done = sh_done;
            __syncthreads();
            if (done) {  // block-uniform: keep every key >= prefix (remaining digits 0), no ties to break
                prefix -= 1;
                need = 0;
                break;
            }
        }
        const uint64_t T = prefix;
        int c_eq_local = 0;
        for (int e = tid; e < deg; e += 256) c_eq_local += keys[e] == T ? 1 : 0;
        int c_eq;
        (void)block_excl_scan(c_eq_local, wsum, c_eq);
        const int tie_skip = c_eq - need;  // ties are kept from the END of the row (stable ascending argsort, [-max_acts:])

        if (!(fix_bug & 1)) {
            int run_keep = 0, run_tie = 0;
            for (int base = 0; base < deg; base += 256) {
                const int e = base + tid;
                const uint64_t k = e < deg ? keys[e] : 0ull;
                const int tie = (e < deg && k == T) ? 1 : 0;
                int tot_tie = 0, tpos = 0;
                if (c_eq > 0) tpos = block_excl_scan(tie, wsum, tot_tie);  // block-uniform: most rows have no key == T
                const int keep = (e < deg && (k > T || (tie && run_tie + tpos >= tie_skip))) ? 1 : 0;
                int tot_keep;
                const int kpos = block_excl_scan(keep, wsum, tot_keep);
                if (keep) {
                    arow[2 * (1 + run_keep + kpos)] = erel[e];
                    arow[2 * (1 + run_keep + kpos) + 1] = edst[e];
                }
                run_keep += tot_keep;
                run_tie += tot_tie;
            }
        } else {
            // fix_scores_sorting_bug: descending (score, position) order (pgpr.py:286).  Collect the survivors'
            // (key, edge) in LDS, then rank each against all others.
            __shared__ int sel_e[1024];
            __shared__ uint64_t sel_k[1024];
            int run_keep = 0, run_tie = 0;
            for (int base = 0; base < deg; base += 256) {
                const int e = base + tid;
                const uint64_t k = e < deg ? keys[e] : 0ull;
                const int tie = (e < deg && k == T) ? 1 : 0;
                int tot_tie;
                const int tpos = block_excl_scan(tie, wsum, tot_tie);
                const int keep = (e < deg && (k > T || (tie && run_tie + tpos >= tie_skip))) ? 1 : 0;
                int tot_keep;
                const int kpos = block_excl_scan(keep, wsum, tot_keep);
                if (keep) sel_e[run_keep + kpos] = e, sel_k[run_keep + kpos] = k;
                run_keep += tot_keep;
                run_tie += tot_tie;
            }
            __syncthreads();
            for (int i = tid; i < max_acts; i += 256) {
                const uint64_t ki = sel_k[i];
                const int ei = sel_e[i];
                int rank = 0;
                for (int j = 0; j < max_acts; ++j) rank += (sel_k[j] > ki || (sel_k[j] == ki && sel_e[j] > ei)) ? 1 : 0;
                arow[2 * (1 + rank)] = erel[ei];
                arow[2 * (1 + rank) + 1] = edst[ei];
            }
        }
        if (tid == 0) n_acts[row] = 1 + max_acts;
    }
}

// ---- float64 score tables for the trim --------------------------------------------------------------------
// src[s, :] rows: 0 = U[user_s] (mode 0), U[user_s] + R[ui] (mode 1)
// V64 (nullable): per-user replacement of the R[ui] row (TPRec's temporal vector sum_c w[u][c] C[c], tprec.py:454-460)
__global__ void pgpr_user_src_kernel(const double* __restrict__ U64, const double* __restrict__ R64,
                                     const double* __restrict__ V64, int d, int ui_relation,
                                     const int64_t* __restrict__ users, int n, int n_users_table,
                                     double* __restrict__ src0, double* __restrict__ src1) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * d) return;
    const int s = i / d, k = i % d;
    int64_t u = users[s];
    u = u < 0 ? 0 : (u >= n_users_table ? n_users_table - 1 : u);
    const double v = U64[(size_t)u * d + k];
    src0[i] = v;
    src1[i] = v + (V64 ? V64[(size_t)u * d + k] : R64[(size_t)ui_relation * d + k]);
}

// out[s, n] = sum_k src[s, k] * T[n, k]   (float64, k ascending: one fixed summation order)
constexpr int DOT_SB = 8;
__global__ __launch_bounds__(256) void batched_dot64_kernel(const double* __restrict__ src, int n_src,
                                                            const double* __restrict__ T, int n_rows, int d,
                                                            double* __restrict__ out) {
    __shared__ double sv[DOT_SB][HW_PGPR_MAX_DIM];
    const int s0 = blockIdx.y * DOT_SB;
    const int ns = min(DOT_SB, n_src - s0);
    for (int i = threadIdx.x; i < ns * d; i += 256) sv[i / d][i % d] = src[(size_t)(s0 + i / d) * d + i % d];
    __syncthreads();
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= n_rows) return;
    double acc[DOT_SB];
#pragma unroll
    for (int q = 0; q < DOT_SB; ++q) acc[q] = 0.0;
    const double* __restrict__ t = T + (size_t)n * d;
    for (int k = 0; k < d; ++k) {
        const double tv = t[k];
#pragma unroll
        for (int q = 0; q < DOT_SB; ++q) acc[q] = fma(sv[q][k], tv, acc[q]);
    }
    for (int q = 0; q < ns; ++q) out[(size_t)(s0 + q) * n_rows + n] = acc[q];
}

// The same sums for many sources (a PGPR training batch: 2 048 users) as a register-tiled product: the kernel above gives
// every lane its own table row to walk (64 cache lines per load instruction) and re-reads the table once per 8 sources --
// 0.65 ms of a 1.8 ms training step for the two [users, n_users | n_entities] tables.  Here a block owns 128 sources x 64
// table rows, both tiles staged k-major in LDS with coalesced loads, a thread 8 sources x 4 rows in registers.  Every output
// still runs ONE fma chain over ascending k with the same operands, so the tables are bit-identical (the trims compare them).
constexpr int DRT_S = 128, DRT_N = 64, DRT_KC = 16;
__global__ __launch_bounds__(256) void batched_dot64_rt_kernel(const double* __restrict__ src, int n_src,
                                                               const double* __restrict__ T, int n_rows, int d,
                                                               double* __restrict__ out) {
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    __shared__ __attribute__((aligned(16))) double sS[DRT_KC][DRT_S + 2];
    __shared__ __attribute__((aligned(16))) double sT[DRT_KC][DRT_N + 2];
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    const int s0 = blockIdx.y * DRT_S, row0 = blockIdx.x * DRT_N;
    double acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
    for (int k0 = 0; k0 < d; k0 += DRT_KC) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {  // 128 sources x 16 k: a 128-byte piece of a source row per 16 threads
            const int idx = j * 256 + tid, sl = idx >> 4, kk = idx & 15;
            const int sg = s0 + sl, k = k0 + kk;
            sS[kk][sl] = (sg < n_src && k < d) ? src[(size_t)sg * d + k] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = j * 256 + tid, rl = idx >> 4, kk = idx & 15;
            const int rg = row0 + rl, k = k0 + kk;
            sT[kk][rl] = (rg < n_rows && k < d) ? T[(size_t)rg * d + k] : 0.0;
        }
        __syncthreads();
        const int kn = d - k0 < DRT_KC ? d - k0 : DRT_KC;
        for (int kk = 0; kk < kn; ++kk) {
            double a[8], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f64x2 v = *reinterpret_cast<const f64x2*>(&sS[kk][8 * ty + 2 * i]);
                a[2 * i] = v.x, a[2 * i + 1] = v.y;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f64x2 v = *reinterpret_cast<const f64x2*>(&sT[kk][4 * tx + 2 * j]);
                b[2 * j] = v.x, b[2 * j + 1] = v.y;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int sg = s0 + 8 * ty + i;
        if (sg >= n_src) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int rg = row0 + 4 * tx + j;
            if (rg < n_rows) out[(size_t)sg * n_rows + rg] = acc[i][j];
        }
    }
}

// ---- A11: state ---------------------------------------------------------------------------------------
// state[row] = [u, cur, last_node, last_rel (, older_node, older_rel)] gathered from the float32 tables
// (zeros for missing history and for the self-loop relation), pgpr.py:296-333 + KGState :568-580.
__global__ __launch_bounds__(256) void pgpr_state_kernel(hw_pgpr_graph g, const float* __restrict__ U32,
                                                         const float* __restrict__ E32, const float* __restrict__ R32,
                                                         int d, int n_parts, const int32_t* __restrict__ paths,
                                                         const int32_t* __restrict__ n_rows_ptr,
                                                         float* __restrict__ state) {
    const int n_rows = n_rows_ptr[0];
    const int sd = n_parts * d;
    const int64_t total = (int64_t)n_rows * sd;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / sd);
        const int c = (int)(i % sd);
        const int part = c / d, k = c % d;
        const int32_t* p = paths + (size_t)row * PATH_INTS;
        const int uid = p[0], len = p[1];
        float v = 0.f;
        auto node_emb = [&](int j) -> float {  // node j of the path (0 = the user)
            if (j == 0) return U32[(size_t)uid * d + k];
            const int t = p[2 + 3 * (j - 1) + 1], n = p[2 + 3 * (j - 1) + 2];
            return t == T_USER ? U32[(size_t)n * d + k] : E32[(size_t)n * d + k];
        };
        auto rel_emb = [&](int j) -> float {  // relation that led to node j (j >= 1)
            const int r = p[2 + 3 * (j - 1)];
            return r == SELF_LOOP ? 0.f : R32[(size_t)r * d + k];
        };
        if (part == 0) v = U32[(size_t)uid * d + k];
        else if (len == 1) v = part == 1 ? U32[(size_t)uid * d + k] : 0.f;
        else if (part == 1) v = node_emb(len - 1);
        else if (part == 2) v = node_emb(len - 2);
        else if (part == 3) v = rel_emb(len - 1);
        else if (len >= 3 && part == 4) v = node_emb(len - 3);
        else if (len >= 3 && part == 5) v = rel_emb(len - 2);
        state[i] = v;
    }
}

// ---- A12, first layer by table lookup ------------------------------------------------------------------------------
// l1(state) = W1 . [u, cur, last, rel, older, older_rel] + b splits into one term per slot, and every slot is a row of a
// small table (users, entities, relations): T_slot = table . W1[:, slot]^T is computed once per weight version
// ([rows, H] each, ~90 MB for ml-1m-small), after which a beam row's first layer is a sum of 4 (2, 6) gathered H-float
// rows instead of a 400-long dot product per output -- 2 KB gathers from L2 / Infinity Cache against 2 x 400 x 512 flops.
// Same ELU / dropout epilogue (and the same Philox words) as the GEMM path; the state tensor is never formed.
struct L1Tables {
    const float* user0;                         // slot 0: the path's user
    const float *cur_u, *cur_e;                 // slot 1: current node, by node type
    const float *last_u, *last_e, *last_r;      // slots 2, 3
    const float *old_u, *old_e, *old_r;         // slots 4, 5
};

constexpr int L1_RUN = 16, L1_MAX_H = 512;

// ELU + dropout of four consecutive outputs (row, c .. c + 3) and their store
__device__ __forceinline__ void l1_finish(f32x4 acc, int row, int c, int H, const float* __restrict__ drop_mask, int drop_mode,
                                          uint64_t seed, uint64_t offset, float* __restrict__ out, int64_t ld) {
    float v[4] = {acc.x, acc.y, acc.z, acc.w};
    // row * H + c is a multiple of 4 (H % 4 == 0, c = 4 * lane + 256 * trip): the four dropout bits of this chunk are the
    // four words of ONE Philox block (the compiler cannot see that and would evaluate it four times)
    uint32_t pw[4] = {0u, 0u, 0u, 0u};
    if (drop_mode == 2) philox_bits4(seed, offset, ((uint64_t)row * H + c) >> 2, pw);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float x = v[k];
        x = x > 0.f ? x : expm1f(x);
        const uint64_t idx = (uint64_t)row * H + c + k;
        if (drop_mode == 1) x *= drop_mask[idx];
        else if (drop_mode == 2) x = (pw[k] & 1u) ? x * 2.f : 0.f;
        v[k] = x;
    }
    *reinterpret_cast<f32x4*>(out + (size_t)row * ld + c) = f32x4{v[0], v[1], v[2], v[3]};
}

__global__ __launch_bounds__(256) void pgpr_layer1_lookup_kernel(L1Tables t, const float* __restrict__ bias, int H,
                                                                 int n_parts, const int32_t* __restrict__ paths,
                                                                 const int32_t* __restrict__ n_rows_ptr,
                                                                 const float* __restrict__ drop_mask, int drop_mode,
                                                                 uint64_t seed, uint64_t offset, float* __restrict__ out,
                                                                 int64_t ld, int run) {
    if (drop_mode == 2) offset = philox_offset(offset);
    const int n_rows = n_rows_ptr[0];
    const int lane = threadIdx.x & 63;
    const bool keep = H <= L1_MAX_H;
    // A wave walks `run` consecutive rows (1 for the small first hops, L1_RUN once there are rows to spare).  The children
    // of one parent path are consecutive (pgpr_expand_kernel) and share
    // the user row and the last-node row, so those two gathered rows stay in registers while their addresses repeat: two
    // 2 KB gathers per row instead of four on the last hop, the same additions in the same order.
    const f32x4* kept_ptr[2] = {nullptr, nullptr};  // the rows held in kept[]: slot 0 (user) and slot 2 (last node)
    f32x4 kept[2][L1_MAX_H / 256];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int t = 0; t < L1_MAX_H / 256; ++t) kept[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int chunk = blockIdx.x * 4 + wave_in_block(); chunk * run < n_rows; chunk += gridDim.x * 4) {
        for (int row_v = chunk * run; row_v < min(n_rows, (chunk + 1) * run); ++row_v) {
            // the path record in one go, through the scalar unit (the row index is wave-uniform; saying so lets the compiler
            // use s_load): indexing the record by a length read from it would chain three memory round trips per row
            const int row = __builtin_amdgcn_readfirstlane(row_v);
            static_assert(PATH_INTS == 12, "three 16-byte pieces");
            const int4* pr = reinterpret_cast<const int4*>(paths + (size_t)row * PATH_INTS);
            const int4 q0 = pr[0], q1 = pr[1], q2 = pr[2];
            const int uid = q0.x, len = q0.y;
            // hop j = 1..3: (relation, node type, node id)
            const int hr[3] = {q0.z, q1.y, q2.x}, ht[3] = {q0.w, q1.z, q2.y}, hn[3] = {q1.x, q1.w, q2.z};
            auto pick = [](const int (&a)[3], int j) { return j == 1 ? a[0] : (j == 2 ? a[1] : a[2]); };
            // rows of the slot tables (nullptr = zero contribution), following pgpr_state_kernel / _get_state
            const float* src[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
            src[0] = t.user0 + (size_t)uid * H;
            auto node_row = [&](int j, const float* tu, const float* te) -> const float* {
                if (j == 0) return tu + (size_t)uid * H;
                return (pick(ht, j) == T_USER ? tu : te) + (size_t)pick(hn, j) * H;
            };
            auto rel_row = [&](int j, const float* tr) -> const float* {
                const int r = pick(hr, j);
                return r == SELF_LOOP ? nullptr : tr + (size_t)r * H;
            };
            if (len == 1) {
                src[1] = t.cur_u + (size_t)uid * H;
            } else {
                src[1] = node_row(len - 1, t.cur_u, t.cur_e);
                if (n_parts > 2) {
                    src[2] = node_row(len - 2, t.last_u, t.last_e);
                    src[3] = rel_row(len - 1, t.last_r);
                }
                if (n_parts > 4 && len >= 3) {
                    src[4] = node_row(len - 3, t.old_u, t.old_e);
                    src[5] = rel_row(len - 2, t.old_r);
                }
            }
            // refresh the kept rows whose address changed (wave-uniform: every lane reads the same path record)
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const f32x4* want = reinterpret_cast<const f32x4*>(src[2 * a]);
                if (keep && want != kept_ptr[a]) {
                    kept_ptr[a] = want;
                    if (want) {
#pragma unroll
                        for (int tt = 0; tt < L1_MAX_H / 256; ++tt)
                            if (lane * 4 + 256 * tt < H) kept[a][tt] = want[lane + 64 * tt];
                    }
                }
            }
#pragma unroll
            for (int tt = 0; tt < L1_MAX_H / 256; ++tt) {  // H % 4 == 0, H <= L1_MAX_H when `keep` (checked by the launcher)
                const int c = lane * 4 + 256 * tt;
                if (c >= H) break;
                f32x4 acc = *reinterpret_cast<const f32x4*>(bias + c);
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    if (!src[q]) continue;
                    if (keep && (q == 0 || q == 2)) acc += kept[q >> 1][tt];
                    else acc += *reinterpret_cast<const f32x4*>(src[q] + c);
                }
                l1_finish(acc, row, c, H, drop_mask, drop_mode, seed, offset, out, ld);
            }
            if (H > L1_MAX_H)
                for (int c = lane * 4 + L1_MAX_H; c < H; c += 256) {  // wider hidden layers: the plain loop for the rest
                    f32x4 acc = *reinterpret_cast<const f32x4*>(bias + c);
#pragma unroll
                    for (int q = 0; q < 6; ++q)
                        if (src[q]) acc += *reinterpret_cast<const f32x4*>(src[q] + c);
                    l1_finish(acc, row, c, H, drop_mask, drop_mode, seed, offset, out, ld);
                }
        }
    }
}

// ---- A12: the policy-MLP layers run on the generic fp32-MFMA GEMM of gemm.hip (hw_mlp_layer) ---------------

// ---- A12/A13: masked softmax, +mask, top-k (one wave per row) -------------------------------------------
// probs = softmax(logits masked to the first n_acts entries); value used for ranking and stored as the
// path's hop "prob" is p + 1 for valid actions (pgpr.py:453).  Top-k ordered (value desc, index asc);
// only the first min(k, n_acts) entries are meaningful -- the rest get idx = -1.
template <int NPL>
__global__ __launch_bounds__(256) void policy_topk_kernel(const float* __restrict__ logits, int ld, int act_dim,
                                                          const int32_t* __restrict__ n_acts,
                                                          const int32_t* __restrict__ n_rows_ptr, int k,
                                                          int32_t* __restrict__ out_idx, float* __restrict__ out_val,
                                                          float* __restrict__ probs_out) {
    const int n_rows = n_rows_ptr[0];
    const int lane = threadIdx.x & 63;
    for (int row = blockIdx.x * 4 + wave_in_block(); row < n_rows; row += gridDim.x * 4) {
        const int na = n_acts[row];
        const float* __restrict__ lr = logits + (size_t)row * ld;
        float x[NPL];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const int j = lane + 64 * i;
            x[i] = (j < na && j < act_dim) ? lr[j] : -INFINITY;
            mx = fmaxf(mx, x[i]);
        }
        mx = wave_max_f32(mx);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            x[i] = x[i] == -INFINITY ? 0.f : expf(x[i] - mx);
            sum += x[i];
        }
        sum = group_sum<64>(sum);
        uint64_t key[NPL];
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const int j = lane + 64 * i;
            const float pr = x[i] / sum;
            if (probs_out && j < act_dim) probs_out[(size_t)row * act_dim + j] = j < na ? pr : 0.f;
            const float v = j < na ? pr + 1.0f : 0.f;
            // valid values are >= 1 > 0: positive floats order as their bit patterns
            key[i] = j < act_dim ? (((uint64_t)__float_as_uint(v) << 32) | (uint32_t)(~(uint32_t)j)) : 0ull;
        }
        uint64_t bound = ~0ull;
        for (int r = 0; r < k; ++r) {
            uint64_t best = 0;
#pragma unroll
            for (int i = 0; i < NPL; ++i)
                if (key[i] < bound && key[i] > best) best = key[i];
            best = wave_max_u64(best);
            if (lane == 0) {
                const int idx = (int)(~(uint32_t)(best & 0xffffffffu));
                const bool ok = best != 0 && idx < na;
                out_idx[(size_t)row * k + r] = ok ? idx : -1;
                out_val[(size_t)row * k + r] = ok ? __uint_as_float((uint32_t)(best >> 32)) : 0.f;
            }
            bound = best;
            if (best == 0) bound = 0;
        }
    }
}

// Fast path for act_dim <= 256 (the reference default, max_acts = 250): all 256 slots of a row live in one wave
// (4 per lane) and are ranked by ONE bitonic sort instead of k arg-max rounds.  p + 1 lies in [1, 2], so its 23
// mantissa bits order the values; with the 8-bit action index underneath, a 32-bit key gives exactly
// (value desc, index asc).  p + 1 == 2 only happens for a single valid action (softmax of one logit); that row needs
// no sort.  21 of the 36 compare-exchange steps cross lanes (xor_lane: DPP for 17 of them); the rest are register swaps.
__device__ __forceinline__ void cmpx(uint32_t& a, uint32_t& b, bool desc) {
    const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
    a = desc ? hi : lo;
    b = desc ? lo : hi;
}

// lane ^ LM's value without the LDS crossbar where the data-parallel primitives reach: quad permutes for 1 and 2, a pair of
// row shifts for 4 and 8 (the partner is 4 / 8 lanes up for lanes with that bit clear, down for the others, always inside
// the 16-lane row), ds_swizzle (no address register) for 16; only the 32-lane exchange goes through ds_bpermute.
template <int LM>
__device__ __forceinline__ uint32_t xor_lane(uint32_t v, int lane) {
    const int x = (int)v;
    if constexpr (LM == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true);       // quad_perm [1,0,3,2]
    else if constexpr (LM == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
    else if constexpr (LM == 4 || LM == 8) {
        const int up = __builtin_amdgcn_update_dpp(0, x, 0x100 + LM, 0xF, 0xF, true);    // row_shl: lane i reads lane i + LM
        const int down = __builtin_amdgcn_update_dpp(0, x, 0x110 + LM, 0xF, 0xF, true);  // row_shr: lane i reads lane i - LM
        return (uint32_t)((lane & LM) ? down : up);
    } else if constexpr (LM == 16) return (uint32_t)__builtin_amdgcn_ds_swizzle(x, 0x401F);  // bit mode: and 0x1F, xor 0x10
    else return (uint32_t)__shfl_xor(x, LM, 64);
}

// one compare-exchange step of the bitonic network over e = lane * 4 + i (descending overall), then the smaller strides
template <int SIZE, int STRIDE>
__device__ __forceinline__ void bitonic_steps(uint32_t (&key)[4], int lane) {
    if constexpr (STRIDE >= 4) {
        constexpr int LM = STRIDE >> 2;  // partner lane = lane ^ LM, same register
        const bool upper = (lane & LM) != 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = lane * 4 + i;
            const bool desc = (e & SIZE) == 0;  // blocks alternate direction; the final pass (SIZE 256) is all descending
            const uint32_t o = xor_lane<LM>(key[i], lane);
            const bool take_max = desc ? !upper : upper;
            key[i] = take_max ? (key[i] > o ? key[i] : o) : (key[i] < o ? key[i] : o);
        }
    } else if constexpr (STRIDE == 2) {  // partners inside the lane: registers i and i ^ STRIDE
        cmpx(key[0], key[2], ((lane * 4 + 0) & SIZE) == 0);
        cmpx(key[1], key[3], ((lane * 4 + 1) & SIZE) == 0);
    } else {
        cmpx(key[0], key[1], ((lane * 4 + 0) & SIZE) == 0);
        cmpx(key[2], key[3], ((lane * 4 + 2) & SIZE) == 0);
    }
    if constexpr (STRIDE > 1) bitonic_steps<SIZE, STRIDE / 2>(key, lane);
}

template <int SIZE>
__device__ __forceinline__ void bitonic_sort256(uint32_t (&key)[4], int lane) {
    bitonic_steps<SIZE, SIZE / 2>(key, lane);
    if constexpr (SIZE < 256) bitonic_sort256<SIZE * 2>(key, lane);
}

// bitonic sort of one key per lane, descending
template <int SIZE, int STRIDE>
__device__ __forceinline__ void sort64_steps(uint32_t& key, int lane) {
    const uint32_t o = xor_lane<STRIDE>(key, lane);
    const bool desc = (lane & SIZE) == 0, upper = (lane & STRIDE) != 0;
    const bool take_max = desc ? !upper : upper;
    key = take_max ? (key > o ? key : o) : (key < o ? key : o);
    if constexpr (STRIDE > 1) sort64_steps<SIZE, STRIDE / 2>(key, lane);
}
template <int SIZE>
__device__ __forceinline__ void sort64(uint32_t& key, int lane) {
    sort64_steps<SIZE, SIZE / 2>(key, lane);
    if constexpr (SIZE < 64) sort64<SIZE * 2>(key, lane);
}

// Masked softmax of one row held by one wave (blocked layout: lane owns elements 4 * lane .. 4 * lane + 3) and its ranking keys:
// (mantissa of p + 1) << 8 | (255 - index), top bit set for valid actions, 0 otherwise -- a plain unsigned compare gives
// (value desc, index asc).  p + 1 rounds to exactly 2.0 when p >= 1 - 2^-25; such a row cannot also hold 1.99999988 (the
// probabilities sum to 1), so 2.0 shares the top mantissa code and `row_has_two` tells them apart.
// NPL = slots per lane: 4 covers 256 actions; rows with at most 128 actions (97 % of the last hop of a 64/64/64 beam) are
// ranked with NPL = 2 -- the ranking is bound by vector-instruction issue, and half the slots are half the exps, key builds
// and counting compares.  The sum of a row's exps is taken in a different order in the two forms (an ulp of p): every
// kernel that ranks a row picks the form by the same rule (rank_slots), so they still agree bit for bit.
__device__ __forceinline__ int rank_slots(int na) { return na <= 128 ? 2 : 4; }
// the row's raw logits in the blocked layout of NPL slots per lane (element j = lane * NPL + i; clamped index, select
// afterwards: the loads are issued together)
template <int NPL>
__device__ __forceinline__ void row_logits_load(const float* __restrict__ lr, int act_dim, int lane, float (&x)[NPL]) {
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int j = lane * NPL + i;
        x[i] = lr[j < act_dim ? j : act_dim - 1];
    }
}

// ranking keys of a row from its raw logits (row_logits_load's registers)
template <int NPL>
__device__ __forceinline__ void row_softmax_keys_of(float (&x)[NPL], int act_dim, int na, int lane,
                                                    float* __restrict__ probs_row, uint32_t (&key)[NPL], bool& row_has_two) {
    // p is rounded to float32 before the + 1 (probs + mask.float() on a float32 tensor, pgpr.py:453): no fused multiply-add,
    // or the same row would rank differently depending on which kernel the function was inlined into
#pragma clang fp contract(off)
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int j = lane * NPL + i;
        x[i] = (j < na && j < act_dim) ? x[i] : -INFINITY;
        mx = fmaxf(mx, x[i]);
    }
    mx = wave_max_f32(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        x[i] = x[i] == -INFINITY ? 0.f : __expf(x[i] - mx);  // v_exp_f32: the ranking key keeps 23 bits of p + 1 anyway
        sum += x[i];
    }
    sum = group_sum<64>(sum);
    const float inv_sum = 1.f / sum;  // one division per row
    bool two = false;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int j = lane * NPL + i;
        const float pr = x[i] * inv_sum;
        if (probs_row && j < act_dim) probs_row[j] = j < na ? pr : 0.f;
        const uint32_t bits = __float_as_uint(pr + 1.0f);  // [1, 2]: 0x3F800000 .. 0x40000000
        const bool valid = j < na && j < act_dim;
        two |= valid && bits >= 0x40000000u;
        const uint32_t mant = bits >= 0x40000000u ? 0x7FFFFFu : (bits & 0x7FFFFFu);
        key[i] = valid ? ((mant << 8) | (uint32_t)(255 - j)) + 0x80000000u : 0u;
    }
    row_has_two = __any(two);
}

template <int NPL>
__device__ __forceinline__ void row_softmax_keys(const float* __restrict__ lr, int act_dim, int na, int lane,
                                                 float* __restrict__ probs_row, uint32_t (&key)[NPL], bool& row_has_two) {
    float x[NPL];
    row_logits_load<NPL>(lr, act_dim, lane, x);
    row_softmax_keys_of<NPL>(x, act_dim, na, lane, probs_row, key, row_has_two);
}

__device__ __forceinline__ void decode_topk_key(uint32_t kk, bool row_has_two, int& idx, float& v) {
    idx = 255 - (int)(kk & 0xFFu);
    const uint32_t mant = (kk & 0x7FFFFFFFu) >> 8;
    v = (row_has_two && mant == 0x7FFFFFu) ? 2.0f : __uint_as_float(0x3F800000u | mant);
}

// The k (<= 64) largest keys of a row, one per lane in rank order (lane j = rank j; 0 where the row has fewer valid actions):
// the k-th largest key is found by a bit-by-bit search with wave ballots (4 compares per bit, the counting on the scalar
// unit; the search stops as soon as exactly k keys lie above the candidate), the k survivors are compacted to one per lane
// through `surv` (64 words of LDS owned by the wave) and only those 64 are sorted: ~1/3 of the vector instructions of sorting
// all 256.
template <int NPL>
__device__ __forceinline__ uint32_t select_sort_topk(const uint32_t (&key)[NPL], int k, int lane, uint32_t* surv) {
    auto count_ge = [&](uint32_t c) {
        int n = 0;
#pragma unroll
        for (int i = 0; i < NPL; ++i) n += __builtin_popcountll(__ballot(key[i] >= c));
        return n;  // wave-uniform
    };
    // valid keys have the top bit set and are distinct; T = the k-th largest of them (all of them if fewer than k)
    uint32_t T = 0x80000000u;
    if (count_ge(T) > k) {
        // no key has a bit set above the largest key's highest one: the search starts there (probabilities of a
        // 250-way softmax leave the top ~8 mantissa bits of p + 1 clear, i.e. ~8 of the ~25 rounds)
        uint32_t mk = key[0];
#pragma unroll
        for (int i = 1; i < NPL; ++i) mk = max(mk, key[i]);
        mk = wave_max_u32(mk);
        const int top = 31 - __builtin_clz((__builtin_amdgcn_readfirstlane(mk) & 0x7FFFFFFFu) | 1u);
        for (int bit = top; bit >= 0; --bit) {
            const uint32_t cand = T | (1u << bit);
            const int c = count_ge(cand);
            if (c >= k) T = cand;
            if (c == k) break;
        }
    }
    surv[lane] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int base = 0;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const bool sel = key[i] >= T;  // T >= 0x80000000: invalid slots (key 0) never qualify
        const uint64_t m = __ballot(sel);
        const int pos = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
        if (sel) surv[pos] = key[i];
        base += __builtin_popcountll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint32_t one = surv[lane];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // the next row of this wave overwrites surv
    sort64<2>(one, lane);
    return one;
}

// SELECT (k <= 64): select_sort_topk; otherwise all 256 keys go through one bitonic sort.  Same output.
template <bool SELECT>
__global__ __launch_bounds__(256) void policy_topk_sort_kernel(const float* __restrict__ logits, int ld, int act_dim,
                                                               const int32_t* __restrict__ n_acts,
                                                               const int32_t* __restrict__ n_rows_ptr, int k,
                                                               int32_t* __restrict__ out_idx, float* __restrict__ out_val,
                                                               float* __restrict__ probs_out) {
    const int n_rows = n_rows_ptr[0];
    const int lane = threadIdx.x & 63;
    for (int row = blockIdx.x * 4 + wave_in_block(); row < n_rows; row += gridDim.x * 4) {
        const int na = n_acts[row];
        uint32_t key[4];
        bool row_has_two;
        // the two-slot form leaves probs_out past 2 * 64 unwritten: only without probs_out (the beam search never asks for it)
        const bool two_slots = SELECT && !probs_out && rank_slots(na) == 2;  // wave-uniform
        if (!two_slots)
            row_softmax_keys<4>(logits + (size_t)row * ld, act_dim, na, lane,
                                probs_out ? probs_out + (size_t)row * act_dim : nullptr, key, row_has_two);
        auto emit = [&](uint32_t kk, int rnk) {
            const bool ok = kk != 0u;
            int idx;
            float v;
            decode_topk_key(kk, row_has_two, idx, v);
            out_idx[(size_t)row * k + rnk] = ok ? idx : -1;
            out_val[(size_t)row * k + rnk] = ok ? v : 0.f;
        };
        if constexpr (SELECT) {
            __shared__ uint32_t surv_all[4][64];
            uint32_t one;
            if (two_slots) {
                uint32_t key2[2];
                row_softmax_keys<2>(logits + (size_t)row * ld, act_dim, na, lane, nullptr, key2, row_has_two);
                one = select_sort_topk<2>(key2, k, lane, surv_all[threadIdx.x >> 6]);
            } else {
                one = select_sort_topk<4>(key, k, lane, surv_all[threadIdx.x >> 6]);
            }
            if (lane < k) emit(one, lane);
        } else {
            bitonic_sort256<2>(key, lane);  // descending overall: element index e = lane * 4 + i
            // ranks 4*lane .. 4*lane+3 are in key[0..3]; emit the first k
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (lane * 4 + i < k) emit(key[i], lane * 4 + i);
        }
    }
}

// ---- A13: expand -----------------------------------------------------------------------------------------
// n_child[row] = min(k, n_acts[row]); exclusive scan -> child offsets + next row count.  Three small launches:
// per-block sums (1024 rows per block), one block scans the block sums, every block rescans its rows with its base.
constexpr int OFF_ROWS_PER_BLOCK = 1024;

__global__ __launch_bounds__(256) void pgpr_offsets_partial_kernel(const int32_t* __restrict__ n_acts,
                                                                  const int32_t* __restrict__ n_rows_ptr, int k,
                                                                  int32_t* __restrict__ block_sums) {
    __shared__ int wsum[4];
    const int n = n_rows_ptr[0];
    const int base = blockIdx.x * OFF_ROWS_PER_BLOCK;
    int s = 0;
    for (int i = threadIdx.x; i < OFF_ROWS_PER_BLOCK; i += 256) {
        const int r = base + i;
        s += r < n ? min(k, n_acts[r]) : 0;
    }
    int total;
    (void)block_excl_scan(s, wsum, total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(1024) void pgpr_offsets_scan_kernel(int32_t* __restrict__ block_sums, int n_blocks,
                                                                 int32_t* __restrict__ n_rows_next, int64_t cap) {
    __shared__ int part[1024];
    // n_blocks <= 1024 * 64: each thread owns a contiguous run of block sums
    const int per = (n_blocks + 1023) / 1024;
    const int lo = min((int)threadIdx.x * per, n_blocks), hi = min(lo + per, n_blocks);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += block_sums[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int t = (int)threadIdx.x >= o ? part[threadIdx.x - o] : 0;
        __syncthreads();
        part[threadIdx.x] += t;
        __syncthreads();
    }
    int run = part[threadIdx.x] - s;
    for (int i = lo; i < hi; ++i) {
        const int v = block_sums[i];
        block_sums[i] = run;  // exclusive base of block i
        run += v;
    }
    if (threadIdx.x == 1023) n_rows_next[0] = (int)min((int64_t)part[1023], cap);
}

__global__ __launch_bounds__(256) void pgpr_offsets_final_kernel(const int32_t* __restrict__ n_acts,
                                                                const int32_t* __restrict__ n_rows_ptr, int k,
                                                                const int32_t* __restrict__ block_base,
                                                                int32_t* __restrict__ offsets) {
    __shared__ int wsum[4];
    const int n = n_rows_ptr[0];
    const int base = blockIdx.x * OFF_ROWS_PER_BLOCK;
    if (base >= n) return;
    int run = block_base[blockIdx.x];
    for (int c = 0; c < OFF_ROWS_PER_BLOCK; c += 256) {  // uniform trip count: block_excl_scan synchronises
        const int r = base + c + threadIdx.x;
        const int v = r < n ? min(k, n_acts[r]) : 0;
        int total;
        const int ex = block_excl_scan(v, wsum, total);
        if (r < n) offsets[r] = run + ex;
        run += total;
    }
}

__global__ __launch_bounds__(256) void pgpr_expand_kernel(hw_pgpr_graph g, const int32_t* __restrict__ paths,
                                                          const float* __restrict__ probs,
                                                          const int32_t* __restrict__ acts, int act_dim,
                                                          const int32_t* __restrict__ n_acts,
                                                          const int32_t* __restrict__ n_rows_ptr,
                                                          const int32_t* __restrict__ topk_idx,
                                                          const float* __restrict__ topk_val, int k,
                                                          const int32_t* __restrict__ offsets,
                                                          int32_t* __restrict__ paths_out, float* __restrict__ probs_out,
                                                          int64_t cap) {
    const int n_rows = n_rows_ptr[0];
    const int64_t total = (int64_t)n_rows * k;
    // Children of consecutive (row, j) land on consecutive output slots (offsets is the exclusive scan of the child
    // counts), so the records a wave produces form one contiguous run.  Each lane builds its 48-byte record + 12-byte
    // probability triple in LDS and the wave then streams the run out as consecutive 16-byte / 4-byte pieces: full cache
    // lines per store instruction instead of 64 lanes each touching a third of a line three times.
    __shared__ __attribute__((aligned(16))) int rec_lds[4][64 * PATH_INTS];
    __shared__ float prob_lds[4][64 * 3];
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, w = wave_in_block();
    const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    const int64_t n_iter = (total + 63) / 64;
    for (int64_t it = (int64_t)blockIdx.x * 4 + w; it < n_iter; it += (int64_t)gridDim.x * 4) {  // wave-uniform trips
        const int64_t i = it * 64 + lane;
        bool live = i < total;
        const int row = live ? (int)(i / k) : 0, j = live ? (int)(i % k) : 0;
        live = live && j < min(k, n_acts[row]);
        const int64_t o = live ? (int64_t)offsets[row] + j : 0;
        live = live && o < cap;
        const unsigned long long m = __ballot(live);
        if (m == 0ull) continue;
        const int slot = __popcll(m & lt_mask);  // position inside the wave's run
        const int first = __ffsll((long long)m) - 1;
        const int64_t o0 = __shfl(o, first, 64);  // the run starts at the first live lane's slot
        const int n_live = __popcll(m);
        if (live) {
            const int idx = topk_idx[(size_t)row * k + j];
            const i32x4* __restrict__ pv = reinterpret_cast<const i32x4*>(paths + (size_t)row * PATH_INTS);
            i32x4 rec[3] = {pv[0], pv[1], pv[2]};
            int* f = reinterpret_cast<int*>(rec);
            const int len = f[1];
            const int cur_t = len == 1 ? T_USER : f[2 + 3 * (len - 2) + 1];
            const int2 act = *reinterpret_cast<const int2*>(&acts[((size_t)row * act_dim + idx) * 2]);
            const int rel = act.x, node = act.y;
            const int nt = rel == SELF_LOOP ? cur_t : next_type(g, cur_t, rel);
            f[1] = len + 1;
            f[2 + 3 * (len - 1)] = rel;
            f[2 + 3 * (len - 1) + 1] = nt;
            f[2 + 3 * (len - 1) + 2] = node;
            i32x4* q = reinterpret_cast<i32x4*>(&rec_lds[w][slot * PATH_INTS]);
            q[0] = rec[0], q[1] = rec[1], q[2] = rec[2];
            const float* pp = probs + (size_t)row * 3;
            float pr[3] = {pp[0], pp[1], pp[2]};
            pr[len - 1] = topk_val[(size_t)row * k + j];
            prob_lds[w][slot * 3 + 0] = pr[0], prob_lds[w][slot * 3 + 1] = pr[1], prob_lds[w][slot * 3 + 2] = pr[2];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // live lanes of one trip are consecutive children, but a trip may span two rows whose runs are adjacent as well:
        // slots o0 .. o0 + n_live - 1
        i32x4* __restrict__ qo = reinterpret_cast<i32x4*>(paths_out + (size_t)o0 * PATH_INTS);
        const i32x4* ql = reinterpret_cast<const i32x4*>(&rec_lds[w][0]);
        for (int c = lane; c < n_live * 3; c += 64) qo[c] = ql[c];
        float* __restrict__ po = probs_out + (size_t)o0 * 3;
        for (int c = lane; c < n_live * 3; c += 64) po[c] = prob_lds[w][c];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}


// ---- A13, fused: masked softmax + top-k + expand of a hop in ONE launch (act_dim <= 256, k <= 64) ------------------------
// One wave per beam row: the row's logits are ranked as in policy_topk_sort_kernel<true> (same keys, same select, same
// order), after which lane j holds the row's j-th child -- its action is gathered and the child record written straight to
// its slot offsets[row] + j.  Against topk + expand as two launches: the [rows, k] index / value arrays are never written
// or read back (0.5 GB each way on the last hop of a 64/64/64 beam), the parent record is read once per row through the
// scalar unit instead of once per child, and a launch goes.  Child order and every value are those of the two-launch form.
// LEAF = the last hop's compact output: a child is {parent row, relation, node, bits of its hop value} = 16 bytes instead of
// the 60-byte record that repeats the parent's prefix 64 times; hw_pgpr_materialize_leaves rebuilds records on demand and
// the collector reads the compact form directly.
template <bool LEAF, bool A16>
__global__ __launch_bounds__(256) void pgpr_topk_expand_kernel(
    hw_pgpr_graph g, const int32_t* __restrict__ paths, const float* __restrict__ probs, const int32_t* __restrict__ acts,
    int act_dim, const int32_t* __restrict__ n_acts, const int32_t* __restrict__ n_rows_ptr,
    const float* __restrict__ logits, int ld, int k, const int32_t* __restrict__ offsets, int32_t* __restrict__ paths_out,
    float* __restrict__ probs_out, int4* __restrict__ leaf_out, int64_t cap) {
    __shared__ uint32_t surv_all[4][64];
    __shared__ __attribute__((aligned(16))) int rec_lds[LEAF ? 1 : 4][LEAF ? 4 : 64 * PATH_INTS];
    __shared__ float prob_lds[LEAF ? 1 : 4][LEAF ? 4 : 64 * 3];
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const int n_rows = n_rows_ptr[0];
    const int lane = threadIdx.x & 63, w = wave_in_block();
    const int stride = gridDim.x * 4;
    const int first = blockIdx.x * 4 + w;
    if (first >= n_rows) return;  // no block-wide barrier below: a wave may leave alone
    // A row is a chain of dependent round trips -- counts and parent record (scalar unit) -> logits, offset list, CSR base ->
    // ranking -> edge -> leaf -- and the launch spent two thirds of its wave cycles in s_waitcnt with ten waves per SIMD
    // (profiles/r5/pmc_pgpr_r5b.txt: SQ_WAIT_ANY 8.2e8 of 1.25e9).  So everything that does not depend on a row's ranking is
    // requested AHEAD of the row, branch-free (clamped indices: the last row is simply requested again):
    //   two rows ahead  the scalar reads: action count, the 48-byte parent record, the output offset;
    //   one row ahead   the vector reads that need those: the raw logits (their layout follows the count: rank_slots), and with
    //                   A16 the row's whole offset list (<= 256 x 2 bytes: four entries per lane; the chosen entry is picked
    //                   out of the lanes' registers afterwards) and the current node's CSR base.
    // Within a row the pre-ranking requests had already been hoisted above the ranking (0.91 -> 0.85 ms for the last hop of
    // the 64/64/64 bench in round 4).
    static_assert(PATH_INTS == 12, "three 16-byte pieces");
    struct RowScalars {
        int row, na;
        i32x4 q0, q1, q2;  // the parent record
        int o0;            // first output slot
    };
    auto read_scalars = [&](int row) {  // row is wave-uniform: scalar loads
        RowScalars r;
        r.row = row;
        r.na = n_acts[row];
        const i32x4* pr = reinterpret_cast<const i32x4*>(paths + (size_t)row * PATH_INTS);
        r.q0 = pr[0], r.q1 = pr[1], r.q2 = pr[2];
        r.o0 = offsets[row];
        return r;
    };
    auto request = [&](const RowScalars& r, float (&x)[4], uint2& a16, int64_t& lo_pref) {
        const int npl = rank_slots(r.na);  // wave-uniform: 2 or 4
        const float* __restrict__ lr = logits + (size_t)r.row * ld;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = lane * npl + (i < npl ? i : npl - 1);  // slots 2, 3 of a two-slot row repeat slot 1
            x[i] = lr[j < act_dim ? j : act_dim - 1];
        }
        if constexpr (A16) {
            const int ld16 = act16_ld(act_dim);
            const uint2* arow = reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(acts) + (size_t)r.row * ld16);
            a16 = arow[lane * 4 < ld16 ? lane : 0];  // clamped: lanes past the row re-read its first piece
            const int plen = r.q0.y;
            const int pt = plen == 1 ? T_USER : (plen == 2 ? r.q0.w : (plen == 3 ? r.q1.z : r.q2.y));
            const int pn = plen == 1 ? r.q0.x : (plen == 2 ? r.q1.x : (plen == 3 ? r.q1.w : r.q2.z));
            lo_pref = g.ptr[pt == T_USER ? pn : (int64_t)g.n_users + pn];
        }
    };
    auto clamp_next = [&](int row) { return row + stride < n_rows ? row + stride : row; };
    RowScalars s1 = read_scalars(__builtin_amdgcn_readfirstlane(first));
    RowScalars s2 = read_scalars(clamp_next(s1.row));
    float xn[4];
    uint2 a16n = make_uint2(0u, 0u);
    int64_t lon = 0;
    request(s1, xn, a16n, lon);
    for (bool more = true; more;) {
        const RowScalars cur = s1;
        const int row = cur.row, na = cur.na;
        float xr[4] = {xn[0], xn[1], xn[2], xn[3]};
        const uint2 a16 = a16n;
        const int64_t lo_pref = lon;
        more = row + stride < n_rows;  // wave-uniform
        s1 = s2;
        s2 = read_scalars(clamp_next(s1.row));
        request(s1, xn, a16n, lon);
        __builtin_amdgcn_sched_barrier(0);  // the requests stay above the ranking
        bool row_has_two;
        uint32_t one;
        if (rank_slots(na) == 2) {  // wave-uniform
            uint32_t key[2];
            float x2[2] = {xr[0], xr[1]};
            row_softmax_keys_of<2>(x2, act_dim, na, lane, nullptr, key, row_has_two);
            one = select_sort_topk<2>(key, k, lane, surv_all[w]);
        } else {
            uint32_t key[4];
            row_softmax_keys_of<4>(xr, act_dim, na, lane, nullptr, key, row_has_two);
            one = select_sort_topk<4>(key, k, lane, surv_all[w]);
        }
        const int64_t o0 = cur.o0;
        const int64_t room = cap - o0;  // children past the output capacity are dropped, as in pgpr_expand_kernel
        const int n_child = min(k, na);
        const int n_live = (int)(room < (int64_t)n_child ? (room < 0 ? 0 : room) : (int64_t)n_child);  // wave-uniform
        if (n_live <= 0) continue;
        const bool live = lane < n_live;
        int idx;
        float v;
        decode_topk_key(one, row_has_two, idx, v);
        // the parent record: read once per row through the scalar unit, two rows ago
        const i32x4 q0 = cur.q0, q1 = cur.q1, q2 = cur.q2;
        const int len = q0.y;
        const int cur_t = len == 1 ? T_USER : (len == 2 ? q0.w : (len == 3 ? q1.z : q2.y));
        int rel, node;
        if constexpr (A16) {  // the action is an edge of the current node: (relation, id) from the CSR
            const int cur_n = len == 1 ? q0.x : (len == 2 ? q1.x : (len == 3 ? q1.w : q2.z));
            const int64_t lo = lo_pref;
            const int ia = live ? idx : 0;  // entry ia sits in lane ia / 4, 16-bit piece ia % 4
            const unsigned w0 = (unsigned)__shfl((int)a16.x, ia >> 2, 64), w1 = (unsigned)__shfl((int)a16.y, ia >> 2, 64);
            const unsigned word = (ia & 2) ? w1 : w0;
            const unsigned off = (ia & 1) ? (word >> 16) : (word & 0xFFFFu);
            const bool self = off == ACT16_SELF;
            const int64_t e = self ? 0 : lo + (int64_t)off;  // edge 0 always exists; a node without edges has no slice to read
            // (the CSR's relation / dst arrays interleaved so that this is one 8-byte gather instead of two 4-byte ones:
            // measured on one card, 0.91 = 0.91 ms -- not kept)
            const int er = g.rel[e], ed = g.dst[e];
            rel = self ? SELF_LOOP : er, node = self ? cur_n : ed;
        } else {
            const int2 act = *reinterpret_cast<const int2*>(&acts[((size_t)row * act_dim + (live ? idx : 0)) * 2]);
            rel = act.x, node = act.y;
        }
        if constexpr (LEAF) {
            if (live) leaf_out[o0 + lane] = make_int4(row, rel, node, (int)__float_as_uint(v));
        } else {
            const int nt = rel == SELF_LOOP ? cur_t : next_type(g, cur_t, rel);
            const float* pp = probs + (size_t)row * 3;
            const float p0 = pp[0], p1 = pp[1], p2 = pp[2];
            if (live) {
                int* slot = &rec_lds[w][lane * PATH_INTS];
                i32x4* q = reinterpret_cast<i32x4*>(slot);
                q[0] = q0, q[1] = q1, q[2] = q2;
                slot[1] = len + 1;
                const int at = 2 + 3 * (len - 1);  // wave-uniform position of the new hop inside the record
                slot[at] = rel, slot[at + 1] = nt, slot[at + 2] = node;
                prob_lds[w][lane * 3 + 0] = len == 1 ? v : p0;
                prob_lds[w][lane * 3 + 1] = len == 2 ? v : p1;
                prob_lds[w][lane * 3 + 2] = len == 3 ? v : p2;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // the row's children are the consecutive slots o0 .. o0 + n_live - 1: one contiguous run of 16-byte pieces
            i32x4* __restrict__ qo = reinterpret_cast<i32x4*>(paths_out + (size_t)o0 * PATH_INTS);
            const i32x4* ql = reinterpret_cast<const i32x4*>(&rec_lds[w][0]);
            for (int c = lane; c < n_live * 3; c += 64) qo[c] = ql[c];
            float* __restrict__ po = probs_out + (size_t)o0 * 3;
            for (int c = lane; c < n_live * 3; c += 64) po[c] = prob_lds[w][c];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// Compact leaves -> 12-int records + probability triples (the layout every other kernel and the host decoders read).
// `sel` (optional): output row t = leaf sel[t] (negative entries give an all-zero record); otherwise every leaf in order.
__global__ __launch_bounds__(256) void pgpr_materialize_leaves_kernel(hw_pgpr_graph g, const int32_t* __restrict__ parents,
                                                                      const float* __restrict__ parent_probs,
                                                                      const int4* __restrict__ leaf,
                                                                      const int32_t* __restrict__ n_leaf_ptr,
                                                                      const int32_t* __restrict__ sel, int64_t n_sel,
                                                                      int64_t max_out, int32_t* __restrict__ paths_out,
                                                                      float* __restrict__ probs_out) {
    const int64_t n_leaf = n_leaf_ptr[0];
    const int64_t n = sel ? n_sel : (n_leaf < max_out ? n_leaf : max_out);  // never past the output buffer
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) {
        const int64_t i = sel ? (int64_t)sel[t] : t;
        int32_t* o = paths_out + (size_t)t * PATH_INTS;
        float* q = probs_out + (size_t)t * 3;
        if (i < 0 || i >= n_leaf) {
#pragma unroll
            for (int c = 0; c < PATH_INTS; ++c) o[c] = 0;
            q[0] = q[1] = q[2] = 0.f;
            continue;
        }
        const int4 lf = leaf[i];
        PathRec rec;
        rec.load(parents, lf.x);
        const int len = rec.len;
        const int cur_t = len == 1 ? T_USER : rec.type(len - 1);
        const int rel = lf.y, node = lf.z;
        const int nt = rel == SELF_LOOP ? cur_t : next_type(g, cur_t, rel);
        int f[PATH_INTS] = {rec.uid, len + 1, rec.hr[0], rec.ht[0], rec.hn[0], rec.hr[1], rec.ht[1], rec.hn[1],
                            rec.hr[2], rec.ht[2], rec.hn[2], rec.batch};
#pragma unroll
        for (int j = 1; j <= 3; ++j)
            if (j == len) f[2 + 3 * (j - 1)] = rel, f[3 + 3 * (j - 1)] = nt, f[4 + 3 * (j - 1)] = node;
#pragma unroll
        for (int c = 0; c < PATH_INTS; ++c) o[c] = f[c];
        const float* pp = parent_probs + (size_t)lf.x * 3;
        const float v = __uint_as_float((uint32_t)lf.w);
        q[0] = len == 1 ? v : pp[0];
        q[1] = len == 2 ? v : pp[1];
        q[2] = len == 3 ? v : pp[2];
    }
}

}  // namespace hw

using namespace hw;

static int check_graph(const hw_pgpr_graph* g) {
    if (!g || !g->ptr || !g->rel || !g->dst) return HW_ERR_INVALID_ARG;
    if (g->n_users < 1 || g->n_entities < 1 || g->n_items < 1 || g->n_items > g->n_entities) return HW_ERR_INVALID_ARG;
    return HW_OK;
}

extern "C" int hw_pgpr_init_paths(const int64_t* users, int64_t n_users_in_batch, int32_t* paths, float* probs,
                                  int32_t* n_rows, void* stream) {
    if (!users || !paths || !n_rows || n_users_in_batch < 0 || n_users_in_batch > INT32_MAX) return HW_ERR_INVALID_ARG;
    HW_CLEAR_ERR();
    hipStream_t st = (hipStream_t)stream;
    const int n = (int)n_users_in_batch;
    if (probs && n > 0 && hw::fill_bytes_async(probs, 0, sizeof(float) * 3 * (size_t)n, st) != hipSuccess) return HW_ERR_LAUNCH;
    hipLaunchKernelGGL(pgpr_init_paths_kernel, dim3(n > 0 ? (n + 255) / 256 : 1), dim3(256), 0, st, users, n, paths,
                       n_rows);
    return HW_LAUNCH_RC();
}

static int launch_dot64(const double* src, int64_t n_src, const double* T, int64_t n_rows, int dim, double* out,
                        hipStream_t st) {
    if (n_src >= 64) {  // many sources: the register-tiled product (same sums bit for bit)
        dim3 grid((unsigned)((n_rows + DRT_N - 1) / DRT_N), (unsigned)((n_src + DRT_S - 1) / DRT_S));
        hipLaunchKernelGGL(batched_dot64_rt_kernel, grid, dim3(256), 0, st, src, (int)n_src, T, (int)n_rows, dim, out);
        return HW_OK;
    }
    dim3 grid((unsigned)((n_rows + 255) / 256), (unsigned)((n_src + DOT_SB - 1) / DOT_SB));
    hipLaunchKernelGGL(batched_dot64_kernel, grid, dim3(256), 0, st, src, (int)n_src, T, (int)n_rows, dim, out);
    return HW_OK;
}

extern "C" int hw_pgpr_rel_entity_scores(const double* relation_emb64, const double* entity_emb64, int64_t n_relations,
                                         int64_t n_entities, int32_t dim, double* out, void* stream) {
    if (!relation_emb64 || !entity_emb64 || !out || n_relations < 1 || n_entities < 1 || dim < 1 ||
        dim > HW_PGPR_MAX_DIM)
        return HW_ERR_INVALID_ARG;
    HW_CLEAR_ERR();
    launch_dot64(relation_emb64, n_relations, entity_emb64, n_entities, dim, out, (hipStream_t)stream);
    return HW_LAUNCH_RC();
}

static int user_scores_impl(const hw_pgpr_graph* graph, const double* user_emb64, const double* entity_emb64,
                            const double* relation_emb64, const double* user_vec64, int32_t dim, const int64_t* users,
                            int64_t n_batch, double* src_workspace, double* score_user, double* score_entity,
                            void* stream) {
    int rc = check_graph(graph);
    if (rc) return rc;
    if (!user_emb64 || !entity_emb64 || !relation_emb64 || !users || !src_workspace || !score_user || !score_entity)
        return HW_ERR_INVALID_ARG;
    if (dim < 1 || dim > HW_PGPR_MAX_DIM || n_batch < 0 || n_batch > INT32_MAX / HW_PGPR_MAX_DIM) return HW_ERR_INVALID_ARG;
    if (n_batch == 0) return HW_OK;
    HW_CLEAR_ERR();
    hipStream_t st = (hipStream_t)stream;
    double* src0 = src_workspace;
    double* src1 = src_workspace + (size_t)n_batch * dim;
    const int total = (int)(n_batch * dim);
    hipLaunchKernelGGL(pgpr_user_src_kernel, dim3((total + 255) / 256), dim3(256), 0, st, user_emb64, relation_emb64, user_vec64, dim,
                       graph->ui_relation, users, (int)n_batch, graph->n_users, src0, src1);
    launch_dot64(src0, n_batch, user_emb64, graph->n_users, dim, score_user, st);
    launch_dot64(src1, n_batch, entity_emb64, graph->n_entities, dim, score_entity, st);
    return HW_LAUNCH_RC();
}

extern "C" int hw_pgpr_user_scores(const hw_pgpr_graph* graph, const double* user_emb64, const double* entity_emb64,
                                   const double* relation_emb64, int32_t dim, const int64_t* users, int64_t n_batch,
                                   double* src_workspace, double* score_user, double* score_entity, void* stream) {
    return user_scores_impl(graph, user_emb64, entity_emb64, relation_emb64, nullptr, dim, users, n_batch, src_workspace,
                            score_user, score_entity, stream);
}

extern "C" int hw_pgpr_user_scores_tp(const hw_pgpr_graph* graph, const double* user_emb64, const double* entity_emb64,
                                      const double* relation_emb64, const double* user_vec64, int32_t dim,
                                      const int64_t* users, int64_t n_batch, double* src_workspace, double* score_user,
                                      double* score_entity, void* stream) {
    if (!user_vec64) return HW_ERR_INVALID_ARG;
    return user_scores_impl(graph, user_emb64, entity_emb64, relation_emb64, user_vec64, dim, users, n_batch,
                            src_workspace, score_user, score_entity, stream);
}

extern "C" int64_t hw_pgpr_actions_workspace_ints(int64_t max_rows) { return max_rows < 0 ? 0 : 4 + 3 * max_rows; }

extern "C" int hw_pgpr_actions(const hw_pgpr_graph* graph, const double* score_user, const double* score_entity,
                               const double* score_rel_entity, const int32_t* paths, const int32_t* n_rows,
                               int64_t max_rows, int32_t max_acts, int32_t fix_scores_sorting_bug, int32_t* acts,
                               int32_t* n_acts, uint64_t* scratch, int64_t scratch_blocks, int64_t scratch_stride,
                               int32_t* work, int64_t work_ints, void* stream) {
    int rc = check_graph(graph);
    if (rc) return rc;
    if (!score_user || !score_entity || !score_rel_entity || !paths || !n_rows || !acts || !n_acts)
        return HW_ERR_INVALID_ARG;
    if (max_acts < 1 || max_acts > 1023 || max_rows < 0 || max_rows > INT32_MAX) return HW_ERR_INVALID_ARG;
    if (max_rows == 0) return HW_OK;
    if (work && work_ints < 4 + 3 * max_rows) return HW_ERR_WORKSPACE;
    // bit 2: action lists as uint16 edge offsets (ActOut<true>); only the wave / block kernels write that layout
    const bool a16 = (fix_scores_sorting_bug & 4) != 0;
    if (a16 && ((fix_scores_sorting_bug & 1) || graph->max_degree > TRIM_BLOCK_MAX_DEG)) return HW_ERR_INVALID_ARG;
    // every block that may meet a row with more than ACT_LDS_KEYS neighbours needs a private scratch slice
    int64_t grid = max_rows < 2048 ? max_rows : 2048;
    if (graph->max_degree > ACT_LDS_KEYS) {
        if (!scratch || scratch_blocks < 1 || scratch_stride < graph->max_degree) return HW_ERR_WORKSPACE;
        if (scratch_blocks < grid) grid = scratch_blocks;
    } else {
        scratch = nullptr;
        scratch_stride = 0;
    }
    HW_CLEAR_ERR();
    hipStream_t st = (hipStream_t)stream;
    // work: three queues of rows that need the trim (by degree class) + their lengths; without it the trim kernels scan
    // all rows for n_acts < 0
    const int32_t* lists[3] = {nullptr, nullptr, nullptr};
    const int32_t* lens[3] = {nullptr, nullptr, nullptr};
    if (work) {
        if (hw::fill_bytes_async(work, 0, 4 * sizeof(int32_t), st) != hipSuccess) return HW_ERR_LAUNCH;
        for (int c = 0; c < 3; ++c) lists[c] = work + 4 + (int64_t)c * max_rows, lens[c] = work + c;
    }
    {
        int64_t g1 = (max_rows + 3) / 4;
        if (g1 > 8192) g1 = 8192;
        if (a16)
            hipLaunchKernelGGL(pgpr_actions_small_kernel<true>, dim3((unsigned)g1), dim3(256), 0, st, *graph, paths, n_rows,
                               max_acts, fix_scores_sorting_bug, acts, n_acts, work, max_rows);
        else
            hipLaunchKernelGGL(pgpr_actions_small_kernel<false>, dim3((unsigned)g1), dim3(256), 0, st, *graph, paths, n_rows,
                               max_acts, fix_scores_sorting_bug, acts, n_acts, work, max_rows);
    }
    const bool both_trims = !(fix_scores_sorting_bug & 1) && graph->max_degree > TRIM_WAVE_MAX_DEG && max_rows <= 65536;
    if (both_trims) {  // small hops and training batches: the two trims side by side in one launch
        const int g2 = (int)((max_rows + 3) / 4 > 4096 ? 4096 : (max_rows + 3) / 4);
        const int g3 = (int)(max_rows < 2048 ? max_rows : 2048);
        if (a16)
            hipLaunchKernelGGL(pgpr_actions_trim_kernel<true>, dim3((unsigned)(g2 + g3)), dim3(256), 0, st, *graph, score_user,
                               score_entity, score_rel_entity, paths, n_rows, max_acts, fix_scores_sorting_bug, acts, n_acts,
                               lists[0], lens[0], lists[1], lens[1], g2, g3);
        else
            hipLaunchKernelGGL(pgpr_actions_trim_kernel<false>, dim3((unsigned)(g2 + g3)), dim3(256), 0, st, *graph, score_user,
                               score_entity, score_rel_entity, paths, n_rows, max_acts, fix_scores_sorting_bug, acts, n_acts,
                               lists[0], lens[0], lists[1], lens[1], g2, g3);
    } else if (!(fix_scores_sorting_bug & 1)) {
        int64_t g2 = (max_rows + 3) / 4;
        if (g2 > 4096) g2 = 4096;
        if (a16)
            hipLaunchKernelGGL(pgpr_actions_trim_wave_kernel<true>, dim3((unsigned)g2), dim3(256), 0, st, *graph, score_user,
                               score_entity, score_rel_entity, paths, n_rows, max_acts, fix_scores_sorting_bug, acts, n_acts,
                               lists[0], lens[0]);
        else
            hipLaunchKernelGGL(pgpr_actions_trim_wave_kernel<false>, dim3((unsigned)g2), dim3(256), 0, st, *graph, score_user,
                               score_entity, score_rel_entity, paths, n_rows, max_acts, fix_scores_sorting_bug, acts, n_acts,
                               lists[0], lens[0]);
        if (graph->max_degree > TRIM_WAVE_MAX_DEG) {
            const int64_t g3 = max_rows < 2048 ? max_rows : 2048;
            if (a16)
                hipLaunchKernelGGL(pgpr_actions_trim_block_kernel<true>, dim3((unsigned)g3), dim3(256), 0, st, *graph,
                                   score_user, score_entity, score_rel_entity, paths, n_rows, max_acts,
                                   fix_scores_sorting_bug, acts, n_acts, lists[1], lens[1]);
            else
                hipLaunchKernelGGL(pgpr_actions_trim_block_kernel<false>, dim3((unsigned)g3), dim3(256), 0, st, *graph,
                                   score_user, score_entity, score_rel_entity, paths, n_rows, max_acts,
                                   fix_scores_sorting_bug, acts, n_acts, lists[1], lens[1]);
        }
    }
    if ((fix_scores_sorting_bug & 1) || graph->max_degree > TRIM_BLOCK_MAX_DEG)
        hipLaunchKernelGGL(pgpr_actions_kernel, dim3((unsigned)grid), dim3(256), 0, st, *graph, score_user, score_entity,
                           score_rel_entity, paths, n_rows, max_acts, fix_scores_sorting_bug, acts, n_acts, scratch,
                           scratch_stride, lists[2], lens[2]);
    return HW_LAUNCH_RC();
}

extern "C" int hw_pgpr_state(const hw_pgpr_graph* graph, const float* user_emb32, const float* entity_emb32,
                             const float* relation_emb32, int32_t dim, int32_t state_history, const int32_t* paths,
                             const int32_t* n_rows, int64_t max_rows, float* state, void* stream) {
    int rc = check_graph(graph);
    if (rc) return rc;
    if (!user_emb32 || !entity_emb32 || !relation_emb32 || !paths || !n_rows || !state) return HW_ERR_INVALID_ARG;
    if (dim < 1 || state_history < 0 || state_history > 2 || max_rows < 0) return HW_ERR_INVALID_ARG;
    if (max_rows == 0) return HW_OK;
    const int n_parts = 2 + 2 * state_history;
    const int64_t total = max_rows * n_parts * dim;
    int64_t grid = (total + 255) / 256;
    if (grid > 8192) grid = 8192;
    HW_CLEAR_ERR();
    hipLaunchKernelGGL(pgpr_state_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, *graph, user_emb32,
                       entity_emb32, relation_emb32, dim, n_parts, paths, n_rows, state);
    return HW_LAUNCH_RC();
}

extern "C" int hw_pgpr_layer1_lookup(const float* const* slot_tables, const float* bias, int32_t hidden,
                                     int32_t state_history, const int32_t* paths, const int32_t* n_rows, int64_t max_rows,
                                     const float* drop_mask, int32_t drop_mode, uint64_t seed, uint64_t offset, float* out,
                                     int64_t ld, void* stream) {
    if (!slot_tables || !bias || !paths || !n_rows || !out || hidden < 4 || (hidden & 3) || ld < hidden)
        return HW_ERR_INVALID_ARG;
    if (state_history < 0 || state_history > 2 || drop_mode < 0 || drop_mode > 2 || (drop_mode == 1 && !drop_mask))
        return HW_ERR_INVALID_ARG;
    const int need = state_history == 0 ? 3 : (state_history == 1 ? 6 : 9);
    for (int i = 0; i < need; ++i)
        if (!slot_tables[i]) return HW_ERR_INVALID_ARG;
    if (((uintptr_t)bias | (uintptr_t)out) & 15 || (ld & 3)) return HW_ERR_INVALID_ARG;
    if (max_rows == 0) return HW_OK;
    L1Tables t;
    t.user0 = slot_tables[0], t.cur_u = slot_tables[1], t.cur_e = slot_tables[2];
    t.last_u = need > 3 ? slot_tables[3] : nullptr, t.last_e = need > 3 ? slot_tables[4] : nullptr;
    t.last_r = need > 3 ? slot_tables[5] : nullptr;
    t.old_u = need > 6 ? slot_tables[6] : nullptr, t.old_e = need > 6 ? slot_tables[7] : nullptr;
    t.old_r = need > 6 ? slot_tables[8] : nullptr;
    const int run = max_rows >= 65536 ? L1_RUN : 1;  // rows per wave visit: sibling reuse needs runs, small hops need waves
    int64_t grid = (max_rows + 4 * run - 1) / (4 * run);
    if (grid > 16384) grid = 16384;
    HW_CLEAR_ERR();
    hipLaunchKernelGGL(pgpr_layer1_lookup_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, t, bias,
                       hidden, 2 + 2 * state_history, paths, n_rows, drop_mask, drop_mode, seed, offset, out, ld, run);
    return HW_LAUNCH_RC();
}

extern "C" int hw_pgpr_policy_topk(const float* logits, int64_t ld, int32_t act_dim, const int32_t* n_acts,
                                   const int32_t* n_rows, int64_t max_rows, int32_t k, int32_t* out_idx, float* out_val,
                                   float* probs_out, void* stream) {
    if (!logits || !n_acts || !n_rows || !out_idx || !out_val) return HW_ERR_INVALID_ARG;
    if (act_dim < 1 || act_dim > 1024 || ld < act_dim || k < 1 || k > act_dim || max_rows < 0) return HW_ERR_INVALID_ARG;
    if (max_rows == 0) return HW_OK;
    int64_t grid = (max_rows + 3) / 4;
    if (grid > 16384) grid = 16384;
    HW_CLEAR_ERR();
    hipStream_t st = (hipStream_t)stream;
    if (act_dim <= 256 && k > 8 && k <= 64)
        hipLaunchKernelGGL(policy_topk_sort_kernel<true>, dim3((unsigned)grid), dim3(256), 0, st, logits, (int)ld, act_dim,
                           n_acts, n_rows, k, out_idx, out_val, probs_out);
    else if (act_dim <= 256 && k > 8)
        hipLaunchKernelGGL(policy_topk_sort_kernel<false>, dim3((unsigned)grid), dim3(256), 0, st, logits, (int)ld, act_dim,
                           n_acts, n_rows, k, out_idx, out_val, probs_out);
    else if (act_dim <= 256)
        hipLaunchKernelGGL((policy_topk_kernel<4>), dim3((unsigned)grid), dim3(256), 0, st, logits, (int)ld, act_dim,
                           n_acts, n_rows, k, out_idx, out_val, probs_out);
    else
        hipLaunchKernelGGL((policy_topk_kernel<16>), dim3((unsigned)grid), dim3(256), 0, st, logits, (int)ld, act_dim,
                           n_acts, n_rows, k, out_idx, out_val, probs_out);
    return HW_LAUNCH_RC();
}

extern "C" int hw_pgpr_expand(const hw_pgpr_graph* graph, const int32_t* paths, const float* probs,
                              const int32_t* acts, int32_t act_dim, const int32_t* n_acts, const int32_t* n_rows,
                              int64_t max_rows, const int32_t* topk_idx, const float* topk_val, int32_t k,
                              int32_t* offsets, int32_t* block_sums_ws, int32_t* paths_out, float* probs_out,
                              int32_t* n_rows_out, int64_t out_capacity, void* stream) {
    int rc = check_graph(graph);
    if (rc) return rc;
    if (!paths || !probs || !acts || !n_acts || !n_rows || !topk_idx || !topk_val || !offsets || !paths_out ||
        !probs_out || !n_rows_out)
        return HW_ERR_INVALID_ARG;
    if (act_dim < 1 || k < 1 || max_rows < 0 || out_capacity < 0 || out_capacity > INT32_MAX) return HW_ERR_INVALID_ARG;
    if (max_rows * (int64_t)k > INT32_MAX) return HW_ERR_INVALID_ARG;  // child offsets are 32-bit
    HW_CLEAR_ERR();
    hipStream_t st = (hipStream_t)stream;
    {
        const int nb = (int)((max_rows + OFF_ROWS_PER_BLOCK - 1) / OFF_ROWS_PER_BLOCK);
        if (nb > 65536 || !block_sums_ws) return HW_ERR_WORKSPACE;
        const int nb1 = nb < 1 ? 1 : nb;
        hipLaunchKernelGGL(pgpr_offsets_partial_kernel, dim3(nb1), dim3(256), 0, st, n_acts, n_rows, k, block_sums_ws);
        hipLaunchKernelGGL(pgpr_offsets_scan_kernel, dim3(1), dim3(1024), 0, st, block_sums_ws, nb1, n_rows_out,
                           out_capacity);
        hipLaunchKernelGGL(pgpr_offsets_final_kernel, dim3(nb1), dim3(256), 0, st, n_acts, n_rows, k, block_sums_ws,
                           offsets);
    }
    if (max_rows > 0) {
        int64_t grid = (max_rows * k + 255) / 256;
        if (grid > 16384) grid = 16384;
        hipLaunchKernelGGL(pgpr_expand_kernel, dim3((unsigned)grid), dim3(256), 0, st, *graph, paths, probs, acts,
                           act_dim, n_acts, n_rows, topk_idx, topk_val, k, offsets, paths_out, probs_out, out_capacity);
    }
    return HW_LAUNCH_RC();
}


static int launch_child_offsets(const int32_t* n_acts, const int32_t* n_rows, int64_t max_rows, int32_t k, int32_t* offsets,
                                int32_t* block_sums_ws, int32_t* n_rows_out, int64_t out_capacity, hipStream_t st) {
    const int nb = (int)((max_rows + OFF_ROWS_PER_BLOCK - 1) / OFF_ROWS_PER_BLOCK);
    if (nb > 65536 || !block_sums_ws) return HW_ERR_WORKSPACE;
    const int nb1 = nb < 1 ? 1 : nb;
    hipLaunchKernelGGL(pgpr_offsets_partial_kernel, dim3(nb1), dim3(256), 0, st, n_acts, n_rows, k, block_sums_ws);
    hipLaunchKernelGGL(pgpr_offsets_scan_kernel, dim3(1), dim3(1024), 0, st, block_sums_ws, nb1, n_rows_out, out_capacity);
    hipLaunchKernelGGL(pgpr_offsets_final_kernel, dim3(nb1), dim3(256), 0, st, n_acts, n_rows, k, block_sums_ws, offsets);
    return HW_OK;
}

extern "C" int hw_pgpr_topk_expand(const hw_pgpr_graph* graph, const int32_t* paths, const float* probs, const int32_t* acts,
                                   int32_t act_dim, const int32_t* n_acts, const int32_t* n_rows, int64_t max_rows,
                                   const float* logits, int64_t ld, int32_t k, int32_t* offsets, int32_t* block_sums_ws,
                                   int32_t* paths_out, float* probs_out, int32_t* leaf_out, int32_t* n_rows_out,
                                   int64_t out_capacity, int32_t acts_u16, void* stream) {
    int rc = check_graph(graph);
    if (rc) return rc;
    if (!paths || !probs || !acts || !n_acts || !n_rows || !logits || !offsets || !n_rows_out) return HW_ERR_INVALID_ARG;
    if (!leaf_out && (!paths_out || !probs_out)) return HW_ERR_INVALID_ARG;
    if (act_dim < 1 || act_dim > 256 || ld < act_dim || k < 1 || k > 64 || k > act_dim || max_rows < 0 || out_capacity < 0 ||
        out_capacity > INT32_MAX)
        return HW_ERR_INVALID_ARG;
    if (max_rows * (int64_t)k > INT32_MAX) return HW_ERR_INVALID_ARG;  // child offsets are 32-bit
    if ((((uintptr_t)paths | (uintptr_t)paths_out | (uintptr_t)leaf_out) & 15) != 0) return HW_ERR_INVALID_ARG;
    HW_CLEAR_ERR();
    hipStream_t st = (hipStream_t)stream;
    rc = launch_child_offsets(n_acts, n_rows, max_rows, k, offsets, block_sums_ws, n_rows_out, out_capacity, st);
    if (rc) return rc;
    if (max_rows > 0) {
        int64_t grid = (max_rows + 3) / 4;
        if (grid > 16384) grid = 16384;
#define TAIL(LEAF_, A16_)                                                                                                 \
    hipLaunchKernelGGL((pgpr_topk_expand_kernel<LEAF_, A16_>), dim3((unsigned)grid), dim3(256), 0, st, *graph, paths, probs, \
                       acts, act_dim, n_acts, n_rows, logits, (int)ld, k, offsets, paths_out, probs_out,                    \
                       reinterpret_cast<int4*>(leaf_out), out_capacity)
        if (leaf_out && acts_u16) TAIL(true, true);
        else if (leaf_out) TAIL(true, false);
        else if (acts_u16) TAIL(false, true);
        else TAIL(false, false);
#undef TAIL
    }
    return HW_LAUNCH_RC();
}

extern "C" int hw_pgpr_materialize_leaves(const hw_pgpr_graph* graph, const int32_t* parents, const float* parent_probs,
                                          const int32_t* leaf, const int32_t* n_leaf, int64_t max_leaf, const int32_t* sel,
                                          int64_t n_sel, int32_t* paths_out, float* probs_out, void* stream) {
    int rc = check_graph(graph);
    if (rc) return rc;
    if (!parents || !parent_probs || !leaf || !n_leaf || !paths_out || !probs_out || max_leaf < 0 || n_sel < 0 ||
        (((uintptr_t)parents | (uintptr_t)leaf) & 15) != 0)
        return HW_ERR_INVALID_ARG;
    const int64_t n = sel ? n_sel : max_leaf;
    if (n == 0) return HW_OK;
    int64_t grid = (n + 255) / 256;
    if (grid > 16384) grid = 16384;
    HW_CLEAR_ERR();
    hipLaunchKernelGGL(pgpr_materialize_leaves_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, *graph,
                       parents, parent_probs, reinterpret_cast<const int4*>(leaf), n_leaf, sel, n_sel, max_leaf, paths_out,
                       probs_out);
    return HW_LAUNCH_RC();
}
