// KGE scoring kernels for gfx950: grouped (1 positive + K negatives per (h, r)) and full sort.
//
// Both are HBM/L2-gather bound: ~0.5 flop per byte.  The design goal is bytes in flight, not flops:
//   * one embedding row is spread over G lanes (G = 64 for d = 200/256: one 16-byte load per lane per
//     table covers the row; G = 16/32 for short rows so that a wave carries 4/2 rows per instruction);
//   * every sub-wave group keeps U tail rows in flight (U x NT x 16-byte loads per lane) before the
//     first one is consumed;
//   * the (h, r) query is built once per wave task and reused for all its tails (kge_model.h).
// No LDS is used: a tail row is consumed by exactly one group, so staging it would only add a hop
// (cdna_hip_programming.md, 'glds vs register staging', GEMV row).
#pragma once
#include "kge_model.h"

namespace hw {

// ------------------------------------------------------------------------------------------------
// grouped scorer.  wave task = (group b, split s of S); the NS = 64/G sub-groups of the wave and the
// S splits interleave over the group's tails:  j = (k*S + s)*NS + sub,  k = 0, 1, ...
// ------------------------------------------------------------------------------------------------
// SAMPLE = true: columns 1.. of the tail-id matrix are DRAWN here (uniform over [1, value_num) minus the key's used set,
// the Philox counters and the rejection rule of neg_sample_kernel, so the ids equal hw_neg_sample's bit for bit) and
// written back to t_idx for the gradient pass; column 0 (the positive) is read.  The key's used set is an LDS bitmap per
// wave, built once per task.  Saves the sampler launch and the read of the id matrix.
// LOSS = true (with SAMPLE, S == 4 only: the block's four waves are the four splits of ONE group): the group's 1 + K scores
// are also kept in LDS, and wave 0 evaluates the self-adversarial term of hw_loss_self_adversarial for the group (same
// arithmetic, same order) -- the per-group pass of the loss without its launch and without re-reading the score matrix;
// selfadv_finish_kernel then adds the n_groups terms.
// loss = mean of the per-group terms, added in finish_mean_kernel's order (kge_loss.hip): thread t takes t, t + 256, ...,
// then the wave sums, then the four wave totals in order.  Whole block (256 threads).
__device__ __forceinline__ void selfadv_finish_block(const float* __restrict__ group_loss, int n, float inv_n,
                                                     float* __restrict__ loss_out) {
    __shared__ float red[4];
    float v = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) v += group_loss[i];
    v = group_sum<64>(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x / HW_WAVE] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) t += red[i];
        loss_out[0] = t * inv_n;
    }
}

constexpr int LOSS_MAX_TAILS = 1024;
// 8 waves per SIMD (<= 64 VGPRs: the 2048 blocks of a launch resident as one generation, like the plain scorer) is asked of
// the compiler only where it costs no scratch: whole-row shapes (G = 64) with one load unit per table.  ComplEx's fused-loss
// form would spill one register there and the short-row shapes 2-11 (tools/kernel_regs.sh kge_score): they keep the
// compiler's own allocation (67-88 VGPRs, 5-7 waves per SIMD).
template <int MODEL, int G, int NE, bool SAMPLE, bool LOSS>
constexpr int grouped_min_waves() {
    return (SAMPLE && NE <= 4 && G == 64 && !(MODEL == HW_COMPLEX && LOSS)) ? 8 : 1;
}
template <int MODEL, int G, int NE, bool VEC, int U, bool SAMPLE = false, bool LOSS = false>
__global__ __launch_bounds__(256, (grouped_min_waves<MODEL, G, NE, SAMPLE, LOSS>())) void kge_score_grouped_kernel(
    hw_kge_desc d, const int64_t* __restrict__ h_idx, const int64_t* __restrict__ r_idx, int64_t r_fixed,
    int64_t* t_idx, int64_t tsb, int64_t tsj, int64_t n_groups, int ntail, int S,
    float* __restrict__ out, int64_t osb, int64_t osj, int32_t* oob, SamplerArgs sa, SelfAdvArgs la = SelfAdvArgs{}) {
    using SC = Scorer<MODEL, G, NE, VEC>;
    using MT = ModelTraits<MODEL>;
    using R = Row<G, NE, VEC>;
    constexpr int NS = HW_WAVE / G;
    extern __shared__ uint32_t used_bits_all[];
    __shared__ float sh_scores[(SAMPLE && LOSS) ? LOSS_MAX_TAILS : 1];  // the group's scores, for the fused loss

    const int lane = threadIdx.x & (HW_WAVE - 1);
    const int sub = lane / G, gl = lane % G;
    // the wave's number read from lane 0: task, group, split, the group's ids and slice bounds and every row base derived
    // from them are then scalars for the compiler too (scalar loads, SGPRs) instead of 64 equal lanes of vector registers
    const int64_t task = (int64_t)blockIdx.x * (blockDim.x / HW_WAVE) + wave_in_block();
    if (task >= n_groups * S) return;  // wave-uniform
    if constexpr (SAMPLE && LOSS) {
        // chained launches: block 0 (all four waves live: S == 4) first adds the group terms the previous launch on this
        // stream left in prev_group_loss -- the one-block finish of that launch without a launch of its own (4.6 us of a
        // 54 us batch).  Kernel boundaries order the two launches; nothing here crosses workgroups.
        if (blockIdx.x == 0 && la.prev_group_loss)
            selfadv_finish_block(la.prev_group_loss, la.prev_n, 1.0f / (float)la.prev_n, la.prev_loss_out);
    }
    const int64_t b = task / S;
    const int s = (int)(task % S);
    if constexpr (SAMPLE) {
        // The eight waves a SIMD holds belong to eight blocks that start together and all spend their first ~1 000 cycles of
        // issue on Philox rounds (v_mul_hi / v_mul_lo at quarter rate) before any tail row is requested: ~3.5 us in which the
        // chip fetches nothing.  Issue priority by co-resident block (blocks b, b + 256, ... share a CU) lets them finish the
        // draws one after the other instead of together, so the first waves' row loads travel under the others' draws.
        // HW_SCORER_STAGGER=0 (host) launches without it.
        if (sa.stagger) {  // s_setprio takes an immediate
            const unsigned cls = (blockIdx.x >> 8) & 3u;
            if (cls == 0) __builtin_amdgcn_s_setprio(3);
            else if (cls == 1) __builtin_amdgcn_s_setprio(2);
            else if (cls == 2) __builtin_amdgcn_s_setprio(1);
        }
    }
    // EARLY (fused sampler, element-wise queries): the launch is ONE generation of blocks that all start together, so every
    // memory round trip ahead of the tail loop is a stretch in which the chip fetches nothing.  The sampler's chain (key ->
    // slice bounds -> slice ids) and the query's chain (h, r ids -> rows) are independent: their loads are issued side by
    // side and the rows are folded into the query after the slice is staged -- four dependent trips ahead of the first tail
    // row instead of six.
    constexpr bool EARLY = SAMPLE && SC::SPLIT_BUILD;
    // POS_FIRST (fused loss: S == 4, the block is one group): the positive (column 0) is scored ahead of the loop by every
    // wave from the early loads (wave 0 writes it) and the waves share the negatives, j = 1 + 4 q + s.  With 1 + 256 tails
    // dealt j = 4 q + s, wave 0 owned a 65th tail: a second chunk -- one more Philox draw, one more exposed round trip and
    // three dummy rows -- at the end of every block of the launch.
    constexpr bool POS_FIRST = SAMPLE && LOSS && SC::SPLIT_BUILD;
    bool bad = false;
    int h = 0, r = 0;
    SC sc;
    R hh[EARLY ? MT::NH : 1], rr[EARLY ? MT::NR : 1], tpos[POS_FIRST ? MT::NT : 1];
    uint32_t* bits = nullptr;
    bool bad_key = false;
    int64_t ulo = 0, uhi = 0;  // the key's slice of the used-set CSR
    if constexpr (SAMPLE) {
        const int64_t key = sa.keys[b];
        int64_t hid = 0, rid = 0, pid = 0;
        if constexpr (EARLY) hid = h_idx[b], rid = r_idx ? r_idx[b] : r_fixed;
        if constexpr (POS_FIRST) pid = t_idx[b * tsb];
        bad_key = key < 0 || key >= sa.n_key_space;
        // branch-free (a load under a divergent branch is waited for on the spot, and with it everything issued before it)
        const int64_t kk = bad_key ? 0 : key;
        ulo = sa.used_ptr[kk], uhi = sa.used_ptr[kk + 1];
        if constexpr (EARLY) {
            h = clamp_row_flag(hid, d.n_head, bad);
            r = clamp_row_flag(rid, d.n_rel, bad);
            sc.issue(d, h, r, gl, hh, rr);
        }
        if constexpr (POS_FIRST) {
            const int tp = clamp_row_flag(pid, d.n_tail, bad);
#pragma unroll
            for (int kx = 0; kx < MT::NT; ++kx) tpos[kx].load_raw(d.tail[kx] + (size_t)tp * d.dim, d.dim, gl);
        }
        if (bad_key) ulo = uhi = 0;
        // sa.words == 0: a value space beyond the LDS bitmap (> 65 536 ids: Yelp2018's entities, HBM-sized tables); the
        // draws then test membership by binary search in the key's sorted slice, as hw_neg_sample's generic kernel does
        if (sa.words > 0 && S == 4) {
            // the four waves of the block are the four splits of ONE group (task = 4 * block + wave): one bitmap of the
            // key's used set for all of them, built by the whole block (the launcher only sends full blocks here)
            bits = used_bits_all;
            for (int w = threadIdx.x; w < sa.words; w += 256) bits[w] = 0u;
            __syncthreads();
            for (int64_t q = ulo + threadIdx.x; q < uhi; q += 256) {
                const int64_t u = sa.used_idx[q];
                if (u >= 0 && u < sa.value_num) atomicOr(&bits[u >> 5], 1u << (u & 31));
            }
            __syncthreads();
        } else if (sa.words > 0) {
            bits = used_bits_all + (threadIdx.x / HW_WAVE) * sa.words;
            for (int w = lane; w < sa.words; w += HW_WAVE) bits[w] = 0u;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int64_t q = ulo + lane; q < uhi; q += HW_WAVE) {
                const int64_t u = sa.used_idx[q];
                if (u >= 0 && u < sa.value_num) atomicOr(&bits[u >> 5], 1u << (u & 31));
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    // beyond the bitmap: the block's key (S == 4: one group per block) gets its sorted slice staged in LDS as 32-bit ids when it
    // fits -- eight dependent LDS reads per draw instead of eight dependent trips to L2 at the start of every block of a
    // launch that is a single generation of blocks (measured on the 4.2 GB roofline table: 176.6 -> us per launch)
    const int32_t* staged = nullptr;
    if constexpr (SAMPLE) {
        if (sa.words == 0 && S == 4 && uhi - ulo <= SAMPLER_STAGE_IDS && sa.value_num <= INT32_MAX) {  // block-uniform
            int32_t* st = reinterpret_cast<int32_t*>(used_bits_all);
            for (int64_t q = ulo + threadIdx.x; q < uhi; q += 256) st[q - ulo] = (int32_t)sa.used_idx[q];
            __syncthreads();
            staged = st;
        }
    }
    auto is_used = [&](int64_t id) -> bool {
        if (bits) return (bits[id >> 5] >> (id & 31)) & 1u;
        if (staged) {
            int a = 0, e = (int)(uhi - ulo);
            const int n = e, want = (int)id;
            while (a < e) {
                const int m = (a + e) >> 1;
                if (staged[m] < want) a = m + 1;
                else e = m;
            }
            return a < n && staged[a] == want;
        }
        int64_t a = ulo, e = uhi;
        while (a < e) {
            const int64_t m = (a + e) >> 1;
            if (sa.used_idx[m] < id) a = m + 1;
            else e = m;
        }
        return a < uhi && sa.used_idx[a] == id;
    };

    if constexpr (EARLY) {
        sc.template fold<false>(d.dim, gl, hh, rr, nullptr, nullptr);
    } else {
        h = clamp_row_flag(h_idx[b], d.n_head, bad);
        r = clamp_row_flag(r_idx ? r_idx[b] : r_fixed, d.n_rel, bad);
        sc.template build<false>(d, h, r, gl, nullptr, nullptr);
    }

    const int64_t* __restrict__ tb = t_idx + b * tsb;
    float* __restrict__ ob = out + b * osb;
    if constexpr (POS_FIRST) {
        // the positive: slot 0 of a four-row reduction whose other slots are zero (x + 0 is exact: the score a loop trip
        // would have produced, whatever slot it had there)
        float p[U];
#pragma unroll
        for (int u = 0; u < U; ++u) p[u] = 0.f;
        float aux = 0.f;
        if constexpr (MT::TWO_PASS) aux = group_sum<G>(sc.template pre<true>(tpos, d.dim, gl));
        p[0] = sc.template partial<true>(tpos, d.eps, d.dim, gl, aux);
        int which;
        const float tot = group_reduce_multi<G, U>(p, gl, which);
        if (s == 0 && sub == 0 && which == 0 && (gl & (G / U - 1)) == 0) {
            const float sv = SC::finalize(tot, d.margin);
            ob[0] = sv;
            sh_scores[0] = sv;
        }
    }

    // This wave's tails, in processing order: q = 0, 1, ... -> j(q) = ((q / NS) * S + s) * NS + q % NS.
    // Lane l prefetches the id of tail q = chunk * 64 + l (one coalesced-ish load per 64 tails, off the
    // critical path), the row loads then pick their id with a cross-lane read.
    auto j_of = [&](int q) { return ((q / NS) * S + s) * NS + (q % NS) + (POS_FIRST ? 1 : 0); };
    for (int q0 = 0; j_of(q0) < ntail; q0 += HW_WAVE) {  // wave-uniform: j_of is monotone in q for q % NS == 0
        const int jq = j_of(q0 + lane);
        int tpre = 0;
        if (jq < ntail) {
            int64_t id;
            if (SAMPLE && jq > 0) {
                id = 0;
                if (!bad_key) {
                    const int64_t e = (int64_t)(jq - 1) * n_groups + b;  // hw_neg_sample's position: draw * n_keys + key
                    const uint32_t span = (uint32_t)(sa.value_num - 1);
                    bool ok = false;
                    for (uint32_t attempt = 0; attempt < (uint32_t)SAMPLER_MAX_ATTEMPTS_FUSED && !ok; ++attempt) {
                        uint32_t c[4] = {(uint32_t)e, (uint32_t)((uint64_t)e >> 32) ^ (attempt << 8), (uint32_t)sa.offset,
                                         (uint32_t)(sa.offset >> 32)};
                        philox4(sa.seed, c);
#pragma unroll
                        for (int t = 0; t < 4 && !ok; ++t) {
                            id = 1 + (int64_t)(((uint64_t)c[t] * (uint64_t)span) >> 32);
                            ok = !is_used(id);
                        }
                    }
                    bad |= !ok;
                } else {
                    bad = true;
                }
                t_idx[b * tsb + (int64_t)jq * tsj] = id;
            } else {
                id = tb[(int64_t)jq * tsj];
            }
            tpre = clamp_row_flag(id, d.n_tail, bad);
        }
        if constexpr (SAMPLE) __builtin_amdgcn_s_setprio(0);
#pragma unroll 1
        for (int k = 0; k < HW_WAVE / NS; k += U) {  // k counts rows of NS tails inside the chunk
            if (j_of(q0 + k * NS) >= ntail) break;   // wave-uniform
            R t[U][MT::NT];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int ql = (k + u) * NS + sub;  // position inside the 64-tail chunk
                int tj;
                if constexpr (NS == 1) tj = __builtin_amdgcn_readlane(tpre, ql);  // wave-uniform -> scalar row base
                else tj = __shfl(tpre, ql, HW_WAVE);
#pragma unroll
                for (int kk = 0; kk < MT::NT; ++kk) t[u][kk].load_raw(d.tail[kk] + (size_t)tj * d.dim, d.dim, gl);
            }
            float p[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float aux = 0.f;
                if constexpr (MT::TWO_PASS) aux = group_sum<G>(sc.template pre<true>(t[u], d.dim, gl));
                p[u] = sc.template partial<true>(t[u], d.eps, d.dim, gl, aux);
            }
            int which;
            const float tot = group_reduce_multi<G, U>(p, gl, which);
            // one writer lane per tail: the first lane of the sub-block that ended up owning it
            const int jw = j_of(q0 + (k + which) * NS + sub);
            if ((gl & (G / U - 1)) == 0 && jw < ntail) {
                const float sv = SC::finalize(tot, d.margin);
                ob[(int64_t)jw * osj] = sv;
                if constexpr (SAMPLE && LOSS) sh_scores[jw] = sv;
            }
        }
    }
    if (oob && __any(bad)) {
        if (lane == 0) atomicAdd(oob, 1);
    }
    if constexpr (SAMPLE && LOSS) {
        // The transcendental work (one exp and one log-sigmoid per negative) is spread over the block's 256 threads; the
        // sums stay in wave 0, in self_adv_kernel's order (lane l adds j = 1 + l, 65 + l, ... then the wave butterfly), so
        // the group term equals that kernel's bit for bit.
        __shared__ float sh_e[LOSS_MAX_TAILS], sh_ls[LOSS_MAX_TAILS], sh_mx[4];
        __syncthreads();  // every wave's scores of this group sit in sh_scores (written next to the global stores)
        float mx = -INFINITY;
        for (int j = 1 + (int)threadIdx.x; j < ntail; j += 256) mx = fmaxf(mx, la.alpha * sh_scores[j]);
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
        if (lane == 0) sh_mx[threadIdx.x / HW_WAVE] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(sh_mx[0], sh_mx[1]), fmaxf(sh_mx[2], sh_mx[3]));
        for (int j = 1 + (int)threadIdx.x; j < ntail; j += 256) {
            const float sj = sh_scores[j];
            sh_e[j] = expf(la.alpha * sj - mx);
            sh_ls[j] = log_sigmoid(-sj);
        }
        __syncthreads();
        if (threadIdx.x < HW_WAVE) {
            float z = 0.f;
            for (int j = 1 + lane; j < ntail; j += 64) z += sh_e[j];
            z = group_sum<64>(z);
            const float inv_z = z > 0.f ? 1.f / z : 0.f;
            float acc = 0.f;
            for (int j = 1 + lane; j < ntail; j += 64) {
                const float p = sh_e[j] * inv_z;
                acc -= p * sh_ls[j];
            }
            acc = group_sum<64>(acc);
            if (lane == 0) la.group_loss[b] = acc - log_sigmoid(sh_scores[0]);
        }
    }
}

// The finish as a launch of its own (the unchained form, and the flush of a chain): a cross-workgroup hand-off inside the
// scorer would need an agent-scope release per block, which on this part writes the XCD's L2 back (measured: 47 -> 118 us).
static __global__ __launch_bounds__(256) void selfadv_finish_kernel(const float* __restrict__ group_loss, int n, float inv_n,
                                                             float* __restrict__ loss_out) {
    selfadv_finish_block(group_loss, n, inv_n, loss_out);
}

// ------------------------------------------------------------------------------------------------
// grouped scorer, XCD-partitioned: tails are routed by id so that each XCD's private L2 only ever sees
// 1/8 of the tail table.  Block bid works for partition p = bid % 8 (blocks are dealt round-robin over the
// 8 XCDs, MI355X_MICROARCH.md 'Workgroup dispatch'); its 4 waves take groups (bid / 8) * 4 + wave and score
// only the tails with (id & 7) == p.  Placement is a speed assumption only: every (group, tail) pair is
// scored exactly once whatever XCD a block lands on.  ml-1m-small: 22 MB of entity rows -> 2.75 MB per
// XCD, inside the 4 MB L2, instead of streaming from Infinity Cache.
// Each wave compacts the ids of its partition into a small LDS list (ballot + prefix popcount) and drains
// it 64 entries at a time with the same U-rows-in-flight loop as the plain kernel.
// ------------------------------------------------------------------------------------------------
template <int MODEL, int G, int NE, bool VEC, int U>
__global__ __launch_bounds__(256) void kge_score_grouped_xcd_kernel(
    hw_kge_desc d, const int64_t* __restrict__ h_idx, const int64_t* __restrict__ r_idx, int64_t r_fixed,
    const int64_t* __restrict__ t_idx, int64_t tsb, int64_t tsj, int64_t n_groups, int ntail,
    float* __restrict__ out, int64_t osb, int64_t osj, int32_t* oob) {
    using SC = Scorer<MODEL, G, NE, VEC>;
    using MT = ModelTraits<MODEL>;
    using R = Row<G, NE, VEC>;
    constexpr int NS = HW_WAVE / G;
    __shared__ int sel_id[4][128];
    __shared__ int sel_j[4][128];

    const int lane = threadIdx.x & (HW_WAVE - 1);
    const int wave = wave_in_block();
    const int sub = lane / G, gl = lane % G;
    const int part = blockIdx.x & 7;
    const int64_t b = (int64_t)(blockIdx.x >> 3) * 4 + wave;
    if (b >= n_groups) return;  // wave-uniform; no block-wide barrier is used below

    bool bad = false;
    const int h = clamp_row_flag(h_idx[b], d.n_head, bad);
    const int r = clamp_row_flag(r_idx ? r_idx[b] : r_fixed, d.n_rel, bad);
    SC sc;
    sc.template build<false>(d, h, r, gl, nullptr, nullptr);

    const int64_t* __restrict__ tb = t_idx + b * tsb;
    float* __restrict__ ob = out + b * osb;
    int* __restrict__ lid = sel_id[wave];
    int* __restrict__ lj = sel_j[wave];
    int cnt = 0;  // entries waiting in the list (wave-uniform)

    for (int j0 = 0; j0 < ntail; j0 += HW_WAVE) {
        const int j = j0 + lane;
        int id = 0;
        bool mine = false;
        if (j < ntail) {
            id = clamp_row_flag(tb[(int64_t)j * tsj], d.n_tail, bad);
            mine = (id & 7) == part;
        }
        const unsigned long long mask = __ballot(mine);
        if (mine) {
            const int pos = cnt + __popcll(mask & ((1ull << lane) - 1ull));
            lid[pos] = id;
            lj[pos] = j;
        }
        cnt += __popcll(mask);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const bool last = j0 + HW_WAVE >= ntail;
        while (cnt >= HW_WAVE || (last && cnt > 0)) {
            const int n_now = cnt < HW_WAVE ? cnt : HW_WAVE;
#pragma unroll 1
            for (int k = 0; k * NS < n_now; k += U) {
                int jj[U];
                bool ok[U];
                R t[U][MT::NT];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int ql = (k + u) * NS + sub;
                    ok[u] = ql < n_now;
                    const int tj = ok[u] ? lid[ql] : 0;
                    jj[u] = ok[u] ? lj[ql] : 0;
#pragma unroll
                    for (int kk = 0; kk < MT::NT; ++kk) t[u][kk].load_raw(d.tail[kk] + (size_t)tj * d.dim, d.dim, gl);
                }
                float p[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                float aux = 0.f;
                if constexpr (MT::TWO_PASS) aux = group_sum<G>(sc.template pre<true>(t[u], d.dim, gl));
                p[u] = sc.template partial<true>(t[u], d.eps, d.dim, gl, aux);
            }
                int which;
                const float tot = group_reduce_multi<G, U>(p, gl, which);
                int jw = jj[0];
                bool okw = ok[0];
#pragma unroll
                for (int u = 1; u < U; ++u) {
                    jw = which == u ? jj[u] : jw;
                    okw = which == u ? ok[u] : okw;
                }
                if ((gl & (G / U - 1)) == 0 && okw) ob[(int64_t)jw * osj] = SC::finalize(tot, d.margin);
            }
            // keep the (at most 63) entries that did not fit this round
            const int rest = cnt - n_now;
            int mid = 0, mj = 0;
            if (lane < rest) mid = lid[n_now + lane], mj = lj[n_now + lane];
            __builtin_amdgcn_wave_barrier();
            if (lane < rest) lid[lane] = mid, lj[lane] = mj;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            cnt = rest;
        }
    }
    if (oob && __any(bad)) {
        if (lane == 0) atomicAdd(oob, 1);
    }
}

// ------------------------------------------------------------------------------------------------
// full sort.  wave task = (tail chunk c, head block hb): HB queries live in registers, the chunk's
// tail rows stream through once and are scored against all HB of them.  Head block is the fastest
// varying task index so the waves that share a tail chunk run together (L1/L2 reuse).
// ------------------------------------------------------------------------------------------------
template <int MODEL, int G, int NE, bool VEC, int HB, int U>
__global__ __launch_bounds__(256) void kge_full_sort_kernel(
    hw_kge_desc d, const int64_t* __restrict__ h_idx, const int64_t* __restrict__ r_idx, int64_t r_fixed,
    int n_heads, int64_t lo, int64_t hi, int chunk, float* __restrict__ out, int64_t ld, int32_t* oob) {
    using SC = Scorer<MODEL, G, NE, VEC>;
    using MT = ModelTraits<MODEL>;
    using R = Row<G, NE, VEC>;
    constexpr int NS = HW_WAVE / G;

    const int lane = threadIdx.x & (HW_WAVE - 1);
    const int sub = lane / G, gl = lane % G;
    const int n_hb = (n_heads + HB - 1) / HB;
    const int64_t n_chunks = (hi - lo + chunk - 1) / chunk;
    const int64_t task = (int64_t)blockIdx.x * (blockDim.x / HW_WAVE) + wave_in_block();
    if (task >= n_chunks * n_hb) return;
    const int hb = (int)(task % n_hb);
    const int64_t c = task / n_hb;

    SC sc[HB];
#pragma unroll
    for (int q = 0; q < HB; ++q) {
        const int bi = min(hb * HB + q, n_heads - 1);
        const int h = clamp_row(h_idx[bi], d.n_head, oob);
        const int r = clamp_row(r_idx ? r_idx[bi] : r_fixed, d.n_rel, oob);
        sc[q].template build<false>(d, h, r, gl, nullptr, nullptr);
    }

    const int64_t n0 = lo + c * chunk;
    const int64_t n1 = min(n0 + chunk, hi);
    for (int64_t nb = n0; nb < n1; nb += NS * U) {
        int64_t n[U];
        R t[U][MT::NT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            n[u] = nb + u * NS + sub;
            const int64_t row = n[u] < n1 ? n[u] : n0;
#pragma unroll
            for (int k = 0; k < MT::NT; ++k) t[u][k].load_raw(d.tail[k] + (size_t)row * d.dim, d.dim, gl);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float aux = 0.f;  // tail-only pre-reduction (TransD: t . t_vec), shared by the HB queries
            if constexpr (MT::TWO_PASS) aux = group_sum<G>(sc[0].template pre<true>(t[u], d.dim, gl));
            if constexpr (HB == 4) {
                // the four heads' partial sums of this tail share one reduce-scatter butterfly (4 + 4 cross-lane steps
                // instead of 4 x 6); four lanes of the group end up owning one head each
                float p[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) p[q] = sc[q].template partial<true>(t[u], d.eps, d.dim, gl, aux);
                int which;
                const float tot = group_reduce_multi<G, 4>(p, gl, which);
                const int bi = hb * HB + which;
                if ((gl & (G / 4 - 1)) == 0 && n[u] < n1 && bi < n_heads)
                    out[(int64_t)bi * ld + (n[u] - lo)] = SC::finalize(tot, d.margin);
            } else {
#pragma unroll
                for (int q = 0; q < HB; ++q) {
                    float p = sc[q].template partial<true>(t[u], d.eps, d.dim, gl, aux);
                    p = group_sum<G>(p);
                    const int bi = hb * HB + q;
                    if (gl == 0 && n[u] < n1 && bi < n_heads) out[(int64_t)bi * ld + (n[u] - lo)] = SC::finalize(p, d.margin);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// full sort fused with the evaluation mask and a first-stage top-k: the [B, N] score matrix
// is never written.  A wave scores a sub-chunk of SUB = 128 consecutive tails for HB heads into LDS, applies the
// mask (column `mask_col` and the history bitmap -> -inf, trainer.py:731-734), and keeps the k best of the
// sub-chunk per head by k wave arg-max rounds on (value desc, index asc) keys.  A second small kernel ranks the
// N / 128 * k candidates of every head.  Exactly the top-k of the full masked row as long as k <= 128.
// ------------------------------------------------------------------------------------------------
constexpr int FS_SUB = 128;

__device__ __forceinline__ uint64_t score_key(float v, uint32_t idx) {
    uint32_t u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((uint64_t)u << 32) | (uint32_t)(~idx);
}

template <int MODEL, int G, int NE, bool VEC, int HB>
__global__ __launch_bounds__(256) void kge_full_sort_topk_kernel(
    hw_kge_desc d, const int64_t* __restrict__ h_idx, const int64_t* __restrict__ r_idx, int64_t r_fixed,
    int n_heads, int64_t lo, int64_t hi, const uint32_t* __restrict__ mask_bits, int64_t mask_words, int64_t mask_col,
    int k, float* __restrict__ cand_val, int64_t* __restrict__ cand_idx, int64_t n_sub,
    const int32_t* __restrict__ run_if) {
    using SC = Scorer<MODEL, G, NE, VEC>;
    using MT = ModelTraits<MODEL>;
    using R = Row<G, NE, VEC>;
    constexpr int NS = HW_WAVE / G;
    constexpr int U = 2;
    __shared__ float lds_sc[4][HB][FS_SUB];
    if (run_if && *run_if == 0) return;  // predicated launch (hw_kge_full_sort_topk_if): decided on the device, no host read

    const int lane = threadIdx.x & (HW_WAVE - 1);
    const int wave = wave_in_block();
    const int sub = lane / G, gl = lane % G;
    const int n_hb = (n_heads + HB - 1) / HB;
    const int64_t task = (int64_t)blockIdx.x * 4 + wave;
    if (task >= n_sub * n_hb) return;
    const int hb = (int)(task % n_hb);
    const int64_t c = task / n_hb;

    SC sc[HB];
#pragma unroll
    for (int q = 0; q < HB; ++q) {
        const int bi = min(hb * HB + q, n_heads - 1);
        const int h = clamp_row(h_idx[bi], d.n_head, nullptr);
        const int r = clamp_row(r_idx ? r_idx[bi] : r_fixed, d.n_rel, nullptr);
        sc[q].template build<false>(d, h, r, gl, nullptr, nullptr);
    }
    const int64_t n0 = lo + c * FS_SUB;
    const int64_t n1 = min(n0 + FS_SUB, hi);
    float(*my)[FS_SUB] = lds_sc[wave];
    for (int64_t nb = n0; nb < n1; nb += NS * U) {
        int64_t n[U];
        R t[U][MT::NT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            n[u] = nb + u * NS + sub;
            const int64_t row = n[u] < n1 ? n[u] : n0;
#pragma unroll
            for (int kk = 0; kk < MT::NT; ++kk) t[u][kk].load_raw(d.tail[kk] + (size_t)row * d.dim, d.dim, gl);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float aux = 0.f;  // tail-only pre-reduction (TransD: t . t_vec), shared by the HB queries
            if constexpr (MT::TWO_PASS) aux = group_sum<G>(sc[0].template pre<true>(t[u], d.dim, gl));
            if constexpr (HB == 4) {
                float p[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) p[q] = sc[q].template partial<true>(t[u], d.eps, d.dim, gl, aux);
                int which;
                const float tot = group_reduce_multi<G, 4>(p, gl, which);
                if ((gl & (G / 4 - 1)) == 0 && n[u] < n1) my[which][n[u] - n0] = SC::finalize(tot, d.margin);
            } else {
#pragma unroll
                for (int q = 0; q < HB; ++q) {
                    float p = sc[q].template partial<true>(t[u], d.eps, d.dim, gl, aux);
                    p = group_sum<G>(p);
                    if (gl == 0 && n[u] < n1) my[q][n[u] - n0] = SC::finalize(p, d.margin);
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // mask + k arg-max rounds per head; lane l owns sub-chunk positions l and l + 64
    const int len = (int)(n1 - n0);
#pragma unroll
    for (int q = 0; q < HB; ++q) {
        const int bi = hb * HB + q;
        if (bi >= n_heads) break;  // wave-uniform
        uint64_t key[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int pos = lane + 64 * e;
            key[e] = 0ull;
            if (pos < len) {
                const int64_t col = n0 + pos - lo;  // column of the [B, N] matrix
                float v = my[q][pos];
                const bool masked = (col == mask_col) ||
                                    (mask_bits && ((mask_bits[(size_t)bi * mask_words + (col >> 5)] >> (col & 31)) & 1u));
                if (masked) v = -INFINITY;
                key[e] = score_key(v, (uint32_t)col);
            }
        }
        uint64_t bound = ~0ull;
        for (int r = 0; r < k; ++r) {
            uint64_t best = 0;
#pragma unroll
            for (int e = 0; e < 2; ++e)
                if (key[e] < bound && key[e] > best) best = key[e];
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                const uint64_t o = __shfl_xor(best, m, 64);
                best = o > best ? o : best;
            }
            if (lane == 0) {
                const size_t o = ((size_t)bi * n_sub + c) * k + r;
                if (best == 0) {
                    cand_val[o] = -INFINITY;
                    cand_idx[o] = -1;
                } else {
                    const uint32_t col = ~(uint32_t)(best & 0xffffffffu);
                    uint32_t u = (uint32_t)(best >> 32);  // undo the order-preserving map: the masked value
                    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
                    cand_val[o] = __uint_as_float(u);
                    cand_idx[o] = col;
                }
            }
            bound = best;
            if (best == 0) bound = 0;
        }
    }
}

// rank the candidates of one head: block per row, k rounds, key = (value, ~global index)
static __global__ __launch_bounds__(256) void topk_candidates_kernel(const float* __restrict__ cand_val,
                                                              const int64_t* __restrict__ cand_idx, int64_t n_cand,
                                                              int k, float* __restrict__ out_val,
                                                              int64_t* __restrict__ out_idx,
                                                              const int32_t* __restrict__ run_if) {
    __shared__ uint64_t red[4];
    __shared__ uint64_t winner;
    if (run_if && *run_if == 0) return;
    const float* __restrict__ v = cand_val + (size_t)blockIdx.x * n_cand;
    const int64_t* __restrict__ ix = cand_idx + (size_t)blockIdx.x * n_cand;
    uint64_t bound = ~0ull;
    for (int r = 0; r < k; ++r) {
        uint64_t best = 0;
        for (int64_t i = threadIdx.x; i < n_cand; i += 256) {
            if (ix[i] < 0) continue;
            const uint64_t key = score_key(v[i], (uint32_t)ix[i]);
            if (key < bound && key > best) best = key;
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const uint64_t o = __shfl_xor(best, m, 64);
            best = o > best ? o : best;
        }
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint64_t w = red[0];
            for (int i = 1; i < 4; ++i) w = red[i] > w ? red[i] : w;
            winner = w;
            const size_t o = (size_t)blockIdx.x * k + r;
            if (w == 0) {
                out_val[o] = -INFINITY;
                out_idx[o] = -1;
            } else {
                uint32_t u = (uint32_t)(w >> 32);
                u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
                out_val[o] = __uint_as_float(u);
                out_idx[o] = (int64_t)(~(uint32_t)(w & 0xffffffffu));
            }
        }
        __syncthreads();
        bound = winner;
    }
}

static __global__ void build_mask_bits_kernel(const int64_t* __restrict__ hr, const int64_t* __restrict__ hc, int64_t n_hist,
                                       int64_t n_rows, int64_t n_cols, int64_t words, uint32_t* __restrict__ bits) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_hist) return;
    const int64_t r = hr[i], c = hc[i];
    if (r >= 0 && r < n_rows && c >= 0 && c < n_cols) atomicOr(bits + (size_t)r * words + (c >> 5), 1u << (c & 31));
}

// ------------------------------------------------------------------------------------------------
// host-side dispatch
// ------------------------------------------------------------------------------------------------
static int check_desc(const hw_kge_desc* d) {
    if (!d) return HW_ERR_INVALID_ARG;
    if (d->dim < 1 || d->dim > HW_MAX_DIM) return HW_ERR_INVALID_ARG;
    if (d->n_head < 1 || d->n_rel < 1 || d->n_tail < 1) return HW_ERR_INVALID_ARG;
    int nh, nr, nt;
    switch (d->model) {
        case HW_TRANSE:
        case HW_DISTMULT:
        case HW_TORUSE:
        case HW_HOLE:
        case HW_RESCAL: nh = 1, nr = 1, nt = 1; break;
        case HW_ROTATE: nh = 2, nr = 1, nt = 2; break;
        case HW_COMPLEX: nh = 2, nr = 2, nt = 2; break;
        case HW_TRANSH: nh = 1, nr = 2, nt = 1; break;
        case HW_TRANSD: nh = 2, nr = 2, nt = 2; break;
        case HW_ANALOGY: nh = 3, nr = 3, nt = 3; break;
        default: return HW_ERR_UNSUPPORTED;
    }
    for (int k = 0; k < nh; ++k)
        if (!d->head[k]) return HW_ERR_INVALID_ARG;
    for (int k = 0; k < nr; ++k)
        if (!d->rel[k]) return HW_ERR_INVALID_ARG;
    for (int k = 0; k < nt; ++k)
        if (!d->tail[k]) return HW_ERR_INVALID_ARG;
    // 16-byte loads need 16-byte aligned rows: dim % 4 == 0 and aligned table bases
    return HW_OK;
}

static bool tables_aligned16(const hw_kge_desc* d) {
    uintptr_t m = 0;
    for (int k = 0; k < HW_KGE_MAX_TABLES; ++k) m |= (uintptr_t)d->head[k] | (uintptr_t)d->rel[k] | (uintptr_t)d->tail[k];
    return (m & 15) == 0;
}

#define HW_DISPATCH_SHAPE(MODEL, shape, CALL)                                   \
    do {                                                                        \
        if (shape.vec) {                                                        \
            if (shape.G == 16) { CALL(MODEL, 16, 4, true); }                    \
            else if (shape.G == 32) { CALL(MODEL, 32, 4, true); }               \
            else if (shape.NE == 4) { CALL(MODEL, 64, 4, true); }               \
            else if (shape.NE == 8) { CALL(MODEL, 64, 8, true); }               \
            else { CALL(MODEL, 64, 16, true); }                                 \
        } else {                                                                \
            if (shape.NE == 1) { CALL(MODEL, 64, 1, false); }                   \
            else if (shape.NE == 4) { CALL(MODEL, 64, 4, false); }              \
            else { CALL(MODEL, 64, 16, false); }                                \
        }                                                                       \
    } while (0)

// FAMILY 0 = the five hot-path scorers (kge_score.hip), FAMILY 1 = the widened set (kge_score_ext.hip): the impl
// templates below are instantiated once per family so that the two translation units compile in parallel.
#define HW_FAMILY_OF(model) ((model) <= HW_RESCAL ? 0 : 1)
#define HW_DISPATCH_ONE(M, shape, CALL)                                         \
    case M:                                                                     \
        if constexpr (FAMILY == HW_FAMILY_OF(M)) HW_DISPATCH_SHAPE(M, shape, CALL); \
        break;
#define HW_DISPATCH_MODEL(model, shape, CALL)                                   \
    do {                                                                        \
        switch (model) {                                                        \
            HW_DISPATCH_ONE(HW_TRANSE, shape, CALL)                             \
            HW_DISPATCH_ONE(HW_ROTATE, shape, CALL)                             \
            HW_DISPATCH_ONE(HW_COMPLEX, shape, CALL)                            \
            HW_DISPATCH_ONE(HW_DISTMULT, shape, CALL)                           \
            HW_DISPATCH_ONE(HW_RESCAL, shape, CALL)                             \
            HW_DISPATCH_ONE(HW_TORUSE, shape, CALL)                             \
            HW_DISPATCH_ONE(HW_TRANSH, shape, CALL)                             \
            HW_DISPATCH_ONE(HW_TRANSD, shape, CALL)                             \
            HW_DISPATCH_ONE(HW_ANALOGY, shape, CALL)                            \
            HW_DISPATCH_ONE(HW_HOLE, shape, CALL)                               \
            default: return HW_ERR_UNSUPPORTED;                                 \
        }                                                                       \
    } while (0)

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

static inline int64_t full_sort_topk_workspace_bytes(int64_t n_heads, int64_t n_tails, int32_t k) {
    const int64_t words = (n_tails + 31) / 32, n_sub = (n_tails + FS_SUB - 1) / FS_SUB;
    return n_heads * words * 4 + n_heads * n_sub * k * (4 + 8) + 64;
}
template <int FAMILY>
static int score_grouped_impl(const hw_kge_desc* desc, const int64_t* h_idx, const int64_t* r_idx,
                                    int64_t r_fixed, const int64_t* t_idx, int64_t t_stride_b, int64_t t_stride_j,
                                    int64_t n_groups, int64_t n_tail_per_group, float* out, int64_t o_stride_b,
                                    int64_t o_stride_j, int32_t* oob_counter, void* stream,
                                    const SamplerArgs* sampler = nullptr, const SelfAdvArgs* selfadv = nullptr) {
    int rc = check_desc(desc);
    if (rc) return rc;
    if (!h_idx || !t_idx || !out || n_groups < 0 || n_tail_per_group < 0) return HW_ERR_INVALID_ARG;
    if (n_tail_per_group > INT32_MAX / 2) return HW_ERR_INVALID_ARG;
    if (n_groups == 0 || n_tail_per_group == 0) return HW_OK;
    if (sampler && FAMILY != 0) return HW_ERR_UNSUPPORTED;
    const RowShape shape = row_shape(desc->dim, tables_aligned16(desc));
    const int ntail = (int)n_tail_per_group;
    const int NS = HW_WAVE / shape.G;
    hipStream_t st = (hipStream_t)stream;
    HW_CLEAR_ERR();
    // Variant: groups with many tails are routed by tail id to XCD-local L2 slices; short groups (predict,
    // pos/neg pairs) use the plain kernel.  HW_GROUPED_VARIANT = 0 (plain) / 1 (xcd) overrides for experiments.
    int variant = sampler ? 0 : env_int("HW_GROUPED_VARIANT", -1);  // the in-kernel sampler lives in the plain kernel
    if (variant < 0) {
        // tail rows the launch can touch: up to 4 MiB they sit in every XCD's L2 anyway; beyond the 256 MiB
        // Infinity Cache nothing is resident and the plain kernel's lower overhead wins (measured, profiles/r1)
        const int nt = (desc->model == HW_ROTATE || desc->model == HW_COMPLEX || desc->model == HW_TRANSD) ? 2
                       : desc->model == HW_ANALOGY ? 3 : 1;
        const double footprint = (double)desc->n_tail * desc->dim * 4.0 * nt;
        variant = (ntail >= 64 && footprint > 6.0e6 && footprint <= 256.0e6) ? 1 : 0;
        // the XCD kernel rebuilds the query in each of its 8 partitions: too dear for the d^2 builds
        if (desc->model == HW_HOLE || desc->model == HW_RESCAL) variant = 0;
    }
    if (variant == 1) {
        const int64_t n_blocks = ((n_groups + 3) / 4) * 8;
        if (n_blocks > INT32_MAX) return HW_ERR_INVALID_ARG;
#define CALLX(MODEL, G_, NE_, VEC_)                                                                                 \
    hipLaunchKernelGGL((kge_score_grouped_xcd_kernel<MODEL, G_, NE_, VEC_, 4>), dim3((unsigned)n_blocks), dim3(256), \
                       0, st, *desc, h_idx, r_idx, r_fixed, t_idx, t_stride_b, t_stride_j, n_groups, ntail, out,    \
                       o_stride_b, o_stride_j, oob_counter)
        HW_DISPATCH_MODEL(desc->model, shape, CALLX);
#undef CALLX
        return HW_LAUNCH_RC();
    }
    const bool u8 = !sampler && env_int("HW_GROUPED_U", 4) == 8;
    constexpr int U = 4;
    // splits per group: enough waves to fill 256 CUs x 32 wave slots, but keep >= U*NS tails per wave
    // so the (h, r) query build is amortised.
    int S = env_int("HW_GROUPED_SPLITS", 0);
    if (S <= 0) {
        const int per_wave = U * NS * 2;
        const int s_max = (ntail + per_wave - 1) / per_wave;
        const int64_t want = (16384 + n_groups - 1) / n_groups;
        S = (int)(want < 1 ? 1 : (want > s_max ? s_max : want));
        // HolE / RESCAL pay d^2 per query build: every extra split repeats it, so only split when the launch would
        // otherwise leave most of the chip idle
        // the in-kernel sampler rebuilds the key's used-set bitmap in every split: half the splits measured 2 % faster
        // (B = 2048, K = 256: 53.8 us at S = 4, 54.9 at S = 8, 63.8 at S = 16)
        if (sampler && S > 4) S = 4;
        // (measured at B = 2048, K = 256: RESCAL d = 64 16.6 / 20.3 / 29.5 us and HolE d = 200 37 / 33 / 40 us at S = 1 / 2 / 4)
        if ((desc->model == HW_HOLE || desc->model == HW_RESCAL) && n_groups >= 1024)
            S = desc->model == HW_HOLE && s_max >= 2 ? 2 : 1;
    }
    const int64_t n_tasks = n_groups * S;
    const int64_t n_blocks = (n_tasks + 3) / 4;
    if (n_blocks > INT32_MAX) return HW_ERR_INVALID_ARG;
    // the fused loss needs a block to own a whole group (the four splits of it) and int32 group counts
    if (selfadv && (!sampler || S != 4 || n_groups > INT32_MAX || ntail > LOSS_MAX_TAILS)) return HW_ERR_UNSUPPORTED;
#define CALL(MODEL, G_, NE_, VEC_)                                                                              \
    do {                                                                                                        \
        if constexpr (FAMILY == 0) {                                                                            \
            if (sampler && selfadv) {                                                                           \
                hipLaunchKernelGGL((kge_score_grouped_kernel<MODEL, G_, NE_, VEC_, U, true, true>),             \
                                   dim3((unsigned)n_blocks), dim3(256), (size_t)(sampler->words ? sampler->words * 16 : SAMPLER_STAGE_IDS * 4), st,\
                                   *desc,  \
                                   h_idx, r_idx, r_fixed, const_cast<int64_t*>(t_idx), t_stride_b, t_stride_j,   \
                                   n_groups, ntail, S, out, o_stride_b, o_stride_j, oob_counter, *sampler,       \
                                   *selfadv);                                                                    \
                break;                                                                                          \
            }                                                                                                   \
            if (sampler) {                                                                                      \
                hipLaunchKernelGGL((kge_score_grouped_kernel<MODEL, G_, NE_, VEC_, U, true>),                   \
                                   dim3((unsigned)n_blocks), dim3(256), (size_t)(sampler->words ? sampler->words * 16 : SAMPLER_STAGE_IDS * 4), st,\
                                   *desc,  \
                                   h_idx, r_idx, r_fixed, const_cast<int64_t*>(t_idx), t_stride_b, t_stride_j,   \
                                   n_groups, ntail, S, out, o_stride_b, o_stride_j, oob_counter, *sampler);      \
                break;                                                                                          \
            }                                                                                                   \
        }                                                                                                       \
            hipLaunchKernelGGL((kge_score_grouped_kernel<MODEL, G_, NE_, VEC_, U>), dim3((unsigned)n_blocks),    \
                               dim3(256), 0, st, *desc, h_idx, r_idx, r_fixed, const_cast<int64_t*>(t_idx),      \
                               t_stride_b, t_stride_j, n_groups, ntail, S, out, o_stride_b, o_stride_j,          \
                               oob_counter, SamplerArgs{});                                                      \
    } while (0)
    if (u8 && desc->model == HW_ROTATE && shape.vec && shape.G == 64 && shape.NE == 4) {  // experiment hook
        hipLaunchKernelGGL((kge_score_grouped_kernel<HW_ROTATE, 64, 4, true, 8>), dim3((unsigned)n_blocks), dim3(256), 0,
                           st, *desc, h_idx, r_idx, r_fixed, const_cast<int64_t*>(t_idx), t_stride_b, t_stride_j, n_groups,
                           ntail, S, out, o_stride_b, o_stride_j, oob_counter, SamplerArgs{});
        return HW_LAUNCH_RC();
    }
    HW_DISPATCH_MODEL(desc->model, shape, CALL);
#undef CALL
    if (selfadv && selfadv->loss_out)
        hipLaunchKernelGGL(selfadv_finish_kernel, dim3(1), dim3(256), 0, st, selfadv->group_loss, (int)n_groups,
                           selfadv->inv_b, selfadv->loss_out);
    return HW_LAUNCH_RC();
}

template <int FAMILY>
static int full_sort_impl(const hw_kge_desc* desc, const int64_t* h_idx, const int64_t* r_idx, int64_t r_fixed,
                                int64_t n_heads, int64_t tail_lo, int64_t tail_hi, float* out, int64_t out_ld,
                                int32_t* oob_counter, void* stream) {
    int rc = check_desc(desc);
    if (rc) return rc;
    if (!h_idx || !out || n_heads < 0 || n_heads > INT32_MAX) return HW_ERR_INVALID_ARG;
    if (tail_lo < 0 || tail_hi > desc->n_tail || tail_lo > tail_hi || out_ld < tail_hi - tail_lo)
        return HW_ERR_INVALID_ARG;
    if (n_heads == 0 || tail_lo == tail_hi) return HW_OK;
    const RowShape shape = row_shape(desc->dim, tables_aligned16(desc));
    const int NS = HW_WAVE / shape.G;
    const int64_t n = tail_hi - tail_lo;
    constexpr int U = 2;
    const bool hb4 = n_heads > 1;
    const int n_hb = hb4 ? (int)((n_heads + 3) / 4) : 1;
    // tail rows per wave task: aim for >= 8192 tasks, at least one U-batch, at most 256 rows
    int chunk = env_int("HW_FULLSORT_CHUNK", 0);
    if (chunk <= 0) {
        int64_t c = (n * n_hb + 8191) / 8192;
        const int unit = NS * U;
        c = ((c + unit - 1) / unit) * unit;
        if (c < unit) c = unit;
        if (c > 256) c = 256;
        chunk = (int)c;
    }
    const int64_t n_chunks = (n + chunk - 1) / chunk;
    const int64_t n_tasks = n_chunks * n_hb;
    const int64_t n_blocks = (n_tasks + 3) / 4;
    if (n_blocks > INT32_MAX) return HW_ERR_INVALID_ARG;
    HW_CLEAR_ERR();
    hipStream_t st = (hipStream_t)stream;
    const int nh = (int)n_heads;
#define CALL(MODEL, G_, NE_, VEC_)                                                                                  \
    do {                                                                                                            \
        if (hb4)                                                                                                    \
            hipLaunchKernelGGL((kge_full_sort_kernel<MODEL, G_, NE_, VEC_, 4, U>), dim3((unsigned)n_blocks),         \
                               dim3(256), 0, st, *desc, h_idx, r_idx, r_fixed, nh, tail_lo, tail_hi, chunk, out,     \
                               out_ld, oob_counter);                                                                 \
        else                                                                                                        \
            hipLaunchKernelGGL((kge_full_sort_kernel<MODEL, G_, NE_, VEC_, 1, U>), dim3((unsigned)n_blocks),         \
                               dim3(256), 0, st, *desc, h_idx, r_idx, r_fixed, nh, tail_lo, tail_hi, chunk, out,     \
                               out_ld, oob_counter);                                                                 \
    } while (0)
    HW_DISPATCH_MODEL(desc->model, shape, CALL);
#undef CALL
    return HW_LAUNCH_RC();
}

template <int FAMILY>
static int full_sort_topk_impl(const hw_kge_desc* desc, const int64_t* h_idx, const int64_t* r_idx, int64_t r_fixed,
                                     int64_t n_heads, int64_t tail_lo, int64_t tail_hi, const int64_t* hist_rows,
                                     const int64_t* hist_cols, int64_t n_hist, int32_t mask_first_col, int32_t k,
                                     void* workspace, int64_t workspace_bytes, float* out_val, int64_t* out_idx,
                                     void* stream, const int32_t* run_if = nullptr) {
    int rc = check_desc(desc);
    if (rc) return rc;
    if (!h_idx || !out_val || !out_idx || !workspace || n_heads < 1 || n_heads > INT32_MAX) return HW_ERR_INVALID_ARG;
    if (tail_lo < 0 || tail_hi > desc->n_tail || tail_lo >= tail_hi || k < 1 || k > FS_SUB) return HW_ERR_INVALID_ARG;
    if (n_hist < 0 || (n_hist > 0 && (!hist_rows || !hist_cols))) return HW_ERR_INVALID_ARG;
    const int64_t n = tail_hi - tail_lo;
    if (n > 0xffffffffll) return HW_ERR_INVALID_ARG;
    if (workspace_bytes < full_sort_topk_workspace_bytes(n_heads, n, k)) return HW_ERR_WORKSPACE;
    const int64_t words = (n + 31) / 32, n_sub = (n + FS_SUB - 1) / FS_SUB;
    uint32_t* bits = (uint32_t*)workspace;
    int64_t* cand_idx = (int64_t*)((char*)workspace + ((n_heads * words * 4 + 15) / 16) * 16);
    float* cand_val = (float*)(cand_idx + n_heads * n_sub * k);
    const RowShape shape = row_shape(desc->dim, tables_aligned16(desc));
    hipStream_t st = (hipStream_t)stream;
    HW_CLEAR_ERR();
    if (hw::fill_bytes_async(bits, 0, (size_t)n_heads * words * 4, st) != hipSuccess) return HW_ERR_LAUNCH;
    if (n_hist > 0)
        hipLaunchKernelGGL(build_mask_bits_kernel, dim3((unsigned)((n_hist + 255) / 256)), dim3(256), 0, st, hist_rows,
                           hist_cols, n_hist, n_heads, n, words, bits);
    const bool hb4 = n_heads > 1;
    const int n_hb = hb4 ? (int)((n_heads + 3) / 4) : 1;
    const int64_t n_blocks = (n_sub * n_hb + 3) / 4;
    if (n_blocks > INT32_MAX) return HW_ERR_INVALID_ARG;
    const int nh = (int)n_heads;
    const int64_t mask_col = mask_first_col ? 0 : -1;
#define CALL(MODEL, G_, NE_, VEC_)                                                                                    \
    do {                                                                                                              \
        if (hb4)                                                                                                      \
            hipLaunchKernelGGL((kge_full_sort_topk_kernel<MODEL, G_, NE_, VEC_, 4>), dim3((unsigned)n_blocks),         \
                               dim3(256), 0, st, *desc, h_idx, r_idx, r_fixed, nh, tail_lo, tail_hi, bits, words,      \
                               mask_col, k, cand_val, cand_idx, n_sub, run_if);                                        \
        else                                                                                                          \
            hipLaunchKernelGGL((kge_full_sort_topk_kernel<MODEL, G_, NE_, VEC_, 1>), dim3((unsigned)n_blocks),         \
                               dim3(256), 0, st, *desc, h_idx, r_idx, r_fixed, nh, tail_lo, tail_hi, bits, words,      \
                               mask_col, k, cand_val, cand_idx, n_sub, run_if);                                        \
    } while (0)
    HW_DISPATCH_MODEL(desc->model, shape, CALL);
#undef CALL
    hipLaunchKernelGGL(topk_candidates_kernel, dim3((unsigned)n_heads), dim3(256), 0, st, cand_val, cand_idx, n_sub * k, k,
                       out_val, out_idx, run_if);
    return HW_LAUNCH_RC();
}

// widened-family entry points (kge_score_ext.hip)
int ext_score_grouped(const hw_kge_desc* desc, const int64_t* h_idx, const int64_t* r_idx, int64_t r_fixed,
                      const int64_t* t_idx, int64_t t_stride_b, int64_t t_stride_j, int64_t n_groups,
                      int64_t n_tail_per_group, float* out, int64_t o_stride_b, int64_t o_stride_j, int32_t* oob_counter,
                      void* stream);
int ext_full_sort(const hw_kge_desc* desc, const int64_t* h_idx, const int64_t* r_idx, int64_t r_fixed, int64_t n_heads,
                  int64_t tail_lo, int64_t tail_hi, float* out, int64_t out_ld, int32_t* oob_counter, void* stream);
int ext_full_sort_topk(const hw_kge_desc* desc, const int64_t* h_idx, const int64_t* r_idx, int64_t r_fixed,
                       int64_t n_heads, int64_t tail_lo, int64_t tail_hi, const int64_t* hist_rows,
                       const int64_t* hist_cols, int64_t n_hist, int32_t mask_first_col, int32_t k, void* workspace,
                       int64_t workspace_bytes, float* out_val, int64_t* out_idx, void* stream,
                       const int32_t* run_if = nullptr);

}  // namespace hw
