// EXPERIMENT, not part of the product build (tools/pf_abl/build8.sh compiles it; tools/pf_abl/run8.py and stamps8.py run it).
//
// The fused PGPR policy kernel (hopwise_amd/csrc/pgpr_policy.hip: one wave per SIMD, 392-426 registers) rebuilt with TWO waves per
// SIMD: a block is eight waves over the same 128 beam rows, waves w and w + 4 share 32 rows and split the output tiles of both
// products, activations travel through LDS.  Logits bit-identical to the product kernel in all dropout modes and state widths.
// Measured (round 4, 1 M synthetic beam rows, one card, alternating): 2.35-2.59 ms against 2.39-2.58 ms for the product kernel,
// i.e. nothing -- DESIGN.md section 8 has the stamps and ablations that say why (vector work beside the partner's MFMA stream runs
// 2.6x slower than alone on this SIMD: the two waves' phases still add up).
//
// Build switches: PF8_SWAP (waves 0-3 prepare operands first instead of waves 4-7), PF8_W_PLANES (weights as the product's
// pre-split bf16 planes, 48 KB per chunk, instead of fp32 images split on the way into LDS, 32 KB), PF8_PRIO = 0 | 1 | 2
// (s_setprio: none / static for waves 4-7 / raised for vector phases), PF_ABL_W_L1 / PF_ABL_X_L1 (every weight load / every gather
// hits the CU's L1: timing only), PF8_STAMPS (s_memtime stamps of one block's waves, read back with pf8_read_stamps).
#include "../../hopwise_amd/csrc/pgpr_policy.hip"

namespace hw {

// fp32 image of a chunk for the 8-wave kernel: 16-byte slot S of the plane image (S = byte offset / 16 = 4 row + physical slot)
// belongs to thread S & 511 of a 512-thread block on its pass S >> 9; its eight floats travel as two fully coalesced 16-byte
// loads per wave (half 0: values 0..3, half 1: values 4..7).  Index of value v of slot S, in floats from the chunk's base:
constexpr int PF8_F32_CHUNK_FLOATS = 256 * PF_KC;  // 32 KB
__device__ __host__ __forceinline__ int pf8_f32_index(int S, int v) {
    const int pass = S >> 9, tid = S & 511, wave = tid >> 6, lane = tid & 63;
    return (((pass * 8 + wave) * 2 + (v >> 2)) * 64 + lane) * 4 + (v & 3);
}


// fp32 images of all chunks, the values hw_pgpr_policy_prepare splits (same element order as its planes)
__global__ __launch_bounds__(256) void pf8_prep_f32_kernel(const float* __restrict__ w2, int H1, const float* __restrict__ wa,
                                                           int act_dim, float* __restrict__ out) {
    const int q = blockIdx.x;
    const int n2 = H1 / PF_KC;
    for (int e = threadIdx.x; e < 256 * 4 * 4; e += 256) {  // (row, slot, pair of k): pgpr_policy_prep_kernel's walk
        const int r = e >> 4, slot = (e >> 2) & 3, pr = e & 3;
        const int g = slot >> 1, th = slot & 1;
        float x0, x1;
        if (q < n2) {
            const int k = PF_KC * q + 16 * g + 8 * th + 2 * pr;
            x0 = w2[(size_t)r * H1 + k], x1 = w2[(size_t)r * H1 + k + 1];
        } else {
            const int T = q - n2, j = 2 * pr;
            const int k = 32 * T + 16 * th + 4 * g + (j < 4 ? j : 8 + (j - 4));
            x0 = r < act_dim ? wa[(size_t)r * PF_H2 + k] : 0.f;
            x1 = r < act_dim ? wa[(size_t)r * PF_H2 + k + 1] : 0.f;
        }
        float* f = out + (size_t)q * PF8_F32_CHUNK_FLOATS;
        const int at = pf8_f32_index(pf_off(r, slot) >> 4, 2 * pr);
        f[at] = x0, f[at + 1] = x1;
    }
}

// ---- the same network with TWO waves per SIMD (round 4) -------------------------------------------------------------------
// The kernel above owns a SIMD with one wave (392-426 registers: both layers' accumulators of 32 beam rows x 256 outputs), and
// a wave that owns a SIMD alone hides nothing: its phases add up (tools/pf_abl: matrix work alone 1.70 ms, everything else
// alone 1.71, together 2.5-2.7).  Here a block is EIGHT waves over the same 128 beam rows: waves w and w + 4 (the pair that
// shares SIMD w) work for the same 32 rows and split the OUTPUT rows of both products -- n-tiles / action tiles 4 hf .. 4 hf + 3
// for hf = w >> 2 -- so that each holds 64 + 64 accumulator registers instead of 128 + 128 and two waves fit a SIMD: while one
// waits for its fragments or prepares operands, the other's MFMAs run.
//   * B operands (activations) travel through LDS now: for chunk c the wave of half hf produces k-step t = hf (its lane's
//     8 of the 16 values: a 32-byte piece of each slot-table row, ELU, dropout, split) and stores the three planes, 16 bytes
//     per lane, into the pair's exchange slot; both waves read both k-steps back.  The gather and the operand preparation
//     are therefore NOT duplicated (each value is still produced once), only the 6 Philox blocks per row are.
//   * layer 3: the accumulator tile T of layer 2 is still a B operand of actor chunk T in the register order the weight
//     preparation assumes; its owner (hf = T >> 2) converts it and passes it through the same exchange slots.
//   * per accumulator the order of the MFMAs is the one of the kernel above (chunk, k-step, six terms): the logits are
//     bit-identical (tests/test_gpu_pgpr_policy8.py).
// LDS: 2 x 48 KB weight stages + 48 KB exchange (2 stages x 4 pairs x 2 k-steps x 3 planes x 1 KB) + 5 KB biases = 149 KB.
#ifdef PF8_STAMPS  // timing experiment (tools/pf_abl): s_memtime stamps of the eight waves of one block
__device__ unsigned long long pf8_stamps[8 * 256];
#define PF8_STAMP()                                                                 \
    do {                                                                            \
        if (blockIdx.x == 4097 && lane == 0 && stamp_n < 256) pf8_stamps[wave * 256 + stamp_n] = __builtin_readcyclecounter(); \
        ++stamp_n;                                                                  \
    } while (0)
#else
#define PF8_STAMP() do {} while (0)
#endif
// issue priority of the pair (PF8_PRIO): 0 none; 1 static -- the younger half (waves 4-7) at priority 1 for the whole kernel;
// 2 per phase -- a wave raises its priority for its vector phases (operand preparation, weight split) and drops it for its
// MFMA phases, so that the partner's back-to-back MFMAs (which need the issue port 8 cycles in 32) do not starve it
#ifndef PF8_PRIO
#define PF8_PRIO 2
#endif
#if PF8_PRIO == 2
#define PF8_PRIO_VECTOR() __builtin_amdgcn_s_setprio(2)
#define PF8_PRIO_MATRIX() __builtin_amdgcn_s_setprio(0)
#else
#define PF8_PRIO_VECTOR() do {} while (0)
#define PF8_PRIO_MATRIX() do {} while (0)
#endif
#ifndef PF8_SUB_FENCE  // keeps the fragment reads of a later sub-group from being hoisted over an earlier one's MFMAs (spills)
#define PF8_SUB_FENCE __builtin_amdgcn_sched_barrier(0)
#endif
constexpr int PF8_THREADS = 512;
constexpr int PF8_XSLOT = 3 * 1024;                   // one k-step of one pair: three planes x 64 lanes x 16 B
constexpr int PF8_XSTAGE = 4 * 2 * PF8_XSLOT;         // 24 KB
constexpr int PF8_XBUF = 2 * PF8_XSTAGE;              // 48 KB

template <int NPARTS, int MODE>
__global__ __launch_bounds__(PF8_THREADS, 1) void pgpr_policy_fused8_kernel(
    PfTables tb, const float* __restrict__ b1, int H1, const uint8_t* __restrict__ planes, const float* __restrict__ b2,
    const float* __restrict__ ba, int act_dim, const int32_t* __restrict__ paths, const int32_t* __restrict__ n_rows_ptr,
    PfDrop dr, float* __restrict__ logits, int64_t ld, const uint8_t* __restrict__ planes_f32) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[2 * PF_CHUNK + PF8_XBUF + (1024 + PF_H2) * 4];
    uint8_t* const xbuf = lds + 2 * PF_CHUNK;
    float* const bias_lds = reinterpret_cast<float*>(lds + 2 * PF_CHUNK + PF8_XBUF);
    const int n_rows = n_rows_ptr[0];
    const int row0 = blockIdx.x * PF_ROWS;
    if (row0 >= n_rows) return;
    const int tid = threadIdx.x, wave = wave_in_block(), lane = tid & 63, b = lane & 31, g = lane >> 5;
    const int pair = wave & 3, hf = wave >> 2;  // scalars: every branch on hf is wave-uniform
#ifdef PF8_STAMPS
    int stamp_n = 0;
#endif
    const int row = row0 + pair * 32 + b;
    const int rowc = row < n_rows ? row : n_rows - 1;
    const int n2 = H1 / PF_KC;
    for (int i = tid; i < H1; i += PF8_THREADS) bias_lds[i] = b1[i];
    if (tid < PF_H2) bias_lds[1024 + tid] = b2[tid];

    const float* src[NPARTS];
    {
        typedef int pf_i32x4 __attribute__((ext_vector_type(4)));
        const pf_i32x4* pr = reinterpret_cast<const pf_i32x4*>(paths + (size_t)rowc * PF_PATH_INTS);
        const pf_i32x4 q0 = pr[0], q1 = pr[1], q2 = pr[2];
        const int uid = q0.x, len = q0.y;
        const int r1 = q0.z, t1 = q0.w, d1 = q1.x, r2 = q1.y, t2 = q1.z, d2 = q1.w, r3 = q2.x, t3 = q2.y, d3 = q2.z;
        auto sel = [](int j, int v1, int v2, int v3) { return j == 1 ? v1 : (j == 2 ? v2 : v3); };
        auto node_row = [&](int j, const float* tu, const float* te) -> const float* {
            const int ty = j == 0 ? PF_T_USER : sel(j, t1, t2, t3);
            const int id = j == 0 ? uid : sel(j, d1, d2, d3);
            return (ty == PF_T_USER ? tu : te) + (size_t)id * H1;
        };
        auto rel_row = [&](int j, const float* tr) -> const float* {
            const int r = sel(j, r1, r2, r3);
            return r == PF_SELF_LOOP ? tb.zero : tr + (size_t)r * H1;
        };
#pragma unroll
        for (int q = 0; q < NPARTS; ++q) src[q] = tb.zero;
        src[0] = tb.user0 + (size_t)uid * H1;
        const int lc = len < 1 ? 1 : (len > 4 ? 4 : len);
        src[1] = node_row(lc - 1, tb.cur_u, tb.cur_e);
        if constexpr (NPARTS > 2) {
            if (lc >= 2) {
                src[2] = node_row(lc - 2, tb.last_u, tb.last_e);
                src[3] = rel_row(lc - 1, tb.last_r);
            }
        }
        if constexpr (NPARTS > 4) {
            if (lc >= 3) {
                src[4] = node_row(lc - 3, tb.old_u, tb.old_e);
                src[5] = rel_row(lc - 2, tb.old_r);
            }
        }
#ifdef PF_ABL_X_L1  // timing experiment: every gather hits the CU's L1 (all lanes read the zero row)
#pragma unroll
        for (int q = 0; q < NPARTS; ++q) src[q] = tb.zero;
#endif
        // this wave's half of the lane's 64-byte piece of every slot row
#pragma unroll
        for (int q = 0; q < NPARTS; ++q) src[q] += 16 * g + 8 * hf;
    }

#ifdef PF8_W_PLANES  // the pre-split planes of the 4-wave kernel: 48 KB per chunk, no split here
    pf_u32x4 wreg[6];
    auto w_fetch = [&](int q) {
        const pf_u32x4* s = reinterpret_cast<const pf_u32x4*>(planes + (size_t)q * PF_CHUNK);
#pragma unroll
        for (int i = 0; i < 6; ++i) wreg[i] = s[tid + PF8_THREADS * i];
    };
    auto w_put = [&](int q) {
        pf_u32x4* d = reinterpret_cast<pf_u32x4*>(lds + (q & 1) * PF_CHUNK);
#pragma unroll
        for (int i = 0; i < 6; ++i) d[tid + PF8_THREADS * i] = wreg[i];
    };
#else
    // ---- weight chunk pipeline: 32 KB of fp32 in (512 threads x 4 x 16 B), split into the three plane images on the way into
    // LDS (the split is pf_split_pair's, as in hw_pgpr_policy_prepare: the same bits as the pre-split planes) ----
    const f32x4* const wf = reinterpret_cast<const f32x4*>(planes_f32);
    f32x4 wreg[4];
    auto w_fetch = [&](int q) {
#ifdef PF_ABL_W_L1  // timing experiment: every weight load hits the CU's L1
        const f32x4* s = wf;
#else
        const f32x4* s = wf + (size_t)q * (PF8_F32_CHUNK_FLOATS / 4);
#endif
#pragma unroll
        for (int i = 0; i < 4; ++i) wreg[i] = s[(wave * 2 + 16 * (i >> 1) + (i & 1)) * 64 + lane];
    };
    auto w_put = [&](int q) {
        uint8_t* d = lds + (q & 1) * PF_CHUNK + tid * 16;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const f32x4 a = wreg[2 * i], c = wreg[2 * i + 1];
            uint32_t h[4], m[4], l[4];
            pf_split_pair(a.x, a.y, h[0], m[0], l[0]);
            pf_split_pair(a.z, a.w, h[1], m[1], l[1]);
            pf_split_pair(c.x, c.y, h[2], m[2], l[2]);
            pf_split_pair(c.z, c.w, h[3], m[3], l[3]);
            *reinterpret_cast<pf_u32x4*>(d + i * 8192) = pf_u32x4{h[0], h[1], h[2], h[3]};
            *reinterpret_cast<pf_u32x4*>(d + i * 8192 + PF_PLANE) = pf_u32x4{m[0], m[1], m[2], m[3]};
            *reinterpret_cast<pf_u32x4*>(d + i * 8192 + 2 * PF_PLANE) = pf_u32x4{l[0], l[1], l[2], l[3]};
        }
    };
#endif
    // exchange slot of (stage, k-step t) of this pair, this lane's 16 bytes of plane p
    auto xslot = [&](int stage, int t, int p) -> uint8_t* {
        return xbuf + stage * PF8_XSTAGE + (pair * 2 + t) * PF8_XSLOT + p * 1024 + lane * 16;
    };
    auto put_planes = [&](int stage, int t, const float (&e)[8]) {
        uint32_t h[4], m[4], l[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) pf_split_pair(e[2 * p], e[2 * p + 1], h[p], m[p], l[p]);
        *reinterpret_cast<pf_u32x4*>(xslot(stage, t, 0)) = pf_u32x4{h[0], h[1], h[2], h[3]};
        *reinterpret_cast<pf_u32x4*>(xslot(stage, t, 1)) = pf_u32x4{m[0], m[1], m[2], m[3]};
        *reinterpret_cast<pf_u32x4*>(xslot(stage, t, 2)) = pf_u32x4{l[0], l[1], l[2], l[3]};
    };
    auto get_planes = [&](int stage, pf_bf16x8 (&bop)[2][3]) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int p = 0; p < 3; ++p) bop[t][p] = *reinterpret_cast<const pf_bf16x8*>(xslot(stage, t, p));
    };

    // ---- this wave's 8 values of chunk c: k = 32 c + 16 g + 8 hf + 0..7 ----
    // ONE gather buffer per wave: half 0 fetches chunk c + 1 at the head of trip c and turns it into operands at the tail
    // (the loads fly under its own MFMAs), half 1 turns chunk c + 1 into operands at the head of trip c and fetches chunk
    // c + 2 right after (the loads fly under its MFMAs and the barrier)
    f32x4 xs[NPARTS][2];
    f32x4 xm[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) xm[i] = f32x4{1.f, 1.f, 1.f, 1.f};
    auto x_fetch = [&](int c, f32x4 (&xq)[NPARTS][2], f32x4 (&mq)[2]) {
        const int k0 = PF_KC * c;
#pragma unroll
        for (int q = 0; q < NPARTS; ++q)
#pragma unroll
            for (int i = 0; i < 2; ++i) xq[q][i] = *reinterpret_cast<const f32x4*>(src[q] + k0 + 4 * i);
        if constexpr (MODE == 1)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                mq[i] = *reinterpret_cast<const f32x4*>(dr.mask1 + (size_t)rowc * H1 + k0 + 16 * g + 8 * hf + 4 * i);
    };
    uint32_t pw[4] = {0u, 0u, 0u, 0u};
    auto x_keep = [&](int c, int cu) -> uint32_t {  // this wave's 8 keep bits of chunk c (see the kernel above)
        uint32_t keep = 0xFFu;
        if constexpr (MODE == 3) {
            if (cu == 0) philox_bits4(dr.seed, dr.offset1, ((uint64_t)rowc * H1 + PF_KC * c) >> 7, pw);
            const uint32_t word = cu == 0 ? pw[0] : (cu == 1 ? pw[1] : (cu == 2 ? pw[2] : pw[3]));
            keep = (word >> (16 * g + 8 * hf)) & 0xFFu;
        }
        return keep;
    };
    auto x_piece = [&](int c, uint32_t keep, const f32x4 (&xq)[NPARTS][2], const f32x4 (&mq)[2], int stage) {
        const int k0 = PF_KC * c + 16 * g + 8 * hf;
        float e[8];
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            f32x4 acc = *reinterpret_cast<const f32x4*>(bias_lds + k0 + 4 * ii);
#pragma unroll
            for (int q = 0; q < NPARTS; ++q) acc += xq[q][ii];
            const float v[4] = {acc.x, acc.y, acc.z, acc.w};
            const float mk[4] = {mq[ii].x, mq[ii].y, mq[ii].z, mq[ii].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x = pf_elu(v[j]);
                if constexpr (MODE == 1) x *= mk[j];
                else if constexpr (MODE == 3) x = ((keep >> (4 * ii + j)) & 1u) ? x * 2.f : 0.f;
                e[4 * ii + j] = x;
            }
        }
        put_planes(stage, hf, e);
    };
    // an eighth of a chunk's matrix work of the block = a quarter of this wave's: k-step t, this wave's tiles 2 j, 2 j + 1
    auto mma_sub = [&](const uint8_t* stage, f32x16 (&acc)[4], const pf_bf16x8 (&bop)[2][3], int t, int j) {
        pf_bf16x8 fa[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p)
                fa[i][p] = *reinterpret_cast<const pf_bf16x8*>(stage + p * PF_PLANE +
                                                               pf_off(32 * (4 * hf + 2 * j + i) + b, 2 * g + t));
#define PF_TERM(PA_, PB_)                                                                     \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) acc[2 * j + i] =                            \
        __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][PA_], bop[t][PB_], acc[2 * j + i], 0, 0, 0);
        PF_TERM(1, 1)
        PF_TERM(2, 0)
        PF_TERM(0, 2)
        PF_TERM(1, 0)
        PF_TERM(0, 1)
        PF_TERM(0, 0)
#undef PF_TERM
    };

    // The two waves of a SIMD run the same program between the same barriers: left alone they do their matrix work together
    // (one pipe: it serialises) and their vector work together (one issue port), i.e. nothing overlaps (MI355X_MICROARCH.md,
    // 'Two waves per SIMD', item 9).  The halves therefore take every trip in OPPOSITE order -- half 0 multiplies first and
    // prepares the next chunk's operands afterwards, half 1 prepares first and multiplies afterwards -- so that one wave's
    // vector work sits beside the other's MFMAs.  Each half runs its own copy of the loops (HF is a compile-time constant in
    // `body`): a per-trip `if (hf)` left values merging at every trip and cost ~50 registers.
    auto body = [&](auto hf_tag) {
        constexpr int HF = decltype(hf_tag)::value;
#ifdef PF8_SWAP  // which half prepares operands first: waves 0-3 (PF8_SWAP) or waves 4-7
        constexpr bool VF = HF == 0;
#else
        constexpr bool VF = HF == 1;
#endif
#if PF8_PRIO == 1
        if constexpr (HF == 1) __builtin_amdgcn_s_setprio(1);
#endif
        f32x16 acc2[4];
#pragma unroll
        for (int T = 0; T < 4; ++T) acc2[T] = f32x16{0};
        auto mma_trip = [&](const uint8_t* st, f32x16 (&acc)[4], int stage) {
            pf_bf16x8 bop[2][3];
            get_planes(stage, bop);
            mma_sub(st, acc, bop, 0, 0);
            PF8_SUB_FENCE;
            mma_sub(st, acc, bop, 0, 1);
            PF8_SUB_FENCE;
            mma_sub(st, acc, bop, 1, 0);
            PF8_SUB_FENCE;
            mma_sub(st, acc, bop, 1, 1);
        };

        w_fetch(0);
        x_fetch(0, xs, xm);
        w_put(0);
        __syncthreads();  // the biases are in LDS
        x_piece(0, x_keep(0, 0), xs, xm, 0);
        if constexpr (VF) x_fetch(1, xs, xm);  // n2 >= 4
        __syncthreads();  // weight chunk 0 and both halves of operand chunk 0 are in LDS
        for (int c0 = 0; c0 < n2; c0 += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + u;
                const int c2 = c + 2 < n2 ? c + 2 : n2 - 1, c1 = c + 1 < n2 ? c + 1 : n2 - 1;
                const uint8_t* st = lds + (u & 1) * PF_CHUNK;
                PF8_STAMP();
                w_fetch(c + 1);  // chunk n2 is the first actor chunk: always exists
                if constexpr (!VF) {
                    x_fetch(c1, xs, xm);
                    mma_trip(st, acc2, u & 1);
                    __builtin_amdgcn_sched_barrier(0);
                    PF8_PRIO_VECTOR();
                    PF8_STAMP();
                    const uint32_t keep = x_keep(c1, (u + 1) & 3);
                    x_piece(c1, keep, xs, xm, (u + 1) & 1);  // the last trip's lands in a slot nobody reads
                    __builtin_amdgcn_sched_barrier(0);
                    PF8_STAMP();
                    w_put(c + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    PF8_PRIO_MATRIX();
                } else {
                    PF8_PRIO_VECTOR();
                    const uint32_t keep = x_keep(c1, (u + 1) & 3);
                    x_piece(c1, keep, xs, xm, (u + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
                    PF8_PRIO_MATRIX();
                    PF8_STAMP();
                    x_fetch(c2, xs, xm);
                    mma_trip(st, acc2, u & 1);
                    __builtin_amdgcn_sched_barrier(0);
                    PF8_PRIO_VECTOR();
                    PF8_STAMP();
                    w_put(c + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    PF8_PRIO_MATRIX();
                }
                __builtin_amdgcn_sched_barrier(0);
                PF8_STAMP();
                __syncthreads();
            }
        }

        // ---- layer 3: h2 tile T (owner: half T >> 2; this lane: n = 32 T + 8 (v >> 2) + 4 g + (v & 3)) is the B operand of
        // actor chunk T; the owner of tile T + 1 converts it while its partner multiplies, and multiplies afterwards
        f32x16 acc3[4];
#pragma unroll
        for (int T = 0; T < 4; ++T) acc3[T] = f32x16{0};
        uint32_t pw2[4] = {0u, 0u, 0u, 0u};
        auto h_word = [&](int T) -> uint32_t {
            uint32_t word = 0u;
            if constexpr (MODE == 3) {
                const uint64_t idx0 = (uint64_t)rowc * PF_H2 + 32 * T;
                if ((T & 3) == 0) philox_bits4(dr.seed, dr.offset2, idx0 >> 7, pw2);
                word = (T & 3) == 0 ? pw2[0] : ((T & 3) == 1 ? pw2[1] : ((T & 3) == 2 ? pw2[2] : pw2[3]));
            }
            return word;
        };
        auto h_piece = [&](int T, int h, uint32_t word, int stage) {  // acc2[T & 3] is this wave's copy of tile T
            float e[8];
#pragma unroll
            for (int vv = 0; vv < 8; ++vv) {
                const int v = 8 * h + vv;
                const int nl = 8 * (v >> 2) + 4 * g + (v & 3);
                float x = pf_elu(acc2[T & 3][v] + bias_lds[1024 + 32 * T + nl]);
                if constexpr (MODE == 1) x *= dr.mask2[(size_t)rowc * PF_H2 + 32 * T + nl];
                else if constexpr (MODE == 3) x = ((word >> nl) & 1u) ? x * 2.f : 0.f;
                e[vv] = x;
            }
            put_planes(stage, h, e);
        };
        if constexpr (HF == 0) {  // the junk the last hidden trip left in stage 0 was written before that trip's barrier
            const uint32_t word = h_word(0);
            h_piece(0, 0, word, 0);
            h_piece(0, 1, word, 0);
        }
        __syncthreads();
#pragma unroll
        for (int T = 0; T < 8; ++T) {
            const int q = n2 + T;  // n2 is even: weight stage q & 1 == T & 1
            const uint8_t* st = lds + (T & 1) * PF_CHUNK;
            if (T + 1 < 8) w_fetch(q + 1);
            if (T + 1 < 8 && ((T + 1) >> 2) == HF) {
                PF8_PRIO_VECTOR();
                const uint32_t word = h_word(T + 1);
                h_piece(T + 1, 0, word, (T + 1) & 1);
                h_piece(T + 1, 1, word, (T + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
                PF8_PRIO_MATRIX();
            }
            mma_trip(st, acc3, T & 1);
            __builtin_amdgcn_sched_barrier(0);
            if (T + 1 < 8) {
                PF8_PRIO_VECTOR();
                w_put(q + 1);
                __builtin_amdgcn_sched_barrier(0);
                PF8_PRIO_MATRIX();
            }
            __builtin_amdgcn_sched_barrier(0);
            if (T + 1 < 8) __syncthreads();
        }

        // ---- logits: this wave's action tiles Ta = 4 HF + i ----
        if (row < n_rows) {
            float* out = logits + (size_t)row * ld;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int vq = 0; vq < 4; ++vq) {
                    const int a0 = 32 * (4 * HF + i) + 8 * vq + 4 * g;
                    const f32x4 bias = *reinterpret_cast<const f32x4*>(ba + a0);
                    f32x4 o = {acc3[i][4 * vq] + bias.x, acc3[i][4 * vq + 1] + bias.y, acc3[i][4 * vq + 2] + bias.z,
                               acc3[i][4 * vq + 3] + bias.w};
                    if (a0 + 3 < ld) *reinterpret_cast<f32x4*>(out + a0) = o;
                }
        }
    };
    if (hf == 0) body(std::integral_constant<int, 0>{});
    else body(std::integral_constant<int, 1>{});
}

}  // namespace hw

using namespace hw;

#ifdef PF8_STAMPS
extern "C" int pf8_read_stamps(unsigned long long* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(hw::pf8_stamps), sizeof(unsigned long long) * 8 * 256);
}
#endif

extern "C" int64_t pf8_f32_bytes(int32_t hidden1) { return (int64_t)(hidden1 / PF_KC + PF_H2 / 32) * PF8_F32_CHUNK_FLOATS * 4; }

extern "C" int pf8_prepare_f32(const float* w2, int32_t hidden1, const float* wa, int32_t act_dim, void* out, void* stream) {
    hipLaunchKernelGGL(pf8_prep_f32_kernel, dim3((unsigned)(hidden1 / PF_KC + PF_H2 / 32)), dim3(256), 0, (hipStream_t)stream, w2,
                       hidden1, wa, act_dim, (float*)out);
    return (int)hipGetLastError();
}

// hw_pgpr_policy_fused's arguments + the fp32 images; state_history 1, Philox dropout (the instantiation the bench runs)
extern "C" int pf8_policy_fused(const float* const* slot_tables, const float* zero_row, const float* b1, int32_t hidden1,
                                const void* planes, const void* planes_f32, const float* b2, const float* ba_padded,
                                int32_t act_dim, const int32_t* paths, const int32_t* n_rows, int64_t max_rows, uint64_t seed,
                                uint64_t offset1, uint64_t offset2, float* logits, int64_t ld, void* stream) {
    PfTables t;
    t.user0 = slot_tables[0], t.cur_u = slot_tables[1], t.cur_e = slot_tables[2];
    t.last_u = slot_tables[3], t.last_e = slot_tables[4], t.last_r = slot_tables[5];
    t.old_u = t.old_e = t.old_r = nullptr;
    t.zero = zero_row;
    PfDrop dr{3, nullptr, nullptr, seed, offset1, offset2};
    const int64_t grid = (max_rows + PF_ROWS - 1) / PF_ROWS;
    hipLaunchKernelGGL((pgpr_policy_fused8_kernel<4, 3>), dim3((unsigned)grid), dim3(PF8_THREADS), 0, (hipStream_t)stream, t, b1,
                       hidden1, (const uint8_t*)planes, b2, ba_padded, act_dim, paths, n_rows, dr, logits, ld,
                       (const uint8_t*)planes_f32);
    return (int)hipGetLastError();
}
