// EXPERIMENT (round 5), not part of the product build: the forms of the policy kernel's main loop behind flags.  The default
// (PF_GLDS: weight chunks by global_load_lds DMA with counted vmcnt + raw s_barrier, one gather buffer) was the product kernel
// of the first half of round 5; -DPF_HALFX=1 -DPF_BWAIT (gather buffer refilled in halves, the closing wait as builtins the
// compiler's counter model sees) IS the product kernel now (csrc/pgpr_policy.hip carries it with the switches resolved);
// PF_HALFX=2, PF_DMA_AUX, PF_ASMDMA (+ PF_NSTAGE=3) are the forms tried on top of it without gain
// (profiles/r5/policy_kernel_experiments_r5b.txt); -DPF_GLDS=0 is the round-4 form (weights staged through 48
// registers, two gather buffers), PF_SINGLE_X its one-buffer variant, PF_NSTAGE=3 DMAs two chunks ahead, PF_PIPE sixteen
// one-tile steps per trip with the fragments of the next step read a step ahead and the operand preparation in two-value
// slices; the PF_ABL_* switches take parts of the work out for timing.  Every form computes the same bits.  On 1 M synthetic
// rows they measure 2.38-2.47 ms against 2.51-2.56 for the round-4 form (tools/pf_abl/run.py); in the real 64/64/64 search
// the lone policy stage stays at 2.78-2.82 ms, and what the smaller register footprint buys is room for other kernels' waves
// beside it (three-stream search 4.25 -> 3.9 ms, tools/pf_abl/ab_streams.sh): DESIGN.md section 8 has the numbers.
// PGPR policy network of one beam-search hop in ONE launch (pgpr.py:175-187 + the state of :296-333):
//   first layer as a slot-table lookup (hw_pgpr_layer1_lookup's sum) -> ELU -> dropout -> [rows, H1] x W2^T on the bf16 matrix
//   pipes (three-way split = fp32-level error, gemm.hip) -> ELU -> dropout -> x Wa^T -> logits.
// The unfused pipeline writes and re-reads x1 [rows, 512] and x2 [rows, 256] (6 GB per 1 M beam rows) between its three
// launches; here neither ever leaves the registers of the wave that owns the row.
//
// Shape of the computation (everything transposed, so that a lane owns a beam row from the gather to the logits):
//   * block = 4 waves x 32 beam rows.  Lane (b, g) of a wave (b = lane & 31, g = lane >> 5) works for beam row b.
//   * layer 2, D2' [n, b] = W2 [n, k] . x1^T [k, b]:  the MFMA B operand of lane (b, g) at k-step t of chunk c is
//     x1[b][32 c + 16 g + 8 t + 0..7] -- over a chunk the lane needs 64 contiguous bytes of each slot-table row, the two
//     half-waves together one whole 128-B line per row.  x1 is produced in registers straight into MFMA operand layout:
//     no LDS, no shuffle.  The A operand (weights, shared by all rows) streams through LDS as pre-split, pre-swizzled bf16
//     plane images prepared once per weight version (hw_pgpr_policy_prepare).
//   * the accumulator of a 32x32 tile leaves lane (b, g) with h2[b][n], n = 32 T + 8 (v >> 2) + 4 g + (v & 3): exactly a
//     B operand of layer 3 (D3' [a, b] = Wa [a, n] . h2^T [n, b]) once the k-slots of Wa are permuted to that order --
//     which hw_pgpr_policy_prepare does.  Layer 2's output feeds layer 3 from the same registers.
//   * dropout (reference semantics: always on, p = 0.5): one Philox4x32-10 block yields 128 keep bits here (element idx of
//     the [rows, H] activation: block idx >> 7, word (idx >> 5) & 3, bit idx & 31) -- 6 Philox evaluations per beam row for
//     both layers; the unfused kernels' stream (hw_common.h:philox_bits, one block per 4 elements) costs 192 per row, about
//     as many cycles as the matrix products.  Mode 1 (explicit masks, the parity tests) and mode 0 (identity) as before.
#include <type_traits>

#include "../../hopwise_amd/csrc/hw_common.h"

namespace hw {

typedef __bf16 pf_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 pf_bf16x2 __attribute__((ext_vector_type(2)));
typedef float pf_f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t pf_u32x4 __attribute__((ext_vector_type(4)));  // HIP's uint4 is a struct the optimiser keeps in memory

constexpr int PF_ROWS = 128;                     // beam rows per block
constexpr int PF_H2 = 256;                       // second hidden width = rows of W2 = k of the actor layer
constexpr int PF_A = 256;                        // action slots (act_dim padded)
constexpr int PF_KC = 32;                        // k per weight chunk
constexpr int PF_PLANE = 256 * PF_KC * 2;        // [256 rows][32 k] bf16 = 16 KB
constexpr int PF_CHUNK = 3 * PF_PLANE;           // hi / mid / lo = 48 KB
constexpr int PF_PATH_INTS = HW_PGPR_ROW_INTS;
constexpr int PF_SELF_LOOP = -1, PF_T_USER = 0;

__device__ __forceinline__ void pf_split_pair(float x0, float x1, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    const pf_f32x2 v = {x0, x1};
    const pf_bf16x2 h = __builtin_convertvector(v, pf_bf16x2);
    const pf_f32x2 r1 = v - __builtin_convertvector(h, pf_f32x2);
    const pf_bf16x2 m = __builtin_convertvector(r1, pf_bf16x2);
    const pf_f32x2 r2 = r1 - __builtin_convertvector(m, pf_f32x2);
    const pf_bf16x2 l = __builtin_convertvector(r2, pf_bf16x2);
    hi = __builtin_bit_cast(uint32_t, h), mid = __builtin_bit_cast(uint32_t, m), lo = __builtin_bit_cast(uint32_t, l);
}

// byte offset of fragment slot `slot` (8 k = 16 B) of row r inside one plane image: 64-byte rows, the four slots XOR-swizzled
// by (r >> 2) so that the 16 lanes a ds_read_b128 serves together (16 consecutive rows, one logical slot) touch the 16
// different 16-byte slots of the 256-byte bank row (r & 3 picks the 64-byte quarter, (r >> 2) & 3 the slot inside it);
// swizzling by (r >> 1) measured SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.39
__device__ __host__ __forceinline__ int pf_off(int r, int slot) { return r * 64 + (((slot ^ (r >> 2)) & 3) << 4); }

// ---- weight preparation: fp32 -> three bf16 planes per chunk, already in LDS image order ---------------------------------
// chunk q < H1 / 32: W2 [256, H1], slot (2 g + t) of row n holds k = 32 q + 16 g + 8 t + 0..7
// chunk H1 / 32 + T: Wa [act_dim, 256] padded to 256 rows, slot (2 g + h) of row a holds n = 32 T + 16 h + 4 g + 0..3 and
//                    32 T + 16 h + 8 + 4 g + 0..3: the order in which lane (b, g) holds layer 2's output tile T
__global__ __launch_bounds__(256) void pgpr_policy_prep_kernel(const float* __restrict__ w2, int H1,
                                                               const float* __restrict__ wa, int act_dim,
                                                               uint8_t* __restrict__ planes) {
    const int q = blockIdx.x;        // chunk
    const int n2 = H1 / PF_KC;
    for (int e = threadIdx.x; e < 256 * 4 * 4; e += 256) {  // (row, slot, pair of k)
        const int r = e >> 4, slot = (e >> 2) & 3, pr = e & 3;
        const int g = slot >> 1, th = slot & 1;
        float x0, x1;
        if (q < n2) {
            const int k = PF_KC * q + 16 * g + 8 * th + 2 * pr;
            x0 = w2[(size_t)r * H1 + k], x1 = w2[(size_t)r * H1 + k + 1];
        } else {
            const int T = q - n2, j = 2 * pr;  // j = 0, 2: first four; 4, 6: second four
            const int k = 32 * T + 16 * th + 4 * g + (j < 4 ? j : 8 + (j - 4));
            x0 = r < act_dim ? wa[(size_t)r * PF_H2 + k] : 0.f;
            x1 = r < act_dim ? wa[(size_t)r * PF_H2 + k + 1] : 0.f;
        }
        uint32_t h, m, l;
        pf_split_pair(x0, x1, h, m, l);
        uint8_t* dst = planes + (size_t)q * PF_CHUNK + pf_off(r, slot) + 4 * pr;
        *reinterpret_cast<uint32_t*>(dst) = h;
        *reinterpret_cast<uint32_t*>(dst + PF_PLANE) = m;
        *reinterpret_cast<uint32_t*>(dst + 2 * PF_PLANE) = l;
    }
}

struct PfTables {
    const float* user0;
    const float *cur_u, *cur_e;
    const float *last_u, *last_e, *last_r;
    const float *old_u, *old_e, *old_r;
    const float* zero;  // H1 zeros: the row a missing slot contributes
};

struct PfDrop {
    int mode;                 // 0 identity, 1 explicit masks, 3 Philox, 128 keep bits per block
    const float* mask1;       // [rows, H1]
    const float* mask2;       // [rows, 256]
    uint64_t seed, offset1, offset2;
};

// ELU without a branch and without expm1f (a library call with divergent paths; a wave that also has to keep the matrix
// pipe fed cannot afford it per element): exp(x) - 1 through v_exp_f32, absolute error ~1e-7 -- the size of the fp32 rounding
// of the O(1) sums these values enter.
__device__ __forceinline__ float pf_elu(float x) {
    const float u = __builtin_amdgcn_exp2f(fminf(x, 0.f) * 1.44269504088896341f) - 1.f;
    return x > 0.f ? x : u;
}

#ifndef PF_HALFX
#define PF_HALFX 0
#endif
#ifndef PF_DMA_AUX  // cache policy bits of the weight DMAs (gfx940 family: 1 = sc0, 2 = nt, 16 = sc1)
#define PF_DMA_AUX 0
#endif
#ifndef PF_INTERLEAVE
#define PF_INTERLEAVE 1
#endif
// PF_GLDS: the weight chunks travel global -> LDS directly (global_load_lds_dwordx4: no vector registers, no ds_write pass)
// instead of through 48 registers per lane.  The kernel sits at the 256-VGPR cap with one wave per SIMD -- the compiler was
// recycling ONE register quad for most fragment reads (read, wait a full LDS round trip, multiply, read ...: the ISA showed
// it, and SQ_WAIT_INST_ANY was 43 % of the wave cycles) and parking addresses in accumulator registers.  With the weights
// out of the register file (and one gather buffer instead of two, PF_SINGLE_X) the fragments get registers of their own.
// Rules followed (cdna_hip_programming.md 5, 'Pipelining across barriers'): the LDS image is lane-linear per wave instruction;
// no __syncthreads() while a DMA is in flight (it would wait vmcnt(0)) -- a counted s_waitcnt vmcnt(N) that leaves the
// younger gathers in flight, lgkmcnt(0), raw s_barrier; the DMAs are issued BEFORE the gathers of the same trip, so that the
// vmcnt(0) hipcc puts in front of the gathers' first use never waits for a DMA younger than them.
#ifndef PF_GLDS
#define PF_GLDS 1
#endif
// PF_PIPE (with PF_GLDS): the matrix work of a trip as SIXTEEN steps of one accumulator tile each (k-step t, tile T: three
// fragment reads, six MFMAs on acc[T]) with the fragments of step s + 1 read at the top of step s into a second register
// triple -- a fragment is asked for one whole step (192 MFMA cycles) before its first use, so no MFMA waits for LDS.  The
// vector work of the trip (next chunk's operands: ELU, dropout, three-way split) is cut into eight two-value slices that sit
// in steps of their own behind the MFMAs.  sched_barrier(0) pins the step boundaries: hipcc otherwise sinks the reads to
// their first use.  Per accumulator the six terms of a k-step run in the same order as in the grouped form: same bits.
#ifndef PF_PIPE
#define PF_PIPE 0
#endif
#ifndef PF_XPS   // pairs of the next chunk's operands per step (1, 2, 4) and the step they start at
#define PF_XPS 2
#endif
#ifndef PF_X0
#define PF_X0 2
#endif
#if PF_PIPE
#undef PF_GLDS
#define PF_GLDS 1
#endif
#if PF_GLDS
#undef PF_SINGLE_X
#define PF_SINGLE_X 1
#endif

template <int NPARTS, int MODE>
__global__ __launch_bounds__(256, 1) void pgpr_policy_fused_kernel(PfTables tb, const float* __restrict__ b1, int H1,
                                                                   const uint8_t* __restrict__ planes,
                                                                   const float* __restrict__ b2,
                                                                   const float* __restrict__ ba, int act_dim,
                                                                   const int32_t* __restrict__ paths,
                                                                   const int32_t* __restrict__ n_rows_ptr, PfDrop dr,
                                                                   float* __restrict__ logits, int64_t ld) {
#ifndef PF_NSTAGE
#define PF_NSTAGE 2
#endif
    // PF_NSTAGE 3 (with PF_GLDS): the DMAs run TWO chunks ahead, so that the wait in front of a trip's barrier is for DMAs issued
    // a whole trip earlier -- nobody waits for memory there; 3 x 48 KB + biases = 149 KB of the CU's 160
    constexpr int PF_STAGES = PF_NSTAGE;
    static_assert(PF_STAGES == 2 || (PF_STAGES == 3 && PF_GLDS && !PF_PIPE), "three stages: the DMA form of the grouped trip");
#if defined(PF_ASMDMA)
    static_assert(PF_HALFX == 1 && PF_GLDS, "the invisible DMAs belong to the half-refill trip");
#endif
    // ONE __shared__ object: a second one beside an array that direct-to-LDS loads write makes hipcc wait vmcnt(0) before the
    // first ds_read of every k-step (cdna_hip_programming.md, 'Projection GEMM at M = 256' item 4(a))
    __shared__ __attribute__((aligned(16))) uint8_t lds[PF_STAGES * PF_CHUNK + (1024 + PF_H2) * 4];
    // the two hidden biases: read per chunk inside the MFMA shadow, where a global load would have to be waited for with a
    // vmcnt that also drains the weight prefetch issued before it (vector-memory operations return in order)
    float* const bias_lds = reinterpret_cast<float*>(lds + PF_STAGES * PF_CHUNK);
    const int n_rows = n_rows_ptr[0];
    const int row0 = blockIdx.x * PF_ROWS;
    if (row0 >= n_rows) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, b = lane & 31, g = lane >> 5;
    const int row = row0 + wave * 32 + b;
    const int rowc = row < n_rows ? row : n_rows - 1;
    const int n2 = H1 / PF_KC;
    for (int i = tid; i < H1; i += 256) bias_lds[i] = b1[i];
    bias_lds[1024 + tid] = b2[tid];  // 256 threads, PF_H2 = 256

    // ---- slot-table rows of this lane's beam row (pgpr_layer1_lookup_kernel's selection, per lane) ----
    const float* src[NPARTS];
    {
        // the whole record as three native 16-byte vectors, every field selected by value (v_cndmask): picking fields through
        // array references let the compiler fold the selection into a dependent load at a computed address inside the
        // record -- a second memory round trip per row, and the NPARTS = 2 instantiation of that form faulted on four-node
        // paths (entity id into the user table)
        typedef int pf_i32x4 __attribute__((ext_vector_type(4)));
        const pf_i32x4* pr = reinterpret_cast<const pf_i32x4*>(paths + (size_t)rowc * PF_PATH_INTS);
        const pf_i32x4 q0 = pr[0], q1 = pr[1], q2 = pr[2];
        const int uid = q0.x, len = q0.y;
        // hop j = 1..3: relation, node type, node id
        const int r1 = q0.z, t1 = q0.w, d1 = q1.x, r2 = q1.y, t2 = q1.z, d2 = q1.w, r3 = q2.x, t3 = q2.y, d3 = q2.z;
        auto sel = [](int j, int v1, int v2, int v3) { return j == 1 ? v1 : (j == 2 ? v2 : v3); };
        auto node_row = [&](int j, const float* tu, const float* te) -> const float* {  // j >= 0
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
        const int lc = len < 1 ? 1 : (len > 4 ? 4 : len);  // a record always holds 1..4 nodes; clamp against garbage rows
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
    }

#ifdef PF_ABL_X_L1  // timing experiment: every gather reads the same (L1-resident) row
#pragma unroll
    for (int q = 0; q < NPARTS; ++q) src[q] = tb.zero;
#endif
#ifdef PF_ABL_CUR_L1  // timing experiments: which of the gathered parts costs what the all-L1 form saves
    src[1] = tb.zero;
#endif
#ifdef PF_ABL_CUR_L2  // the current node's row from a 512 KB window of its table: distinct rows per lane, L2-resident
    src[1] = tb.cur_e + (size_t)(rowc & 255) * H1;
#endif
#ifdef PF_ABL_USER_L1
    src[0] = tb.zero;
#endif
#ifdef PF_ABL_LAST_L1
    if constexpr (NPARTS > 2) src[2] = tb.zero;
#endif
#ifdef PF_ABL_REL_L1
    if constexpr (NPARTS > 2) src[3] = tb.zero;
#endif
#ifdef PF_ABL_SHARED_L1  // the parts siblings share (user, last node, last relation) from one row
    src[0] = tb.zero;
    if constexpr (NPARTS > 2) src[2] = tb.zero, src[3] = tb.zero;
#endif
    // ---- weight chunk pipeline: global -> registers -> LDS stage (q & 1) ----
#if PF_GLDS
    const int wave_u = wave_in_block();  // wave-uniform for the compiler too: the DMA's LDS base travels in M0
    auto w_glds = [&](int q) {  // chunk q -> stage q & 1: twelve 1-KB pieces per wave, destination = base + lane * 16
#ifdef PF_ABL_NO_W
        return;
#endif
        typedef __attribute__((address_space(1))) const void gptr_t;
        typedef __attribute__((address_space(3))) void lptr_t;
        const uint8_t* src = planes + (size_t)q * PF_CHUNK + (size_t)tid * 16;
        uint8_t* dst = lds + (q % PF_STAGES) * PF_CHUNK + wave_u * 1024;
#ifdef PF_ASMDMA
        // the DMAs as inline assembly: the compiler's counter model does not see them, so a use of a gathered register with a
        // DMA in flight gets a counted wait instead of the forced vmcnt(0).  Safe: an operation the model does not know only
        // ADDS to the outstanding count, so every wait the compiler computes waits for at least what it meant to; the DMAs'
        // own completion is waited for by the explicit counted waits of the loop (and, in order, by the waits for younger
        // gathers).
        const uint32_t dst_u = (uint32_t)(uintptr_t)dst;
#pragma unroll
        for (int i = 0; i < 12; ++i)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst_u + 4096u * i),
                         "v"(src + 4096 * i)
                         : "memory", "m0");
#else
#pragma unroll
        for (int i = 0; i < 12; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t*)(src + 4096 * i), (lptr_t*)(dst + 4096 * i), 16, 0, PF_DMA_AUX);
#endif
    };
    auto w_fetch = [&](int) {};
    auto w_put = [&](int) {};
#else
    pf_u32x4 wreg[12];
    auto w_fetch = [&](int q) {
#ifdef PF_ABL_NO_W
        return;
#endif
        const pf_u32x4* s = reinterpret_cast<const pf_u32x4*>(planes + (size_t)q * PF_CHUNK);
#pragma unroll
        for (int i = 0; i < 12; ++i) wreg[i] = s[tid + 256 * i];
    };
    auto w_put = [&](int q) {
#ifdef PF_ABL_NO_W
        return;
#endif
        pf_u32x4* d = reinterpret_cast<pf_u32x4*>(lds + (q & 1) * PF_CHUNK);
#pragma unroll
        for (int i = 0; i < 12; ++i) d[tid + 256 * i] = wreg[i];
    };
    auto w_glds = [&](int) {};
#endif
    // end of a trip: the DMAs of this trip have landed (counted wait: the N younger vector loads -- this trip's gathers -- stay
    // in flight), this wave's own LDS reads are done, then the block meets.  Register staging: a plain __syncthreads().
    auto trip_sync = [&](auto n_younger) {
#ifdef PF_ABL_NO_SYNC
        return;
#endif
#if PF_GLDS && defined(PF_BWAIT)
        // the wait as an instruction the compiler can see (hipcc's counter model reads pre-existing s_waitcnt): an LDS DMA it
        // believes pending makes it force vmcnt(0) at the next use of ANY loaded register ("pending flat" rule)
        constexpr int n = decltype(n_younger)::value;
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt((n & 15) | (7 << 4) | (0 << 8) | ((n >> 4) << 14));
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#elif PF_GLDS
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(decltype(n_younger)::value) : "memory");
#else
        __syncthreads();
#endif
    };

    // ---- x1 chunk of this lane: 16 values k = 32 c + 16 g + 0..15, as gathered sums (before ELU) ----
    // Two gather buffers and two operand sets: while the matrix pipe works on chunk c, the same wave turns the gathered rows
    // of chunk c + 1 into MFMA operands (ELU, dropout, split: ~600 vector instructions against 96 MFMAs of 32 cycles -- with
    // one wave per SIMD nothing else can fill the MFMA shadow) and the gathers of chunk c + 2 are in flight.
#ifndef PF_SINGLE_X
#define PF_SINGLE_X 0
#endif
    // PF_SINGLE_X: ONE gather buffer instead of two (64 vector registers fewer at NPARTS = 4).  The gathers of chunk c + 2 are
    // then issued in the MIDDLE of trip c, right behind the last piece of chunk c + 1's finish (which empties the buffer), and
    // have the second half of the trip plus the first quarter of the next one (~3 000 cycles) to arrive.
    static_assert(!PF_SINGLE_X || PF_INTERLEAVE, "the single gather buffer is scheduled inside the interleaved trip");
    constexpr int PF_XB = PF_SINGLE_X ? 1 : 2;
    f32x4 xs[PF_XB][NPARTS][4];
    f32x4 xm[PF_XB][4];  // explicit mask (mode 1)
#pragma unroll
    for (int u = 0; u < PF_XB; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) xm[u][i] = f32x4{1.f, 1.f, 1.f, 1.f};
    auto x_fetch = [&](int c, f32x4 (&xq)[NPARTS][4], f32x4 (&mq)[4]) {
#ifdef PF_ABL_NO_X
        return;
#endif
        const int k0 = PF_KC * c + 16 * g;
#pragma unroll
        for (int q = 0; q < NPARTS; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#ifdef PF_NT_X
                xq[q][i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src[q] + k0 + 4 * i));
#else
                xq[q][i] = *reinterpret_cast<const f32x4*>(src[q] + k0 + 4 * i);
#endif
            }
        if constexpr (MODE == 1)
#pragma unroll
            for (int i = 0; i < 4; ++i) mq[i] = *reinterpret_cast<const f32x4*>(dr.mask1 + (size_t)rowc * H1 + k0 + 4 * i);
    };
    // PF_HALFX: the gather buffer refilled in halves -- the eight values k-step t of the next chunk reads are free as soon as
    // piece t has been prepared, so the gathers of chunk c + 2's half t are issued right there, a WHOLE trip before their use
    // (the one-piece form issues all sixteen loads after piece 1 and the compiler wants them back ~15 MFMAs into the next trip:
    // ~50 MFMAs = 0.7 us of cover, less than a loaded L2 round trip)
    auto x_fetch_half = [&](int c, int h, f32x4 (&xq)[NPARTS][4], f32x4 (&mq)[4]) {
#ifdef PF_ABL_NO_X
        return;
#endif
        const int k0 = PF_KC * c + 16 * g;
#pragma unroll
        for (int q = 0; q < NPARTS; ++q)
#pragma unroll
            for (int i = 2 * h; i < 2 * h + 2; ++i) xq[q][i] = *reinterpret_cast<const f32x4*>(src[q] + k0 + 4 * i);
        if constexpr (MODE == 1)
#pragma unroll
            for (int i = 2 * h; i < 2 * h + 2; ++i)
                mq[i] = *reinterpret_cast<const f32x4*>(dr.mask1 + (size_t)rowc * H1 + k0 + 4 * i);
    };
    uint32_t pw[4] = {0u, 0u, 0u, 0u};
    pf_bf16x8 xb[2][2][3];  // B operands: [set][k-step][plane]
    // `fresh`: chunk c starts a new 128-element Philox block of this row (c % 4 == 0: H1 % 128 == 0) -- known at compile time
    // in the unrolled loop, so the region stays one basic block
    // The finish of a chunk is cut in two pieces (k-step t = 0, 1: eight of the lane's sixteen values each) so that the main
    // loop can place them BETWEEN the four MFMA groups of the chunk being multiplied: with one wave per SIMD only the wave's
    // own instruction stream can fill the matrix pipe's shadow, and a finish issued as one block before (or after) the 96
    // MFMAs was time added, not hidden (matrix work alone 1.70 ms, the rest alone 1.71, together 2.73: tools/pf_abl).
    auto x_keep = [&](int c, int cu) -> uint32_t {
        // c: chunk (clamped, addresses only); cu = chunk & 3 as the unrolled loop knows it: word cu of the row's current
        // Philox block holds this chunk's 32 keep bits, this lane's 16 at bit 16 g; a new block every fourth chunk
        uint32_t keep = 0xFFFFu;
        if constexpr (MODE == 3) {
            if (cu == 0) philox_bits4(dr.seed, dr.offset1, ((uint64_t)rowc * H1 + PF_KC * c) >> 7, pw);
            const uint32_t word = cu == 0 ? pw[0] : (cu == 1 ? pw[1] : (cu == 2 ? pw[2] : pw[3]));
            keep = (word >> (16 * g)) & 0xFFFFu;
        }
        return keep;
    };
    auto x_piece = [&](int c, int t, uint32_t keep, const f32x4 (&xq)[NPARTS][4], const f32x4 (&mq)[4], pf_bf16x8 (&out)[2][3]) {
#ifdef PF_ABL_NO_X
        return;
#endif
        const int k0 = PF_KC * c + 16 * g;
        float e[8];
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int i = 2 * t + ii;
            f32x4 acc = *reinterpret_cast<const f32x4*>(bias_lds + k0 + 4 * i);
#pragma unroll
            for (int q = 0; q < NPARTS; ++q) acc += xq[q][i];
            const float v[4] = {acc.x, acc.y, acc.z, acc.w};
            const float mk[4] = {mq[i].x, mq[i].y, mq[i].z, mq[i].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x = pf_elu(v[j]);
                if constexpr (MODE == 1) x *= mk[j];
                else if constexpr (MODE == 3) x = ((keep >> (4 * i + j)) & 1u) ? x * 2.f : 0.f;
                e[4 * ii + j] = x;
            }
        }
        uint32_t h[4], m[4], l[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) pf_split_pair(e[2 * p], e[2 * p + 1], h[p], m[p], l[p]);
        out[t][0] = __builtin_bit_cast(pf_bf16x8, pf_u32x4{h[0], h[1], h[2], h[3]});
        out[t][1] = __builtin_bit_cast(pf_bf16x8, pf_u32x4{m[0], m[1], m[2], m[3]});
        out[t][2] = __builtin_bit_cast(pf_bf16x8, pf_u32x4{l[0], l[1], l[2], l[3]});
    };
    auto x_finish = [&](int c, int cu, const f32x4 (&xq)[NPARTS][4], const f32x4 (&mq)[4], pf_bf16x8 (&out)[2][3]) {
        const uint32_t keep = x_keep(c, cu);
        x_piece(c, 0, keep, xq, mq, out);
        x_piece(c, 1, keep, xq, mq, out);
    };
    // two of a piece's eight values (pair p of k-step t: float4 i = 2 t + (p >> 1), elements 2 (p & 1), + 1) -> their split
    // (the chunk's sixteen bias values are read from LDS once, ahead of the steps: an LDS read inside a slice would be waited
    // for with lgkmcnt(0), i.e. together with the fragment reads issued just before it)
    auto x_pair = [&](const f32x4 (&bq)[4], int t, int p, uint32_t keep, const f32x4 (&xq)[NPARTS][4], const f32x4 (&mq)[4],
                      uint32_t& h, uint32_t& m, uint32_t& l) {
#ifdef PF_ABL_NO_X
        h = m = l = 0u;
        return;
#endif
        const int i = 2 * t + (p >> 1), j0 = 2 * (p & 1);
        pf_f32x2 acc = {j0 ? bq[i].z : bq[i].x, j0 ? bq[i].w : bq[i].y};
#pragma unroll
        for (int q = 0; q < NPARTS; ++q) acc += pf_f32x2{j0 ? xq[q][i].z : xq[q][i].x, j0 ? xq[q][i].w : xq[q][i].y};
        float e[2] = {acc.x, acc.y};
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            float x = pf_elu(e[jj]);
            if constexpr (MODE == 1) x *= (j0 + jj == 0 ? mq[i].x : (j0 + jj == 1 ? mq[i].y : (j0 + jj == 2 ? mq[i].z : mq[i].w)));
            else if constexpr (MODE == 3) x = ((keep >> (4 * i + j0 + jj)) & 1u) ? x * 2.f : 0.f;
            e[jj] = x;
        }
        pf_split_pair(e[0], e[1], h, m, l);
    };
    auto frag_load = [&](const uint8_t* stage, int t, int T, pf_bf16x8 (&f)[3]) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            f[pl] = *reinterpret_cast<const pf_bf16x8*>(stage + pl * PF_PLANE + pf_off(32 * T + b, 2 * g + t));
    };
    auto tile_mma = [&](f32x16& acc, const pf_bf16x8 (&f)[3], const pf_bf16x8 (&bo)[3]) {
#ifdef PF_ABL_NO_MMA
        acc[0] += __uint_as_float(__builtin_bit_cast(pf_u32x4, f[0]).x ^ __builtin_bit_cast(pf_u32x4, f[1]).x ^
                                  __builtin_bit_cast(pf_u32x4, f[2]).x ^ __builtin_bit_cast(pf_u32x4, bo[0]).x);
        return;
#endif
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[1], bo[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[2], bo[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[0], bo[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[1], bo[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[0], bo[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[0], bo[0], acc, 0, 0, 0);
    };
    // a quarter of a chunk's matrix work: k-step t, tiles T = 4 half .. 4 half + 3 (12 fragment reads, 24 MFMAs)
    auto mma_group = [&](const uint8_t* stage, f32x16 (&acc)[8], const pf_bf16x8 (&bop)[2][3], int t, int half) {
#ifdef PF_ABL_NO_MMA
        return;
#endif
        pf_bf16x8 fa[4][3];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p)
                fa[i][p] = *reinterpret_cast<const pf_bf16x8*>(stage + p * PF_PLANE + pf_off(32 * (4 * half + i) + b, 2 * g + t));
#define PF_TERM(PA_, PB_)                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) acc[4 * half + i] =                                              \
        __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][PA_], bop[t][PB_], acc[4 * half + i], 0, 0, 0);
        PF_TERM(1, 1)
        PF_TERM(2, 0)
        PF_TERM(0, 2)
        PF_TERM(1, 0)
        PF_TERM(0, 1)
        PF_TERM(0, 0)
#undef PF_TERM
    };
    // one chunk of matrix work: acc[T] += W[chunk] (rows 32 T ..) . B operands, T = 0..7, both k-steps
    auto mma_chunk = [&](const uint8_t* stage, f32x16 (&acc)[8], const pf_bf16x8 (&bop)[2][3]) {
#ifdef PF_ABL_NO_MMA  // timing experiment: one FMA per (tile, step) instead of six MFMAs, operands still read
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const pf_u32x4 w = *reinterpret_cast<const pf_u32x4*>(stage + pf_off(32 * i + b, 2 * g + t));
                acc[i][0] = fmaf(__uint_as_float(w.x), __uint_as_float(__builtin_bit_cast(pf_u32x4, bop[t][0]).x), acc[i][0]);
            }
        return;
#endif
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int half = 0; half < 2; ++half) mma_group(stage, acc, bop, t, half);
    };
    f32x16 acc2[8];
#pragma unroll
    for (int T = 0; T < 8; ++T) acc2[T] = f32x16{0};

    w_fetch(0);
    w_glds(0);
    if (PF_STAGES == 3) w_glds(1);
    x_fetch(0, xs[0], xm[0]);
#if !PF_SINGLE_X
    x_fetch(1, xs[1], xm[1]);  // n2 >= 4
#endif
    w_put(0);
#if PF_GLDS
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
    __syncthreads();  // weight chunk 0 and the biases are in LDS
#endif
    x_finish(0, 0, xs[0], xm[0], xb[0]);
#if PF_SINGLE_X
    x_fetch(1, xs[0], xm[0]);
#endif
    for (int c0 = 0; c0 < n2; c0 += 4) {  // n2 % 4 == 0; unrolled by four so that buffer parity and `fresh` are static
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + u;
            // no test around the gathers (a load that leaves its block through a phi is waited for at the end of that
            // block, i.e. in front of the MFMAs): the last trips re-read the last chunk and throw the result away
            const int c2 = c + 2 < n2 ? c + 2 : n2 - 1, c1 = c + 1 < n2 ? c + 1 : n2 - 1;
#if PF_PIPE
            {
                const uint8_t* st = lds + (u & 1) * PF_CHUNK;
                const uint32_t keep = x_keep(c1, (u + 1) & 3);
                pf_bf16x8 fr[2][3];
                uint32_t sh[4], sm[4], sl[4];
                f32x4 bq[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) bq[i] = *reinterpret_cast<const f32x4*>(bias_lds + PF_KC * c1 + 16 * g + 4 * i);
                frag_load(st, 0, 0, fr[0]);
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const int t = s >> 3, T = s & 7;
                    if (s + 1 < 16) frag_load(st, (s + 1) >> 3, (s + 1) & 7, fr[(s + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    tile_mma(acc2[T], fr[s & 1], xb[u & 1][t]);
                    if (s >= PF_X0 && s < PF_X0 + 8 / PF_XPS) {  // the next chunk's operands, 2 PF_XPS values per step
#pragma unroll
                        for (int e = 0; e < PF_XPS; ++e) {
                            const int pi = (s - PF_X0) * PF_XPS + e, pt = pi >> 2, pp = pi & 3;
                            x_pair(bq, pt, pp, keep, xs[0], xm[0], sh[pp], sm[pp], sl[pp]);
                            if (pp == 3) {
                                xb[(u + 1) & 1][pt][0] = __builtin_bit_cast(pf_bf16x8, pf_u32x4{sh[0], sh[1], sh[2], sh[3]});
                                xb[(u + 1) & 1][pt][1] = __builtin_bit_cast(pf_bf16x8, pf_u32x4{sm[0], sm[1], sm[2], sm[3]});
                                xb[(u + 1) & 1][pt][2] = __builtin_bit_cast(pf_bf16x8, pf_u32x4{sl[0], sl[1], sl[2], sl[3]});
                            }
                        }
                    }
                    // the DMAs before the gathers: older than them on the vector-memory counter; the gathers as soon as the
                    // pairs above have emptied the buffer -- they are wanted back at step PF_X0 of the next trip
                    if (s == PF_X0 + 8 / PF_XPS) w_glds(c + 1);
                    if (s == PF_X0 + 8 / PF_XPS) x_fetch(c2, xs[0], xm[0]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#elif PF_INTERLEAVE
            {
                // ONE scheduling region per trip, in the order wanted: the weight loads of chunk c + 1 and the gathers of
                // chunk c + 2 ride in the first MFMA groups, the finish of chunk c + 1 in the middle ones, the LDS stores
                // of the weights (the other stage: every wave left it at the barrier that ended the previous trip) in the last.
                // The compiler's own order inside the region is kept: sched_group_barrier patterns measured slower (2.82 vs 2.51 ms).
                const uint8_t* st = lds + (PF_STAGES == 3 ? c % 3 : (u & 1)) * PF_CHUNK;
                const uint32_t keep = x_keep(c1, (u + 1) & 3);
#if PF_HALFX
                // The DMAs stay behind the last LDS read of the biases in the trip: hipcc makes every LDS load that carries a
                // memory operand wait for ALL pending LDS DMAs (it cannot tell the stages from the bias rows), so a DMA issued
                // at the top of the trip was waited for, vmcnt(0), at the first bias read.
#if PF_HALFX == 2
                // both halves of chunk c + 1 waited for HERE, by an instruction the compiler sees: no use of a gathered register
                // is left behind the DMA below (a use behind a pending LDS DMA is forced to vmcnt(0)), so the DMA can go out a
                // quarter of a trip in: it gets 72 MFMAs to land, each half 72 (second) or 96 (first)
                mma_group(st, acc2, xb[u & 1], 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));  // vmcnt(0)
                x_piece(c1, 0, keep, xs[0], xm[0], xb[(u + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                w_glds(c + 1);
                x_fetch_half(c2, 0, xs[0], xm[0]);
                mma_group(st, acc2, xb[u & 1], 0, 1);
                x_piece(c1, 1, keep, xs[0], xm[0], xb[(u + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                x_fetch_half(c2, 1, xs[0], xm[0]);
                mma_group(st, acc2, xb[u & 1], 1, 0);
                mma_group(st, acc2, xb[u & 1], 1, 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            trip_sync(std::integral_constant<int, NPARTS * 4 + (MODE == 1 ? 4 : 0)>{});
#else
                mma_group(st, acc2, xb[u & 1], 0, 0);
                x_piece(c1, 0, keep, xs[0], xm[0], xb[(u + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);  // the gathers below overwrite what the piece above read
                x_fetch_half(c2, 0, xs[0], xm[0]);
                mma_group(st, acc2, xb[u & 1], 0, 1);
                x_piece(c1, 1, keep, xs[0], xm[0], xb[(u + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                w_glds(c + PF_STAGES - 1);
                x_fetch_half(c2, 1, xs[0], xm[0]);
                mma_group(st, acc2, xb[u & 1], 1, 0);
                mma_group(st, acc2, xb[u & 1], 1, 1);
            }
            __builtin_amdgcn_sched_barrier(0);
#if defined(PF_ASMDMA) && PF_NSTAGE == 3
            // three stages, DMAs two chunks ahead and invisible to the compiler: the DMA of chunk c + 1 (issued in trip c - 1)
            // is older than the second half of chunk c + 1's gathers, whose use in THIS trip has already waited it out (vector
            // memory returns in order) -- nothing to wait for here but the wave's own LDS reads
            trip_sync(std::integral_constant<int, 63>{});
#else
            trip_sync(std::integral_constant<int, NPARTS * 2 + (MODE == 1 ? 2 : 0)>{});
#endif
#endif
            continue;
            {
                const uint8_t* st = lds;
                const uint32_t keep = 0;
#endif
                mma_group(st, acc2, xb[u & 1], 0, 0);
                w_fetch(c + 1);  // chunk n2 is the first actor chunk: always exists
#if PF_SINGLE_X
                x_piece(c1, 0, keep, xs[0], xm[0], xb[(u + 1) & 1]);
                mma_group(st, acc2, xb[u & 1], 0, 1);
                x_piece(c1, 1, keep, xs[0], xm[0], xb[(u + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);  // the gathers below overwrite what the two pieces above read
                w_glds(c + PF_STAGES - 1);  // before the gathers: older than them on the vector-memory counter
                x_fetch(c2, xs[0], xm[0]);
#else
                x_fetch(c2, xs[u & 1], xm[u & 1]);  // two chunks ahead; the finish of the previous trip emptied this buffer
                x_piece(c1, 0, keep, xs[(u + 1) & 1], xm[(u + 1) & 1], xb[(u + 1) & 1]);
                mma_group(st, acc2, xb[u & 1], 0, 1);
                x_piece(c1, 1, keep, xs[(u + 1) & 1], xm[(u + 1) & 1], xb[(u + 1) & 1]);
#endif
                mma_group(st, acc2, xb[u & 1], 1, 0);
                w_put(c + 1);
                mma_group(st, acc2, xb[u & 1], 1, 1);
            }
            __builtin_amdgcn_sched_barrier(0);
#else
            w_fetch(c + 1);  // chunk n2 is the first actor chunk: always exists
            x_fetch(c2, xs[u & 1], xm[u & 1]);  // two chunks ahead; x_finish of the previous trip emptied this buffer
            __builtin_amdgcn_sched_barrier(0);
            x_finish(c1, (u + 1) & 3, xs[(u + 1) & 1], xm[(u + 1) & 1], xb[(u + 1) & 1]);
            mma_chunk(lds + (u & 1) * PF_CHUNK, acc2, xb[u & 1]);
            __builtin_amdgcn_sched_barrier(0);
            w_put(c + 1);
#endif
            trip_sync(std::integral_constant<int, NPARTS * 4 + (MODE == 1 ? 4 : 0) + (PF_STAGES == 3 ? 12 : 0)>{});
        }
    }

    // ---- layer 3: h2 tile T (this lane: n = 32 T + 8 (v >> 2) + 4 g + (v & 3)) is the B operand of actor chunk T ----
    f32x16 acc3[8];
#pragma unroll
    for (int T = 0; T < 8; ++T) acc3[T] = f32x16{0};
    uint32_t pw2[4] = {0u, 0u, 0u, 0u};
    pf_bf16x8 hb[2][2][3];
    // bias, ELU, dropout of the 16 values of tile T -> operand set; element (row, n): keep bit of idx = row * 256 + n
    auto h_word = [&](int T) -> uint32_t {
        uint32_t word = 0u;
        if constexpr (MODE == 3) {  // block (row * 256 + 32 T) >> 7 (new at T = 0, 4), word T & 3, bit n & 31
            const uint64_t idx0 = (uint64_t)rowc * PF_H2 + 32 * T;
            if ((T & 3) == 0) philox_bits4(dr.seed, dr.offset2, idx0 >> 7, pw2);
            word = (T & 3) == 0 ? pw2[0] : ((T & 3) == 1 ? pw2[1] : ((T & 3) == 2 ? pw2[2] : pw2[3]));
        }
        return word;
    };
    // half h of tile T's sixteen values (v = 8 h .. 8 h + 7) -> operand planes of k-step h
    auto h_piece = [&](int T, int h, uint32_t word, pf_bf16x8 (&out)[2][3]) {
        float e[8];
#pragma unroll
        for (int vv = 0; vv < 8; ++vv) {
            const int v = 8 * h + vv;
            const int nl = 8 * (v >> 2) + 4 * g + (v & 3);  // n - 32 T
            float x = pf_elu(acc2[T][v] + bias_lds[1024 + 32 * T + nl]);
            if constexpr (MODE == 1) x *= dr.mask2[(size_t)rowc * PF_H2 + 32 * T + nl];
            else if constexpr (MODE == 3) x = ((word >> nl) & 1u) ? x * 2.f : 0.f;
            e[vv] = x;
        }
        uint32_t hh[4], mm[4], ll[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) pf_split_pair(e[2 * p], e[2 * p + 1], hh[p], mm[p], ll[p]);
        out[h][0] = __builtin_bit_cast(pf_bf16x8, pf_u32x4{hh[0], hh[1], hh[2], hh[3]});
        out[h][1] = __builtin_bit_cast(pf_bf16x8, pf_u32x4{mm[0], mm[1], mm[2], mm[3]});
        out[h][2] = __builtin_bit_cast(pf_bf16x8, pf_u32x4{ll[0], ll[1], ll[2], ll[3]});
    };
    // two of a tile's sixteen values (pair p of half h: v = 8 h + 2 p, + 1) -> their split
    auto h_pair = [&](const float (&hbias)[16], int T, int h, int p, uint32_t word, uint32_t& hh, uint32_t& mm, uint32_t& ll) {
        float e[2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int v = 8 * h + 2 * p + jj;
            const int nl = 8 * (v >> 2) + 4 * g + (v & 3);
            float x = pf_elu(acc2[T][v] + hbias[v]);
            if constexpr (MODE == 1) x *= dr.mask2[(size_t)rowc * PF_H2 + 32 * T + nl];
            else if constexpr (MODE == 3) x = ((word >> nl) & 1u) ? x * 2.f : 0.f;
            e[jj] = x;
        }
        pf_split_pair(e[0], e[1], hh, mm, ll);
    };
    auto h_make = [&](int T, pf_bf16x8 (&out)[2][3]) {
        const uint32_t word = h_word(T);
        h_piece(T, 0, word, out);
        h_piece(T, 1, word, out);
    };
    h_make(0, hb[0]);
#pragma unroll
    for (int T = 0; T < 8; ++T) {
        const int q = n2 + T;
#if PF_PIPE
        {
            const uint8_t* st = lds + (q & 1) * PF_CHUNK;
            pf_bf16x8 fr[2][3];
            uint32_t sh[4], sm[4], sl[4];
            uint32_t word = 0u;
            float hbias[16];
            if (T + 1 < 8) {
#pragma unroll
                for (int vq = 0; vq < 4; ++vq) {  // v = 4 vq .. 4 vq + 3 are four consecutive n: one 16-byte read
                    const f32x4 x4 = *reinterpret_cast<const f32x4*>(bias_lds + 1024 + 32 * (T + 1) + 8 * vq + 4 * g);
                    hbias[4 * vq] = x4.x, hbias[4 * vq + 1] = x4.y, hbias[4 * vq + 2] = x4.z, hbias[4 * vq + 3] = x4.w;
                }
            }
            frag_load(st, 0, 0, fr[0]);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int t = s >> 3, Ta = s & 7;
                if (s + 1 < 16) frag_load(st, (s + 1) >> 3, (s + 1) & 7, fr[(s + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                tile_mma(acc3[Ta], fr[s & 1], hb[T & 1][t]);
                if (T + 1 < 8) {
                    if (s == 0) {
                        w_glds(q + 1);
                        word = h_word(T + 1);
                    }
                    if (s >= 1 && s <= 8) {
                        const int ph = (s - 1) >> 2, pp = (s - 1) & 3;
                        h_pair(hbias, T + 1, ph, pp, word, sh[pp], sm[pp], sl[pp]);
                        if (pp == 3) {
                            hb[(T + 1) & 1][ph][0] = __builtin_bit_cast(pf_bf16x8, pf_u32x4{sh[0], sh[1], sh[2], sh[3]});
                            hb[(T + 1) & 1][ph][1] = __builtin_bit_cast(pf_bf16x8, pf_u32x4{sm[0], sm[1], sm[2], sm[3]});
                            hb[(T + 1) & 1][ph][2] = __builtin_bit_cast(pf_bf16x8, pf_u32x4{sl[0], sl[1], sl[2], sl[3]});
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (T + 1 < 8) trip_sync(std::integral_constant<int, 0>{});
#elif PF_INTERLEAVE
        {  // one scheduling region per trip, as in the hidden-layer loop
            const uint8_t* st = lds + (q % PF_STAGES) * PF_CHUNK;
            mma_group(st, acc3, hb[T & 1], 0, 0);
            if (T + 1 < 8) {
                w_fetch(q + 1);
                if (T + PF_STAGES - 1 < 8) w_glds(q + PF_STAGES - 1);
                const uint32_t word = h_word(T + 1);
                h_piece(T + 1, 0, word, hb[(T + 1) & 1]);
                mma_group(st, acc3, hb[T & 1], 0, 1);
                h_piece(T + 1, 1, word, hb[(T + 1) & 1]);
                mma_group(st, acc3, hb[T & 1], 1, 0);
                w_put(q + 1);
                mma_group(st, acc3, hb[T & 1], 1, 1);
            } else {
                mma_group(st, acc3, hb[T & 1], 0, 1);
                mma_group(st, acc3, hb[T & 1], 1, 0);
                mma_group(st, acc3, hb[T & 1], 1, 1);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (T + 1 < 8) {
            if (PF_STAGES == 3 && T + 2 < 8) trip_sync(std::integral_constant<int, 12>{});
            else trip_sync(std::integral_constant<int, 0>{});
        }
#else
        if (T + 1 < 8) w_fetch(q + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (T + 1 < 8) h_make(T + 1, hb[(T + 1) & 1]);
        mma_chunk(lds + (q & 1) * PF_CHUNK, acc3, hb[T & 1]);
        __builtin_amdgcn_sched_barrier(0);
        if (T + 1 < 8) {
            w_put(q + 1);
#ifndef PF_ABL_NO_SYNC
            __syncthreads();
#endif
        }
#endif
    }

    // ---- logits: lane (b, g) holds actions a = 32 Ta + 8 (v >> 2) + 4 g + (v & 3) of its beam row ----
    if (row < n_rows) {
        float* out = logits + (size_t)row * ld;
#pragma unroll
        for (int Ta = 0; Ta < 8; ++Ta)
#pragma unroll
            for (int vq = 0; vq < 4; ++vq) {
                const int a0 = 32 * Ta + 8 * vq + 4 * g;
                const f32x4 bias = *reinterpret_cast<const f32x4*>(ba + a0);  // padded to 256 by the caller
                f32x4 o = {acc3[Ta][4 * vq] + bias.x, acc3[Ta][4 * vq + 1] + bias.y, acc3[Ta][4 * vq + 2] + bias.z,
                           acc3[Ta][4 * vq + 3] + bias.w};
                if (a0 + 3 < ld) *reinterpret_cast<f32x4*>(out + a0) = o;
            }
    }
}

}  // namespace hw

using namespace hw;

extern "C" int64_t hw_pgpr_policy_planes_bytes(int32_t hidden1) {
    if (hidden1 < 32 || hidden1 % 128) return 0;
    return (int64_t)(hidden1 / PF_KC + PF_H2 / 32) * PF_CHUNK;
}

extern "C" int hw_pgpr_policy_prepare(const float* w2, int32_t hidden1, int32_t hidden2, const float* wa, int32_t act_dim,
                                      void* planes, void* stream) {
    if (!w2 || !wa || !planes || hidden2 != PF_H2 || hidden1 < 128 || hidden1 % 128 || hidden1 > 1024 || act_dim < 1 ||
        act_dim > PF_A)
        return HW_ERR_UNSUPPORTED;
    HW_CLEAR_ERR();
    hipLaunchKernelGGL(pgpr_policy_prep_kernel, dim3((unsigned)(hidden1 / PF_KC + PF_H2 / 32)), dim3(256), 0,
                       (hipStream_t)stream, w2, hidden1, wa, act_dim, (uint8_t*)planes);
    return HW_LAUNCH_RC();
}

extern "C" int hw_pgpr_policy_fused(const float* const* slot_tables, const float* zero_row, const float* b1,
                                    int32_t hidden1, const void* planes, const float* b2, const float* ba_padded,
                                    int32_t act_dim, int32_t state_history, const int32_t* paths, const int32_t* n_rows,
                                    int64_t max_rows, int32_t drop_mode, const float* mask1, const float* mask2,
                                    uint64_t seed, uint64_t offset1, uint64_t offset2, float* logits, int64_t ld,
                                    void* stream) {
    if (!slot_tables || !zero_row || !b1 || !planes || !b2 || !ba_padded || !paths || !n_rows || !logits)
        return HW_ERR_INVALID_ARG;
    if (hidden1 < 128 || hidden1 % 128 || hidden1 > 1024 || act_dim < 1 || act_dim > PF_A) return HW_ERR_UNSUPPORTED;
    if (state_history < 0 || state_history > 2) return HW_ERR_INVALID_ARG;
    if (!(drop_mode == 0 || drop_mode == 1 || drop_mode == 3) || (drop_mode == 1 && (!mask1 || !mask2)))
        return HW_ERR_INVALID_ARG;
    if (drop_mode == 3 && ((offset1 | offset2) >> 63)) return HW_ERR_UNSUPPORTED;  // plain offsets only (evaluation path)
    if (ld < PF_A || (ld & 3) || (((uintptr_t)logits | (uintptr_t)b1 | (uintptr_t)ba_padded | (uintptr_t)planes |
                                   (uintptr_t)zero_row) & 15))
        return HW_ERR_INVALID_ARG;
    if (drop_mode == 1 && (((uintptr_t)mask1) & 15)) return HW_ERR_INVALID_ARG;
    const int need = state_history == 0 ? 3 : (state_history == 1 ? 6 : 9);
    for (int i = 0; i < need; ++i)
        if (!slot_tables[i] || (((uintptr_t)slot_tables[i]) & 15)) return HW_ERR_INVALID_ARG;
    if (max_rows <= 0) return HW_OK;
    PfTables t;
    t.user0 = slot_tables[0], t.cur_u = slot_tables[1], t.cur_e = slot_tables[2];
    t.last_u = need > 3 ? slot_tables[3] : nullptr, t.last_e = need > 3 ? slot_tables[4] : nullptr;
    t.last_r = need > 3 ? slot_tables[5] : nullptr;
    t.old_u = need > 6 ? slot_tables[6] : nullptr, t.old_e = need > 6 ? slot_tables[7] : nullptr;
    t.old_r = need > 6 ? slot_tables[8] : nullptr;
    t.zero = zero_row;
    PfDrop dr{drop_mode, mask1, mask2, seed, offset1, offset2};
    const int64_t grid = (max_rows + PF_ROWS - 1) / PF_ROWS;
    if (grid > INT32_MAX) return HW_ERR_INVALID_ARG;
    HW_CLEAR_ERR();
    hipStream_t st = (hipStream_t)stream;
#define PF_GO(NP, MD)                                                                                                   \
    hipLaunchKernelGGL((pgpr_policy_fused_kernel<NP, MD>), dim3((unsigned)grid), dim3(256), 0, st, t, b1, hidden1,      \
                       (const uint8_t*)planes, b2, ba_padded, act_dim, paths, n_rows, dr, logits, ld)
#ifdef PF_ONLY_43
#define PF_BY_MODE(NP) PF_GO(4, 3)
#else
#define PF_BY_MODE(NP)                 \
    do {                               \
        if (drop_mode == 0) PF_GO(NP, 0);      \
        else if (drop_mode == 1) PF_GO(NP, 1); \
        else PF_GO(NP, 3);                     \
    } while (0)
#endif
    if (state_history == 0) PF_BY_MODE(2);
    else if (state_history == 1) PF_BY_MODE(4);
    else PF_BY_MODE(6);
#undef PF_BY_MODE
#undef PF_GO
    return HW_LAUNCH_RC();
}
