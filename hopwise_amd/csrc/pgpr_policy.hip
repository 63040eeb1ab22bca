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

#include "hw_common.h"

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

// Register budget and the weight path.  The kernel runs one wave per SIMD and its accumulators alone are 256 registers per
// lane, so what the rest of the loop keeps live decides two things: whether the fragment reads get registers of their own
// (a first form staged the weight chunks through 48 registers per lane and kept two gather buffers; hipcc then recycled ONE
// register quad for most fragment reads -- read, wait a full LDS round trip, multiply, read ... -- and SQ_WAIT_INST_ANY was
// 43 % of the wave cycles), and how many registers per SIMD are left for OTHER kernels' waves: the beam search's action and
// selection kernels run beside this one on other streams, and a wave of theirs can only become resident on a SIMD whose
// register file has room for it (432 of 512 taken: none of the trim kernels fits; 376 taken: they do -- the three-stream search
// went from 4.25 to 3.9 ms with nothing else changed, profiles/r5/policy_kernel_experiments_r5.txt).  Hence:
//   * the weight chunks travel global -> LDS directly (global_load_lds_dwordx4: no vector registers, no ds_write pass);
//   * ONE gather buffer, refilled in halves: the gathers of chunk c + 2's half t are issued in trip c right behind piece t of
//     chunk c + 1's operand preparation (which empties that half).
// Rules followed for the DMA (cdna_hip_programming.md 5, 'Pipelining across barriers'): the LDS image is lane-linear per wave
// instruction; no __syncthreads() while a DMA is in flight (it would wait vmcnt(0)) -- a counted s_waitcnt vmcnt(N) that
// leaves the younger gathers in flight, lgkmcnt(0), raw s_barrier, all three as builtins the compiler's counter model can
// see; the DMAs of a trip are issued behind the trip's last use of a gathered register and land by its closing wait, so that
// no use of a loaded register ever has a DMA pending in front of it (hipcc answers that with vmcnt(0): trip_sync below).
// tools/pf_abl/policy_pipe.hip keeps the other forms tried (three stages, sixteen one-tile steps per trip, the whole-buffer
// refill) behind flags.

template <int NPARTS, int MODE>
__global__ __launch_bounds__(256, 1) void pgpr_policy_fused_kernel(PfTables tb, const float* __restrict__ b1, int H1,
                                                                   const uint8_t* __restrict__ planes,
                                                                   const float* __restrict__ b2,
                                                                   const float* __restrict__ ba, int act_dim,
                                                                   const int32_t* __restrict__ paths,
                                                                   const int32_t* __restrict__ n_rows_ptr, PfDrop dr,
                                                                   float* __restrict__ logits, int64_t ld) {
    // ONE __shared__ object: a second one beside an array that direct-to-LDS loads write makes hipcc wait vmcnt(0) before the
    // first ds_read of every k-step (cdna_hip_programming.md, 'Projection GEMM at M = 256' item 4(a))
    __shared__ __attribute__((aligned(16))) uint8_t lds[2 * PF_CHUNK + (1024 + PF_H2) * 4];
    // the two hidden biases: read per chunk inside the MFMA shadow, where a global load would have to be waited for with a
    // vmcnt that also drains the weight prefetch issued before it (vector-memory operations return in order)
    float* const bias_lds = reinterpret_cast<float*>(lds + 2 * PF_CHUNK);
    const int n_rows = n_rows_ptr[0];
    const int row0 = blockIdx.x * PF_ROWS;
    if (row0 >= n_rows) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, b = lane & 31, g = lane >> 5;
    const int row = row0 + wave * 32 + b;
    const int rowc = row < n_rows ? row : n_rows - 1;
    const int n2 = H1 / PF_KC;
    for (int i = tid; i < H1; i += 256) bias_lds[i] = b1[i];
    bias_lds[1024 + tid] = b2[tid];  // 256 threads, PF_H2 = 256

    // ---- weight chunk pipeline: global -> LDS stage (q & 1), no registers in between ----
    const int wave_u = wave_in_block();  // wave-uniform for the compiler too: the DMA's LDS base travels in M0
    auto w_glds = [&](int q) {  // chunk q -> stage q & 1: twelve 1-KB pieces per wave, destination = base + lane * 16
        typedef __attribute__((address_space(1))) const void gptr_t;
        typedef __attribute__((address_space(3))) void lptr_t;
        const uint8_t* src = planes + (size_t)q * PF_CHUNK + (size_t)tid * 16;
        uint8_t* dst = lds + (q & 1) * PF_CHUNK + wave_u * 1024;
#pragma unroll
        for (int i = 0; i < 12; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t*)(src + 4096 * i), (lptr_t*)(dst + 4096 * i), 16, 0, 0);
    };
    w_glds(0);  // first of all: the chunk travels while the path records are read and turned into row addresses

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

    // end of a trip: the DMAs of this trip have landed (counted wait: the N younger vector loads -- this trip's gathers -- stay
    // in flight), this wave's own LDS reads are done, then the block meets.
    // The wait is a builtin, not inline assembly, so that hipcc's own counter model sees it: an LDS DMA the compiler believes
    // pending makes it force vmcnt(0) at the next use of ANY loaded register (its "pending flat" rule -- a DMA counts on both
    // the vector-memory and the LDS counter), which is how the first form of this loop came to wait for all sixteen gathers of
    // a chunk a few MFMAs into the trip that uses them.  With the wait visible the uses get counted waits (vmcnt(14) .. (8)).
    auto trip_sync = [&](auto n_younger) {
        constexpr int n = decltype(n_younger)::value;
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt((n & 15) | (7 << 4) | (0 << 8) | ((n >> 4) << 14));  // vmcnt(n) lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // ---- x1 chunk of this lane: 16 values k = 32 c + 16 g + 0..15, as gathered sums (before ELU) ----
    // One gather buffer and two operand sets: while the matrix pipe works on chunk c, the same wave turns the gathered rows of
    // chunk c + 1 into MFMA operands (ELU, dropout, split: ~600 vector instructions against 96 MFMAs of 32 cycles -- with one
    // wave per SIMD nothing else can fill the MFMA shadow); the gathers of chunk c + 2 are issued in trip c, half by half, as
    // soon as that preparation has emptied a half (x_fetch_half below).
    f32x4 xs[NPARTS][4];
    f32x4 xm[4];  // explicit mask (mode 1)
#pragma unroll
    for (int i = 0; i < 4; ++i) xm[i] = f32x4{1.f, 1.f, 1.f, 1.f};
    auto x_fetch = [&](int c, f32x4 (&xq)[NPARTS][4], f32x4 (&mq)[4]) {
        const int k0 = PF_KC * c + 16 * g;
#pragma unroll
        for (int q = 0; q < NPARTS; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) xq[q][i] = *reinterpret_cast<const f32x4*>(src[q] + k0 + 4 * i);
        if constexpr (MODE == 1)
#pragma unroll
            for (int i = 0; i < 4; ++i) mq[i] = *reinterpret_cast<const f32x4*>(dr.mask1 + (size_t)rowc * H1 + k0 + 4 * i);
    };
    // the buffer is refilled in halves: the eight values k-step t reads are free as soon as piece t has been prepared, so the
    // gathers of chunk c + 2's half t go out right there -- 72 (first half: the trip's closing wait is for a DMA younger than
    // it) and ~94 (second half) MFMAs before their use, against ~45 for sixteen loads issued behind piece 1 (2.60 -> 2.43 ms on
    // 1 M rows, bit-identical; tools/pf_abl PF_HALFX)
    auto x_fetch_half = [&](int c, int h, f32x4 (&xq)[NPARTS][4], f32x4 (&mq)[4]) {
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
    // a quarter of a chunk's matrix work: k-step t, tiles T = 4 half .. 4 half + 3 (12 fragment reads, 24 MFMAs)
    auto mma_group = [&](const uint8_t* stage, f32x16 (&acc)[8], const pf_bf16x8 (&bop)[2][3], int t, int half) {
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
    f32x16 acc2[8];
#pragma unroll
    for (int T = 0; T < 8; ++T) acc2[T] = f32x16{0};

    x_fetch(0, xs, xm);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // chunk 0 and the biases are in LDS
    x_finish(0, 0, xs, xm, xb[0]);
    x_fetch(1, xs, xm);
    for (int c0 = 0; c0 < n2; c0 += 4) {  // n2 % 4 == 0; unrolled by four so that buffer parity and `fresh` are static
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + u;
            // no test around the gathers (a load that leaves its block through a phi is waited for at the end of that
            // block, i.e. in front of the MFMAs): the last trips re-read the last chunk and throw the result away
            const int c2 = c + 2 < n2 ? c + 2 : n2 - 1, c1 = c + 1 < n2 ? c + 1 : n2 - 1;
            {
                // The order wanted, pinned by scheduling barriers where a load would overwrite what a piece still reads: piece
                // t of chunk c + 1's operand preparation behind MFMA group t, the gathers of chunk c + 2's half t right behind
                // it, the DMAs of weight chunk c + 1 (the other stage: every wave left it at the barrier that ended the previous
                // trip) behind piece 1 -- behind the LAST read of a gathered register in the trip: a use behind a pending DMA
                // would be a vmcnt(0) (see trip_sync) -- and older than the second half's gathers, which stay in flight across
                // the closing wait.  Inside a region the compiler's own order is kept: sched_group_barrier patterns measured
                // slower (2.82 vs 2.51 ms).
                const uint8_t* st = lds + (u & 1) * PF_CHUNK;
                const uint32_t keep = x_keep(c1, (u + 1) & 3);
                mma_group(st, acc2, xb[u & 1], 0, 0);
                x_piece(c1, 0, keep, xs, xm, xb[(u + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                x_fetch_half(c2, 0, xs, xm);
                mma_group(st, acc2, xb[u & 1], 0, 1);
                x_piece(c1, 1, keep, xs, xm, xb[(u + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                w_glds(c + 1);  // chunk n2 is the first actor chunk: always exists
                x_fetch_half(c2, 1, xs, xm);
                mma_group(st, acc2, xb[u & 1], 1, 0);
                mma_group(st, acc2, xb[u & 1], 1, 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            trip_sync(std::integral_constant<int, NPARTS * 2 + (MODE == 1 ? 2 : 0)>{});
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
    auto h_make = [&](int T, pf_bf16x8 (&out)[2][3]) {
        const uint32_t word = h_word(T);
        h_piece(T, 0, word, out);
        h_piece(T, 1, word, out);
    };
    h_make(0, hb[0]);
#pragma unroll
    for (int T = 0; T < 8; ++T) {
        const int q = n2 + T;
        {  // one scheduling region per trip, as in the hidden-layer loop
            const uint8_t* st = lds + (q & 1) * PF_CHUNK;
            mma_group(st, acc3, hb[T & 1], 0, 0);
            if (T + 1 < 8) {
                w_glds(q + 1);
                const uint32_t word = h_word(T + 1);
                h_piece(T + 1, 0, word, hb[(T + 1) & 1]);
                mma_group(st, acc3, hb[T & 1], 0, 1);
                h_piece(T + 1, 1, word, hb[(T + 1) & 1]);
                mma_group(st, acc3, hb[T & 1], 1, 0);
                mma_group(st, acc3, hb[T & 1], 1, 1);
            } else {
                mma_group(st, acc3, hb[T & 1], 0, 1);
                mma_group(st, acc3, hb[T & 1], 1, 0);
                mma_group(st, acc3, hb[T & 1], 1, 1);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (T + 1 < 8) trip_sync(std::integral_constant<int, 0>{});
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
#define PF_BY_MODE(NP)                 \
    do {                               \
        if (drop_mode == 0) PF_GO(NP, 0);      \
        else if (drop_mode == 1) PF_GO(NP, 1); \
        else PF_GO(NP, 3);                     \
    } while (0)
    if (state_history == 0) PF_BY_MODE(2);
    else if (state_history == 1) PF_BY_MODE(4);
    else PF_BY_MODE(6);
#undef PF_BY_MODE
#undef PF_GO
    return HW_LAUNCH_RC();
}
