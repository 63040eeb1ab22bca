// fp32 GEMM on the CDNA4 matrix cores.  Two main loops behind one epilogue: the three-way bf16 split on the bf16 pipes
// (default for 16-byte aligned operands: six v_mfma_f32_32x32x16_bf16 per 16 k, fp32-level error, "fp32 product on the bf16
// matrix pipes" below) and exact fp32 FMA chains (v_mfma_f32_32x32x2_f32; gfx950 has no TF32/xf32 path), which also serve
// unaligned operands and HW_GEMM_SPLIT=0.
//
//   C[M, N] = epilogue( op(A)[M, K] . op(B)[K, N] )
// A is either K-contiguous ([M, K] row-major) or M-contiguous ([K, M] row-major, i.e. A^T stored); B is either
// K-contiguous ([N, K] row-major, the nn.Linear weight layout) or N-contiguous ([K, N] row-major).  That covers
// x @ W^T, x @ W, dY^T @ X and dY @ W without materialising a transpose.
//
// Block = 256 threads = 2 x 2 waves, tile 128 x 128 x 16; each wave owns 64 x 64 = four 32x32 accumulators
// (64 accumulator registers).  FMA loop: LDS tiles are k-major ([k][m], [k][n], row stride 132 floats) so the MFMA operand
// read (lane l -> row l & 31, k = l >> 5) is a conflict-free ds_read_b32 and M/N-contiguous operands are staged with
// ds_write_b128.  Two blocks fit a CU, which hides one block's staging behind the other's MFMA burst.
// Epilogue: + bias[n], ELU, multiplicative dropout (explicit mask or Philox keep-1/2), all optional.
#include <type_traits>

#include "hw_common.h"
#include "rank_epilogue.h"

namespace hw {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 16, LDT = 132;

// Stage one 128 x 16 operand tile into LDS (k-major).  X(row, k): row in [0, rows), k in [0, K).
//   KC = true : X stored [rows, K] (K contiguous, leading dim ld)
//   KC = false: X stored [K, rows] (row index contiguous, leading dim ld)
template <bool KC>
__device__ __forceinline__ void stage_tile(const float* __restrict__ X, int64_t ld, int rows, int K, int r0, int k0,
                                           bool vec_ok, float (*T)[LDT]) {
    const int tid = threadIdx.x;
    if constexpr (KC) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = tid + 256 * i;  // float4 id: 128 rows x 4 k-quads
            const int r = f >> 2, kq = (f & 3) * 4;
            const int gr = r0 + r, gk = k0 + kq;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (gr < rows) {
                const float* __restrict__ p = X + (size_t)gr * ld + gk;
                if (vec_ok && gk + 3 < K) {
                    const f32x4 x = *reinterpret_cast<const f32x4*>(p);
                    v[0] = x.x, v[1] = x.y, v[2] = x.z, v[3] = x.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = (gk + j < K) ? p[j] : 0.f;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) T[kq + j][r] = v[j];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = tid + 256 * i;  // float4 id: 16 k x 32 row-quads
            const int k = f >> 5, rq = (f & 31) * 4;
            const int gk = k0 + k, gr = r0 + rq;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (gk < K) {
                const float* __restrict__ p = X + (size_t)gk * ld + gr;
                if (vec_ok && gr + 3 < rows) {
                    const f32x4 x = *reinterpret_cast<const f32x4*>(p);
                    v[0] = x.x, v[1] = x.y, v[2] = x.z, v[3] = x.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = (gr + j < rows) ? p[j] : 0.f;
                }
            }
            *reinterpret_cast<f32x4*>(&T[k][rq]) = f32x4{v[0], v[1], v[2], v[3]};
        }
    }
}

// ---- software-pipelined tile movement (16-byte aligned operands) -------------------------------------------------------
// fetch_tile brings this thread's 2 x 16 bytes of the NEXT k-tile into registers while the current one is multiplied,
// put_tile writes them to LDS afterwards.  The fetch is straight-line code: hipcc waits for a load at the end of the
// basic block that issued it when its value leaves the block through a phi, so any edge test around the loads would put
// s_waitcnt vmcnt(0) in front of the MFMAs and undo the overlap.  Edges are handled by clamping addresses instead (a
// 16-byte aligned operand has ld % 4 == 0 >= its inner extent, so the last partial float4 of a row stays inside the row's
// allocation) and by zeroing k >= K when the registers are stored; rows / columns past M / N only feed outputs that are
// never written.
template <bool KC>
__device__ __forceinline__ void fetch_tile(const float* __restrict__ X, int64_t ld, int rows, int K, int r0, int k0,
                                           f32x4 (&v)[2]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int f = tid + 256 * i;
        if constexpr (KC) {
            const int r = f >> 2, kq = (f & 3) * 4;
            const int gr = min(r0 + r, rows - 1), gk = min(k0 + kq, ((K + 3) & ~3) - 4);
            v[i] = *reinterpret_cast<const f32x4*>(X + (size_t)gr * ld + gk);
        } else {
            const int k = f >> 5, rq = (f & 31) * 4;
            const int gk = min(k0 + k, K - 1), gr = min(r0 + rq, ((rows + 3) & ~3) - 4);
            v[i] = *reinterpret_cast<const f32x4*>(X + (size_t)gk * ld + gr);
        }
    }
}

template <bool KC>
__device__ __forceinline__ void put_tile(const f32x4 (&v)[2], int K, int k0, float (*T)[LDT]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int f = tid + 256 * i;
        const float x[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
        if constexpr (KC) {
            const int r = f >> 2, kq = (f & 3) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) T[kq + j][r] = k0 + kq + j < K ? x[j] : 0.f;
        } else {
            const int k = f >> 5, rq = (f & 31) * 4;
            const bool live = k0 + k < K;
            *reinterpret_cast<f32x4*>(&T[k][rq]) =
                f32x4{live ? x[0] : 0.f, live ? x[1] : 0.f, live ? x[2] : 0.f, live ? x[3] : 0.f};
        }
    }
}

// ---- fp32 product on the bf16 matrix pipes: three-way split (SPLIT variant of gemm_f32_kernel) --------------------------
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 MFMA rate.  Every fp32 value is the exact sum of three bf16 values
// (x = hi + mid + lo: 3 x 8 significand bits, each taken round-to-nearest from what the previous ones left), bf16 x bf16
// products are exact in fp32 and the MFMA accumulates in fp32, so
//     a b  =  ah bh + (ah bm + am bh) + (ah bl + al bh + am bm)  +  O(2^-24 |a b|)
// -- six bf16 MFMAs (v_mfma_f32_32x32x16_bf16) per 16 k instead of eight fp32 ones per 2 k x 8: 2.67 x fewer matrix-pipe
// cycles, and an error per product (<= ~2^-23) of the size of the fp32 rounding of the sum it is added to (measured against
// float64 next to the FMA loop: tools/check_gemm_split.py).  HW_GEMM_SPLIT=0 keeps the exact-FMA loop.  Inf operands give NaN (inf - inf in
// the split), which the FMA loop would not: operands are embeddings and activations.
// LDS: per stage and operand three plane images [128 rows][16 k] bf16 (32-byte rows); the MFMA fragment of lane (r, h) is the
// 16 bytes k = 8 h .. 8 h + 7 of row r, stored at half h ^ ((r >> 3) & 1) so that the four 16-lane groups of a ds_read_b128
// touch 16 distinct 16-byte slots of the 256-byte bank row.  2 stages x 2 operands x 3 planes x 4 KB = 48 KB: 3 blocks per CU.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// A k-major operand (stored [K, rows]) keeps k-major plane images [16 k][128 rows] bf16 -- a thread's float4 (four rows at
// one k) is one 8-byte store per plane -- and its fragments come back through ds_read_b64_tr_b16, the transposing read: a
// 16-lane group fetches 4 k-rows x 16 columns and lane i receives column i's four k, so two reads (k = 8 h + 0..3, + 4..7)
// make the lane's 8-k fragment (bank placement: below).
typedef short s16x4 __attribute__((ext_vector_type(4)));
// A k-major image's k-rows are 256 bytes (one bank row), so the 64-byte chunks of k-row k are stored at chunk ^ (k & 3): the
// four k-rows of a transposing read then fall on disjoint 16-dword bank ranges, and an 8-byte piece stays whole.
constexpr int SPLIT_PLANE = 128 * 16 * 2, SPLIT_KM_ROW = 256;
template <bool KC>
struct SplitImg {
    static constexpr int PLANE = SPLIT_PLANE;
    static constexpr int OPERAND = 3 * PLANE;
};
__device__ __forceinline__ int split_km_off(int k, int col) { return k * SPLIT_KM_ROW + ((col * 2) ^ ((k & 3) << 6)); }

__device__ __forceinline__ void split_pair(float x0, float x1, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    const f32x2 v = {x0, x1};
    const bf16x2 h = __builtin_convertvector(v, bf16x2);
    const f32x2 r1 = v - __builtin_convertvector(h, f32x2);
    const bf16x2 m = __builtin_convertvector(r1, bf16x2);
    const f32x2 r2 = r1 - __builtin_convertvector(m, f32x2);
    const bf16x2 l = __builtin_convertvector(r2, bf16x2);
    hi = __builtin_bit_cast(uint32_t, h), mid = __builtin_bit_cast(uint32_t, m), lo = __builtin_bit_cast(uint32_t, l);
}

__device__ __forceinline__ int split_off(int r, int half) { return r * 32 + ((half ^ ((r >> 3) & 1)) << 4); }

// registers of fetch_tile<KC> -> the three plane images of one operand.  KC: rows f >> 2, four consecutive k at (f & 3) * 4;
// k-major: k = f >> 5, four consecutive rows at (f & 31) * 4.  FULL: the whole k-tile lies below K (no zeroing).
template <bool KC, bool FULL, bool TWO_PLANES = false>
__device__ __forceinline__ void put_piece_split(const f32x4& v, int i, int K, int k0, uint8_t* __restrict__ img) {
    const int tid = threadIdx.x;
    constexpr int PLANE = SplitImg<KC>::PLANE;
    const int f = tid + 256 * i;
    float x0, x1, x2, x3;
    uint8_t* dst;
    if constexpr (KC) {
        const int r = f >> 2, kq = (f & 3) * 4, kg = k0 + kq;
        x0 = FULL || kg < K ? v.x : 0.f, x1 = FULL || kg + 1 < K ? v.y : 0.f;
        x2 = FULL || kg + 2 < K ? v.z : 0.f, x3 = FULL || kg + 3 < K ? v.w : 0.f;
        dst = img + split_off(r, kq >> 3) + ((kq & 4) << 1);
    } else {
        const int k = f >> 5, rq = (f & 31) * 4;
        const bool live = FULL || k0 + k < K;
        x0 = live ? v.x : 0.f, x1 = live ? v.y : 0.f, x2 = live ? v.z : 0.f, x3 = live ? v.w : 0.f;
        dst = img + split_km_off(k, rq);
    }
    uint32_t h0, m0, l0, h1, m1, l1;
    split_pair(x0, x1, h0, m0, l0);
    split_pair(x2, x3, h1, m1, l1);
    *reinterpret_cast<uint2*>(dst) = uint2{h0, h1};
    *reinterpret_cast<uint2*>(dst + PLANE) = uint2{m0, m1};
    // the ranking product never reads the lo plane (the compiler drops the second residual with the store)
    if constexpr (!TWO_PLANES) *reinterpret_cast<uint2*>(dst + 2 * PLANE) = uint2{l0, l1};
}

template <bool KC, bool FULL>
__device__ __forceinline__ void put_tile_split(const f32x4 (&v)[2], int K, int k0, uint8_t* __restrict__ img) {
    put_piece_split<KC, FULL>(v[0], 0, K, k0, img);
    put_piece_split<KC, FULL>(v[1], 1, K, k0, img);
}

// the MFMA fragment (k = 8 h .. 8 h + 7 of tile row lane & 31, h = lane >> 5) of the 32-row tile that starts at `base`
template <bool KC>
__device__ __forceinline__ bf16x8 read_frag_split(const uint8_t* __restrict__ plane, int base, int lane) {
    if constexpr (KC) {
        return *reinterpret_cast<const bf16x8*>(plane + split_off(base + (lane & 31), lane >> 5));
    } else {
        // group g = lane >> 4 covers tile rows 16 (g & 1) .. + 15 at k = 8 (g >> 1) ..; lane 4 q + p of the group addresses
        // k-row q, columns 4 p .. 4 p + 3 of the block
        const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
        const uint8_t* a = plane + split_km_off(8 * (g >> 1) + q, base + 16 * (g & 1) + 4 * pp);  // k + 4: same k & 3
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 4 * SPLIT_KM_ROW));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, both);
    }
}

// ACT == 3: sigmoid + BCELoss epilogue (TuckER / ConvE 1-vs-all scores, tucker.py:83,124-125): the tile never leaves as
// scores -- each element becomes dLoss/dscore (what the backward GEMMs consume) and the block's loss terms are summed into
// loss_partial[linear block id]; target[c] = scale * flag[c] + add.
struct GemmBce {
    const uint8_t* flag;
    float scale, add, inv_numel;
    float* loss_partial;
    // ACT == 5 (ranking product, hw_gemm_rank), both optional:
    //   rank_mask      bit (row, col) set = the entry leaves as -inf (a user's history): [rows, rank_mask_ld] 32-bit words
    //   rank_tile_max  the maximum of every 64-column stretch of every (masked) row: [rows, rank_tile_ld] floats -- what
    //                  hw_topk_from_tile_max needs to find a row's best columns while reading a few of its stretches only
    const uint32_t* rank_mask;
    int64_t rank_mask_ld;
    float* rank_tile_max;
    int64_t rank_tile_ld;
    //   rank_tile_sum  (with rank_tile_max) sum over the stretch of exp(x - that maximum): the pieces of the row's log-sum-exp
    //                  (hw_gemm_f32_lse + hw_lse_rows: a vocabulary log-softmax without a pass over the logits)
    float* rank_tile_sum;
};

// Epilogue: one wave's 32-row x 64-column slab (accumulator tiles [ti][0], [ti][1]) leaves through a wave-private 8 KB LDS
// image so that the global stores are dwordx4 -- 4 rows x 256 contiguous bytes per instruction instead of 2 rows x 128 as
// dword stores straight from the accumulator layout (64 store instructions per lane and tile; the store tail was 0.19 ms of
// the score product's 0.85, tools/gemm_abl).  LDS executes a wave's instructions in order: no wait between the write and the
// read-back, none before the next slab overwrites the image.
__device__ __forceinline__ void store_slab_via_lds(float* __restrict__ slab, const float (&v)[2][16], float* __restrict__ crow,
                                                   int64_t ld, int lane) {
    const int c = lane & 31, g = lane >> 5;
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) slab[((rg & 3) + 8 * (rg >> 2) + 4 * g) * 64 + tj * 32 + c] = v[tj][rg];
    const int r = lane >> 4, q = (lane & 15) * 4;
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(&slab[(pass * 4 + r) * 64 + q]);
        *reinterpret_cast<f32x4*>(crow + (size_t)(pass * 4 + r) * ld + q) = x;
    }
}

#include "gemm_small.h"

template <bool A_KC, bool B_KC, int ACT, bool SPLIT = false>
__global__ __launch_bounds__(256, SPLIT ? 3 : 2) void gemm_f32_kernel(const float* __restrict__ A, int64_t lda,
                                                          const float* __restrict__ B, int64_t ldb,
                                                          float* __restrict__ C, int64_t ldc,
                                                          const int32_t* __restrict__ m_ptr, int m_max, int N, int K,
                                                          const float* __restrict__ bias,
                                                          const float* __restrict__ drop_mask, int drop_mode,
                                                          uint64_t seed, uint64_t offset, int vec_a, int vec_b,
                                                          int k_len, float* __restrict__ partial, int pipe,
                                                          GemmBce bce) {
    // two LDS stages: the pipelined loop writes tile k+1 into the stage nobody reads while tile k is multiplied, which
    // leaves one barrier per k-step (the unpipelined fallback uses stage 0 only)
    // (the SPLIT loop: two stages of six bf16 plane images, see above)
    constexpr int SPLIT_STAGE = SplitImg<A_KC>::OPERAND + SplitImg<B_KC>::OPERAND;
    __shared__ __attribute__((aligned(16))) float lds[SPLIT ? 2 * SPLIT_STAGE / 4 : 4 * BK * LDT];
    float (*As2)[BK][LDT] = reinterpret_cast<float (*)[BK][LDT]>(lds);
    float (*Bs2)[BK][LDT] = reinterpret_cast<float (*)[BK][LDT]>(lds + 2 * BK * LDT);
    float (*As)[LDT] = As2[0];
    float (*Bs)[LDT] = Bs2[0];
    const int M = m_ptr ? min(m_ptr[0], m_max) : m_max;
    // Block coordinates.  pipe bit 2: XCD-aware order -- workgroups go to the 8 XCDs round-robin by linear id, so the blocks
    // that read the SAME B tile (all row tiles of one column tile / one k-range) are renumbered onto ONE XCD: the tile then
    // fills one L2 instead of eight (gridDim.y, or gridDim.z with split-K, is a multiple of 8 when the bit is set).
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (pipe & 4) {
        const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const unsigned xcd = lin & 7u, j = lin >> 3;
        bx = j % gridDim.x;
        if (gridDim.z > 1) {
            by = (j / gridDim.x) % gridDim.y;
            bz = (j / (gridDim.x * gridDim.y)) * 8u + xcd;
        } else {
            by = (j / gridDim.x) * 8u + xcd;
            // pipe bits 8..15: `group` row tiles (not all) share a column tile at a time inside the XCD -- the blocks of one
            // column tile run in step and ask for the same operand bytes, so fewer of them per tile means more distinct tiles
            // in flight towards HBM (gemm_rank.hip: what paced that product's main loop)
            const unsigned group = (unsigned)(pipe >> 8) & 255u;
            if (group && group < gridDim.x) {
                const unsigned n_col = gridDim.y >> 3, per_sweep = group * n_col;
                const unsigned sweep = j / per_sweep, in = j % per_sweep;
                const unsigned g_here = min(group, gridDim.x - sweep * group);
                bx = sweep * group + in % g_here, by = (in / g_here) * 8u + xcd;
            }
        }
    }
    const int m0 = bx * BM, n0 = by * BN;
    if (m0 >= M || n0 >= N) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x16{0};

    // split-K: blockIdx.z owns k in [z * k_len, min(K, (z + 1) * k_len)); partial sums go to `partial[z]`
    const int k_begin = bz * k_len;
    const int k_end = min(K, k_begin + k_len);
    if constexpr (SPLIT) {
        constexpr int PA = SplitImg<A_KC>::PLANE, PB = SplitImg<B_KC>::PLANE, OA = SplitImg<A_KC>::OPERAND;
        uint8_t* const img = reinterpret_cast<uint8_t*>(lds);
        // TWO k-tiles of register prefetch (r?[0] / r?[1] alternate): one k-step of MFMAs is 768 cycles, an operand tile that
        // misses L2 (the long-K backward products stream dS from HBM; the score product's table tiles come from the Infinity
        // Cache) is ~2 000 away.  Four tiles of look-ahead measured the same as two (tools/gemm_abl, round 3).
        f32x4 ra[2][2], rb[2][2];
        fetch_tile<A_KC>(A, lda, M, k_end, m0, k_begin, ra[0]);
        fetch_tile<B_KC>(B, ldb, N, k_end, n0, k_begin, rb[0]);
        fetch_tile<A_KC>(A, lda, M, k_end, m0, k_begin + BK, ra[1]);
        fetch_tile<B_KC>(B, ldb, N, k_end, n0, k_begin + BK, rb[1]);
        put_tile_split<A_KC, false>(ra[0], k_end, k_begin, img);
        put_tile_split<B_KC, false>(rb[0], k_end, k_begin, img + OA);
        __syncthreads();
        // one k-step: multiply LDS stage `cur`, store the tile held in (na, nb) = k0 + BK to the other stage, refill (fa_, fb_)
        // with k0 + 2 BK (clamped past the end, stored as zeros).  The conversion of the next tile is cut into its four
        // register pieces and placed BETWEEN the wave's own MFMA groups, and the scheduler is told to keep that order: with
        // the phases apart (24 MFMAs, then the conversion) the parts of the score product ADDED UP -- MFMA 0.40 + conversion
        // 0.23 + fragment reads 0.19 + output stores 0.28 ms of a 1.03 ms launch (tools/gemm_abl) -- although two other waves
        // share the SIMD; interleaved it runs 0.83 ms (the backward products 1.02 -> 0.79, 1.08 -> 0.83).  Measured and not kept:
        // fragment sets double-buffered across a mid-step barrier at two blocks per CU (0.98 / 0.83 / 0.88 ms), split-K with
        // interleaved instead of contiguous k-ranges, one-time start offsets between co-resident blocks (no change).
        auto kstep = [&](auto full_tag, int k0, int cur, f32x4 (&na)[2], f32x4 (&nb)[2], f32x4 (&fa_)[2], f32x4 (&fb_)[2]) {
            constexpr bool FULL = decltype(full_tag)::value;  // the next tile is a whole one: no zeroing
            // the loads of tile k0 + 2 BK: at the top of the step behind a scheduling barrier -- or, when both operands are
            // K-contiguous (the score products), after the first MFMA group inside the step's one region, with four
            // load slots in the group pattern below (measured: -2 % there, +2 % on the k-major layouts)
            constexpr bool FETCH_IN = A_KC && B_KC;
            auto fetch_next = [&]() {
#if !defined(GEMM_ABL_NO_FETCH) && !defined(GEMM_ABL_NO_FETCH_A)  // timing ablations (tools/gemm_abl): never in the product
                fetch_tile<A_KC>(A, lda, M, k_end, m0, k0 + 2 * BK, fa_);
#endif
#if !defined(GEMM_ABL_NO_FETCH) && !defined(GEMM_ABL_NO_FETCH_B)
                fetch_tile<B_KC>(B, ldb, N, k_end, n0, k0 + 2 * BK, fb_);
#endif
            };
            if constexpr (!FETCH_IN) {
                fetch_next();
                __builtin_amdgcn_sched_barrier(0);
            }
            const uint8_t* Ai = img + cur * SPLIT_STAGE;
            const uint8_t* Bi = Ai + OA;
            uint8_t* An = img + (cur ^ 1) * SPLIT_STAGE;
            // ACT == 5, the ranking product (hw_gemm_rank): the three largest terms only -- hi.hi, hi.mid, mid.hi -- for products
            // whose consumer re-scores its candidates exactly (hw_topk_rerank); the lo planes are not read.  Dropped: mid.mid,
            // hi.lo, lo.hi <= 3 x 2^-18 |a b| per product (round-to-nearest splits).
            constexpr bool LITE = ACT == 5;  // ACT == 6: all six terms, the same epilogue (hw_gemm_f32_masked)
            constexpr int NPLANES = LITE ? 2 : 3;
            bf16x8 fa[2][3], fb[2][3];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int p = 0; p < NPLANES; ++p) {
#ifdef GEMM_ABL_NO_READ
                    fa[t][p] = fb[t][p] = __builtin_bit_cast(bf16x8, f32x4{(float)lane, (float)t, (float)p, (float)cur});
#else
                    fa[t][p] = read_frag_split<A_KC>(Ai + p * PA, wm + 32 * t, lane);
                    fb[t][p] = read_frag_split<B_KC>(Bi + p * PB, wn + 32 * t, lane);
#endif
                }
            // smallest terms first; the four accumulators alternate so that no MFMA waits for the one before it
#ifdef GEMM_ABL_NO_MMA
#define HW_SPLIT_TERM(PA_, PB_)                                                                                      \
    _Pragma("unroll") for (int ti = 0; ti < 2; ++ti) _Pragma("unroll") for (int tj = 0; tj < 2; ++tj)               \
        asm volatile("" ::"v"(fa[ti][PA_]), "v"(fb[tj][PB_]));
#else
#define HW_SPLIT_TERM(PA_, PB_)                                                                                      \
    _Pragma("unroll") for (int ti = 0; ti < 2; ++ti) _Pragma("unroll") for (int tj = 0; tj < 2; ++tj)               \
        acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ti][PA_], fb[tj][PB_], acc[ti][tj], 0, 0, 0);
#endif
#ifdef GEMM_ABL_NO_PUT
#define HW_PIECE(KC_, V_, I_, OFF_) asm volatile("" ::"v"(V_), "v"(An));
#else
#define HW_PIECE(KC_, V_, I_, OFF_) put_piece_split<KC_, FULL, ACT == 5>(V_, I_, k_end, k0 + BK, An + OFF_);
#endif
            if constexpr (LITE) {
                HW_SPLIT_TERM(1, 0)
                if constexpr (FETCH_IN) fetch_next();
                HW_PIECE(A_KC, na[0], 0, 0)
                HW_PIECE(A_KC, na[1], 1, 0)
                HW_SPLIT_TERM(0, 1)
                HW_PIECE(B_KC, nb[0], 0, OA)
                HW_PIECE(B_KC, nb[1], 1, OA)
                HW_SPLIT_TERM(0, 0)
                // half the MFMAs carry the same conversion: eight vector instructions and an LDS store per MFMA
#pragma unroll
                for (int q = 0; q < 12; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                    if (FETCH_IN && q >= 1 && q <= 4) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
            } else {
                HW_SPLIT_TERM(1, 1)
                if constexpr (FETCH_IN) fetch_next();
                HW_PIECE(A_KC, na[0], 0, 0)
                HW_SPLIT_TERM(2, 0)
                HW_PIECE(A_KC, na[1], 1, 0)
                HW_SPLIT_TERM(0, 2)
                HW_PIECE(B_KC, nb[0], 0, OA)
                HW_SPLIT_TERM(1, 0)
                HW_PIECE(B_KC, nb[1], 1, OA)
                HW_SPLIT_TERM(0, 1)
                HW_SPLIT_TERM(0, 0)
                // the scheduler's order: per MFMA up to four vector instructions of the conversion, every other one an LDS store
#pragma unroll
                for (int q = 0; q < 24; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                    if (FETCH_IN && q >= 1 && q <= 4) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    if (q % 2 == 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#undef HW_SPLIT_TERM
#undef HW_PIECE
#ifndef GEMM_ABL_NO_SYNC
            __syncthreads();
#endif
        };
        const std::true_type whole{};
        const std::false_type ragged{};
        int k0 = k_begin;
        // two steps per trip (stages and register sets are fixed in each half); the trips whose two stored tiles are whole
        // ones run without a branch inside the loop, the last one or two steps take the zeroing form
        for (; k0 + 3 * BK <= k_end; k0 += 2 * BK) {
            kstep(whole, k0, 0, ra[1], rb[1], ra[0], rb[0]);
            kstep(whole, k0 + BK, 1, ra[0], rb[0], ra[1], rb[1]);
        }
        if (k0 + BK < k_end) {
            kstep(ragged, k0, 0, ra[1], rb[1], ra[0], rb[0]);
            kstep(ragged, k0 + BK, 1, ra[0], rb[0], ra[1], rb[1]);
            k0 += 2 * BK;
        }
        if (k0 < k_end) kstep(ragged, k0, 0, ra[1], rb[1], ra[0], rb[0]);
    } else if (pipe & 1) {
        f32x4 ra[2], rb[2];
        fetch_tile<A_KC>(A, lda, M, k_end, m0, k_begin, ra);
        fetch_tile<B_KC>(B, ldb, N, k_end, n0, k_begin, rb);
        put_tile<A_KC>(ra, k_end, k_begin, As2[0]);
        put_tile<B_KC>(rb, k_end, k_begin, Bs2[0]);
        __syncthreads();
        int cur = 0;
        for (int k0 = k_begin; k0 < k_end; k0 += BK, cur ^= 1) {
            // unconditional: past the last tile the clamped addresses re-read valid memory and put_tile stores zeros
            fetch_tile<A_KC>(A, lda, M, k_end, m0, k0 + BK, ra);
            fetch_tile<B_KC>(B, ldb, N, k_end, n0, k0 + BK, rb);
            __builtin_amdgcn_sched_barrier(0);  // or the scheduler sinks the loads below the MFMAs to shorten live ranges
            float (*Ac)[LDT] = As2[cur];
            float (*Bc)[LDT] = Bs2[cur];
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                const int kr = kk + (lane >> 5);
                const float a0 = Ac[kr][wm + (lane & 31)];
                const float a1 = Ac[kr][wm + 32 + (lane & 31)];
                const float b0 = Bc[kr][wn + (lane & 31)];
                const float b1 = Bc[kr][wn + 32 + (lane & 31)];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);  // ... and hoists put_tile's selects (first use of the loads) above them
            // the other stage was last read one k-step ago, and every wave has passed the barrier that ended that step
            put_tile<A_KC>(ra, k_end, k0 + BK, As2[cur ^ 1]);
            put_tile<B_KC>(rb, k_end, k0 + BK, Bs2[cur ^ 1]);
            __syncthreads();
        }
    } else
    for (int k0 = k_begin; k0 < k_end; k0 += BK) {
        stage_tile<A_KC>(A, lda, M, k_end, m0, k0, vec_a != 0, As);
        stage_tile<B_KC>(B, ldb, N, k_end, n0, k0, vec_b != 0, Bs);
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const int kr = kk + (lane >> 5);
            const float a0 = As[kr][wm + (lane & 31)];
            const float a1 = As[kr][wm + 32 + (lane & 31)];
            const float b0 = Bs[kr][wn + (lane & 31)];
            const float b1 = Bs[kr][wn + 32 + (lane & 31)];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
    // C/D map of a 32x32 tile: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    if constexpr (ACT == 3) {
        // 64 elements per lane: this epilogue is as many VALU cycles as the k-loop is MFMA cycles unless it is kept lean.  A
        // 16-element group whose |x| all lie below 16 (block-uniform interior tile, wave-uniform vote) takes the form without
        // BCELoss's clamps and without per-element predicates -- 3 transcendentals + ~12 VALU per element; any other group
        // keeps the general element with its bounds tests.
        float loss = 0.f;
        const bool interior = m0 + BM <= M && n0 + BN <= N;
#ifdef GEMM_NO_LDS_EPI
        const bool via_lds = false;
#else
        const bool via_lds = interior && (pipe & 2);
#endif
        float* const slab = lds + wave * 2048;  // every loop ended with a barrier: no wave still reads the tiles
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
            const int gm0 = m0 + wm + ti * 32 + 4 * (lane >> 5);
            float gv[2][16];
            bool fast[2];
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) {
                const int gn = n0 + wn + tj * 32 + (lane & 31);
                const bool col_ok = gn < N;
                const float bv = (bias && col_ok) ? bias[gn] : 0.f;
                const float t = (bce.flag && col_ok) ? fmaf(bce.scale, (float)bce.flag[gn], bce.add) : bce.add;
                float x[16], mx = 0.f;
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    x[rg] = acc[ti][tj][rg] + bv;
                    mx = fmaxf(mx, fabsf(x[rg]));
                }
                fast[tj] = interior && __builtin_amdgcn_ballot_w64(!(mx <= 16.f)) == 0;  // NaN fails the test: general element
                if (fast[tj]) {
#pragma unroll
                    for (int rg = 0; rg < 16; ++rg)
                        loss += bce_elem_unclamped(x[rg], 1.f - t, t * bce.inv_numel, bce.inv_numel, &gv[tj][rg]);
                } else {
#pragma unroll
                    for (int rg = 0; rg < 16; ++rg) {
                        const int gm = gm0 + (rg & 3) + 8 * (rg >> 2);
                        float pv;
                        const float l = bce_elem(x[rg], t, bce.inv_numel, &pv, &gv[tj][rg]);
                        if (gm < M && col_ok) loss += l;
                    }
                }
            }
            if (via_lds) {
                store_slab_via_lds(slab, gv, C + (size_t)(m0 + wm + ti * 32) * ldc + n0 + wn, ldc, lane);
            } else {
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) {
                    const int gn = n0 + wn + tj * 32 + (lane & 31);
#pragma unroll
                    for (int rg = 0; rg < 16; ++rg) {
                        const int gm = gm0 + (rg & 3) + 8 * (rg >> 2);
                        if (gm < M && gn < N) C[(size_t)gm * ldc + gn] = gv[tj][rg];
                    }
                }
            }
        }
        loss = group_sum<64>(loss);
        float* red = lds + 8192;  // past the four slabs
        if (lane == 0) red[wave] = loss;
        __syncthreads();
        if (tid == 0) bce.loss_partial[bx + gridDim.x * by] = (red[0] + red[1]) + (red[2] + red[3]);
        return;
    }
    if constexpr (ACT == 4) {
        // Distance epilogue (hw_gemm_dist): the tile holds <q_i, t_j>; out = base - sqrt(max(||q_i||^2 + ||t_j||^2 - 2 <q_i, t_j>, 0)).
        // Arguments reuse the epilogue's slots: bias = ||t||^2 per column, drop_mask = ||q||^2 per row, bce.scale = base,
        // bce.add = the cancellation threshold.  An entry whose d^2 falls below threshold x (||q||^2 + ||t||^2) -- where the
        // expansion has lost its digits: the query's nearest neighbours, a handful per row -- is recomputed by its lane as
        // sum_k (q_k - t_k)^2 from the two operand rows (four partial sums), so every score that leaves is fp32-accurate.
        const float* __restrict__ tn = bias;
        const float* __restrict__ qn = drop_mask;
        const float base = bce.scale, frac = bce.add;
        const bool interior = m0 + BM <= M && n0 + BN <= N;
        const bool via_lds = interior && (pipe & 2);
        float* const slab = lds + wave * 2048;  // every loop ended with a barrier: no wave still reads the tiles
        const bool vec_rows = vec_a && vec_b && (K & 3) == 0;
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
            const int gm0 = m0 + wm + ti * 32 + 4 * (lane >> 5);
            float qv[16];
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) qv[rg] = qn[min(gm0 + (rg & 3) + 8 * (rg >> 2), M - 1)];
            float v[2][16];
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) {
                const int gn = n0 + wn + tj * 32 + (lane & 31);
                const bool col_ok = gn < N;
                const float tv = col_ok ? tn[gn] : 0.f;
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    const int gm = gm0 + (rg & 3) + 8 * (rg >> 2);
                    const float sum = qv[rg] + tv;
                    float d2 = fmaf(-2.f, acc[ti][tj][rg], sum);
                    if (col_ok && gm < M && d2 < frac * sum) {
                        const float* __restrict__ qr = A + (size_t)gm * lda;
                        const float* __restrict__ tr = B + (size_t)gn * ldb;
                        float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
                        if (vec_rows) {
                            for (int kk = 0; kk < K; kk += 4) {
                                const f32x4 a4 = *reinterpret_cast<const f32x4*>(qr + kk);
                                const f32x4 b4 = *reinterpret_cast<const f32x4*>(tr + kk);
                                const float e0 = a4.x - b4.x, e1 = a4.y - b4.y, e2 = a4.z - b4.z, e3 = a4.w - b4.w;
                                p0 = fmaf(e0, e0, p0), p1 = fmaf(e1, e1, p1), p2 = fmaf(e2, e2, p2), p3 = fmaf(e3, e3, p3);
                            }
                        } else {
                            for (int kk = 0; kk < K; ++kk) {
                                const float e0 = qr[kk] - tr[kk];
                                p0 = fmaf(e0, e0, p0);
                            }
                        }
                        d2 = (p0 + p1) + (p2 + p3);
                    }
                    v[tj][rg] = base - __builtin_amdgcn_sqrtf(fmaxf(d2, 0.f));
                }
            }
            if (via_lds) {
                store_slab_via_lds(slab, v, C + (size_t)(m0 + wm + ti * 32) * ldc + n0 + wn, ldc, lane);
            } else {
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) {
                    const int gn = n0 + wn + tj * 32 + (lane & 31);
#pragma unroll
                    for (int rg = 0; rg < 16; ++rg) {
                        const int gm = gm0 + (rg & 3) + 8 * (rg >> 2);
                        if (gm < M && gn < N) C[(size_t)gm * ldc + gn] = v[tj][rg];
                    }
                }
            }
        }
        return;
    }
    if constexpr (ACT == 5 || ACT == 6) {
        // ranking product: + bias, the history mask and the per-stretch maxima on the way out.  A wave's slab is 32 rows x 64
        // columns = exactly one 64-column stretch of 32 rows (n0 + wn is a multiple of 64), its two 32-column halves one mask
        // word each: lane l of the wave fetches the word of (row l >> 1, half l & 1) -- one load for the slab -- and every lane
        // picks its rows' words out of the wave with readlane.
        const bool interior = m0 + BM <= M && n0 + BN <= N;
        const bool via_lds = interior && (pipe & 2);
        float* const slab = lds + wave * 2048;  // every loop ended with a barrier: no wave still reads the tiles
        const int g5 = lane >> 5;
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
            const int gm0 = m0 + wm + ti * 32;
            uint32_t mword = 0u;
            if (bce.rank_mask) {
                const int mr = gm0 + (lane >> 1);
                const int64_t wcol = ((int64_t)(n0 + wn) >> 5) + (lane & 1);
                if (mr < M && wcol < bce.rank_mask_ld) mword = bce.rank_mask[(size_t)mr * bce.rank_mask_ld + wcol];
            }
            float v[2][16];
            float rmax[16];
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) rmax[rg] = -INFINITY;
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) {
                const int gn = n0 + wn + tj * 32 + (lane & 31);
                const float bv = (bias && gn < N) ? bias[gn] : 0.f;
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    const int r0 = (rg & 3) + 8 * (rg >> 2);  // the row of the lanes with g5 == 0; g5 == 1: r0 + 4
                    const uint32_t wa = (uint32_t)__builtin_amdgcn_readlane((int)mword, 2 * r0 + tj);
                    const uint32_t wb = (uint32_t)__builtin_amdgcn_readlane((int)mword, 2 * (r0 + 4) + tj);
                    const uint32_t wsel = g5 ? wb : wa;
                    float x = acc[ti][tj][rg] + bv;
                    if ((wsel >> (lane & 31)) & 1u) x = -INFINITY;
                    v[tj][rg] = x;
                    rmax[rg] = fmaxf(rmax[rg], gn < N ? x : -INFINITY);
                }
            }
            if (bce.rank_tile_sum && n0 + wn < N) {  // (max, sum exp) of the stretch: the lane's two columns first
                float rsum[16];
                const bool in0 = n0 + wn + (lane & 31) < N, in1 = n0 + wn + 32 + (lane & 31) < N;
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    const float x0 = in0 ? v[0][rg] : -INFINITY, x1 = in1 ? v[1][rg] : -INFINITY;
                    // exp(larger - max) = 1, exp(smaller - max) = exp(-|x0 - x1|); a lone column: exp(-inf) = 0; none: 0
                    rsum[rg] = rmax[rg] == -INFINITY ? 0.f : 1.f + (x0 == x1 ? 1.f : __expf(-fabsf(x0 - x1)));
                }
                int rg;
                float mx, sm;
                half_wave_row_lse(rmax, rsum, lane, rg, mx, sm);  // rank_epilogue.h
                const int gm = gm0 + (rg & 3) + 8 * (rg >> 2) + 4 * g5;
                if ((lane & 1) == 0 && gm < M) {
                    const size_t at = (size_t)gm * bce.rank_tile_ld + ((n0 + wn) >> 6);
                    bce.rank_tile_max[at] = mx, bce.rank_tile_sum[at] = sm;
                }
            } else if (bce.rank_tile_max && n0 + wn < N) {  // a stretch that starts inside the row
                int rg;
                const float mx = half_wave_row_max(rmax, lane, rg);  // rank_epilogue.h
                const int gm = gm0 + (rg & 3) + 8 * (rg >> 2) + 4 * g5;
                if ((lane & 1) == 0 && gm < M) bce.rank_tile_max[(size_t)gm * bce.rank_tile_ld + ((n0 + wn) >> 6)] = mx;
            }
            if (via_lds) {
                store_slab_via_lds(slab, v, C + (size_t)gm0 * ldc + n0 + wn, ldc, lane);
            } else {
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) {
                    const int gn = n0 + wn + tj * 32 + (lane & 31);
#pragma unroll
                    for (int rg = 0; rg < 16; ++rg) {
                        const int gm = gm0 + (rg & 3) + 8 * (rg >> 2) + 4 * g5;
                        if (gm < M && gn < N) C[(size_t)gm * ldc + gn] = v[tj][rg];
                    }
                }
            }
        }
        return;
    }
#ifndef GEMM_ABL_NO_STORE
    if ((pipe & 2) && drop_mode == 0 && m0 + BM <= M && n0 + BN <= N) {  // interior tile, 16-byte rows: dwordx4 stores
        float* const slab = lds + wave * 2048;  // every loop ended with a barrier: no wave still reads the tiles
        float* const dst = partial ? partial + (size_t)bz * m_max * N : C;
        const int64_t ld = partial ? (int64_t)N : ldc;
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
            float v[2][16];
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) {
                const float bv = (bias && !partial) ? bias[n0 + wn + tj * 32 + (lane & 31)] : 0.f;
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    float x = acc[ti][tj][rg] + bv;
                    if (!partial) {
                        if (ACT == 1) x = x > 0.f ? x : expm1f(x);
                        if (ACT == 2) x = fmaxf(x, 0.f);
                    }
                    v[tj][rg] = x;
                }
            }
            store_slab_via_lds(slab, v, dst + (size_t)(m0 + wm + ti * 32) * ld + n0 + wn, ld, lane);
        }
        return;
    }
#endif
    if (drop_mode == 2) offset = philox_offset(offset);
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
            const int gn = n0 + wn + tj * 32 + (lane & 31);
            const float bv = (bias && gn < N) ? bias[gn] : 0.f;
            // Philox dropout: the four lanes of a quad hold four consecutive columns of one row, i.e. the four words of ONE
            // Philox block when N % 4 == 0.  Quad lane q evaluates the block of row (rg & 3) == q once and the keep bits
            // travel by DPP quad broadcast: one evaluation per four elements instead of four.
            const bool quad_drop = drop_mode == 2 && !partial && (N & 3) == 0;
            unsigned keep16 = 0u;  // bit rg: keep element rg of this lane
            if (quad_drop) {
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int gmq = m0 + wm + ti * 32 + (lane & 3) + 8 * r4 + 4 * (lane >> 5);
                    uint32_t w[4];
                    philox_bits4(seed, offset, ((uint64_t)gmq * N + (gn & ~3)) >> 2, w);
                    const int nib = (int)((w[0] & 1u) | ((w[1] & 1u) << 1) | ((w[2] & 1u) << 2) | ((w[3] & 1u) << 3));
                    const int b0 = __builtin_amdgcn_update_dpp(0, nib, 0x00, 0xF, 0xF, true);  // quad_perm: lane 0's
                    const int b1 = __builtin_amdgcn_update_dpp(0, nib, 0x55, 0xF, 0xF, true);  // lane 1's
                    const int b2 = __builtin_amdgcn_update_dpp(0, nib, 0xAA, 0xF, 0xF, true);
                    const int b3 = __builtin_amdgcn_update_dpp(0, nib, 0xFF, 0xF, 0xF, true);
                    const int sh = lane & 3;
                    keep16 |= (unsigned)((b0 >> sh) & 1) << (4 * r4 + 0);
                    keep16 |= (unsigned)((b1 >> sh) & 1) << (4 * r4 + 1);
                    keep16 |= (unsigned)((b2 >> sh) & 1) << (4 * r4 + 2);
                    keep16 |= (unsigned)((b3 >> sh) & 1) << (4 * r4 + 3);
                }
            }
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) {
                const int gm = m0 + wm + ti * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * (lane >> 5);
                if (gm < M && gn < N && partial) {
                    partial[((size_t)bz * m_max + gm) * N + gn] = acc[ti][tj][rg];
                } else if (gm < M && gn < N) {
                    float v = acc[ti][tj][rg] + bv;
                    if (ACT == 1) v = v > 0.f ? v : expm1f(v);
                    if (ACT == 2) v = fmaxf(v, 0.f);
                    if (drop_mode == 1) v *= drop_mask[(size_t)gm * N + gn];
                    else if (quad_drop) v = ((keep16 >> rg) & 1u) ? v * 2.f : 0.f;
                    else if (drop_mode == 2) v = (philox_bits(seed, offset, (uint64_t)gm * N + gn) & 1u) ? v * 2.f : 0.f;
#ifdef GEMM_ABL_NO_STORE
                    if (v == 123.456f)
#endif
                    C[(size_t)gm * ldc + gn] = v;
                }
            }
        }
}

// ---- wide tiles for the narrow-output products: 128 x 256 per block, 512 threads ------------------------------------------
// The TuckER / ConvE backward products (dX = dS . E, dE = dS^T . x) have N = d = 200 output columns.  With 128-wide tiles two
// blocks cover a row tile and BOTH stream the same dS tile; measured with everything but the matrix pipe and the global
// loads removed, those launches already ran at their full 0.79 ms (matrix pipe alone: 0.45) at ~12 bytes per cycle and CU of
// L1 fill -- the rate a CU sustains against far memory (MI355X_MICROARCH.md, 'prologue HBM burst').  One block per row tile
// with all 256 columns reads dS once (-25 % bytes per MFMA), converts a quarter less per MFMA and meets half as many
// barriers per MFMA.  Eight waves as 2 x 4, each 64 x 64 as above; LDS 2 x 36 KB; one block per CU (two waves per SIMD).
// Plain epilogue only (split-K partials or C); same k-step as gemm_f32_kernel's split loop.
#ifndef GEMM_WIDE_RING
#define GEMM_WIDE_RING 2
#endif
constexpr int WBN = 256, WNT = 512, W_PB = WBN * 32;
template <bool KC>
__device__ __forceinline__ void fetch_b256(const float* __restrict__ X, int64_t ld, int rows, int K, int r0, int k0,
                                           f32x4 (&v)[2]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int f = tid + WNT * i;
        if constexpr (KC) {
            const int r = f >> 2, kq = (f & 3) * 4;
            const int gr = min(r0 + r, rows - 1), gk = min(k0 + kq, ((K + 3) & ~3) - 4);
            v[i] = *reinterpret_cast<const f32x4*>(X + (size_t)gr * ld + gk);
        } else {
            const int k = f >> 6, rq = (f & 63) * 4;
            const int gk = min(k0 + k, K - 1), gr = min(r0 + rq, ((rows + 3) & ~3) - 4);
            v[i] = *reinterpret_cast<const f32x4*>(X + (size_t)gk * ld + gr);
        }
    }
}
// k-major image of a 256-row operand: 512-byte k-rows, 64-byte chunks at chunk ^ (k & 3) as in split_km_off
__device__ __forceinline__ int split_km_off256(int k, int col) { return k * 512 + ((col * 2) ^ ((k & 3) << 6)); }
template <bool KC, bool FULL>
__device__ __forceinline__ void put_piece_b256(const f32x4& v, int i, int K, int k0, uint8_t* __restrict__ img) {
    const int f = threadIdx.x + WNT * i;
    float x0, x1, x2, x3;
    uint8_t* dst;
    if constexpr (KC) {
        const int r = f >> 2, kq = (f & 3) * 4, kg = k0 + kq;
        x0 = FULL || kg < K ? v.x : 0.f, x1 = FULL || kg + 1 < K ? v.y : 0.f;
        x2 = FULL || kg + 2 < K ? v.z : 0.f, x3 = FULL || kg + 3 < K ? v.w : 0.f;
        dst = img + split_off(r, kq >> 3) + ((kq & 4) << 1);
    } else {
        const int k = f >> 6, rq = (f & 63) * 4;
        const bool live = FULL || k0 + k < K;
        x0 = live ? v.x : 0.f, x1 = live ? v.y : 0.f, x2 = live ? v.z : 0.f, x3 = live ? v.w : 0.f;
        dst = img + split_km_off256(k, rq);
    }
    uint32_t h0, m0, l0, h1, m1, l1;
    split_pair(x0, x1, h0, m0, l0);
    split_pair(x2, x3, h1, m1, l1);
    *reinterpret_cast<uint2*>(dst) = uint2{h0, h1};
    *reinterpret_cast<uint2*>(dst + W_PB) = uint2{m0, m1};
    *reinterpret_cast<uint2*>(dst + 2 * W_PB) = uint2{l0, l1};
}
template <bool KC>
__device__ __forceinline__ bf16x8 read_frag_b256(const uint8_t* __restrict__ plane, int base, int lane) {
    if constexpr (KC) {
        return *reinterpret_cast<const bf16x8*>(plane + split_off(base + (lane & 31), lane >> 5));
    } else {
        const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
        const uint8_t* a = plane + split_km_off256(8 * (g >> 1) + q, base + 16 * (g & 1) + 4 * pp);
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 4 * 512));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, both);
    }
}

// WM = 128: the tile above.  WM = 256 (K-contiguous A only): 256 x 256 per block, each wave 128 x 64 = eight accumulators --
// for the products whose row tiles are few and whose K is long (dX = dS . E: 8 row tiles x 32 split-K ranges = 256 blocks):
// L1 fill per MFMA another third lower (16 + 16 KB per 384 MFMAs against 8 + 16 KB per 192), four float4 converted per
// thread and 48 MFMAs instead of three and 24, 18 fragment reads per 48 MFMAs instead of 12 per 24.  LDS 2 x 48 KB.
template <bool A_KC, bool B_KC, int WM>
__global__ __launch_bounds__(512, 2) void gemm_split_wide_kernel(const float* __restrict__ A, int64_t lda,
                                                                 const float* __restrict__ B, int64_t ldb,
                                                                 float* __restrict__ C, int64_t ldc,
                                                                 const int32_t* __restrict__ m_ptr, int m_max, int N, int K,
                                                                 int k_len, float* __restrict__ partial, int flags) {
    static_assert(WM == 128 || (WM == 256 && A_KC), "256-row tiles: K-contiguous A");
    constexpr int TI = WM / 64, NA = WM / 128;                       // accumulator row tiles per wave, float4 of A per thread
    constexpr int PA = WM == 128 ? SplitImg<A_KC>::PLANE : W_PB, OA = 3 * PA, STAGE = OA + 3 * W_PB;
    __shared__ __attribute__((aligned(16))) uint8_t img[2 * STAGE];
    const int M = m_ptr ? min(m_ptr[0], m_max) : m_max;
    unsigned bx = blockIdx.x, bz = blockIdx.z;
    if (flags & 4) {  // XCD-aware order: the row tiles of one k-range on one XCD (gridDim.z is a multiple of 8)
        const unsigned lin = blockIdx.x + gridDim.x * blockIdx.z;
        bx = (lin >> 3) % gridDim.x;
        bz = ((lin >> 3) / gridDim.x) * 8u + (lin & 7u);
    }
    const int m0 = bx * WM;
    if (m0 >= M) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = (wave >> 2) * (WM / 2), wn = (wave & 3) * 64;
    f32x16 acc[TI][2];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x16{0};
    const int k_begin = bz * k_len;
    const int k_end = min(K, k_begin + k_len);
    // this thread's float4s of the A tile (128 rows: the 256-thread mapping of fetch_tile with f = tid)
    auto fetch_a = [&](int k0, f32x4 (&v)[NA]) {
        if constexpr (WM == 256) {
            fetch_b256<true>(A, lda, M, k_end, m0, k0, v);
        } else if constexpr (A_KC) {
            const int r = tid >> 2, kq = (tid & 3) * 4;
            const int gr = min(m0 + r, M - 1), gk = min(k0 + kq, ((k_end + 3) & ~3) - 4);
            v[0] = *reinterpret_cast<const f32x4*>(A + (size_t)gr * lda + gk);
        } else {
            const int k = tid >> 5, rq = (tid & 31) * 4;
            const int gk = min(k0 + k, k_end - 1), gr = min(m0 + rq, ((M + 3) & ~3) - 4);
            v[0] = *reinterpret_cast<const f32x4*>(A + (size_t)gk * lda + gr);
        }
    };
    auto put_a = [&](auto full_tag, const f32x4& v, int i, int kt, uint8_t* dst) {
        constexpr bool FULL = decltype(full_tag)::value;
        if constexpr (WM == 256) put_piece_b256<true, FULL>(v, i, k_end, kt, dst);
        else put_piece_split<A_KC, FULL>(v, i, k_end, kt, dst);
    };
    auto read_a = [&](const uint8_t* plane, int base) -> bf16x8 {
        if constexpr (WM == 256) return read_frag_b256<true>(plane, base, lane);
        else return read_frag_split<A_KC>(plane, base, lane);
    };
    // ring of WD register sets: the loads of tile j + WD are issued at the top of step j (WD = 4 measured the same as 2 on
    // both backward products: the loop is paced by the CU's fill rate, not by the latency of a miss)
    constexpr int WD = GEMM_WIDE_RING;
    static_assert(WD == 2 || WD == 4, "ring depth");
    const std::true_type whole{};
    const std::false_type ragged{};
    f32x4 ra[WD][NA], rb[WD][2];
#pragma unroll
    for (int d = 0; d < WD; ++d) {
        fetch_a(k_begin + d * BK, ra[d]);
        fetch_b256<B_KC>(B, ldb, N, k_end, 0, k_begin + d * BK, rb[d]);
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) put_a(ragged, ra[0][i], i, k_begin, img);
    put_piece_b256<B_KC, false>(rb[0][0], 0, k_end, k_begin, img + OA);
    put_piece_b256<B_KC, false>(rb[0][1], 1, k_end, k_begin, img + OA);
    __syncthreads();
    auto kstep = [&](auto full_tag, int k0, int cur, f32x4 (&na)[NA], f32x4 (&nb)[2], f32x4 (&fa_)[NA], f32x4 (&fb_)[2]) {
        constexpr bool FULL = decltype(full_tag)::value;
        fetch_a(k0 + WD * BK, fa_);
        fetch_b256<B_KC>(B, ldb, N, k_end, 0, k0 + WD * BK, fb_);
        __builtin_amdgcn_sched_barrier(0);
        const uint8_t* Ai = img + cur * STAGE;
        const uint8_t* Bi = Ai + OA;
        uint8_t* An = img + (cur ^ 1) * STAGE;
        bf16x8 fb[2][3];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int p = 0; p < 3; ++p) fb[t][p] = read_frag_b256<B_KC>(Bi + p * W_PB, wn + 32 * t, lane);
#define HW_WTERM(H_, PA_, PB_)                                                                                       \
    _Pragma("unroll") for (int ti = 0; ti < 2; ++ti) _Pragma("unroll") for (int tj = 0; tj < 2; ++tj)               \
        acc[2 * (H_) + ti][tj] =                                                                                     \
            __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ti][PA_], fb[tj][PB_], acc[2 * (H_) + ti][tj], 0, 0, 0);
        // pass h: accumulator row tiles 2 h, 2 h + 1 (their A fragments live only inside the pass)
#pragma unroll
        for (int h = 0; h < TI / 2; ++h) {
            bf16x8 fa[2][3];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int p = 0; p < 3; ++p) fa[t][p] = read_a(Ai + p * PA, wm + 64 * h + 32 * t);
            HW_WTERM(h, 1, 1)
            if (h == 0) put_a(full_tag, na[0], 0, k0 + BK, An);
            else put_piece_b256<B_KC, FULL>(nb[1], 1, k_end, k0 + BK, An + OA);
            HW_WTERM(h, 2, 0)
            if (h == 0 && NA == 2) put_a(full_tag, na[NA - 1], NA - 1, k0 + BK, An);
            if (h == 0 && NA == 1) put_piece_b256<B_KC, FULL>(nb[0], 0, k_end, k0 + BK, An + OA);
            HW_WTERM(h, 0, 2)
            if (h == 0 && NA == 2) put_piece_b256<B_KC, FULL>(nb[0], 0, k_end, k0 + BK, An + OA);
            if (h == 0 && NA == 1) put_piece_b256<B_KC, FULL>(nb[1], 1, k_end, k0 + BK, An + OA);
            HW_WTERM(h, 1, 0)
            HW_WTERM(h, 0, 1)
            HW_WTERM(h, 0, 0)
        }
#undef HW_WTERM
#pragma unroll
        for (int q = 0; q < 12 * TI; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            if (q % 2 == 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    };
    int k0 = k_begin;
    if constexpr (WD == 2) {
        for (; k0 + 3 * BK <= k_end; k0 += 2 * BK) {
            kstep(whole, k0, 0, ra[1], rb[1], ra[0], rb[0]);
            kstep(whole, k0 + BK, 1, ra[0], rb[0], ra[1], rb[1]);
        }
        if (k0 + BK < k_end) {
            kstep(ragged, k0, 0, ra[1], rb[1], ra[0], rb[0]);
            kstep(ragged, k0 + BK, 1, ra[0], rb[0], ra[1], rb[1]);
            k0 += 2 * BK;
        }
        if (k0 < k_end) kstep(ragged, k0, 0, ra[1], rb[1], ra[0], rb[0]);
    } else {
        for (; k0 + 5 * BK <= k_end; k0 += 4 * BK) {
            kstep(whole, k0, 0, ra[1], rb[1], ra[0], rb[0]);
            kstep(whole, k0 + BK, 1, ra[2], rb[2], ra[1], rb[1]);
            kstep(whole, k0 + 2 * BK, 0, ra[3], rb[3], ra[2], rb[2]);
            kstep(whole, k0 + 3 * BK, 1, ra[0], rb[0], ra[3], rb[3]);
        }
        for (; k0 < k_end; k0 += 4 * BK) {  // at most two trips: the last steps, in the zeroing form
            kstep(ragged, k0, 0, ra[1], rb[1], ra[0], rb[0]);
            if (k0 + BK < k_end) kstep(ragged, k0 + BK, 1, ra[2], rb[2], ra[1], rb[1]);
            if (k0 + 2 * BK < k_end) kstep(ragged, k0 + 2 * BK, 0, ra[3], rb[3], ra[2], rb[2]);
            if (k0 + 3 * BK < k_end) kstep(ragged, k0 + 3 * BK, 1, ra[0], rb[0], ra[3], rb[3]);
        }
    }
    // epilogue: split-K partial [z][m_max][N] or C; dwordx4 through the LDS transposition where the slab is whole and aligned
    float* const dst = partial ? partial + (size_t)bz * m_max * N : C;
    const int64_t ld = partial ? (int64_t)N : ldc;
    float* const slab = reinterpret_cast<float*>(img) + wave * 2048;
#pragma unroll
    for (int ti = 0; ti < TI; ++ti) {
        const int r0 = m0 + wm + ti * 32;
        if ((flags & 2) && r0 + 32 <= M && wn + 64 <= N) {
            float v[2][16];
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) v[tj][rg] = acc[ti][tj][rg];
            store_slab_via_lds(slab, v, dst + (size_t)r0 * ld + wn, ld, lane);
        } else {
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) {
                const int gn = wn + tj * 32 + (lane & 31);
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    const int gm = r0 + (rg & 3) + 8 * (rg >> 2) + 4 * (lane >> 5);
                    if (gm < M && gn < N) dst[(size_t)gm * ld + gn] = acc[ti][tj][rg];
                }
            }
        }
    }
}

// ---- narrow-N variant: 192 < N <= 208 ------------------------------------------------------------------------------------
// The TuckER / ConvE backward products have N = d = 200 output columns (dX = dS . E, dE = dS^T . x): 128-wide tiles pad
// that to 256 and spend 22 % of the MFMA passes on zeros.  This kernel covers the whole row with 13 tiles of
// v_mfma_f32_16x16x4_f32 (208 columns, 4 % padding; same FLOP rate as the 32x32x2 form): block = 512 threads = 8 waves,
// tile 128 x 208 x 16, wave w owns rows 16 w .. 16 w + 15 and all 13 column tiles (52 accumulator registers), so A is
// read once instead of once per 128-column block.  LDS tiles are k-major with row strides 144 and 208 floats (= 16 mod 64):
// the operand read of a 16x16x4 MFMA (lane l -> element l & 15 of k-row l >> 4) touches 64 distinct banks.  Same two-stage
// software pipeline as gemm_f32_kernel; plain epilogue only (no bias / activation / dropout), split-K through `partial`.
typedef float f32x4v __attribute__((ext_vector_type(4)));
constexpr int NB_M = 128, NB_N = 208, NB_LDA = 144, NB_LDB = 208, NB_TILES = 13;

template <bool KC, int ROWS, int LD>
__device__ __forceinline__ void nb_fetch(const float* __restrict__ X, int64_t ld, int rows, int K, int r0, int k0, int f,
                                         f32x4& v) {
    if constexpr (KC) {  // X [rows, K], K contiguous: float4 along k
        const int r = f >> 2, kq = (f & 3) * 4;
        const int gr = min(r0 + r, rows - 1), gk = min(k0 + kq, ((K + 3) & ~3) - 4);
        v = *reinterpret_cast<const f32x4*>(X + (size_t)gr * ld + gk);
    } else {  // X [K, rows], row index contiguous: float4 along rows
        constexpr int Q = ROWS / 4;
        const int k = f / Q, rq = (f % Q) * 4;
        const int gk = min(k0 + k, K - 1), gr = min(r0 + rq, ((rows + 3) & ~3) - 4);
        v = *reinterpret_cast<const f32x4*>(X + (size_t)gk * ld + gr);
    }
}

// SWZ: element (k, col) lives at column col ^ (((k >> 2) & 3) << 4).  With a row stride of 16 mod 64 the MFMA operand read
// (k-rows kk .. kk + 3, 16 columns) is conflict-free either way; the K-contiguous staging write (k-rows j, 4 + j, 8 + j,
// 12 + j of one 16-column block in one instruction) would hit the same 16 banks four times -- the swizzle sends the four
// k-groups to four different column blocks.  Needs the tile width to be a multiple of 64 columns (the A tile; not B's 208).
template <bool SWZ>
__device__ __forceinline__ int nb_col(int k, int col) {
    return SWZ ? col ^ (((k >> 2) & 3) << 4) : col;
}

template <bool KC, int ROWS, int LD, bool SWZ>
__device__ __forceinline__ void nb_put(const f32x4& v, int K, int k0, int f, float (*T)[LD]) {
    const float x[4] = {v.x, v.y, v.z, v.w};
    if constexpr (KC) {
        const int r = f >> 2, kq = (f & 3) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) T[kq + j][nb_col<SWZ>(kq + j, r)] = k0 + kq + j < K ? x[j] : 0.f;
    } else {
        constexpr int Q = ROWS / 4;
        const int k = f / Q, rq = (f % Q) * 4;
        const bool live = k0 + k < K;
        *reinterpret_cast<f32x4*>(&T[k][nb_col<SWZ>(k, rq)]) =
            f32x4{live ? x[0] : 0.f, live ? x[1] : 0.f, live ? x[2] : 0.f, live ? x[3] : 0.f};
    }
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(512, 2) void gemm_f32_n208_kernel(const float* __restrict__ A, int64_t lda,
                                                               const float* __restrict__ B, int64_t ldb,
                                                               float* __restrict__ C, int64_t ldc,
                                                               const int32_t* __restrict__ m_ptr, int m_max, int N, int K,
                                                               int k_len, float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float As2[2][BK][NB_LDA];
    __shared__ __attribute__((aligned(16))) float Bs2[2][BK][NB_LDB];
    const int M = m_ptr ? min(m_ptr[0], m_max) : m_max;
    const int m0 = blockIdx.x * NB_M;
    if (m0 >= M) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave * 16;
    f32x4v acc[NB_TILES];
#pragma unroll
    for (int t = 0; t < NB_TILES; ++t) acc[t] = f32x4v{0.f, 0.f, 0.f, 0.f};
    const int k_begin = blockIdx.z * k_len;
    const int k_end = min(K, k_begin + k_len);
    // B tile = 208 x 16 floats = 832 float4: threads 0..319 carry a second one (the others re-fetch slot 831: same data,
    // same LDS address -- branch-free so the loads stay in flight across the MFMAs)
    const int fb1 = min(tid + 512, NB_N * BK / 4 - 1);
    f32x4 ra, rb0, rb1;
    nb_fetch<A_KC, NB_M, NB_LDA>(A, lda, M, k_end, m0, k_begin, tid, ra);
    nb_fetch<B_KC, NB_N, NB_LDB>(B, ldb, N, k_end, 0, k_begin, tid, rb0);
    nb_fetch<B_KC, NB_N, NB_LDB>(B, ldb, N, k_end, 0, k_begin, fb1, rb1);
    nb_put<A_KC, NB_M, NB_LDA, true>(ra, k_end, k_begin, tid, As2[0]);
    nb_put<B_KC, NB_N, NB_LDB, false>(rb0, k_end, k_begin, tid, Bs2[0]);
    nb_put<B_KC, NB_N, NB_LDB, false>(rb1, k_end, k_begin, fb1, Bs2[0]);
    __syncthreads();
    int cur = 0;
    for (int k0 = k_begin; k0 < k_end; k0 += BK, cur ^= 1) {
        nb_fetch<A_KC, NB_M, NB_LDA>(A, lda, M, k_end, m0, k0 + BK, tid, ra);
        nb_fetch<B_KC, NB_N, NB_LDB>(B, ldb, N, k_end, 0, k0 + BK, tid, rb0);
        nb_fetch<B_KC, NB_N, NB_LDB>(B, ldb, N, k_end, 0, k0 + BK, fb1, rb1);
        __builtin_amdgcn_sched_barrier(0);
        float (*Ac)[NB_LDA] = As2[cur];
        float (*Bc)[NB_LDB] = Bs2[cur];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 4) {
            const int kr = kk + (lane >> 4);
            const float a = Ac[kr][nb_col<true>(kr, wm + (lane & 15))];
            float b[NB_TILES];
#pragma unroll
            for (int t = 0; t < NB_TILES; ++t) b[t] = Bc[kr][t * 16 + (lane & 15)];
#pragma unroll
            for (int t = 0; t < NB_TILES; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[t], acc[t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        nb_put<A_KC, NB_M, NB_LDA, true>(ra, k_end, k0 + BK, tid, As2[cur ^ 1]);
        nb_put<B_KC, NB_N, NB_LDB, false>(rb0, k_end, k0 + BK, tid, Bs2[cur ^ 1]);
        nb_put<B_KC, NB_N, NB_LDB, false>(rb1, k_end, k0 + BK, fb1, Bs2[cur ^ 1]);
        __syncthreads();
    }
    // C/D map of a 16x16 tile: col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
    for (int t = 0; t < NB_TILES; ++t) {
        const int gn = t * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gm = m0 + wm + 4 * (lane >> 4) + r;
            if (gm < M && gn < N) {
                if (partial) partial[((size_t)blockIdx.z * m_max + gm) * N + gn] = acc[t][r];
                else C[(size_t)gm * ldc + gn] = acc[t][r];
            }
        }
    }
}

// second stage of split-K: C = epilogue(sum_z partial[z]) in fixed z order (deterministic)
template <int ACT>
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const float* __restrict__ partial, int splits,
                                                                 const int32_t* __restrict__ m_ptr, int m_max, int N,
                                                                 float* __restrict__ C, int64_t ldc,
                                                                 const float* __restrict__ bias,
                                                                 const float* __restrict__ drop_mask, int drop_mode,
                                                                 uint64_t seed, uint64_t offset) {
    const int M = m_ptr ? min(m_ptr[0], m_max) : m_max;
    const int64_t total = (int64_t)M * N;
    if (drop_mode == 2) offset = philox_offset(offset);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int gm = (int)(i / N), gn = (int)(i % N);
        float v = 0.f;
        const float* __restrict__ pz = partial + (size_t)gm * N + gn;
        const size_t zs = (size_t)m_max * N;
        int z = 0;
        for (; z + 8 <= splits; z += 8) {  // eight loads in flight, the additions in the same fixed order as one by one
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = pz[(size_t)(z + u) * zs];
#pragma unroll
            for (int u = 0; u < 8; ++u) v += t[u];
        }
        for (; z < splits; ++z) v += pz[(size_t)z * zs];
        v += bias ? bias[gn] : 0.f;
        if (ACT == 1) v = v > 0.f ? v : expm1f(v);
        if (ACT == 2) v = fmaxf(v, 0.f);
        if (drop_mode == 1) v *= drop_mask[(size_t)gm * N + gn];
        else if (drop_mode == 2) v = (philox_bits(seed, offset, (uint64_t)gm * N + gn) & 1u) ? v * 2.f : 0.f;
        C[(size_t)gm * ldc + gn] = v;
    }
}

__global__ __launch_bounds__(256) void gemm_bce_finish_kernel(const float* __restrict__ partial, int n, float inv_numel,
                                                              float* __restrict__ loss) {
    __shared__ float red[4];
    float v = 0.f;
    int i = threadIdx.x;
    for (; i + 7 * 256 < n; i += 8 * 256) {  // fixed order (deterministic), eight loads in flight
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = partial[i + u * 256];
#pragma unroll
        for (int u = 0; u < 8; ++u) v += t[u];
    }
    for (; i < n; i += 256) v += partial[i];
    v = group_sum<64>(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = ((red[0] + red[1]) + (red[2] + red[3])) * inv_numel;
}

static bool aligned16(const void* p, int64_t ld) { return (((uintptr_t)p) & 15) == 0 && (ld % 4) == 0; }

static int launch_gemm(const float* a, int64_t lda, bool a_kc, const float* b, int64_t ldb, bool b_kc, float* c,
                       int64_t ldc, const int32_t* m_rows, int64_t m_max, int64_t n, int64_t k, const float* bias,
                       int act, const float* drop_mask, int drop_mode, uint64_t seed, uint64_t offset,
                       float* workspace, int64_t workspace_floats, hipStream_t st, GemmBce bce = GemmBce{}) {
    {
        // short products on few tiles (gemm_small.h): a latency chain on the 128-wide tiles, 64 x 64 tiles with the whole K (or
        // two chunks of it) in flight here.  Measured on MI355X, CAFE's module products at 512 paths: [512, 200] x [256, 200]^T
        // 26 -> 7 us, its weight gradient (K = 512) 33 -> 9 us (tools/bench_gemm_small.py).
        const char* sme = getenv("HW_GEMM_SMALL");
        const char* se2 = getenv("HW_GEMM_SPLIT");
        const bool on = (sme ? atoi(sme) != 0 : true) && (se2 ? atoi(se2) != 0 : true);
        const int64_t big_tiles = ((m_max + BM - 1) / BM) * ((n + BN - 1) / BN);
        if (on && act >= 0 && act <= 2 && drop_mode >= 0 && drop_mode <= 2 && k >= 4 && k <= 1024 && big_tiles <= 32 && aligned16(a, lda) &&
            aligned16(b, ldb) && (a_kc || m_max >= 4) && (b_kc || n >= 4)) {
            dim3 grid((unsigned)((m_max + SM_T - 1) / SM_T), (unsigned)((n + SM_T - 1) / SM_T));
#define GOS(AK, BK_, ACT_) hipLaunchKernelGGL((gemm_small_kernel<AK, BK_, ACT_>), grid, dim3(256), 0, st, a, lda, b, ldb, c, ldc, m_rows, (int)m_max, (int)n, (int)k, bias, drop_mask, drop_mode, seed, offset)
#define GOS3(AK, BK_)                     \
    do {                                  \
        if (act == 1) GOS(AK, BK_, 1);      \
        else if (act == 2) GOS(AK, BK_, 2); \
        else GOS(AK, BK_, 0);               \
    } while (0)
            if (a_kc && b_kc) GOS3(true, true);
            else if (a_kc && !b_kc) GOS3(true, false);
            else if (!a_kc && !b_kc) GOS3(false, false);
            else GOS3(false, true);
#undef GOS3
#undef GOS
            return HW_OK;
        }
    }
    // narrow-N products (TuckER / ConvE backward: N = d = 200): whole-row tiles of 16x16x4 MFMAs, see gemm_f32_n208_kernel
    {
        // Measured on MI355X (tools/bench_gemm_tucker.py, iFashion shapes): with an M-contiguous A (dE = dS^T . x) the
        // whole-row kernel runs 104-107 TF/s against 85-94 for the 128-wide tiles; with a K-contiguous A streamed over a
        // long K (dX = dS . E, 16 strided rows of 64 B per wave load) it is 84-86 against 88-93, so that case stays on the
        // wide tiles unless HW_GEMM_NARROW=2 forces it.  With the three-way bf16 split (the default, below) the 128-wide
        // tiles win both ways -- dE 117-119 TF/s, dX 120-125 although they pad N = 200 to 256 -- so this exact-FMA kernel
        // only runs with HW_GEMM_SPLIT=0 or when forced.
        const char* ne = getenv("HW_GEMM_NARROW");
        const char* se0 = getenv("HW_GEMM_SPLIT");
        const bool split_on = se0 ? atoi(se0) != 0 : true;
        const int nmode = ne ? atoi(ne) : 1;
        const bool want = nmode == 2 || (nmode == 1 && !a_kc && !split_on);
        if (want && act == 0 && !bias && drop_mode == 0 && n > 192 && n <= NB_N && aligned16(a, lda) && aligned16(b, ldb) &&
            k >= 4 && m_max >= 4) {
            const int64_t gmn = (m_max + NB_M - 1) / NB_M;
            int splits = 1;
            if (workspace && gmn < 512 && k >= 16 * BK) {
                int64_t wantz = (768 + gmn - 1) / gmn;
                const int64_t kmax = k / (8 * BK);
                if (wantz > kmax) wantz = kmax;
                const int64_t fit = workspace_floats / (m_max * n);
                if (wantz > fit) wantz = fit;
                if (wantz > 1) splits = (int)wantz;
            }
            int k_len = (int)k;
            if (splits > 1) {
                k_len = (int)(((k + splits - 1) / splits + BK - 1) / BK * BK);
                splits = (int)((k + k_len - 1) / k_len);
            }
            float* partial = splits > 1 ? workspace : nullptr;
            dim3 grid((unsigned)gmn, 1, (unsigned)splits);
#define GON(AK, BK_) hipLaunchKernelGGL((gemm_f32_n208_kernel<AK, BK_>), grid, dim3(512), 0, st, a, lda, b, ldb, c, ldc, m_rows, (int)m_max, (int)n, (int)k, k_len, partial)
            if (a_kc && b_kc) GON(true, true);
            else if (a_kc && !b_kc) GON(true, false);
            else if (!a_kc && !b_kc) GON(false, false);
            else GON(false, true);
#undef GON
            if (splits > 1) {
                int64_t rg = (m_max * n + 255) / 256;
                if (rg > 4096) rg = 4096;
                hipLaunchKernelGGL((gemm_splitk_reduce_kernel<0>), dim3((unsigned)rg), dim3(256), 0, st, partial, splits, m_rows,
                                   (int)m_max, (int)n, c, ldc, bias, drop_mask, drop_mode, seed, offset);
            }
            return HW_OK;
        }
    }
    {
        // wide tiles (gemm_split_wide_kernel): plain products with 128 < N <= 256 on 16-byte aligned operands
        const char* we = getenv("HW_GEMM_WIDE");
        const char* se1 = getenv("HW_GEMM_SPLIT");
        const char* pe1 = getenv("HW_GEMM_PIPE");
        const bool on = (we ? atoi(we) != 0 : true) && (se1 ? atoi(se1) != 0 : true) && (pe1 ? atoi(pe1) != 0 : true);
        if (on && act == 0 && !bias && drop_mode == 0 && n > BN && n <= WBN && aligned16(a, lda) && aligned16(b, ldb) && k >= 4 &&
            (a_kc || m_max >= 4) && (a_kc || !b_kc)) {
            // 256-row tiles when the row tiles are few and K is long (split-K fills the chip either way), else 128
            const char* te = getenv("HW_GEMM_TALL");
            const bool tall = (te ? atoi(te) != 0 : true) && a_kc && workspace && m_max >= 1024 && m_max <= 8192 && k >= 256 * BK;
            const int wmr = tall ? 256 : BM;
            const int64_t gmw = (m_max + wmr - 1) / wmr;
            int splits = 1;
            if (workspace && gmw < 128 && k >= 64 * BK) {  // one block per CU: 256 slots
                int64_t want = (256 + gmw - 1) / gmw;
                const int64_t kmax = k / (8 * BK);
                if (want > kmax) want = kmax;
                const int64_t fit = workspace_floats / (m_max * n);
                if (want > fit) want = fit;
                if (want >= 8) want = want / 8 * 8;  // a multiple of 8: the XCD-aware order
                if (want > 1) splits = (int)want;
            }
            int k_len = (int)k;
            if (splits > 1) {
                k_len = (int)(((k + splits - 1) / splits + BK - 1) / BK * BK);
                if ((k + k_len - 1) / k_len != splits) splits = (int)((k + k_len - 1) / k_len);
            }
            float* partial = splits > 1 ? workspace : nullptr;
            const int vc = splits > 1 ? ((((uintptr_t)partial) & 15) == 0 && (n % 4) == 0) : aligned16(c, ldc);
            const int flags = (vc << 1) | ((splits > 1 && splits % 8 == 0) ? 4 : 0);
            dim3 grid((unsigned)gmw, 1, (unsigned)splits);
#define GOW(AK, BK_, WM_) hipLaunchKernelGGL((gemm_split_wide_kernel<AK, BK_, WM_>), grid, dim3(WNT), 0, st, a, lda, b, ldb, c, ldc, m_rows, (int)m_max, (int)n, (int)k, k_len, partial, flags)
            if (tall && b_kc) GOW(true, true, 256);
            else if (tall) GOW(true, false, 256);
            else if (a_kc && b_kc) GOW(true, true, 128);
            else if (a_kc) GOW(true, false, 128);
            else GOW(false, false, 128);
#undef GOW
            if (splits > 1) {
                int64_t rg = (m_max * n + 255) / 256;
                if (rg > 4096) rg = 4096;
                hipLaunchKernelGGL((gemm_splitk_reduce_kernel<0>), dim3((unsigned)rg), dim3(256), 0, st, partial, splits, m_rows,
                                   (int)m_max, (int)n, c, ldc, bias, drop_mask, drop_mode, seed, offset);
            }
            return HW_OK;
        }
    }
    const int64_t gm = (m_max + BM - 1) / BM, gn = (n + BN - 1) / BN;
    // split-K when the output grid cannot fill the chip and K is long (e.g. dX = dY [2048, 144779] . E [144779, 200])
    int splits = 1;
    if (act < 3 && workspace && gm * gn < 256 && k >= 16 * BK) {
        // also the short-K, few-row products of a beam search's first hops: a lone tile walks its K / 16 steps at one
        // load latency each (75 us for [64, 512] x [256, 512]^T), so the chain is cut into splits of >= 4 steps
        int64_t want = (768 + gm * gn - 1) / (gm * gn);
        const int64_t kmax = k >= 64 * BK ? k / (8 * BK) : k / (4 * BK);
        if (want > kmax) want = kmax;
        const int64_t fit = workspace_floats / (m_max * n);
        if (want > fit) want = fit;
        if (want > 1) splits = (int)want;
    }
    int k_len = (int)k;
    if (splits > 1) {
        k_len = (int)(((k + splits - 1) / splits + BK - 1) / BK * BK);
        splits = (int)((k + k_len - 1) / k_len);
    }
    float* partial = splits > 1 ? workspace : nullptr;
    dim3 grid((unsigned)gm, (unsigned)gn, (unsigned)splits);
    const int va = aligned16(a, lda), vb = aligned16(b, ldb);
    const char* pe = getenv("HW_GEMM_PIPE");
    // the pipelined loop needs 16-byte loads on both operands and at least one float4 along each inner extent
    // (a K-contiguous operand is fetched along k, whatever its row count; a k-major one along its rows)
    const int pipe1 = (pe ? atoi(pe) : 1) && va && vb && k >= 4 && (a_kc || m_max >= 4) && (b_kc || n >= 4);
    // bit 1: the output rows take 16-byte stores (interior tiles leave through the LDS transposition of the epilogue)
    const int vc = splits > 1 ? ((((uintptr_t)partial) & 15) == 0 && (n % 4) == 0) : aligned16(c, ldc);
    int pipe = pipe1 | (vc << 1);
    {
        // XCD-aware block order (see the kernel): worth it when many row tiles share a B tile; the y (or z) extent is padded to
        // a multiple of 8, surplus blocks exit at once
        const char* xe = getenv("HW_GEMM_XCD");
        const bool on = xe ? atoi(xe) != 0 : true;
        if (on && gm >= 2 && gm * gn * splits >= 512) {
            if (splits > 1 && splits % 8 == 0) pipe |= 4;
            else if (splits == 1 && gn >= 16) {
                pipe |= 4;
                grid.y = (unsigned)((gn + 7) / 8 * 8);
                static const int group = [] { const char* e = getenv("HW_GEMM_GROUP"); return e ? atoi(e) & 255 : 0; }();
                pipe |= group << 8;
            }
        }
    }
    // 16-byte aligned operands: the three-way bf16 split on the bf16 matrix pipes (HW_GEMM_SPLIT=0: the exact-FMA loop)
    const char* se = getenv("HW_GEMM_SPLIT");
    const bool split = pipe1 && (se ? atoi(se) != 0 : true);
#define GO(AK, BK_, EL)                                                                                                  \
    do {                                                                                                                 \
        if (split)                                                                                                       \
            hipLaunchKernelGGL((gemm_f32_kernel<AK, BK_, EL, true>), grid, dim3(256), 0, st, a, lda, b, ldb, c, ldc,     \
                               m_rows, (int)m_max, (int)n, (int)k, bias, drop_mask, drop_mode, seed, offset, va, vb,     \
                               k_len, partial, pipe, bce);                                                               \
        else                                                                                                             \
            hipLaunchKernelGGL((gemm_f32_kernel<AK, BK_, EL, false>), grid, dim3(256), 0, st, a, lda, b, ldb, c, ldc,    \
                               m_rows, (int)m_max, (int)n, (int)k, bias, drop_mask, drop_mode, seed, offset, va, vb,     \
                               k_len, partial, pipe, bce);                                                               \
    } while (0)
#define GO3(AK, BK_)              \
    do {                          \
        if (act == 1) GO(AK, BK_, 1);      \
        else if (act == 2) GO(AK, BK_, 2); \
        else GO(AK, BK_, 0);               \
    } while (0)
#ifdef GEMM_ABL_ONLY  // experiment builds: one instantiation per operand layout (seconds instead of a minute of hipcc)
    if (split && act == 3 && a_kc && b_kc) {
        hipLaunchKernelGGL((gemm_f32_kernel<true, true, 3, true>), grid, dim3(256), 0, st, a, lda, b, ldb, c, ldc, m_rows, (int)m_max, (int)n, (int)k, bias, drop_mask, drop_mode, seed, offset, va, vb, k_len, partial, pipe, bce);
        return HW_OK;
    }
    if (!split || act != 0) return HW_ERR_UNSUPPORTED;
    if (a_kc && b_kc) hipLaunchKernelGGL((gemm_f32_kernel<true, true, 0, true>), grid, dim3(256), 0, st, a, lda, b, ldb, c, ldc, m_rows, (int)m_max, (int)n, (int)k, bias, drop_mask, drop_mode, seed, offset, va, vb, k_len, partial, pipe, bce);
    else if (a_kc && !b_kc) hipLaunchKernelGGL((gemm_f32_kernel<true, false, 0, true>), grid, dim3(256), 0, st, a, lda, b, ldb, c, ldc, m_rows, (int)m_max, (int)n, (int)k, bias, drop_mask, drop_mode, seed, offset, va, vb, k_len, partial, pipe, bce);
    else if (!a_kc && !b_kc) hipLaunchKernelGGL((gemm_f32_kernel<false, false, 0, true>), grid, dim3(256), 0, st, a, lda, b, ldb, c, ldc, m_rows, (int)m_max, (int)n, (int)k, bias, drop_mask, drop_mode, seed, offset, va, vb, k_len, partial, pipe, bce);
    else return HW_ERR_UNSUPPORTED;
#else
    if (act == 3) {  // fused BCE epilogue: x . E^T only, never split
        if (!(a_kc && b_kc) || splits != 1) return HW_ERR_UNSUPPORTED;
        GO(true, true, 3);
    } else if (act == 4) {  // distance epilogue (hw_gemm_dist): q . t^T only, never split
        if (!(a_kc && b_kc) || splits != 1) return HW_ERR_UNSUPPORTED;
        GO(true, true, 4);
    } else if (act == 5) {  // ranking product (hw_gemm_rank): q . t^T + bias from the three largest split terms, never split
        if (!(a_kc && b_kc) || splits != 1 || !split) return HW_ERR_UNSUPPORTED;
        GO(true, true, 5);
    } else if (act == 6) {  // the six-term product with the ranking product's epilogue (hw_gemm_f32_masked)
        if (!(a_kc && b_kc) || splits != 1 || !split) return HW_ERR_UNSUPPORTED;
        GO(true, true, 6);
    } else if (a_kc && b_kc) GO3(true, true);
    else if (a_kc && !b_kc) GO3(true, false);
    else if (!a_kc && !b_kc) GO3(false, false);
    else GO3(false, true);
#endif
#undef GO3
#undef GO
    if (splits > 1) {
        int64_t rg = (m_max * n + 255) / 256;
        if (rg > 4096) rg = 4096;
#define RED(ACT_)                                                                                                  \
    hipLaunchKernelGGL((gemm_splitk_reduce_kernel<ACT_>), dim3((unsigned)rg), dim3(256), 0, st, partial, splits, m_rows, \
                       (int)m_max, (int)n, c, ldc, bias, drop_mask, drop_mode, seed, offset)
        if (act == 1) RED(1);
        else if (act == 2) RED(2);
        else RED(0);
#undef RED
    }
    return HW_OK;
}

}  // namespace hw

using namespace hw;

extern "C" int hw_gemm_f32(const float* a, int64_t lda, int32_t a_k_contiguous, const float* b, int64_t ldb,
                           int32_t b_k_contiguous, float* c, int64_t ldc, const int32_t* m_rows, int64_t m_max,
                           int64_t n, int64_t k, const float* bias, int32_t elu, const float* drop_mask,
                           int32_t drop_mode, uint64_t seed, uint64_t offset, float* workspace,
                           int64_t workspace_floats, void* stream) {
    if (!a || !b || !c || m_max < 0 || n < 1 || k < 1 || ldc < n) return HW_ERR_INVALID_ARG;
    if (lda < (a_k_contiguous ? k : m_max) || ldb < (b_k_contiguous ? k : n)) return HW_ERR_INVALID_ARG;
    if (drop_mode < 0 || drop_mode > 2 || (drop_mode == 1 && !drop_mask) || elu < 0 || elu > 2) return HW_ERR_INVALID_ARG;
    if (m_max > INT32_MAX || n > INT32_MAX || k > INT32_MAX) return HW_ERR_INVALID_ARG;
    if (m_max == 0) return HW_OK;
    HW_CLEAR_ERR();
    launch_gemm(a, lda, a_k_contiguous != 0, b, ldb, b_k_contiguous != 0, c, ldc, m_rows, m_max, n, k, bias, elu,
                drop_mask, drop_mode, seed, offset, workspace, workspace ? workspace_floats : 0, (hipStream_t)stream);
    return HW_LAUNCH_RC();
}

// c[i, j] = <q_i, t_j> + bias[j] to ~2^-16 relative accuracy per product (three of the six split terms: half the matrix
// work): candidate generation for a consumer that re-scores exactly (hw_topk_rerank).  |error| <= 1.2e-5 sum_k |q_ik t_jk|.
extern "C" int hw_gemm_rank(const float* q, int64_t ldq, const float* t, int64_t ldt, const float* bias, float* c, int64_t ldc,
                            int64_t m, int64_t n, int64_t k, const uint32_t* mask_bits, int64_t mask_ld, float* tile_max,
                            int64_t tile_ld, void* stream) {
    if (!q || !t || !c || m < 0 || n < 1 || k < 4 || ldc < n || ldq < k || ldt < k) return HW_ERR_INVALID_ARG;
    if (m > INT32_MAX || n > INT32_MAX || k > INT32_MAX) return HW_ERR_INVALID_ARG;
    if ((mask_bits && mask_ld < (n + 31) / 32) || (tile_max && tile_ld < (n + 63) / 64)) return HW_ERR_INVALID_ARG;
    if (!aligned16(q, ldq) || !aligned16(t, ldt)) return HW_ERR_UNSUPPORTED;  // the split loop's operands
    if (m == 0) return HW_OK;
    HW_CLEAR_ERR();
    GemmBce rk{};
    rk.rank_mask = mask_bits, rk.rank_mask_ld = mask_ld, rk.rank_tile_max = tile_max, rk.rank_tile_ld = tile_ld;
    const int rc = launch_gemm(q, ldq, true, t, ldt, true, c, ldc, nullptr, m, n, k, bias, 5, nullptr, 0, 0, 0, nullptr, 0,
                               (hipStream_t)stream, rk);
    if (rc != HW_OK) return rc;
    return HW_LAUNCH_RC();
}

// hw_gemm_rank's by-products on the full six-term product: c = q . t^T + bias with fp32-level error, history mask applied, stretch
// maxima written -- the evaluation step of the bilinear scorers (whose product IS the score: no re-rank) without its mask pass
// and with hw_topk_from_tile_max instead of a top-k over the whole matrix.
extern "C" int hw_gemm_f32_masked(const float* q, int64_t ldq, const float* t, int64_t ldt, const float* bias, float* c,
                                  int64_t ldc, int64_t m, int64_t n, int64_t k, const uint32_t* mask_bits, int64_t mask_ld,
                                  float* tile_max, int64_t tile_ld, void* stream) {
    if (!q || !t || !c || m < 0 || n < 1 || k < 4 || ldc < n || ldq < k || ldt < k) return HW_ERR_INVALID_ARG;
    if (m > INT32_MAX || n > INT32_MAX || k > INT32_MAX) return HW_ERR_INVALID_ARG;
    if ((mask_bits && mask_ld < (n + 31) / 32) || (tile_max && tile_ld < (n + 63) / 64)) return HW_ERR_INVALID_ARG;
    if (!aligned16(q, ldq) || !aligned16(t, ldt)) return HW_ERR_UNSUPPORTED;  // the split loop's operands
    if (m == 0) return HW_OK;
    HW_CLEAR_ERR();
    GemmBce rk{};
    rk.rank_mask = mask_bits, rk.rank_mask_ld = mask_ld, rk.rank_tile_max = tile_max, rk.rank_tile_ld = tile_ld;
    const int rc = launch_gemm(q, ldq, true, t, ldt, true, c, ldc, nullptr, m, n, k, bias, 6, nullptr, 0, 0, 0, nullptr, 0,
                               (hipStream_t)stream, rk);
    if (rc != HW_OK) return rc;
    return HW_LAUNCH_RC();
}

// c = q . t^T (+ bias) with fp32-level error, and per 64-column stretch of every row the pair (maximum, sum of exp(x - maximum))
// in tile_max / tile_sum [m, tile_ld]: hw_lse_rows turns them into the row's (maximum, log sum exp) -- a log-softmax over the
// whole row is then (x - maximum) - log sum exp wherever a consumer reads x, and the pass that rewrote the matrix is gone.
extern "C" int hw_gemm_f32_lse(const float* q, int64_t ldq, const float* t, int64_t ldt, const float* bias, float* c,
                               int64_t ldc, int64_t m, int64_t n, int64_t k, float* tile_max, float* tile_sum, int64_t tile_ld,
                               void* stream) {
    if (!q || !t || !c || !tile_max || !tile_sum || m < 0 || n < 1 || k < 4 || ldc < n || ldq < k || ldt < k)
        return HW_ERR_INVALID_ARG;
    if (m > INT32_MAX || n > INT32_MAX || k > INT32_MAX || tile_ld < (n + 63) / 64) return HW_ERR_INVALID_ARG;
    if (!aligned16(q, ldq) || !aligned16(t, ldt)) return HW_ERR_UNSUPPORTED;  // the split loop's operands
    if (m == 0) return HW_OK;
    HW_CLEAR_ERR();
    GemmBce rk{};
    rk.rank_tile_max = tile_max, rk.rank_tile_sum = tile_sum, rk.rank_tile_ld = tile_ld;
    const int rc = launch_gemm(q, ldq, true, t, ldt, true, c, ldc, nullptr, m, n, k, bias, 6, nullptr, 0, 0, 0, nullptr, 0,
                               (hipStream_t)stream, rk);
    if (rc != HW_OK) return rc;
    return HW_LAUNCH_RC();
}

// c[i, j] = base - ||q_i - t_j|| from ONE product: the distance epilogue of the kernel above (ACT = 4).
extern "C" int hw_gemm_dist(const float* q, int64_t ldq, const float* t, int64_t ldt, float* c, int64_t ldc, int64_t m,
                            int64_t n, int64_t k, const float* qn, const float* tn, float base, float rescore_frac,
                            void* stream) {
    if (!q || !t || !c || !qn || !tn || m < 0 || n < 1 || k < 1 || ldc < n || ldq < k || ldt < k) return HW_ERR_INVALID_ARG;
    if (!(rescore_frac >= 0.f && rescore_frac <= 1.f)) return HW_ERR_INVALID_ARG;
    if (m > INT32_MAX || n > INT32_MAX || k > INT32_MAX) return HW_ERR_INVALID_ARG;
    if (m == 0) return HW_OK;
    HW_CLEAR_ERR();
    GemmBce ep{nullptr, base, rescore_frac, 0.f, nullptr};
    const int rc = launch_gemm(q, ldq, true, t, ldt, true, c, ldc, nullptr, m, n, k, tn, 4, qn, 0, 0, 0, nullptr, 0,
                               (hipStream_t)stream, ep);
    if (rc) return rc;
    return HW_LAUNCH_RC();
}

extern "C" int hw_mlp_layer(const float* a, int64_t lda, const float* w, const float* bias, float* c, int64_t ldc,
                            const int32_t* m_rows, int64_t m_max, int64_t n, int64_t k, int32_t elu,
                            const float* drop_mask, int32_t drop_mode, uint64_t seed, uint64_t offset, void* stream) {
    return hw_gemm_f32(a, lda, 1, w, k, 1, c, ldc, m_rows, m_max, n, k, bias, elu, drop_mask, drop_mode, seed, offset,
                       nullptr, 0, stream);
}

extern "C" int64_t hw_gemm_bce_workspace_floats(int64_t m, int64_t n) {
    return m < 0 || n < 0 ? 0 : ((m + BM - 1) / BM) * ((n + BN - 1) / BN);
}

extern "C" int hw_gemm_f32_bce(const float* x, int64_t ldx, const float* table, int64_t ldt, const float* bias, float* g_out,
                               int64_t ldg, int64_t m, int64_t n, int64_t k, const uint8_t* pos_flag, float target_scale,
                               float target_add, float* loss_out, float* workspace, int64_t workspace_floats,
                               void* stream) {
    if (!x || !table || !g_out || !loss_out || !workspace || m < 1 || n < 1 || k < 1 || ldg < n || ldx < k || ldt < k)
        return HW_ERR_INVALID_ARG;
    if (m > INT32_MAX || n > INT32_MAX || k > INT32_MAX) return HW_ERR_INVALID_ARG;
    const int64_t need = hw_gemm_bce_workspace_floats(m, n);
    if (workspace_floats < need || need > INT32_MAX) return HW_ERR_WORKSPACE;
    HW_CLEAR_ERR();
    hipStream_t st = (hipStream_t)stream;
    GemmBce bce{pos_flag, target_scale, target_add, (float)(1.0 / ((double)m * (double)n)), workspace};
    const int64_t tiles_m = (m + BM - 1) / BM;
    const int64_t blocks = tiles_m * ((n + BN - 1) / BN);
    const int rc = launch_gemm(x, ldx, true, table, ldt, true, g_out, ldg, nullptr, m, n, k, bias, 3, nullptr, 0, 0, 0, nullptr,
                               0, st, bce);
    if (rc) return rc;
    hipLaunchKernelGGL(gemm_bce_finish_kernel, dim3(1), dim3(256), 0, st, workspace, (int)blocks, bce.inv_numel, loss_out);
    return HW_LAUNCH_RC();
}
