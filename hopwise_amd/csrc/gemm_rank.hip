// The ranking product on pre-split operands: c[i, j] = <q_i, t_j> + bias[j] from the three largest split terms (hi.hi, hi.mid,
// mid.hi), history mask and per-stretch maxima on the way out -- hw_gemm_rank's result, bit for bit, from a kernel shaped for what
// bounds THAT product.
//
// hw_gemm_rank runs on gemm.hip's 128 x 128 tiles, which split their fp32 operands on the way into LDS.  For the evaluation's
// [2 048 users, K = 400] x [45 539 items, K]^T that kernel took the same 0.38 ms with three terms as with six: halving the matrix
// work changed nothing, because every 128 x 128 tile pulls 128 x K x 4 bytes of each operand through L2 (2.3 GB for the whole
// product: the chip's L2 row rate for ~0.2 ms) and converts them again (the user rows 356 times over).  Here:
//   * both operands arrive as bf16 PLANE IMAGES (hi, mid), split once -- the item table once per weight version (an evaluation epoch
//     scores thousands of batches against the same table), the batch's query rows once per call -- and stored in LDS image order,
//     so a k-step's 32 KB travel global -> LDS by DMA (global_load_lds): no conversion, no staging registers in the main loop;
//   * 128 x 256 tiles (four waves of 64 x 128), two blocks per CU (HW_GEMM_RANK_TILE=256: 256 x 256 tiles, one block per CU: least
//     L2 traffic per multiply, 300 against 284 us); XCD-aware block order (the blocks of one item tile on one XCD);
//   * three (four) LDS stages, DMAs two (three) k-steps ahead, one barrier per k-step of 24 MFMAs per wave (counted vmcnt, raw
//     s_barrier: the rules of pgpr_policy.hip's weight path).
// Same terms in the same order per accumulator and k ascending: the sums are hw_gemm_rank's.
#include <cstdlib>
#include <type_traits>

#include "hw_common.h"
#include "rank_epilogue.h"

namespace hw {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int RK_T = 256;                  // tile rows of either operand
constexpr int RK_PLANE = RK_T * 32;        // [256 rows][16 k] bf16: 8 KB
constexpr int RK_OPERAND = 2 * RK_PLANE;   // hi, mid

// byte offset of fragment half h (8 k = 16 B) of row r inside a plane image: gemm.hip's split_off
__device__ __host__ __forceinline__ int rk_off(int r, int half) { return r * 32 + ((half ^ ((r >> 3) & 1)) << 4); }

// fp32 [rows, K] -> plane images: image (row tile rt, k-step ks, plane p) at ((rt * n_ks + ks) * 2 + p) * RK_PLANE, rows / k past
// the matrix as zeros.  One thread per (row, k-quad).
__global__ __launch_bounds__(256) void rank_planes_kernel(const float* __restrict__ x, int64_t ld, int64_t rows, int K,
                                                          int n_ks, uint8_t* __restrict__ planes, int vec) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (padded row, k quad)
    const int quads = n_ks * 4;
    const int64_t r = i / quads;
    const int kq = (int)(i % quads) * 4;
    const int64_t rows_pad = (rows + RK_T - 1) / RK_T * RK_T;
    if (r >= rows_pad) return;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (r < rows) {
        const float* __restrict__ p = x + r * ld + kq;
        if (vec && kq + 3 < K) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(p);
            v[0] = q.x, v[1] = q.y, v[2] = q.z, v[3] = q.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = kq + j < K ? p[j] : 0.f;
        }
    }
    uint32_t hi[2], mid[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {  // gemm.hip: split_pair, first two planes
        const f32x2 a = {v[2 * j], v[2 * j + 1]};
        const bf16x2 h = __builtin_convertvector(a, bf16x2);
        const f32x2 r1 = a - __builtin_convertvector(h, f32x2);
        const bf16x2 m = __builtin_convertvector(r1, bf16x2);
        hi[j] = __builtin_bit_cast(uint32_t, h), mid[j] = __builtin_bit_cast(uint32_t, m);
    }
    const int ks = kq >> 4, kk = kq & 15;
    uint8_t* dst = planes + ((size_t)(r / RK_T) * n_ks + ks) * RK_OPERAND + rk_off((int)(r % RK_T), kk >> 3) + ((kk & 4) << 1);
    *reinterpret_cast<uint2*>(dst) = uint2{hi[0], hi[1]};
    *reinterpret_cast<uint2*>(dst + RK_PLANE) = uint2{mid[0], mid[1]};
}

// TA = rows of the query tile: 256 (eight waves, one block per CU) or 128 (four waves, 72 KB of LDS: two blocks per CU).  The item
// tile is 256 rows either way; a wave owns 64 x 128.  Measured (2 048 x 45 539 x 400, tools/bench_gemm_rank.py): 284 us (128) /
// 300 us (256), of which ~65 us are the 373 MB of stores (the same kernel without them: 200 us; a device fill of the matrix:
// 55 us) and the main loop ~0.45 us per k -- about half the matrix pipes' rate: what was tried on top without effect is listed in
// DESIGN.md section 8 (a half-tile phase shift between the two blocks of a CU; item tiles fastest in the block order).
template <int TA>
__global__ __launch_bounds__(TA * 2, TA == 128 ? 2 : 1) void gemm_rank_planes_kernel(
    const uint8_t* __restrict__ pa, const uint8_t* __restrict__ pb, int n_ks, float* __restrict__ C, int64_t ldc, int M, int N,
    const float* __restrict__ bias, const uint32_t* __restrict__ mask, int64_t mask_ld, float* __restrict__ tile_max,
    int64_t tile_ld, int vec_c, int group) {
    constexpr int NW = TA / 32;                     // waves
    constexpr int A_PLANE = TA * 32;                // the tile's rows of one plane image
    constexpr int STAGE = 2 * A_PLANE + RK_OPERAND; // A (hi, mid) then B (hi, mid)
    constexpr int STAGES = TA == 128 ? 3 : 4;
    constexpr int PIECES = STAGE / 1024, A_PIECES = 2 * A_PLANE / 1024, PER_WAVE = PIECES / NW;
    static_assert(PIECES % NW == 0, "whole pieces per wave");
    // ONE __shared__ object (a second one beside an array that direct-to-LDS loads write makes hipcc wait vmcnt(0) before LDS reads)
    __shared__ __attribute__((aligned(16))) uint8_t lds[STAGES * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = wave_in_block();
    const int wr = wave >> 1, wc = wave & 1;  // wave tile: rows 64 wr .., columns 128 wc ..
    // XCD-aware block order (gemm.hip): workgroups go to the 8 XCDs round-robin by linear id, so the blocks that read the SAME
    // item tile (all query tiles of one column of the grid) are renumbered onto ONE XCD -- the tile is then fetched from HBM into
    // one L2 instead of eight (73 MB of item planes instead of 584 MB per product).  gridDim.y is a multiple of 8.
    // Within an XCD, `group` query tiles (not all of them) share an item tile at a time: the blocks of one item tile run in step
    // and ask for the same bytes, so what an XCD has in flight towards HBM is (resident blocks / blocks per item tile) tiles --
    // with all 16 query tiles on one item tile, four tiles x two k-steps = 128 KB per XCD, i.e. a stream of ~0.3 TB/s at HBM's
    // latency: THAT paced the main loop (1.4 us per k-step against 0.37 us of matrix work).  Smaller groups put more distinct
    // item tiles in flight; the item planes are then read `gridDim.x / group` times (from the Infinity Cache after the first).
    const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;
    const unsigned xcd = lin & 7u, j = lin >> 3;
    const unsigned gx = gridDim.x, n_col = gridDim.y >> 3;  // query tiles; item tiles per XCD
    const unsigned g = (unsigned)group < gx ? (unsigned)group : gx;
    const unsigned per_sweep = g * n_col;                   // blocks of one sweep over the XCD's item tiles with g query tiles
    const unsigned sweep = j / per_sweep, in = j % per_sweep;
    const unsigned g_here = min(g, gx - sweep * g);         // the last sweep may hold fewer query tiles
    const unsigned bm = sweep * g + in % g_here, bn = (in / g_here) * 8u + xcd;
    const int m0 = bm * TA, n0 = bn * RK_T;
    if (n0 >= N) return;  // the padding of gridDim.y (whole block: before any barrier)
    // the query tile's rows inside their 256-row plane image: rows (m0 % 256) .. + TA, contiguous in image order
    const uint8_t* __restrict__ a_img = pa + (size_t)(m0 / RK_T) * n_ks * RK_OPERAND + (m0 % RK_T) * 32;
    const uint8_t* __restrict__ b_img = pb + (size_t)bn * n_ks * RK_OPERAND;
    // k-step s -> stage s % STAGES, in one-KB pieces (destination = base + lane * 16), PER_WAVE per wave
    auto dma = [&](int s) {
        typedef __attribute__((address_space(1))) const void gptr_t;
        typedef __attribute__((address_space(3))) void lptr_t;
        const int sc = s < n_ks ? s : n_ks - 1;  // past the end: the last step again, into a stage nobody reads any more
        uint8_t* dst = lds + (s % STAGES) * STAGE;
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const int piece = wave * PER_WAVE + i;
            const uint8_t* src;
            if (piece < A_PIECES) {  // plane piece / (A_PIECES / 2), KB piece % (A_PIECES / 2) of the tile's rows
                src = a_img + (size_t)sc * RK_OPERAND + (piece / (A_PIECES / 2)) * RK_PLANE + (piece % (A_PIECES / 2)) * 1024;
            } else {
                src = b_img + (size_t)sc * RK_OPERAND + (piece - A_PIECES) * 1024;
            }
            __builtin_amdgcn_global_load_lds((gptr_t*)(src + lane * 16), (lptr_t*)(dst + piece * 1024), 16, 0, 0);
        }
    };
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x16{0};
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s) dma(s);
    for (int s = 0; s < n_ks; ++s) {
        // step s has landed (the younger DMAs of the steps after it may still fly), every wave is done with step s - 1
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"((STAGES - 2) * PER_WAVE) : "memory");
        dma(s + STAGES - 1);  // into the stage step s - 1 was read from
        const uint8_t* st = lds + (s % STAGES) * STAGE;
        bf16x8 fa[2][2], fb[4][2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
                fa[t][p] = *reinterpret_cast<const bf16x8*>(st + p * A_PLANE + rk_off(64 * wr + 32 * t + (lane & 31), lane >> 5));
#pragma unroll
            for (int t = 0; t < 4; ++t)
                fb[t][p] = *reinterpret_cast<const bf16x8*>(st + 2 * A_PLANE + p * RK_PLANE +
                                                            rk_off(128 * wc + 32 * t + (lane & 31), lane >> 5));
        }
        // smallest terms first, as gemm_f32_kernel<.., 5, ..>: (mid, hi), (hi, mid), (hi, hi)
#define RK_TERM(PA_, PB_)                                                                   \
    _Pragma("unroll") for (int ti = 0; ti < 2; ++ti) _Pragma("unroll") for (int tj = 0; tj < 4; ++tj) \
        acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ti][PA_], fb[tj][PB_], acc[ti][tj], 0, 0, 0);
        RK_TERM(1, 0)
        RK_TERM(0, 1)
        RK_TERM(0, 0)
#undef RK_TERM
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the stages become the store slabs

    // ---- epilogue: gemm_f32_kernel's ACT == 5 branch per 32-row x 64-column slab ----
    float* const slab = reinterpret_cast<float*>(lds) + wave * 2048;  // 8 KB per wave (NW x 8 KB <= the stages)
    const int g5 = lane >> 5;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int hc = 0; hc < 2; ++hc) {  // the wave's two 64-column halves
            const int gm0 = m0 + 64 * wr + 32 * ti, gn0 = n0 + 128 * wc + 64 * hc;
            if (gm0 >= M || gn0 >= N) continue;  // wave-uniform
            uint32_t mword = 0u;
            if (mask) {
                const int mr = gm0 + (lane >> 1);
                const int64_t wcol = ((int64_t)gn0 >> 5) + (lane & 1);
                if (mr < M && wcol < mask_ld) mword = mask[(size_t)mr * mask_ld + wcol];
            }
            float v[2][16], rmax[16];
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) rmax[rg] = -INFINITY;
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) {
                const int gn = gn0 + tj * 32 + (lane & 31);
                const float bv = (bias && gn < N) ? bias[gn] : 0.f;
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    const int r0 = (rg & 3) + 8 * (rg >> 2);
                    const uint32_t wa = (uint32_t)__builtin_amdgcn_readlane((int)mword, 2 * r0 + tj);
                    const uint32_t wb = (uint32_t)__builtin_amdgcn_readlane((int)mword, 2 * (r0 + 4) + tj);
                    const uint32_t wsel = g5 ? wb : wa;
                    float x = acc[ti][2 * hc + tj][rg] + bv;
                    if ((wsel >> (lane & 31)) & 1u) x = -INFINITY;
                    v[tj][rg] = x;
                    rmax[rg] = fmaxf(rmax[rg], gn < N ? x : -INFINITY);
                }
            }
            if (tile_max) {
                int rg;
                const float mx = half_wave_row_max(rmax, lane, rg);  // rank_epilogue.h
                const int gm = gm0 + (rg & 3) + 8 * (rg >> 2) + 4 * g5;
                if ((lane & 1) == 0 && gm < M) tile_max[(size_t)gm * tile_ld + (gn0 >> 6)] = mx;
            }
            if (vec_c && gm0 + 32 <= M && gn0 + 64 <= N) {  // whole slab, 16-byte rows: through the wave's LDS slab
                const int c = lane & 31;
#pragma unroll
                for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                    for (int rg = 0; rg < 16; ++rg) slab[((rg & 3) + 8 * (rg >> 2) + 4 * g5) * 64 + tj * 32 + c] = v[tj][rg];
                const int r = lane >> 4, q = (lane & 15) * 4;
                float* crow = C + (size_t)gm0 * ldc + gn0;
#pragma unroll
                for (int pass = 0; pass < 8; ++pass) {
                    const f32x4 x = *reinterpret_cast<const f32x4*>(&slab[(pass * 4 + r) * 64 + q]);
                    *reinterpret_cast<f32x4*>(crow + (size_t)(pass * 4 + r) * ldc + q) = x;
                }
            } else {
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) {
                    const int gn = gn0 + tj * 32 + (lane & 31);
#pragma unroll
                    for (int rg = 0; rg < 16; ++rg) {
                        const int gm = gm0 + (rg & 3) + 8 * (rg >> 2) + 4 * g5;
                        if (gm < M && gn < N) C[(size_t)gm * ldc + gn] = v[tj][rg];
                    }
                }
            }
        }
}

}  // namespace
}  // namespace hw

using namespace hw;

extern "C" int64_t hw_rank_planes_bytes(int64_t rows, int64_t k) {
    if (rows < 1 || k < 1) return 0;
    return (rows + RK_T - 1) / RK_T * ((k + 15) / 16) * (int64_t)RK_OPERAND;
}

extern "C" int hw_rank_planes(const float* x, int64_t ld, int64_t rows, int64_t k, void* planes, void* stream) {
    if (!x || !planes || rows < 1 || k < 1 || ld < k || k > INT32_MAX || (((uintptr_t)planes) & 15)) return HW_ERR_INVALID_ARG;
    const int n_ks = (int)((k + 15) / 16);
    const int64_t rows_pad = (rows + RK_T - 1) / RK_T * RK_T;
    const int64_t n = rows_pad * n_ks * 4;
    if ((n + 255) / 256 > INT32_MAX) return HW_ERR_INVALID_ARG;
    const int vec = (((uintptr_t)x & 15) == 0 && (ld & 3) == 0) ? 1 : 0;
    HW_CLEAR_ERR();
    hipLaunchKernelGGL(rank_planes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ld, rows,
                       (int)k, n_ks, (uint8_t*)planes, vec);
    return HW_LAUNCH_RC();
}

extern "C" int hw_gemm_rank_planes(const void* q_planes, const void* t_planes, int64_t m, int64_t n, int64_t k, const float* bias,
                                   float* c, int64_t ldc, const uint32_t* mask_bits, int64_t mask_ld, float* tile_max,
                                   int64_t tile_ld, void* stream) {
    if (!q_planes || !t_planes || !c || m < 0 || n < 1 || k < 1 || ldc < n) return HW_ERR_INVALID_ARG;
    if (m > INT32_MAX || n > INT32_MAX || k > INT32_MAX) return HW_ERR_INVALID_ARG;
    if ((mask_bits && mask_ld < (n + 31) / 32) || (tile_max && tile_ld < (n + 63) / 64)) return HW_ERR_INVALID_ARG;
    if ((((uintptr_t)q_planes) | ((uintptr_t)t_planes)) & 15) return HW_ERR_INVALID_ARG;
    if (m == 0) return HW_OK;
    const int64_t gn = (n + RK_T - 1) / RK_T;
    if (gn > 65535) return HW_ERR_INVALID_ARG;
    const int vec_c = ((((uintptr_t)c) & 15) == 0 && (ldc & 3) == 0) ? 1 : 0;
    // 128-row query tiles, two blocks per CU (one's epilogue beside the other's main loop), unless told otherwise
    static const bool big = [] { const char* e = getenv("HW_GEMM_RANK_TILE"); return e && atoi(e) == 256; }();
    static const int group = [] { const char* e = getenv("HW_GEMM_RANK_GROUP"); return e && atoi(e) > 0 ? atoi(e) : 4; }();
    HW_CLEAR_ERR();
    if (big)
        hipLaunchKernelGGL(gemm_rank_planes_kernel<256>, dim3((unsigned)((m + 255) / 256), (unsigned)((gn + 7) / 8 * 8)), dim3(512), 0,
                           (hipStream_t)stream, (const uint8_t*)q_planes, (const uint8_t*)t_planes, (int)((k + 15) / 16), c, ldc,
                           (int)m, (int)n, bias, mask_bits, mask_ld, tile_max, tile_ld, vec_c, group);
    else
        hipLaunchKernelGGL(gemm_rank_planes_kernel<128>, dim3((unsigned)((m + 127) / 128), (unsigned)((gn + 7) / 8 * 8)), dim3(256), 0,
                           (hipStream_t)stream, (const uint8_t*)q_planes, (const uint8_t*)t_planes, (int)((k + 15) / 16), c, ldc,
                           (int)m, (int)n, bias, mask_bits, mask_ld, tile_max, tile_ld, vec_c, group);
    return HW_LAUNCH_RC();
}
