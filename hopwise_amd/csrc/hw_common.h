// Device-side helpers shared by the gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/hopwise_amd.h"

#define HW_WAVE 64
#define HW_MAX_DIM 1024

// hipGetLastError() reports the last error of ANY earlier runtime call on this thread (torch leaves benign
// ones behind), so every launcher clears it first and reads it back right after its own launches.
#define HW_CLEAR_ERR() ((void)hipGetLastError())
#define HW_LAUNCH_RC() (hipGetLastError() == hipSuccess ? HW_OK : HW_ERR_LAUNCH)

namespace hw {

// The wave's number inside its block, read from lane 0: a scalar for the compiler too.  Everything derived from it (the row or
// task a wave works on, that row's ids, bounds and base addresses) then lives in SGPRs and is fetched by scalar loads instead
// of 64 equal lanes of vector registers and vector loads (fused RotatE scorer: 79 -> 61 VGPRs from this alone).
__device__ __forceinline__ int wave_in_block() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }


typedef float f32x4 __attribute__((ext_vector_type(4)));

// One 32-bit word of the Philox4x32-10 stream keyed by `seed`: word idx & 3 of block (idx >> 2, offset).
__device__ __forceinline__ uint32_t philox_bits(uint64_t seed, uint64_t offset, uint64_t idx) {
    uint32_t c0 = (uint32_t)(idx >> 2), c1 = (uint32_t)(idx >> 34), c2 = (uint32_t)offset, c3 = (uint32_t)(offset >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0, c1 = n1, c2 = n2, c3 = n3;
        k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
    }
    const uint32_t w[4] = {c0, c1, c2, c3};
    return w[idx & 3];
}



// Philox offsets that live on the device.  Every entry point that takes a (seed, offset) pair accepts, instead of a plain offset,
// an INDIRECT one: bit 63 set, bits 0..47 the address of a uint64 in device memory, bits 48..62 a small addend -- the stream
// offset is then `*address + addend`, read when the kernel RUNS.  A training step captured in a HIP graph freezes its kernel
// arguments; with indirect offsets each replay still draws fresh masks / actions: the graph's last node bumps the word
// (hopwise_amd.philox.PhiloxCounter, driver.GraphedTrainStep).  Kernels resolve the offset once, on entry.
__device__ __forceinline__ uint64_t philox_offset(uint64_t offset) {
    if (offset >> 63) {  // wave-uniform: a kernel argument
        const uint64_t* p = reinterpret_cast<const uint64_t*>(offset & 0x0000FFFFFFFFFFFFull);
        return *p + ((offset >> 48) & 0x7FFFull);
    }
    return offset;
}

// All four words of block `blk` of the same stream: philox_bits(seed, offset, 4 * blk + k) == w[k].  For code that walks
// four consecutive positions starting at a multiple of 4 -- one Philox evaluation instead of four.
__device__ __forceinline__ void philox_bits4(uint64_t seed, uint64_t offset, uint64_t blk, uint32_t (&w)[4]) {
    uint32_t c0 = (uint32_t)blk, c1 = (uint32_t)(blk >> 32), c2 = (uint32_t)offset, c3 = (uint32_t)(offset >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0, c1 = n1, c2 = n2, c3 = n3;
        k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
    }
    w[0] = c0, w[1] = c1, w[2] = c2, w[3] = c3;
}

// Philox4x32-10 on a caller-built counter block (the negative sampler's convention: c = {position lo, position hi ^
// (attempt << 8), offset lo, offset hi}, key = seed); all four output words are used.
__device__ __forceinline__ void philox4(uint64_t seed, uint32_t (&c)[4]) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0, c[1] = n1, c[2] = n2, c[3] = n3;
        k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
    }
}

// Arguments of the in-scorer uniform negative sampler (hw_kge_score_grouped_sampled): same draws as hw_neg_sample.
struct SamplerArgs {
    const int64_t* used_ptr;
    const int64_t* used_idx;
    const int64_t* keys;
    int64_t n_key_space, value_num;
    uint64_t seed, offset;
    int words;  // bitmap words per wave = ceil(value_num / 32); 0: value space beyond the bitmap, membership by binary search
    int stagger;  // fused scorer: issue priority by co-resident block during the draws (kge_score_kernels.h)
};
constexpr int SAMPLER_STAGE_IDS = 1024;  // words == 0: a key's sorted used slice is staged in LDS (4 KB: 8 blocks per CU stay resident) when it has at most this many ids
constexpr int SAMPLER_MAX_ATTEMPTS_FUSED = 1 << 16;

// Self-adversarial loss folded into the sampled grouped scorer (hw_kge_score_grouped_sampled_selfadv): the block that scores
// a group also reduces it; a one-block finish kernel sums the per-group terms.
struct SelfAdvArgs {
    float alpha, inv_b;
    float* group_loss;  // [n_groups]
    float* loss_out;    // [1]; nullptr: the group terms stay in group_loss for a later launch to add (chained form)
    // chained form: the group terms an EARLIER launch on this stream left behind, added by block 0 of this launch
    const float* prev_group_loss = nullptr;
    float* prev_loss_out = nullptr;
    int prev_n = 0;
};

__device__ __forceinline__ float log_sigmoid(float x) {  // -(max(-x,0) + log1p(exp(-|x|)))
    return -(fmaxf(-x, 0.f) + log1pf(expf(-fabsf(x))));
}

// ---- DPP helpers ------------------------------------------------------------------------------------
// Cross-lane adds inside a 16-lane row cost one VALU op with a DPP modifier (no LDS-crossbar round trip):
// quad_perm for xor 1 / xor 2, row_half_mirror / row_mirror to finish the row.  Crossing rows (xor 16 / 32)
// goes through ds_bpermute (__shfl_xor).  Callers keep the whole wave active around these (EXEC all ones).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

// sum over aligned groups of W lanes (W = 1, 2, 4, 8, 16) inside a DPP row; every lane gets its group's total.
// xor 1 / xor 2 are quad_perm; once quads are uniform row_half_mirror (i -> 7 - i) reaches the other quad of an
// 8-lane group, and once 8-lane groups are uniform row_mirror (i -> 15 - i) reaches the other half of the row.
template <int W>
__device__ __forceinline__ float dpp_group_sum(float v) {
    if constexpr (W >= 2) v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    if constexpr (W >= 4) v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    if constexpr (W >= 8) v += dpp_mov<0x141>(v);  // row_half_mirror
    if constexpr (W >= 16) v += dpp_mov<0x140>(v); // row_mirror
    return v;
}

// Wave-wide maximum / sum of a 64-bit value per lane without the LDS crossbar: four DPP steps inside each 16-lane row (quad_perm,
// quad_perm, row_half_mirror, row_mirror: two moves, a compare and two selects per step), then the four row results through
// v_readlane.  A __shfl_xor of a 64-bit value is two ds_bpermute round trips; the six dependent steps of a butterfly measured
// ~1 400 cycles per selection round of the interpreter, 56 % of the kernel (cycle stamps).  Maxima only: a sum taken this way adds in another order than group_sum's.  Whole wave active.
template <int CTRL>
__device__ __forceinline__ uint64_t dpp_mov_u64(uint64_t v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, 0xf, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, 0xf, 0xf, false);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int lane_const) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane_const);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane_const);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
    uint64_t o;
    o = dpp_mov_u64<0xB1>(v), v = o > v ? o : v;
    o = dpp_mov_u64<0x4E>(v), v = o > v ? o : v;
    o = dpp_mov_u64<0x141>(v), v = o > v ? o : v;
    o = dpp_mov_u64<0x140>(v), v = o > v ? o : v;
    const uint64_t a = readlane_u64(v, 0), b = readlane_u64(v, 16), c = readlane_u64(v, 32), d = readlane_u64(v, 48);
    const uint64_t ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}
__device__ __forceinline__ int wave_sum_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
           __builtin_amdgcn_readlane(v, 48);
}

__device__ __forceinline__ float wave_max_f32(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x140>(v));
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    uint32_t o;
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false), v = o > v ? o : v;
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false), v = o > v ? o : v;
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false), v = o > v ? o : v;
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, false), v = o > v ? o : v;
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    const uint32_t ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}

// ---- sub-wave sum over G lanes (G power of two, 1..64); every lane gets the total ----
template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int m = G / 2; m >= 16; m >>= 1) v += __shfl_xor(v, m, HW_WAVE);
    return dpp_group_sum<(G < 16 ? G : 16)>(v);
}

// Reduce U partial sums (one per tail) over the G lanes of a sub-wave group with a reduce-scatter butterfly:
// while more than one value is live, lanes whose bit `m` is clear keep the lower half of the values and ship the
// upper half to lane ^ m (and vice versa).  U = 4, G = 64: 2 + 1 + 1 ds_bpermute and 4 DPP adds instead of 4 x 6
// ds_bpermute.  Returns this lane's total; `which` = index of the tail it belongs to.  The G / U lanes that share
// `which` all hold the same total.
template <int G, int U>
__device__ __forceinline__ float group_reduce_multi(float (&p)[U], int gl, int& which) {
    static_assert((G & (G - 1)) == 0 && (U & (U - 1)) == 0 && U <= G, "shape");
    if constexpr (G == 64 && U == 4) {
        // gfx950 lane-half / row swaps do a butterfly stage of the reduce-scatter in ONE instruction and without
        // selects: v_permlane32_swap exchanges lanes 32..63 of its first operand with lanes 0..31 of its second, so
        // (p0, p2) -> a pair whose sum holds p0's two-lane sums in the lower half and p2's in the upper half;
        // v_permlane16_swap does the same between odd and even 16-lane rows.  After the two stages row r of the wave
        // holds the four-row sums of p[r]; a DPP row reduction finishes.  10 instructions for four tails.
        const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(p[0]), __float_as_uint(p[2]), false, false);
        const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(p[1]), __float_as_uint(p[3]), false, false);
        const float s02 = __uint_as_float(a[0]) + __uint_as_float(a[1]);
        const float s13 = __uint_as_float(b[0]) + __uint_as_float(b[1]);
        const auto c = __builtin_amdgcn_permlane16_swap(__float_as_uint(s02), __float_as_uint(s13), false, false);
        which = gl >> 4;  // rows 0..3 own p[0], p[1], p[2], p[3]
        return dpp_group_sum<16>(__uint_as_float(c[0]) + __uint_as_float(c[1]));
    }
    if constexpr (G == 32 && U == 4) {
        // two 32-lane groups per wave, each reducing its own four values: the row stage is one v_permlane16_swap per pair
        // (odd rows of the first operand <-> even rows of the second), the half-row stage two selects and a DPP rotate
        const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(p[0]), __float_as_uint(p[1]), false, false);
        const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(p[2]), __float_as_uint(p[3]), false, false);
        const float s01 = __uint_as_float(a[0]) + __uint_as_float(a[1]);  // even row: p0, odd row: p1
        const float s23 = __uint_as_float(b[0]) + __uint_as_float(b[1]);  // even row: p2, odd row: p3
        const bool hi8 = (gl & 8) != 0;
        const float keep = hi8 ? s23 : s01, send = hi8 ? s01 : s23;
        which = ((gl >> 4) & 1) + (hi8 ? 2 : 0);
        return dpp_group_sum<8>(keep + dpp_mov<0x128>(send));  // row_ror:8 pairs the two halves of a row
    }
    if constexpr (G == 16 && U == 4) {
        // four 16-lane groups per wave (one DPP row each): halves of the row, then quads of each half -- DPP only
        const bool hi8 = (gl & 8) != 0, hi4 = (gl & 4) != 0;
        const float k0 = hi8 ? p[2] : p[0], s0 = hi8 ? p[0] : p[2];
        const float k1 = hi8 ? p[3] : p[1], s1 = hi8 ? p[1] : p[3];
        const float r0 = k0 + dpp_mov<0x128>(s0), r1 = k1 + dpp_mov<0x128>(s1);
        const float k = hi4 ? r1 : r0, sd = hi4 ? r0 : r1;
        which = (hi8 ? 2 : 0) + (hi4 ? 1 : 0);
        // row_half_mirror (i -> 7 - i) is a bijection between the two quads of a half row: enough, the quad is summed next
        return dpp_group_sum<4>(k + dpp_mov<0x141>(sd));
    }
    constexpr int REM = G / U;  // lanes that share one value after the scatter phase
    which = 0;
    int live = U;
    int m = G / 2;
#pragma unroll
    for (; live > 1; live >>= 1, m >>= 1) {
        const bool upper = (gl & m) != 0;
#pragma unroll
        for (int i = 0; i < live / 2; ++i) {
            const float keep = upper ? p[i + live / 2] : p[i];
            const float send = upper ? p[i] : p[i + live / 2];
            p[i] = keep + __shfl_xor(send, m, HW_WAVE);
        }
        which = which * 2 + (upper ? 1 : 0);
    }
    float v = p[0];
#pragma unroll
    for (int mm = REM / 2; mm >= 16; mm >>= 1) v += __shfl_xor(v, mm, HW_WAVE);  // masks that cross DPP rows
    return dpp_group_sum<(REM < 16 ? REM : 16)>(v);
}

// branch-free clamp into [0, n); `bad` accumulates whether anything was out of range (reported once per wave)
__device__ __forceinline__ int clamp_row_flag(int64_t id, int64_t n, bool& bad) {
    const int64_t c = id < 0 ? 0 : (id >= n ? n - 1 : id);
    bad |= (c != id);
    return (int)c;
}

// clamp an id into [0, n) and count violations (ids outside the table never reach a load address)
__device__ __forceinline__ int clamp_row(int64_t id, int64_t n, int32_t* oob) {
    if (id < 0 || id >= n) {
        if (oob) atomicAdd(oob, 1);
        id = id < 0 ? 0 : n - 1;
    }
    return (int)id;
}

// ---- sigmoid + BCELoss element (tucker.py:83,124-125), shared by tucker.hip and the GEMM's fused epilogue ------------
// One element of the loss / gradient.  The kernel that calls this is transcendental-bound (~3e8 elements per TuckER side),
// so the element is written with ONE exp, ONE log and ONE reciprocal on the hardware units (v_exp_f32 / v_log_f32 /
// v_rcp_f32) instead of the literal sequence sigmoid -> log(p), log(1 - p) (one exp, two logs, one reciprocal):
//     e = exp(-|x|),  r = 1 / (1 + e),  p = sigmoid(x) = x >= 0 ? r : e r
//     -log(1 - p) = softplus(x) = max(x, 0) + log(1 + e),   -log(p) = softplus(x) - x
//     loss = t (-log p) + (1 - t)(-log(1 - p))
// torch.nn.BCELoss clamps both logs at -100, and the reference feeds it a float32 sigmoid: p rounds to exactly 1 for
// x > 16.6355 (1 + 2^-24 rounds to 1), where log(1 - p) = log(0) clamps to -100, and exp(-x) overflows for x < -88.72, where
// p = 0 and log(p) clamps to -100.  Both saturations are reproduced; between them the softplus form is the exact value the
// reference's float32 log(1 - p) approximates (its 1 - p carries a relative rounding error of up to 2^-24 / (1 - p): 3e-4
// at x = 8, 3 % at x = 14).
__device__ __forceinline__ float bce_elem(float x, float t, float inv_numel, float* p_out, float* g_out) {
    // the raw hardware instructions: __logf / __fdividef expand to a denormal-range fix-up and a full IEEE division (12 VALU
    // instructions each, measured in the GEMM epilogue's ISA); 1 + e lies in [1, 2], where neither is needed
    const float e = __builtin_amdgcn_exp2f(fabsf(x) * -1.44269504088896341f);
    const float r = __builtin_amdgcn_rcpf(1.f + e);
    const float p = x >= 0.f ? r : e * r;
    const float sp = fmaf(__builtin_amdgcn_logf(1.f + e), 0.693147180559945309f, fmaxf(x, 0.f));  // -log(1 - p)
    const float nl1p = x > 16.635532f ? 100.f : fminf(sp, 100.f);
    const float nlp = x < -88.72283f ? 100.f : fminf(sp - x, 100.f);
    const float pq = (1.f - p) * p;
    *p_out = p;
    // (p - t) / max(pq, 1e-12) * pq: the quotient and the product cancel unless pq was clamped -- one reciprocal fewer
    *g_out = (pq >= 1e-12f ? p - t : (p - t) * (pq * 1e12f)) * inv_numel;
    return t * nlp + (1.f - t) * nl1p;
}

// The same element when |x| <= 16 is known (no clamp of torch.nn.BCELoss is active, p (1 - p) > 1e-12, exp(-x) neither
// overflows nor vanishes against 1): with e = exp(-x), p = 1 / (1 + e) and -log(1 - p) = x + log(1 + e), so
//     loss = log(1 + e) + (1 - t) x,   dLoss/dx = (p - t) / numel
// -- no |x|, no max, no select: 7 vector instructions + the three transcendentals.  omt = 1 - t, t_inv = t * inv_numel
// (per column, computed by the caller).
__device__ __forceinline__ float bce_elem_unclamped(float x, float omt, float t_inv, float inv_numel, float* g_out) {
    const float s = 1.f + __builtin_amdgcn_exp2f(x * -1.44269504088896341f);
    *g_out = fmaf(__builtin_amdgcn_rcpf(s), inv_numel, -t_inv);
    return fmaf(__builtin_amdgcn_logf(s), 0.693147180559945309f, omt * x);
}

// Flat float4 sweep over the [n_rows, n_cols] matrix (rows are not 16-byte aligned when n_cols is odd, the flat buffer
// is); the column of the first lane element is carried incrementally -- no 64-bit modulo in the loop.
// ---- one embedding row spread over the G lanes of a sub-wave group -----------------------------
// VEC = true : dim % 4 == 0; lane gl owns float4 chunks gl, gl+G, ...  (NE/4 chunks)  -> 16-B loads
// VEC = false: any dim;      lane gl owns elements   gl, gl+G, ...     (NE elements)  -> 4-B loads
// Slots past the end of the row hold 0.
#ifndef HW_ROW_LOAD_AUX  // cache policy bits of the whole-wave row loads (experiment switch, tools/scorer_ab.sh: 2 = nt)
#define HW_ROW_LOAD_AUX 0
#endif
#ifndef HW_ROW_BUFFER_LOAD
#define HW_ROW_BUFFER_LOAD 1
#endif
template <int G, int NE, bool VEC>
struct Row {
    float v[NE];

    // Branch-free: lanes past the end of the row re-read the last chunk / element (always in bounds) and the
    // value is then zeroed with a select.  A load inside a divergent branch makes hipcc wait for it right there
    // (vmcnt(0) before the phi moves), which serialises the U rows a wave is supposed to keep in flight.
    __device__ __forceinline__ void load(const float* __restrict__ base, int dim, int gl) {
        load_raw(base, dim, gl);
#pragma unroll
        for (int i = 0; i < NE; ++i) v[i] = col(i, dim, gl) >= 0 ? v[i] : 0.f;
    }

    // Same loads without the zeroing: padding slots hold copies of valid data; the caller masks their
    // contribution (Scorer::partial with MASK = true).
    __device__ __forceinline__ void load_raw(const float* __restrict__ base, int dim, int gl) {
        if constexpr (VEC && G == 64 && HW_ROW_BUFFER_LOAD) {
            // a whole wave on one row: the row as a raw buffer of dim * 4 bytes -- lanes past its end are out of range and
            // read zeros without an address clamp
            __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, dim * 4, 0x00020000);
#pragma unroll
            for (int c = 0; c < NE / 4; ++c) {
                const f32x4 x = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (gl + c * G) * 16, 0, HW_ROW_LOAD_AUX));
                v[4 * c + 0] = x.x;
                v[4 * c + 1] = x.y;
                v[4 * c + 2] = x.z;
                v[4 * c + 3] = x.w;
            }
        } else if constexpr (VEC) {
            const int d4 = dim >> 2;
            const f32x4* __restrict__ p = reinterpret_cast<const f32x4*>(base);
#pragma unroll
            for (int c = 0; c < NE / 4; ++c) {
                const int q = gl + c * G;
                const f32x4 x = p[q < d4 ? q : d4 - 1];
                v[4 * c + 0] = x.x;
                v[4 * c + 1] = x.y;
                v[4 * c + 2] = x.z;
                v[4 * c + 3] = x.w;
            }
        } else {
#pragma unroll
            for (int c = 0; c < NE; ++c) {
                const int e = gl + c * G;
                v[c] = base[e < dim ? e : dim - 1];
            }
        }
    }

    // validity of the load unit (float4 chunk or element) that slot i belongs to
    __device__ __forceinline__ static bool slot_valid(int i, int dim, int gl) { return col(i, dim, gl) >= 0; }

    // column of slot i, or -1 when the slot lies past the end of the row
    __device__ __forceinline__ static int col(int i, int dim, int gl) {
        int c;
        if constexpr (VEC) {
            c = 4 * (gl + (i >> 2) * G) + (i & 3);
        } else {
            c = gl + i * G;
        }
        return c < dim ? c : -1;
    }
};

// Launch geometry chosen on the host from `dim`.
struct RowShape {
    int G;     // lanes per row (16, 32, 64)
    int NE;    // register slots per lane
    bool vec;  // 16-byte loads
};

inline RowShape row_shape(int dim, bool allow_vec) {
    RowShape s;
    if (allow_vec && (dim % 4) == 0) {
        const int d4 = dim / 4;
        s.vec = true;
        s.G = d4 <= 16 ? 16 : (d4 <= 32 ? 32 : 64);
        const int chunks = (d4 + s.G - 1) / s.G;  // 1..4
        s.NE = chunks <= 1 ? 4 : (chunks <= 2 ? 8 : 16);
    } else {
        s.vec = false;
        s.G = 64;
        s.NE = dim <= 64 ? 1 : (dim <= 256 ? 4 : 16);
    }
    return s;
}

// Stream-ordered fill of a buffer with a byte value, as a KERNEL launch (what hipMemsetAsync is for, without being one).
// Why not hipMemsetAsync: a training step captured in a HIP graph turns it into a memset NODE, and on this stack a memset
// node writes its captured value on the FIRST launch of the instantiated graph only -- from the second launch on it fills
// the destination with a 16-byte pattern read from runtime staging memory that has been reused since (the byte count of
// the fill, a 1 and an address-like word: tools/memset_pattern_probe.py shows it in one process, every replay after the
// first; profiles/r4/memset_pattern_probe*.json).  Viewed as floats that pattern is mostly denormals, which is why the
// relation-replica scratch of hw_kge_grad_grouped (zeroed, accumulated into, reduced) only OCCASIONALLY came back visibly
// wrong.  A kernel node carries its arguments in the graph and has no such state.
// Any size and alignment: unaligned head and tail bytes are written singly, the body as 16-byte or 4-byte words.
static __global__ __launch_bounds__(256) void fill_words_kernel(uint32_t* __restrict__ p, size_t n_words, uint32_t v) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if ((((uintptr_t)p) & 15) == 0) {
        uint4* __restrict__ p4 = reinterpret_cast<uint4*>(p);
        const size_t n4 = n_words >> 2;
        for (size_t q = i; q < n4; q += stride) p4[q] = uint4{v, v, v, v};
        for (size_t q = (n4 << 2) + i; q < n_words; q += stride) p[q] = v;
        return;
    }
    for (; i < n_words; i += stride) p[i] = v;
}

static __global__ __launch_bounds__(64) void fill_edge_bytes_kernel(uint8_t* __restrict__ head, int n_head,
                                                                   uint8_t* __restrict__ tail, int n_tail, uint8_t b) {
    const int t = threadIdx.x;
    if (t < n_head) head[t] = b;
    if (t < n_tail) tail[t] = b;
}

inline hipError_t fill_bytes_async(void* p, int byte_value, size_t bytes, hipStream_t st) {
    if (bytes == 0) return hipSuccess;
    if (!p) return hipErrorInvalidValue;
    const uint32_t b = (uint32_t)(byte_value & 0xff), v = b | (b << 8) | (b << 16) | (b << 24);
    uint8_t* base = (uint8_t*)p;
    size_t n_head = (size_t)((4 - ((uintptr_t)base & 3)) & 3);  // bytes up to the first 4-byte boundary
    if (n_head > bytes) n_head = bytes;
    const size_t n_words = (bytes - n_head) >> 2, n_tail = bytes - n_head - (n_words << 2);
    if (n_head || n_tail)
        hipLaunchKernelGGL(fill_edge_bytes_kernel, dim3(1), dim3(64), 0, st, base, (int)n_head,
                           base + n_head + (n_words << 2), (int)n_tail, (uint8_t)b);
    if (n_words) {
        size_t blocks = (n_words / 4 + 255) / 256;
        if (blocks < 1) blocks = 1;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(fill_words_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (uint32_t*)(base + n_head), n_words, v);
    }
    return hipGetLastError();
}

}  // namespace hw
