// Device negative sampler: rejection sampling against CSR "used" sets.
//
// Restates AbstractSampler.sample_by_key_ids (/root/reference/hopwise/sampler/sampler.py:140-183):
// out[j*n_keys + i] ~ Uniform([1, value_num) \ used(keys[i])) (uniform mode) or the alias-table
// popularity distribution restricted the same way (:68-116).  The reference redraws rejected
// positions round by round on the host; here each output position owns a Philox4x32-10 counter
// stream and redraws in registers until accepted, so the result is distribution-equal (not
// numpy-stream-equal) and independent of launch geometry.
#include <stdlib.h>

#include "hw_common.h"

namespace hw {

struct Philox {
    uint32_t k0, k1;
    __device__ __forceinline__ static void round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c[0] = n0, c[1] = n1, c[2] = n2, c[3] = n3;
    }
    __device__ __forceinline__ void operator()(uint32_t (&c)[4]) const {
        uint32_t a = k0, b = k1;
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            round(c, a, b);
            a += 0x9E3779B9u;
            b += 0xBB67AE85u;
        }
    }
};

__device__ __forceinline__ bool csr_contains(const int64_t* __restrict__ idx, int64_t lo, int64_t hi, int64_t v) {
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        const int64_t x = idx[mid];
        if (x == v) return true;
        if (x < v) lo = mid + 1;
        else hi = mid;
    }
    return false;
}

constexpr int SAMPLER_MAX_ATTEMPTS = 1 << 16;

__global__ __launch_bounds__(256) void neg_sample_kernel(
    const int64_t* __restrict__ used_ptr, const int64_t* __restrict__ used_idx, int64_t n_key_space,
    const int64_t* __restrict__ keys, int64_t n_keys, int64_t total, int64_t value_num,
    const float* __restrict__ alias_prob, const int64_t* __restrict__ alias_keys,
    const int64_t* __restrict__ alias_alias, int64_t n_alias, uint64_t seed, uint64_t offset,
    int64_t* __restrict__ out, int64_t out_stride_key, int64_t out_stride_draw, int32_t* oob) {
    offset = philox_offset(offset);
    const int64_t tpos = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tpos >= total) return;
    // Thread -> (key i, draw j): follow the OUTPUT layout so that stores coalesce and, in the row-per-key layout, a wave
    // works on one key (its used list stays in L1).  The random stream is tied to the reference position
    // e = j * n_keys + i either way, so the values drawn do not depend on the layout or the launch geometry.
    const int64_t num = total / n_keys;
    const bool key_major = out_stride_draw <= out_stride_key;
    const int64_t i = key_major ? tpos / num : tpos % n_keys;
    const int64_t j = key_major ? tpos % num : tpos / n_keys;
    const int64_t e = j * n_keys + i;
    const int64_t o = i * out_stride_key + j * out_stride_draw;
    int64_t key = keys[i];
    if (key < 0 || key >= n_key_space) {  // the reference raises ValueError (sampler.py:286-291,352-357)
        if (oob) atomicAdd(oob, 1);
        out[o] = 0;
        return;
    }
    const int64_t lo = used_ptr[key], hi = used_ptr[key + 1];
    Philox rng{(uint32_t)seed, (uint32_t)(seed >> 32)};
    const uint32_t span = (uint32_t)(value_num - 1);
    int64_t v = 0;
    bool ok = false;
    for (uint32_t attempt = 0; attempt < SAMPLER_MAX_ATTEMPTS && !ok; ++attempt) {
        uint32_t c[4] = {(uint32_t)e, (uint32_t)((uint64_t)e >> 32) ^ (attempt << 8), (uint32_t)offset,
                         (uint32_t)(offset >> 32)};
        rng(c);
        if (alias_prob) {
            // two draws per philox block: (index, uniform) pairs (sampler.py:105-116)
#pragma unroll
            for (int t = 0; t < 2 && !ok; ++t) {
                const int64_t a = (int64_t)(((uint64_t)c[2 * t] * (uint64_t)n_alias) >> 32);
                const float u = (float)(c[2 * t + 1] >> 8) * (1.0f / 16777216.0f);
                v = alias_prob[a] > u ? alias_keys[a] : alias_alias[a];
                ok = !csr_contains(used_idx, lo, hi, v);
            }
        } else {
#pragma unroll
            for (int t = 0; t < 4 && !ok; ++t) {
                v = 1 + (int64_t)(((uint64_t)c[t] * (uint64_t)span) >> 32);
                ok = !csr_contains(used_idx, lo, hi, v);
            }
        }
    }
    if (!ok && oob) atomicAdd(oob, 1);  // a key that has used (almost) every id; the reference would spin
    out[o] = v;
}

// Fast path for the grouped layout (a block of 256 consecutive draws belongs to ONE key, uniform distribution): the
// key's used set becomes an LDS bitmap once per block, membership is one LDS read instead of a dependent binary
// search through global memory.  Same Philox counters as neg_sample_kernel -> bit-identical output.
constexpr int SAMPLER_BITMAP_WORDS = 8192;  // value_num <= 262 144

__global__ __launch_bounds__(256) void neg_sample_bitmap_kernel(
    const int64_t* __restrict__ used_ptr, const int64_t* __restrict__ used_idx, int64_t n_key_space,
    const int64_t* __restrict__ keys, int64_t n_keys, int num, int64_t value_num, uint64_t seed, uint64_t offset,
    int64_t* __restrict__ out, int64_t out_stride_key, int64_t out_stride_draw, int32_t* oob) {
    offset = philox_offset(offset);
    __shared__ uint32_t bits[SAMPLER_BITMAP_WORDS];
    const int chunks = (num + 255) / 256;
    const int64_t i = blockIdx.x / chunks;
    const int j = (int)(blockIdx.x % chunks) * 256 + threadIdx.x;
    const int64_t key = keys[i];
    const bool bad_key = key < 0 || key >= n_key_space;
    const int words = (int)((value_num + 31) >> 5);
    for (int w = threadIdx.x; w < words; w += 256) bits[w] = 0u;
    __syncthreads();
    if (!bad_key) {
        const int64_t lo = used_ptr[key], hi = used_ptr[key + 1];
        for (int64_t q = lo + threadIdx.x; q < hi; q += 256) {
            const int64_t u = used_idx[q];
            if (u >= 0 && u < value_num) atomicOr(&bits[u >> 5], 1u << (u & 31));
        }
    }
    __syncthreads();
    if (j >= num) return;
    const int64_t o = i * out_stride_key + (int64_t)j * out_stride_draw;
    if (bad_key) {
        if (oob && threadIdx.x == 0) atomicAdd(oob, 1);
        out[o] = 0;
        return;
    }
    const int64_t e = (int64_t)j * n_keys + i;
    Philox rng{(uint32_t)seed, (uint32_t)(seed >> 32)};
    const uint32_t span = (uint32_t)(value_num - 1);
    int64_t v = 0;
    bool ok = false;
    for (uint32_t attempt = 0; attempt < SAMPLER_MAX_ATTEMPTS && !ok; ++attempt) {
        uint32_t c[4] = {(uint32_t)e, (uint32_t)((uint64_t)e >> 32) ^ (attempt << 8), (uint32_t)offset,
                         (uint32_t)(offset >> 32)};
        rng(c);
#pragma unroll
        for (int t = 0; t < 4 && !ok; ++t) {
            v = 1 + (int64_t)(((uint64_t)c[t] * (uint64_t)span) >> 32);
            ok = !((bits[v >> 5] >> (v & 31)) & 1u);
        }
    }
    if (!ok && oob) atomicAdd(oob, 1);
    out[o] = v;
}

}  // namespace hw

using namespace hw;

extern "C" int hw_neg_sample(const int64_t* used_ptr, const int64_t* used_idx, int64_t n_key_space,
                             const int64_t* keys, int64_t n_keys, int32_t num, int64_t value_num,
                             const float* alias_prob, const int64_t* alias_keys, const int64_t* alias_alias,
                             int64_t n_alias, uint64_t seed, uint64_t offset, int64_t* out, int64_t out_stride_key,
                             int64_t out_stride_draw, int32_t* oob_counter, void* stream) {
    if (!used_ptr || !keys || !out || n_key_space < 1 || n_keys < 0 || num < 0) return HW_ERR_INVALID_ARG;
    if (value_num < 2 || value_num > 0xffffffffll) return HW_ERR_INVALID_ARG;
    if (alias_prob && (!alias_keys || !alias_alias || n_alias < 1)) return HW_ERR_INVALID_ARG;
    const int64_t total = n_keys * (int64_t)num;
    if (total == 0) return HW_OK;
    if ((total + 255) / 256 > INT32_MAX) return HW_ERR_INVALID_ARG;
    HW_CLEAR_ERR();
    const int64_t chunks = ((int64_t)num + 255) / 256;
    const char* knob = getenv("HW_SAMPLER_BITMAP");  // "0" forces the generic kernel (tests compare the two)
    if (!(knob && knob[0] == '0') && !alias_prob && used_idx && num >= 128 && out_stride_draw <= out_stride_key &&
        value_num <= (int64_t)SAMPLER_BITMAP_WORDS * 32 && n_keys * chunks <= INT32_MAX) {
        hipLaunchKernelGGL(neg_sample_bitmap_kernel, dim3((unsigned)(n_keys * chunks)), dim3(256), 0, (hipStream_t)stream,
                           used_ptr, used_idx, n_key_space, keys, n_keys, (int)num, value_num, seed, offset, out,
                           out_stride_key, out_stride_draw, oob_counter);
        return HW_LAUNCH_RC();
    }
    hipLaunchKernelGGL(neg_sample_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       used_ptr, used_idx, n_key_space, keys, n_keys, total, value_num, alias_prob, alias_keys,
                       alias_alias, n_alias, seed, offset, out, out_stride_key, out_stride_draw, oob_counter);
    return HW_LAUNCH_RC();
}

extern "C" const char* hw_version(void) { return "hopwise_amd 0.1.0 (gfx950)"; }
