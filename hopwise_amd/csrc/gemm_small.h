// Short products on few tiles (included by gemm.hip, inside namespace hw, after the split helpers).
//
// The 128 x 128 kernels walk K in steps of 16 with one tile of register prefetch: a lone tile pays one global-load latency per
// step (~2.3 us: [512, 200] x [256, 200]^T = 13 steps = 33 us for 52 MFLOP), and the layers of the path models are exactly such
// products -- CAFE's relation modules ([paths, 200] x [256, 200]^T and back: 33 launches of a training step, 0.82 of its
// 1.33 ms as a graph replay), PGPR's policy layers during training and on the first hops of a search.  They are latency
// chains, not throughput problems, so this kernel is built the other way round:
//   * 64 x 64 output tiles (four waves, one 32 x 32 accumulator each): 4 x the blocks of the big tiles on an empty chip;
//   * K in chunks of 64 with TWO chunks of register prefetch: every load of a product with K <= 128 is in flight before the
//     first wait, a K = 200 / 256 product waits for memory about twice instead of 13 / 16 times;
//   * one barrier per chunk (two LDS stages of 2 x 3 plane images [64 rows][64 k] bf16 = 48 KB each).
// Same arithmetic as the big kernels: three-way bf16 split, six v_mfma_f32_32x32x16_bf16 per 16 k in the same term order, k
// ascending per accumulator -- same bits as gemm_f32_kernel<.., SPLIT> on the same operands.  Epilogue: bias, ELU / ReLU,
// dropout (explicit mask or Philox keep-1/2, the same bits as the big kernels draw).  Operands 16-byte aligned (as for every
// split kernel); no split-K (long-K products stay on the big tiles).

constexpr int SM_T = 64, SM_KC = 64;
constexpr int SM_SUB = SM_T * 32;          // one [64 rows][16 k] bf16 sub-image (K-contiguous operands): 2 KB
constexpr int SM_PLANE = 4 * SM_SUB;       // a chunk's plane image: 8 KB
constexpr int SM_OPERAND = 3 * SM_PLANE;   // hi / mid / lo
constexpr int SM_STAGE = 2 * SM_OPERAND;   // A and B: 48 KB
constexpr int SM_KM_ROW = SM_T * 2;        // k-major image: [64 k][64 rows] bf16, 128-byte k-rows

// k-major image: the 32-byte pieces of k-row k sit at piece ^ (k & 3): the four k-rows of one transposing read (same column
// range) then fall on four different 32-byte bank ranges; an 8-byte store / read piece stays whole
__device__ __forceinline__ int sm_km_off(int k, int col) { return k * SM_KM_ROW + ((col * 2) ^ ((k & 3) << 5)); }

// this thread's four float4 of chunk c of one operand (clamped addresses, straight-line: see fetch_tile)
template <bool KC>
__device__ __forceinline__ void sm_fetch(const float* __restrict__ X, int64_t ld, int rows, int K, int r0, int k0, f32x4 (&v)[4]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = tid + 256 * i;
        if constexpr (KC) {
            const int r = f >> 4, kq = (f & 15) * 4;
            const int gr = min(r0 + r, rows - 1), gk = min(k0 + kq, ((K + 3) & ~3) - 4);
            v[i] = *reinterpret_cast<const f32x4*>(X + (size_t)gr * ld + gk);
        } else {
            const int k = f >> 4, rq = (f & 15) * 4;
            const int gk = min(k0 + k, K - 1), gr = min(r0 + rq, ((rows + 3) & ~3) - 4);
            v[i] = *reinterpret_cast<const f32x4*>(X + (size_t)gk * ld + gr);
        }
    }
}

// registers of sm_fetch -> the three plane images of one operand's chunk (k >= K stored as zeros)
template <bool KC>
__device__ __forceinline__ void sm_put(const f32x4 (&v)[4], int K, int k0, uint8_t* __restrict__ img) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = tid + 256 * i;
        float x0, x1, x2, x3;
        uint8_t* dst;
        if constexpr (KC) {
            const int r = f >> 4, kq = (f & 15) * 4, kg = k0 + kq;
            x0 = kg < K ? v[i].x : 0.f, x1 = kg + 1 < K ? v[i].y : 0.f;
            x2 = kg + 2 < K ? v[i].z : 0.f, x3 = kg + 3 < K ? v[i].w : 0.f;
            const int kk = kq & 15;  // inside the 16-k sub-image kq >> 4
            dst = img + (kq >> 4) * SM_SUB + split_off(r, kk >> 3) + ((kk & 4) << 1);
        } else {
            const int k = f >> 4, rq = (f & 15) * 4;
            const bool live = k0 + k < K;
            x0 = live ? v[i].x : 0.f, x1 = live ? v[i].y : 0.f, x2 = live ? v[i].z : 0.f, x3 = live ? v[i].w : 0.f;
            dst = img + sm_km_off(k, rq);
        }
        uint32_t h0, m0, l0, h1, m1, l1;
        split_pair(x0, x1, h0, m0, l0);
        split_pair(x2, x3, h1, m1, l1);
        *reinterpret_cast<uint2*>(dst) = uint2{h0, h1};
        *reinterpret_cast<uint2*>(dst + SM_PLANE) = uint2{m0, m1};
        *reinterpret_cast<uint2*>(dst + 2 * SM_PLANE) = uint2{l0, l1};
    }
}

// MFMA fragment (k = 16 s + 8 h .. + 7 of tile row base + (lane & 31), h = lane >> 5) of sub-step s
template <bool KC>
__device__ __forceinline__ bf16x8 sm_frag(const uint8_t* __restrict__ plane, int s, int base, int lane) {
    if constexpr (KC) {
        return *reinterpret_cast<const bf16x8*>(plane + s * SM_SUB + split_off(base + (lane & 31), lane >> 5));
    } else {
        // as read_frag_split<false>: group g = lane >> 4 covers tile rows 16 (g & 1) .. + 15 at k = 8 (g >> 1) ..; lane 4 q + p
        // of the group addresses k-row q, columns 4 p .. 4 p + 3
        const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
        const uint8_t* a = plane + sm_km_off(16 * s + 8 * (g >> 1) + q, base + 16 * (g & 1) + 4 * pp);  // k + 4: same k & 3
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 4 * SM_KM_ROW));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, both);
    }
}

template <bool A_KC, bool B_KC, int ACT>
__global__ __launch_bounds__(256, 1) void gemm_small_kernel(const float* __restrict__ A, int64_t lda,
                                                            const float* __restrict__ B, int64_t ldb,
                                                            float* __restrict__ C, int64_t ldc,
                                                            const int32_t* __restrict__ m_ptr, int m_max, int N, int K,
                                                            const float* __restrict__ bias,
                                                            const float* __restrict__ drop_mask, int drop_mode,
                                                            uint64_t seed, uint64_t offset) {
    __shared__ __attribute__((aligned(16))) uint8_t img[2 * SM_STAGE];
    const int M = m_ptr ? min(m_ptr[0], m_max) : m_max;
    const int m0 = blockIdx.x * SM_T, n0 = blockIdx.y * SM_T;
    if (m0 >= M || n0 >= N) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    f32x16 acc = f32x16{0};
    const int n_chunks = (K + SM_KC - 1) / SM_KC;

    f32x4 ra[2][4], rb[2][4];
    sm_fetch<A_KC>(A, lda, M, K, m0, 0, ra[0]);
    sm_fetch<B_KC>(B, ldb, N, K, n0, 0, rb[0]);
    sm_fetch<A_KC>(A, lda, M, K, m0, SM_KC, ra[1]);  // past the end: clamped re-reads, stored as zeros (never multiplied)
    sm_fetch<B_KC>(B, ldb, N, K, n0, SM_KC, rb[1]);

    // chunk c: its registers -> LDS stage c & 1, barrier, the registers refilled with chunk c + 2, the chunk multiplied.  The
    // stage written here was last read while chunk c - 2 was multiplied, and every wave has passed the barrier of chunk c - 1
    // since: one barrier per chunk.
    auto chunk = [&](int c, f32x4 (&sa)[4], f32x4 (&sb)[4]) {
        uint8_t* const st = img + (c & 1) * SM_STAGE;
        sm_put<A_KC>(sa, K, c * SM_KC, st);
        sm_put<B_KC>(sb, K, c * SM_KC, st + SM_OPERAND);
        __syncthreads();
        sm_fetch<A_KC>(A, lda, M, K, m0, (c + 2) * SM_KC, sa);
        sm_fetch<B_KC>(B, ldb, N, K, n0, (c + 2) * SM_KC, sb);
        __builtin_amdgcn_sched_barrier(0);  // the loads stay above the MFMAs
        // all four sub-steps, whatever K: the k >= K part of a ragged chunk was stored as zeros (a branch here would put the
        // loads above into another basic block than their first use, and hipcc then waits for them before the MFMAs)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 fa[3], fb[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                fa[p] = sm_frag<A_KC>(st + p * SM_PLANE, s, wm, lane);
                fb[p] = sm_frag<B_KC>(st + SM_OPERAND + p * SM_PLANE, s, wn, lane);
            }
            // smallest terms first, the order of gemm_f32_kernel's split loop
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2], fb[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[0], acc, 0, 0, 0);
        }
    };
    int c = 0;
    for (; c + 1 < n_chunks; c += 2) {
        chunk(c, ra[0], rb[0]);
        chunk(c + 1, ra[1], rb[1]);
    }
    if (c < n_chunks) chunk(c, ra[0], rb[0]);

    // C/D map of a 32x32 tile: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    // dropout as in gemm_f32_kernel's epilogue: explicit multiplicative mask (1), or keep-1/2 from the Philox word of element
    // gm * N + gn (2; the big kernel's quad-shared evaluation draws the same bits)
    const int gn = n0 + wn + (lane & 31);
    const float bv = (bias && gn < N) ? bias[gn] : 0.f;
    if (drop_mode == 2) offset = philox_offset(offset);
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) {
        const int gm = m0 + wm + (rg & 3) + 8 * (rg >> 2) + 4 * (lane >> 5);
        if (gm < M && gn < N) {
            float v = acc[rg] + bv;
            if (ACT == 1) v = v > 0.f ? v : expm1f(v);
            if (ACT == 2) v = fmaxf(v, 0.f);
            if (drop_mode == 1) v *= drop_mask[(size_t)gm * N + gn];
            else if (drop_mode == 2) v = (philox_bits(seed, offset, (uint64_t)gm * N + gn) & 1u) ? v * 2.f : 0.f;
            C[(size_t)gm * ldc + gn] = v;
        }
    }
}
