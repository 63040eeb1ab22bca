// TuckER pieces that are not plain GEMMs (tucker.py:69-84, :97-127):
//   * the batched vector-matrix contraction x[b, :] . W_b[:, :] with hidden_dropout1 fused (the [B, d, d] tensor
//     is read exactly once, forward and backward) -- HBM-bound, d*d*4 bytes per row of the batch;
//   * sigmoid + BCELoss against the reference's smoothed "union of the batch's positives" target, fused with its
//     gradient (one pass over the [B, N] logits instead of five in the reference).
// The dense products (r . W, x . E^T and their backward) run on gemm.hip.
#include "hw_common.h"

namespace hw {

// out[b, j] = sum_i x[b, i] * w[b, i, j] * mask[b, i, j]        (mask nullable; mask holds 0 or 1/(1-p))
__global__ __launch_bounds__(256) void tucker_contract_fwd_kernel(const float* __restrict__ x,
                                                                  const float* __restrict__ w,
                                                                  const float* __restrict__ mask, int d,
                                                                  float* __restrict__ out) {
    const int b = blockIdx.x;
    const float* __restrict__ wb = w + (size_t)b * d * d;
    const float* __restrict__ mb = mask ? mask + (size_t)b * d * d : nullptr;
    const float* __restrict__ xb = x + (size_t)b * d;
    for (int j = threadIdx.x; j < d; j += 256) {
        float acc = 0.f;
        if (mb) {
#pragma unroll 4
            for (int i = 0; i < d; ++i) acc = fmaf(xb[i], wb[(size_t)i * d + j] * mb[(size_t)i * d + j], acc);
        } else {
#pragma unroll 4
            for (int i = 0; i < d; ++i) acc = fmaf(xb[i], wb[(size_t)i * d + j], acc);
        }
        out[(size_t)b * d + j] = acc;
    }
}

// gx[b, i] = sum_j go[b, j] w[b, i, j] m[b, i, j];   gw[b, i, j] = x[b, i] go[b, j] m[b, i, j]
__global__ __launch_bounds__(256) void tucker_contract_bwd_kernel(const float* __restrict__ x,
                                                                  const float* __restrict__ w,
                                                                  const float* __restrict__ mask,
                                                                  const float* __restrict__ go, int d,
                                                                  float* __restrict__ gx, float* __restrict__ gw) {
    const int b = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float* __restrict__ wb = w + (size_t)b * d * d;
    const float* __restrict__ mb = mask ? mask + (size_t)b * d * d : nullptr;
    const float* __restrict__ gob = go + (size_t)b * d;
    for (int i = wave; i < d; i += 4) {
        const float xi = x[(size_t)b * d + i];
        float acc = 0.f;
        for (int j = lane; j < d; j += 64) {
            const float m = mb ? mb[(size_t)i * d + j] : 1.f;
            const float g = gob[j] * m;
            acc = fmaf(g, wb[(size_t)i * d + j], acc);
            if (gw) gw[(size_t)b * d * d + (size_t)i * d + j] = xi * g;
        }
        acc = group_sum<64>(acc);
        if (gx && lane == 0) gx[(size_t)b * d + i] = acc;
    }
}

// ---- sigmoid + BCELoss(mean) against t[b, c] = scale * flag[c] + add  ------------------------------------------
// Arithmetic order of the reference: p = sigmoid(x) in fp32, loss terms max(log(p), -100) and max(log(1 - p), -100)
// (torch BCELoss), gradient (p - t) / max((1 - p) p, 1e-12) * p (1 - p) / numel (binary_cross_entropy_backward
// followed by sigmoid_backward).  pred (nullable) receives p; glogits (nullable) the gradient.
constexpr int TB_BLOCKS = 1024;

__global__ __launch_bounds__(256) void tucker_bce_kernel(const float* __restrict__ logits, int64_t n_rows, int64_t n_cols,
                                                         const uint8_t* __restrict__ flag, float scale, float add,
                                                         float inv_numel, float* __restrict__ pred,
                                                         float* __restrict__ glogits, float* __restrict__ partial) {
    __shared__ float red[4];
    const int64_t total = n_rows * n_cols;
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t c = i % n_cols;
        const float x = logits[i];
        const float p = 1.f / (1.f + expf(-x));
        const float t = flag ? fmaf(scale, (float)flag[c], add) : add;
        const float lp = fmaxf(logf(p), -100.f), l1p = fmaxf(logf(1.f - p), -100.f);
        acc -= t * lp + (1.f - t) * l1p;
        if (pred) pred[i] = p;
        if (glogits) {
            const float pq = (1.f - p) * p;
            glogits[i] = (p - t) / fmaxf(pq, 1e-12f) * pq * inv_numel;
        }
    }
    acc = group_sum<64>(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void tucker_bce_finish_kernel(const float* __restrict__ partial, int n, float inv_numel,
                                                                float* __restrict__ loss) {
    __shared__ float red[4];
    float v = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) v += partial[i];
    v = group_sum<64>(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = ((red[0] + red[1]) + (red[2] + red[3])) * inv_numel;
}

__global__ void sigmoid_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = 1.f / (1.f + expf(-x[i]));
}

}  // namespace hw

using namespace hw;

extern "C" int hw_tucker_contract(const float* x, const float* w, const float* mask, int64_t batch, int32_t dim,
                                  float* out, void* stream) {
    if (!x || !w || !out || batch < 0 || dim < 1 || batch > INT32_MAX) return HW_ERR_INVALID_ARG;
    if (batch == 0) return HW_OK;
    HW_CLEAR_ERR();
    hipLaunchKernelGGL(tucker_contract_fwd_kernel, dim3((unsigned)batch), dim3(256), 0, (hipStream_t)stream, x, w, mask,
                       dim, out);
    return HW_LAUNCH_RC();
}

extern "C" int hw_tucker_contract_grad(const float* x, const float* w, const float* mask, const float* gout,
                                       int64_t batch, int32_t dim, float* gx, float* gw, void* stream) {
    if (!x || !w || !gout || batch < 0 || dim < 1 || batch > INT32_MAX) return HW_ERR_INVALID_ARG;
    if (batch == 0) return HW_OK;
    HW_CLEAR_ERR();
    hipLaunchKernelGGL(tucker_contract_bwd_kernel, dim3((unsigned)batch), dim3(256), 0, (hipStream_t)stream, x, w, mask,
                       gout, dim, gx, gw);
    return HW_LAUNCH_RC();
}

extern "C" int64_t hw_tucker_bce_workspace_floats(void) { return TB_BLOCKS; }

extern "C" int hw_tucker_sigmoid_bce(const float* logits, int64_t n_rows, int64_t n_cols, const uint8_t* pos_flag,
                                     float target_scale, float target_add, float* loss_out, float* pred,
                                     float* glogits, float* workspace, void* stream) {
    if (!logits || !loss_out || !workspace || n_rows < 1 || n_cols < 1) return HW_ERR_INVALID_ARG;
    HW_CLEAR_ERR();
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = n_rows * n_cols;
    int64_t grid = (total + 255) / 256;
    if (grid > TB_BLOCKS) grid = TB_BLOCKS;
    const float inv = 1.0f / (float)total;
    hipLaunchKernelGGL(tucker_bce_kernel, dim3((unsigned)grid), dim3(256), 0, st, logits, n_rows, n_cols, pos_flag,
                       target_scale, target_add, inv, pred, glogits, workspace);
    hipLaunchKernelGGL(tucker_bce_finish_kernel, dim3(1), dim3(256), 0, st, workspace, (int)grid, inv, loss_out);
    return HW_LAUNCH_RC();
}

extern "C" int hw_sigmoid(const float* x, int64_t n, float* y, void* stream) {
    if (!x || !y || n < 0) return HW_ERR_INVALID_ARG;
    if (n == 0) return HW_OK;
    HW_CLEAR_ERR();
    int64_t grid = (n + 255) / 256;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(sigmoid_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, n, y);
    return HW_LAUNCH_RC();
}
