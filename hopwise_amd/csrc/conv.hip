// Single-input-channel 2-D convolution (stride 1, no padding) for the two convolutional scorers:
//   ConvE  (conve.py:63,75-80): Conv2d(1, 32, (3, 3)) over the stacked [2*d1, d2] head / relation image
//   ConvKB (convkb.py:52,65-72): Conv2d(1, T, (k, 3)) over the [d, 3] (h, r, t) matrix
// out[b, c, i, j] = bias[c] + sum_{u, v} w[c, u, v] x[b, i + u, j + v].
// A few hundred MFLOP per batch next to the scorers' GEMMs: one thread per output (or input) element, weights through
// the scalar cache; the weight gradient is a two-stage fixed-order reduction (no float atomics -> deterministic).
#include "hw_common.h"

namespace hw {

constexpr int CONV_MAX_TAPS = 32;  // kh * kw

__global__ __launch_bounds__(256) void conv1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, int64_t B, int C, int H, int W,
                                                        int kh, int kw, float* __restrict__ out) {
    const int Ho = H - kh + 1, Wo = W - kw + 1;
    const int64_t total = B * C * Ho * Wo;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int j = (int)(idx % Wo), i = (int)((idx / Wo) % Ho), c = (int)((idx / ((int64_t)Wo * Ho)) % C);
    const int64_t b = idx / ((int64_t)Wo * Ho * C);
    const float* __restrict__ xb = x + b * H * W;
    const float* __restrict__ wc = w + (size_t)c * kh * kw;
    float acc = bias ? bias[c] : 0.f;
    for (int u = 0; u < kh; ++u)
        for (int v = 0; v < kw; ++v) acc = fmaf(wc[u * kw + v], xb[(i + u) * W + (j + v)], acc);
    out[idx] = acc;
}

// dx[b, p, q] = sum_c sum_{u, v} w[c, u, v] dy[b, c, p - u, q - v]
__global__ __launch_bounds__(256) void conv1_bwd_x_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                          int64_t B, int C, int H, int W, int kh, int kw,
                                                          float* __restrict__ dx) {
    const int Ho = H - kh + 1, Wo = W - kw + 1;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * H * W) return;
    const int q = (int)(idx % W), p = (int)((idx / W) % H);
    const int64_t b = idx / ((int64_t)W * H);
    float acc = 0.f;
    for (int c = 0; c < C; ++c) {
        const float* __restrict__ dyc = dy + (b * C + c) * (int64_t)Ho * Wo;
        const float* __restrict__ wc = w + (size_t)c * kh * kw;
        for (int u = 0; u < kh; ++u) {
            const int i = p - u;
            if (i < 0 || i >= Ho) continue;
            for (int v = 0; v < kw; ++v) {
                const int j = q - v;
                if (j < 0 || j >= Wo) continue;
                acc = fmaf(wc[u * kw + v], dyc[i * Wo + j], acc);
            }
        }
    }
    dx[idx] = acc;
}

// partial[s, c, 0 .. taps) = sum over the samples of split s of dy[b, c, i, j] x[b, i + u, j + v]; slot `taps` = sum dy
__global__ __launch_bounds__(256) void conv1_bwd_w_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          int64_t B, int C, int H, int W, int kh, int kw, int n_splits,
                                                          float* __restrict__ partial) {
    __shared__ float red[4][CONV_MAX_TAPS + 1];
    const int c = blockIdx.x, s = blockIdx.y;
    const int Ho = H - kh + 1, Wo = W - kw + 1, taps = kh * kw;
    const int64_t per = (B + n_splits - 1) / n_splits;
    const int64_t b0 = (int64_t)s * per, b1 = min(B, b0 + per);
    float acc[CONV_MAX_TAPS + 1];
#pragma unroll
    for (int t = 0; t <= CONV_MAX_TAPS; ++t) acc[t] = 0.f;
    const int64_t n = (b1 > b0 ? b1 - b0 : 0) * Ho * Wo;
    for (int64_t e = threadIdx.x; e < n; e += 256) {
        const int j = (int)(e % Wo), i = (int)((e / Wo) % Ho);
        const int64_t b = b0 + e / ((int64_t)Wo * Ho);
        const float g = dy[((b * C + c) * Ho + i) * (int64_t)Wo + j];
        const float* __restrict__ xb = x + b * H * W + i * W + j;
        acc[CONV_MAX_TAPS] += g;
#pragma unroll
        for (int t = 0; t < CONV_MAX_TAPS; ++t)
            if (t < taps) acc[t] = fmaf(g, xb[(t / kw) * W + (t % kw)], acc[t]);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t <= CONV_MAX_TAPS; ++t) {
        const float v = group_sum<64>(acc[t]);
        if (lane == 0) red[wave][t] = v;
    }
    __syncthreads();
    if (threadIdx.x <= CONV_MAX_TAPS) {
        const int t = threadIdx.x;
        const float v = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
        if (t < taps) partial[((size_t)s * C + c) * (taps + 1) + t] = v;
        else if (t == CONV_MAX_TAPS) partial[((size_t)s * C + c) * (taps + 1) + taps] = v;
    }
}

__global__ void conv1_bwd_w_finish_kernel(const float* __restrict__ partial, int C, int taps, int n_splits,
                                          float* __restrict__ dw, float* __restrict__ dbias) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= C * (taps + 1)) return;
    const int c = idx / (taps + 1), t = idx % (taps + 1);
    float v = 0.f;
    for (int s = 0; s < n_splits; ++s) v += partial[((size_t)s * C + c) * (taps + 1) + t];
    if (t < taps) dw[(size_t)c * taps + t] = v;
    else if (dbias) dbias[c] = v;
}

}  // namespace hw

using namespace hw;

static int conv_args_ok(int64_t B, int C, int H, int W, int kh, int kw) {
    return B >= 0 && C >= 1 && kh >= 1 && kw >= 1 && H >= kh && W >= kw && kh * kw <= CONV_MAX_TAPS;
}

extern "C" int64_t hw_conv2d_1ch_workspace_floats(int32_t channels, int32_t kh, int32_t kw) {
    return (int64_t)64 * channels * (kh * kw + 1);
}

extern "C" int hw_conv2d_1ch_forward(const float* x, const float* w, const float* bias, int64_t batch, int32_t channels,
                                     int32_t height, int32_t width, int32_t kh, int32_t kw, float* out, void* stream) {
    if (!x || !w || !out || !conv_args_ok(batch, channels, height, width, kh, kw)) return HW_ERR_INVALID_ARG;
    const int64_t total = batch * channels * (height - kh + 1) * (int64_t)(width - kw + 1);
    if (total == 0) return HW_OK;
    if ((total + 255) / 256 > INT32_MAX) return HW_ERR_INVALID_ARG;
    HW_CLEAR_ERR();
    hipLaunchKernelGGL(conv1_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, w,
                       bias, batch, channels, height, width, kh, kw, out);
    return HW_LAUNCH_RC();
}

extern "C" int hw_conv2d_1ch_backward(const float* x, const float* w, const float* dy, int64_t batch, int32_t channels,
                                      int32_t height, int32_t width, int32_t kh, int32_t kw, float* dx, float* dw,
                                      float* dbias, float* workspace, void* stream) {
    if (!x || !w || !dy || !conv_args_ok(batch, channels, height, width, kh, kw)) return HW_ERR_INVALID_ARG;
    if (dw && !workspace) return HW_ERR_WORKSPACE;
    if (batch == 0) return HW_OK;
    HW_CLEAR_ERR();
    hipStream_t st = (hipStream_t)stream;
    if (dx) {
        const int64_t total = batch * height * (int64_t)width;
        hipLaunchKernelGGL(conv1_bwd_x_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dy, w, batch,
                           channels, height, width, kh, kw, dx);
    }
    if (dw) {
        const int taps = kh * kw;
        int splits = (int)(batch < 64 ? batch : 64);
        hipLaunchKernelGGL(conv1_bwd_w_kernel, dim3((unsigned)channels, (unsigned)splits), dim3(256), 0, st, dy, x, batch,
                           channels, height, width, kh, kw, splits, workspace);
        const int n = channels * (taps + 1);
        hipLaunchKernelGGL(conv1_bwd_w_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, workspace,
                           channels, taps, splits, dw, dbias);
    }
    return HW_LAUNCH_RC();
}
