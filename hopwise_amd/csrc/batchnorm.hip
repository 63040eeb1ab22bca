// nn.BatchNorm2d in training mode over NCHW feature maps with few channels (ConvE / ConvKB: conve.py:62-66,
// convkb.py:51-56 -- BatchNorm2d(1) on the stacked input image, BatchNorm2d(32 / T) on the convolution output).
// The library path (MIOpen spatial BN) spends three launches and ~450 us on the 85 MB of ConvE's [2048, 32, 18, 18]
// activations -- a third of the model's step; this is two HBM-bound passes forward and two backward.
//
//   forward : mean_c, var_c (biased) over N x H x W;  y = (x - mean) rstd gamma + beta;
//             running_mean = (1 - m) running_mean + m mean;  running_var likewise with the UNBIASED variance
//   backward: dgamma_c = sum dy xhat, dbeta_c = sum dy,  dx = gamma rstd (dy - dbeta / n - xhat dgamma / n)
// Channel sums are accumulated in float64 (per-thread, then per-block partials, then a fixed-order final sum) so that
// the statistics do not depend on how the reduction is split.
#include "hw_common.h"

namespace hw {

constexpr int BN_SPLITS = 64;  // row-range splits per channel in the reduction passes

__device__ __forceinline__ double block_sum_f64(double v, double* red) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    __syncthreads();  // red may still be read from a previous call
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// partial[c][s] = (sum a, sum a * b') over the samples of split s; MODE 0: (x, x^2); MODE 1: (dy, dy * xhat)
template <int MODE>
__global__ __launch_bounds__(256) void bn2d_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, int64_t N, int C, int HW,
                                                           double* __restrict__ partial) {
    __shared__ double red[4];
    const int c = blockIdx.x, s = blockIdx.y;
    const int64_t per = (N + BN_SPLITS - 1) / BN_SPLITS;
    const int64_t n0 = (int64_t)s * per, n1 = min(N, n0 + per);
    double a = 0.0, b = 0.0;
    float mu = 0.f, rs = 0.f;
    if (MODE == 1) mu = mean[c], rs = rstd[c];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int64_t n = n0 + wave; n < n1; n += 4) {  // one wave per (sample, channel) plane: no index division
        const size_t base = ((size_t)n * C + c) * HW;
        for (int i = lane; i < HW; i += 64) {
            if (MODE == 0) {
                const float v = x[base + i];
                a += v;
                b += (double)v * v;
            } else {
                const float g = dy[base + i];
                a += g;
                b += (double)g * ((x[base + i] - mu) * rs);
            }
        }
    }
    a = block_sum_f64(a, red);
    b = block_sum_f64(b, red);
    if (threadIdx.x == 0) {
        partial[((size_t)c * BN_SPLITS + s) * 2] = a;
        partial[((size_t)c * BN_SPLITS + s) * 2 + 1] = b;
    }
}

// statistics from the partials (every block recomputes its channel's: 64 additions), then the normalisation
__global__ __launch_bounds__(256) void bn2d_forward_kernel(const float* __restrict__ x, const double* __restrict__ partial,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           int64_t N, int C, int HW, float eps, float momentum,
                                                           float* __restrict__ running_mean,
                                                           float* __restrict__ running_var, float* __restrict__ save_mean,
                                                           float* __restrict__ save_rstd, float* __restrict__ y) {
    const int c = blockIdx.x, s = blockIdx.y;
    double sa = 0.0, sb = 0.0;
    for (int q = 0; q < BN_SPLITS; ++q) {
        sa += partial[((size_t)c * BN_SPLITS + q) * 2];
        sb += partial[((size_t)c * BN_SPLITS + q) * 2 + 1];
    }
    const double cnt = (double)N * HW;
    const double mean = sa / cnt;
    double var = sb / cnt - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    const float mu = (float)mean, rs = (float)(1.0 / sqrt(var + (double)eps));
    if (s == 0 && threadIdx.x == 0) {
        save_mean[c] = mu;
        save_rstd[c] = rs;
        if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
        if (running_var) {
            const double unbiased = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
    }
    const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
    const float scale = rs * g, shift = bt - mu * rs * g;
    const int64_t per = (N + BN_SPLITS - 1) / BN_SPLITS;
    const int64_t n0 = (int64_t)s * per, n1 = min(N, n0 + per);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int64_t n = n0 + wave; n < n1; n += 4) {
        const size_t base = ((size_t)n * C + c) * HW;
        for (int i = lane; i < HW; i += 64) y[base + i] = fmaf(x[base + i], scale, shift);
    }
}

__global__ __launch_bounds__(256) void bn2d_backward_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            const double* __restrict__ partial,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            int64_t N, int C, int HW, float* __restrict__ dx,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x, s = blockIdx.y;
    double sa = 0.0, sb = 0.0;
    for (int q = 0; q < BN_SPLITS; ++q) {
        sa += partial[((size_t)c * BN_SPLITS + q) * 2];
        sb += partial[((size_t)c * BN_SPLITS + q) * 2 + 1];
    }
    if (s == 0 && threadIdx.x == 0) {
        if (dbeta) dbeta[c] = (float)sa;
        if (dgamma) dgamma[c] = (float)sb;
    }
    if (!dx) return;
    const double cnt = (double)N * HW;
    const float mu = mean[c], rs = rstd[c], g = gamma ? gamma[c] : 1.f;
    const float k0 = g * rs, mdy = (float)(sa / cnt), mdyx = (float)(sb / cnt);
    const int64_t per = (N + BN_SPLITS - 1) / BN_SPLITS;
    const int64_t n0 = (int64_t)s * per, n1 = min(N, n0 + per);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int64_t n = n0 + wave; n < n1; n += 4) {
        const size_t base = ((size_t)n * C + c) * HW;
        for (int i = lane; i < HW; i += 64) {
            const float xh = (x[base + i] - mu) * rs;
            dx[base + i] = k0 * (dy[base + i] - mdy - xh * mdyx);
        }
    }
}

// ---- nn.BatchNorm1d over [N, C] rows (TuckER bn0 / bn1, ConvE bn2, CAFE's relation modules) -----------------------------
// Same statistics and update rules as above with HW = 1, but the layout asks for a different mapping: thread = channel
// (64 consecutive channels per block: coalesced row reads), 4 row phases per block, BN1_SPLITS row ranges per channel group.
constexpr int BN1_SPLITS = 32;

// partial[c][s] = (sum a, sum a * b'); MODE 0: (x, x^2); MODE 1: (dy, dy * xhat)
template <int MODE>
__global__ __launch_bounds__(256) void bn1d_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           int64_t N, int C, double* __restrict__ partial) {
    __shared__ double ra[4][64], rb[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx, s = blockIdx.y;
    const int64_t per = (N + BN1_SPLITS - 1) / BN1_SPLITS;
    const int64_t n0 = (int64_t)s * per, n1 = min(N, n0 + per);
    double a = 0.0, b = 0.0;
    if (c < C) {
        float mu = 0.f, rs = 0.f;
        if (MODE == 1) mu = mean[c], rs = rstd[c];
        // (unrolled: the rows are 16 dependent-looking loads otherwise -- one round trip each; the sums keep their order)
#pragma unroll 8
        for (int64_t n = n0 + ry; n < n1; n += 4) {
            if (MODE == 0) {
                const float v = x[(size_t)n * C + c];
                a += v;
                b += (double)v * v;
            } else {
                const float g = dy[(size_t)n * C + c];
                a += g;
                b += (double)g * ((x[(size_t)n * C + c] - mu) * rs);
            }
        }
    }
    ra[ry][cx] = a, rb[ry][cx] = b;
    __syncthreads();
    if (ry == 0 && c < C) {
        partial[((size_t)c * BN1_SPLITS + s) * 2] = (ra[0][cx] + ra[1][cx]) + (ra[2][cx] + ra[3][cx]);
        partial[((size_t)c * BN1_SPLITS + s) * 2 + 1] = (rb[0][cx] + rb[1][cx]) + (rb[2][cx] + rb[3][cx]);
    }
}

// MODE 0: training forward (statistics from the partials, running stats updated by split 0);  MODE 1: backward;
// MODE 2: evaluation forward (running statistics)
template <int MODE>
__global__ __launch_bounds__(256) void bn1d_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         const double* __restrict__ partial, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, int64_t N, int C, float eps,
                                                         float momentum, float* __restrict__ running_mean,
                                                         float* __restrict__ running_var, float* __restrict__ save_mean,
                                                         float* __restrict__ save_rstd, float* __restrict__ out,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx, s = blockIdx.y;
    if (c >= C) return;
    const int64_t per = (N + BN1_SPLITS - 1) / BN1_SPLITS;
    const int64_t n0 = (int64_t)s * per, n1 = min(N, n0 + per);
    double sa = 0.0, sb = 0.0;
    if (MODE != 2)
#pragma unroll
        for (int q = 0; q < BN1_SPLITS; ++q) {
            sa += partial[((size_t)c * BN1_SPLITS + q) * 2];
            sb += partial[((size_t)c * BN1_SPLITS + q) * 2 + 1];
        }
    const double cnt = (double)N;
    const float g = gamma ? gamma[c] : 1.f;
    if (MODE == 0) {
        const double mean = sa / cnt;
        double var = sb / cnt - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        const float mu = (float)mean, rs = (float)(1.0 / sqrt(var + (double)eps));
        if (s == 0 && ry == 0) {
            save_mean[c] = mu;
            save_rstd[c] = rs;
            if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
            if (running_var) {
                const double unbiased = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
            }
        }
        const float scale = rs * g, shift = (beta ? beta[c] : 0.f) - mu * rs * g;
#pragma unroll 8
        for (int64_t n = n0 + ry; n < n1; n += 4) out[(size_t)n * C + c] = fmaf(x[(size_t)n * C + c], scale, shift);
    } else if (MODE == 2) {
        const float rs = 1.f / sqrtf(running_var[c] + eps), mu = running_mean[c];
        const float scale = rs * g, shift = (beta ? beta[c] : 0.f) - mu * rs * g;
#pragma unroll 8
        for (int64_t n = n0 + ry; n < n1; n += 4) out[(size_t)n * C + c] = fmaf(x[(size_t)n * C + c], scale, shift);
    } else {
        if (s == 0 && ry == 0) {
            if (dbeta) dbeta[c] = (float)sa;
            if (dgamma) dgamma[c] = (float)sb;
        }
        if (!out) return;
        const float mu = save_mean[c], rs = save_rstd[c];
        const float k0 = g * rs, mdy = (float)(sa / cnt), mdyx = (float)(sb / cnt);
#pragma unroll 8
        for (int64_t n = n0 + ry; n < n1; n += 4) {
            const float xh = (x[(size_t)n * C + c] - mu) * rs;
            out[(size_t)n * C + c] = k0 * (dy[(size_t)n * C + c] - mdy - xh * mdyx);
        }
    }
}

// ---- nn.Dropout on a flat tensor: element i is kept with probability 1 - p (scaled by 1 / (1 - p)) from Philox word i of the
// (seed, offset) stream; the backward is the same call on the upstream gradient.
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, int64_t n, float p, uint64_t seed,
                                                      uint64_t offset, float* __restrict__ y) {
    const float inv_keep = 1.f / (1.f - p);
    const int64_t nq = (n + 3) >> 2;
    offset = philox_offset(offset);
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < nq; q += (int64_t)gridDim.x * 256) {
        uint32_t w[4];
        philox_bits4(seed, offset, (uint64_t)q, w);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t i = 4 * q + k;
            if (i < n) y[i] = ((float)(w[k] >> 8) * (1.0f / 16777216.0f) >= p) ? x[i] * inv_keep : 0.f;
        }
    }
}

}  // namespace hw

using namespace hw;

extern "C" int64_t hw_batchnorm2d_workspace_doubles(int32_t channels) {
    return channels < 0 ? 0 : (int64_t)channels * BN_SPLITS * 2;
}

extern "C" int hw_batchnorm2d_forward(const float* x, int64_t batch, int32_t channels, int32_t hw, const float* gamma,
                                      const float* beta, float eps, float momentum, float* running_mean,
                                      float* running_var, float* save_mean, float* save_rstd, float* y, double* workspace,
                                      void* stream) {
    if (!x || !y || !save_mean || !save_rstd || !workspace || batch < 1 || channels < 1 || hw < 1) return HW_ERR_INVALID_ARG;
    if (channels > 65535 || !(eps >= 0.f)) return HW_ERR_INVALID_ARG;
    HW_CLEAR_ERR();
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)channels, BN_SPLITS);
    hipLaunchKernelGGL((bn2d_partial_kernel<0>), grid, dim3(256), 0, st, x, nullptr, nullptr, nullptr, batch, channels, hw,
                       workspace);
    hipLaunchKernelGGL(bn2d_forward_kernel, grid, dim3(256), 0, st, x, workspace, gamma, beta, batch, channels, hw, eps,
                       momentum, running_mean, running_var, save_mean, save_rstd, y);
    return HW_LAUNCH_RC();
}

extern "C" int hw_batchnorm2d_backward(const float* x, const float* dy, int64_t batch, int32_t channels, int32_t hw,
                                       const float* gamma, const float* save_mean, const float* save_rstd, float* dx,
                                       float* dgamma, float* dbeta, double* workspace, void* stream) {
    if (!x || !dy || !save_mean || !save_rstd || !workspace || batch < 1 || channels < 1 || hw < 1) return HW_ERR_INVALID_ARG;
    if (channels > 65535) return HW_ERR_INVALID_ARG;
    HW_CLEAR_ERR();
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)channels, BN_SPLITS);
    hipLaunchKernelGGL((bn2d_partial_kernel<1>), grid, dim3(256), 0, st, x, dy, save_mean, save_rstd, batch, channels, hw,
                       workspace);
    hipLaunchKernelGGL(bn2d_backward_kernel, grid, dim3(256), 0, st, x, dy, workspace, gamma, save_mean, save_rstd, batch,
                       channels, hw, dx, dgamma, dbeta);
    return HW_LAUNCH_RC();
}

extern "C" int64_t hw_batchnorm1d_workspace_doubles(int32_t channels) {
    return channels < 0 ? 0 : (int64_t)channels * BN1_SPLITS * 2;
}

extern "C" int hw_batchnorm1d_forward(const float* x, int64_t batch, int32_t channels, const float* gamma, const float* beta,
                                      float eps, float momentum, float* running_mean, float* running_var, float* save_mean,
                                      float* save_rstd, float* y, double* workspace, int32_t training, void* stream) {
    if (!x || !y || batch < 1 || channels < 1 || !(eps >= 0.f)) return HW_ERR_INVALID_ARG;
    if (training ? (!save_mean || !save_rstd || !workspace) : (!running_mean || !running_var)) return HW_ERR_INVALID_ARG;
    HW_CLEAR_ERR();
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((channels + 63) / 64), BN1_SPLITS);
    if (training) {
        hipLaunchKernelGGL((bn1d_partial_kernel<0>), grid, dim3(256), 0, st, x, nullptr, nullptr, nullptr, batch, channels,
                           workspace);
        hipLaunchKernelGGL((bn1d_apply_kernel<0>), grid, dim3(256), 0, st, x, nullptr, workspace, gamma, beta, batch, channels,
                           eps, momentum, running_mean, running_var, save_mean, save_rstd, y, nullptr, nullptr);
    } else {
        hipLaunchKernelGGL((bn1d_apply_kernel<2>), grid, dim3(256), 0, st, x, nullptr, nullptr, gamma, beta, batch, channels,
                           eps, momentum, running_mean, running_var, nullptr, nullptr, y, nullptr, nullptr);
    }
    return HW_LAUNCH_RC();
}

extern "C" int hw_batchnorm1d_backward(const float* x, const float* dy, int64_t batch, int32_t channels, const float* gamma,
                                       const float* save_mean, const float* save_rstd, float* dx, float* dgamma,
                                       float* dbeta, double* workspace, void* stream) {
    if (!x || !dy || !save_mean || !save_rstd || !workspace || batch < 1 || channels < 1) return HW_ERR_INVALID_ARG;
    HW_CLEAR_ERR();
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((channels + 63) / 64), BN1_SPLITS);
    hipLaunchKernelGGL((bn1d_partial_kernel<1>), grid, dim3(256), 0, st, x, dy, save_mean, save_rstd, batch, channels,
                       workspace);
    hipLaunchKernelGGL((bn1d_apply_kernel<1>), grid, dim3(256), 0, st, x, dy, workspace, gamma, nullptr, batch, channels, 0.f,
                       0.f, nullptr, nullptr, const_cast<float*>(save_mean), const_cast<float*>(save_rstd), dx, dgamma, dbeta);
    return HW_LAUNCH_RC();
}

extern "C" int hw_dropout(const float* x, int64_t n, float drop_p, uint64_t seed, uint64_t offset, float* y, void* stream) {
    if (!x || !y || n < 0 || !(drop_p >= 0.f && drop_p < 1.f)) return HW_ERR_INVALID_ARG;
    if (n == 0) return HW_OK;
    HW_CLEAR_ERR();
    int64_t grid = ((n + 3) / 4 + 255) / 256;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, n, drop_p, seed, offset, y);
    return HW_LAUNCH_RC();
}
