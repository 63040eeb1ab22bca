// Dense Adam over every table of a model in one launch (trainer.py:189-190: optim.Adam(params, lr, weight_decay) is
// the reference's default learner; its update touches every row each step, so the moments of rows the batch never
// gathered keep decaying -- reproduced, this is not a row-sparse optimizer).
//
//   t = ++step;  g += wd p;  m = m + (1 - b1)(g - m);  v = b2 v + (1 - b2) g^2
//   p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)          (torch/optim/adam.py _single_tensor_adam)
//
// HBM-bound: 16 B read (p, g, m, v) + 12 B written (p, m, v) per element, + 4 B when the gradient is cleared in the
// same pass (zero_grads), which replaces the per-table fills of the next zero_grad.  The step counter lives on the
// device so that the launch can be captured in a HIP graph.
#include "hw_common.h"

namespace hw {

constexpr int ADAM_CHUNK = 256 * 4 * 4;  // elements per block: 256 threads x 4 float4

__global__ void adam_bump_kernel(int64_t* step) { *step += 1; }

__global__ __launch_bounds__(256) void adam_multi_kernel(hw_adam_args a, double lr, double beta1, double beta2,
                                                         float eps, float wd, const int64_t* __restrict__ step,
                                                         int zero_grads) {
    // block -> (segment, chunk): at most HW_ADAM_MAX_SEGS prefix steps
    int64_t c = blockIdx.x;
    int s = 0;
    for (; s < a.n_segs; ++s) {
        const int64_t nc = (a.seg[s].n + ADAM_CHUNK - 1) / ADAM_CHUNK;
        if (c < nc) break;
        c -= nc;
    }
    if (s >= a.n_segs) return;
    const hw_adam_seg sg = a.seg[s];
    __shared__ float bias[2];
    if (threadIdx.x == 0) {  // two double-precision pow: once per block, not per lane
        const double t = (double)*step;
        bias[0] = (float)(lr / (1.0 - pow(beta1, t)));
        bias[1] = (float)(1.0 / sqrt(1.0 - pow(beta2, t)));
    }
    __syncthreads();
    const float step_size = bias[0], inv_sqrt_bc2 = bias[1];
    // 1 - beta in double first: (float)(1 - 0.999) is what torch's python-side scalars give, 1.f - 0.999f is 5e-5 off
    const float b2 = (float)beta2, omb1 = (float)(1.0 - beta1), omb2 = (float)(1.0 - beta2);
    const float decay = 1.f - (float)lr * wd;  // AdamW: p *= 1 - lr wd before the step (torch/optim/adamw.py)
    auto upd = [&](float& p, float& g, float& m, float& v) {
        if (zero_grads & 2) p *= decay;
        else if (wd != 0.f) g = fmaf(wd, p, g);
        m = fmaf(omb1, g - m, m);
        v = fmaf(omb2 * g, g, b2 * v);
        const float denom = fmaf(__builtin_amdgcn_sqrtf(v), inv_sqrt_bc2, eps);
        p -= step_size * (m / denom);
    };
    const int64_t base = c * ADAM_CHUNK;
    const bool vec = ((((uintptr_t)sg.p | (uintptr_t)sg.g | (uintptr_t)sg.m | (uintptr_t)sg.v) & 15) == 0);
    if (vec && base + ADAM_CHUNK <= sg.n) {
        f32x4* __restrict__ P = reinterpret_cast<f32x4*>(sg.p + base);
        f32x4* __restrict__ G = reinterpret_cast<f32x4*>(sg.g + base);
        f32x4* __restrict__ M = reinterpret_cast<f32x4*>(sg.m + base);
        f32x4* __restrict__ V = reinterpret_cast<f32x4*>(sg.v + base);
        f32x4 p[4], g[4], m[4], v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = threadIdx.x + 256 * u;
            p[u] = P[i];
            g[u] = G[i];
            m[u] = M[i];
            v[u] = V[i];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = threadIdx.x + 256 * u;
            float pp[4] = {p[u].x, p[u].y, p[u].z, p[u].w}, gg[4] = {g[u].x, g[u].y, g[u].z, g[u].w};
            float mm[4] = {m[u].x, m[u].y, m[u].z, m[u].w}, vv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) upd(pp[e], gg[e], mm[e], vv[e]);
            P[i] = f32x4{pp[0], pp[1], pp[2], pp[3]};
            M[i] = f32x4{mm[0], mm[1], mm[2], mm[3]};
            V[i] = f32x4{vv[0], vv[1], vv[2], vv[3]};
            if (zero_grads & 1) G[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        return;
    }
    for (int64_t i = base + threadIdx.x; i < base + ADAM_CHUNK && i < sg.n; i += 256) {
        float p = sg.p[i], g = sg.g[i], m = sg.m[i], v = sg.v[i];
        upd(p, g, m, v);
        sg.p[i] = p;
        sg.m[i] = m;
        sg.v[i] = v;
        if (zero_grads & 1) sg.g[i] = 0.f;
    }
}

}  // namespace hw

using namespace hw;

extern "C" int hw_adam_step(const hw_adam_args* args, double lr, double beta1, double beta2, double eps,
                            double weight_decay, int64_t* step, int32_t flags, void* stream) {
    const int32_t zero_grads = flags;  // bit 0: clear the gradients; bit 1: decoupled weight decay (AdamW)
    if (!args || !step || args->n_segs < 0 || args->n_segs > HW_ADAM_MAX_SEGS) return HW_ERR_INVALID_ARG;
    if (!(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0)) return HW_ERR_INVALID_ARG;
    int64_t blocks = 0;
    for (int s = 0; s < args->n_segs; ++s) {
        const hw_adam_seg& sg = args->seg[s];
        if (sg.n < 0 || (sg.n > 0 && (!sg.p || !sg.g || !sg.m || !sg.v))) return HW_ERR_INVALID_ARG;
        blocks += (sg.n + ADAM_CHUNK - 1) / ADAM_CHUNK;
    }
    if (blocks > INT32_MAX) return HW_ERR_INVALID_ARG;
    HW_CLEAR_ERR();
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(adam_bump_kernel, dim3(1), dim3(1), 0, st, step);
    if (blocks > 0)
        hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, st, *args, lr, beta1, beta2,
                           (float)eps, (float)weight_decay, step, (int)zero_grads);
    return HW_LAUNCH_RC();
}
