// Device-side helpers shared by the gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hopwise_amd.h"

#define HW_WAVE 64
#define HW_MAX_DIM 1024

// hipGetLastError() reports the last error of ANY earlier runtime call on this thread (torch leaves benign
// ones behind), so every launcher clears it first and reads it back right after its own launches.
#define HW_CLEAR_ERR() ((void)hipGetLastError())
#define HW_LAUNCH_RC() (hipGetLastError() == hipSuccess ? HW_OK : HW_ERR_LAUNCH)

namespace hw {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- sub-wave butterfly sum over G lanes (G power of two, 1..64); every lane gets the total ----
template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int m = G / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, HW_WAVE);
    return v;
}

// clamp an id into [0, n) and count violations (ids outside the table never reach a load address)
__device__ __forceinline__ int clamp_row(int64_t id, int64_t n, int32_t* oob) {
    if (id < 0 || id >= n) {
        if (oob) atomicAdd(oob, 1);
        id = id < 0 ? 0 : n - 1;
    }
    return (int)id;
}

// ---- one embedding row spread over the G lanes of a sub-wave group -----------------------------
// VEC = true : dim % 4 == 0; lane gl owns float4 chunks gl, gl+G, ...  (NE/4 chunks)  -> 16-B loads
// VEC = false: any dim;      lane gl owns elements   gl, gl+G, ...     (NE elements)  -> 4-B loads
// Slots past the end of the row hold 0.
template <int G, int NE, bool VEC>
struct Row {
    float v[NE];

    __device__ __forceinline__ void load(const float* __restrict__ base, int dim, int gl) {
        if constexpr (VEC) {
            const int d4 = dim >> 2;
            const f32x4* __restrict__ p = reinterpret_cast<const f32x4*>(base);
#pragma unroll
            for (int c = 0; c < NE / 4; ++c) {
                const int q = gl + c * G;
                f32x4 x = {0.f, 0.f, 0.f, 0.f};
                if (q < d4) x = p[q];
                v[4 * c + 0] = x.x;
                v[4 * c + 1] = x.y;
                v[4 * c + 2] = x.z;
                v[4 * c + 3] = x.w;
            }
        } else {
#pragma unroll
            for (int c = 0; c < NE; ++c) {
                const int e = gl + c * G;
                v[c] = (e < dim) ? base[e] : 0.f;
            }
        }
    }

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

}  // namespace hw
