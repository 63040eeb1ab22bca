// Per-model arithmetic of the four gather-bound scorers, written once and shared by the grouped
// scorer, the full-sort scorer and the gradient kernel.
//
// Every model factors as   score(h, r, t) = finalize( sum_d  f(Q(h, r)_d, t_d) )
// where the "query" Q depends only on the (head, relation) pair, so a group of tails that share
// (h, r) -- one positive + K negatives, or all N candidates of a full sort -- pays for the head /
// relation rows (and RotatE's sincos) once.
//
//   TransE   Q0 = h + r                      f = ((Q0 - t) + eps)^2                 -sqrt(s)      transe.py:55-57
//   DistMult Q0 = h * r                      f = Q0 * t                             s             distmult.py:50-51
//   RotatE   Q0 = cos(th) h_re - sin(th) h_im
//            Q1 = cos(th) h_im + sin(th) h_re f = (Q0 - t_re)^2 + (Q1 - t_im)^2     margin-sqrt(s) rotate.py:61-69
//   ComplEx  Q0 = h_re r_re
//            Q1 = h_im r_re + h_re r_im - h_im r_im   f = Q0 t_re + Q1 t_im         s             complex.py:53-62
//            (the 4th term multiplies t_im, exactly as the reference does -- not the paper's t_re)
//   RESCAL   Q0 = h^T M_r  (M_r = relation row viewed [d, d])   f = Q0 * t          s             rescal.py:51-54
//            the query build streams the d x d matrix once per (h, r) task; the per-tail work is DistMult's
// Widened set, same factorisation:
//   TorusE   Q0 = frac(h) + frac(r)          f = min(x^2, 1 - x^2), x = Q0 - frac(t)   -4 s         toruse.py:68-79
//   TransH   c = 1 - sum(w) w; Q0 = h c + r, Q1 = c   f = ((Q0 - t Q1) + eps)^2          -sqrt(s)     transh.py:51-55,74-75
//   TransD   Q0 = h + rv (h.hv) + r, Q1 = rv; per tail sigma = t.tv (a first reduction, Scorer::pre)
//                                             f = ((Q0 - t - Q1 sigma) + eps)^2          -sqrt(s)     transd.py:89-94
//   Analogy  Q0 = h r, Q1 = h_re r_re - h_im r_im, Q2 = h_re r_im + h_im r_re   f = Q0 t + Q1 t_re + Q2 t_im   s   analogy.py:66-70
//   HolE     Q0[j] = sum_i h[i] r[(j - i) mod d]      f = Q0 t                            s            hole.py:53-61
#pragma once
#include "hw_common.h"

namespace hw {

template <int MODEL>
struct ModelTraits;
template <>
struct ModelTraits<HW_TRANSE> {
    static constexpr bool TWO_PASS = false;
    static constexpr int NH = 1, NR = 1, NT = 1, NQ = 1;
};
template <>
struct ModelTraits<HW_DISTMULT> {
    static constexpr bool TWO_PASS = false;
    static constexpr int NH = 1, NR = 1, NT = 1, NQ = 1;
};
template <>
struct ModelTraits<HW_RESCAL> {
    static constexpr bool TWO_PASS = false;
    static constexpr int NH = 1, NR = 1, NT = 1, NQ = 1;
};
template <>
struct ModelTraits<HW_ROTATE> {
    static constexpr bool TWO_PASS = false;
    static constexpr int NH = 2, NR = 1, NT = 2, NQ = 2;
};
template <>
struct ModelTraits<HW_COMPLEX> {
    static constexpr bool TWO_PASS = false;
    static constexpr int NH = 2, NR = 2, NT = 2, NQ = 2;
};

template <>
struct ModelTraits<HW_TORUSE> {
    static constexpr bool TWO_PASS = false;
    static constexpr int NH = 1, NR = 1, NT = 1, NQ = 1;
};
template <>
struct ModelTraits<HW_TRANSH> {
    static constexpr bool TWO_PASS = false;
    static constexpr int NH = 1, NR = 2, NT = 1, NQ = 2;
};
template <>
struct ModelTraits<HW_TRANSD> {
    static constexpr int NH = 2, NR = 2, NT = 2, NQ = 2;
    static constexpr bool TWO_PASS = true;
};
template <>
struct ModelTraits<HW_ANALOGY> {
    static constexpr bool TWO_PASS = false;
    static constexpr int NH = 3, NR = 3, NT = 3, NQ = 3;
};
template <>
struct ModelTraits<HW_HOLE> {
    static constexpr bool TWO_PASS = false;
    static constexpr int NH = 1, NR = 1, NT = 1, NQ = 1;
};

__device__ __forceinline__ float fracf_(float x) { return x - truncf(x); }  // torch.frac: sign of x

template <int MODEL, int G, int NE, bool VEC>
struct Scorer {
    using MT = ModelTraits<MODEL>;
    using R = Row<G, NE, VEC>;

    R q[MT::NQ];
    static constexpr bool SPLIT_BUILD = MODEL != HW_RESCAL && MODEL != HW_HOLE;  // build() = issue() + fold()

    // Load the head / relation rows of one (h, r) pair and fold them into the query.
    // KEEP = true additionally returns the raw rows (the gradient kernel needs them).
    template <bool KEEP>
    __device__ __forceinline__ void build(const hw_kge_desc& d, int h, int r, int gl, R* hraw, R* rraw) {
        if constexpr (MODEL == HW_RESCAL) {
            const float* __restrict__ hrow = d.head[0] + (size_t)h * d.dim;
            const float* __restrict__ mat = d.rel[0] + (size_t)r * d.dim * d.dim;
#pragma unroll
            for (int i = 0; i < NE; ++i) q[0].v[i] = 0.f;
#pragma unroll 4
            for (int i = 0; i < d.dim; ++i) {
                const float hi = hrow[i];  // one address per wave: broadcast load
                R m;
                m.load_raw(mat + (size_t)i * d.dim, d.dim, gl);
#pragma unroll
                for (int k = 0; k < NE; ++k) q[0].v[k] = fmaf(hi, m.v[k], q[0].v[k]);
            }
#pragma unroll
            for (int k = 0; k < NE; ++k) q[0].v[k] = R::slot_valid(k, d.dim, gl) ? q[0].v[k] : 0.f;
            if constexpr (KEEP) hraw[0].load(hrow, d.dim, gl);
            return;
        }
        if constexpr (MODEL == HW_HOLE) {
            // circular convolution: Q[j] = sum_i h[i] r[(j - i) mod d]; h[i] is a wave-uniform load, the rolled
            // relation row a gather inside one L1-resident row
            const float* __restrict__ hrow = d.head[0] + (size_t)h * d.dim;
            const float* __restrict__ rrow = d.rel[0] + (size_t)r * d.dim;
            if constexpr (VEC) {
                // 4 x 4 tiles: the lane's columns j0..j0+3 against h[i0..i0+3] need r[(j0 - i0) - 3 .. (j0 - i0) + 3],
                // i.e. two aligned 16-byte chunks of the relation row (dim % 4 == 0, so the wrap is chunk-aligned too);
                // 16 FMAs per 2 vector loads.  The head row address is wave-uniform (every caller builds one query per wave):
                // say so, and it becomes scalar loads.
                const f32x4* __restrict__ h4 =
                    reinterpret_cast<const f32x4*>(d.head[0] + (size_t)__builtin_amdgcn_readfirstlane(h) * d.dim);
                const f32x4* __restrict__ r4 = reinterpret_cast<const f32x4*>(rrow);
                const int d4 = d.dim >> 2;
#pragma unroll
                for (int c = 0; c < NE / 4; ++c) {
                    const int jc = gl + c * G;
                    const bool valid = jc < d4;
                    int a = valid ? jc : 0;  // chunk index of (j0 - i0) mod dim, walks down as i0 walks up
                    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
                    for (int m = 0; m < d4; ++m) {
                        const f32x4 hv = h4[m];
                        const f32x4 hi_ = r4[a];
                        const f32x4 lo_ = r4[a == 0 ? d4 - 1 : a - 1];
                        const float hh_[4] = {hv.x, hv.y, hv.z, hv.w};
                        const float win[8] = {lo_.x, lo_.y, lo_.z, lo_.w, hi_.x, hi_.y, hi_.z, hi_.w};  // r[(j0-i0) - 4 + t]
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                            for (int ii = 0; ii < 4; ++ii) acc[jj] = fmaf(hh_[ii], win[4 + jj - ii], acc[jj]);
                        a = a == 0 ? d4 - 1 : a - 1;
                    }
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) q[0].v[4 * c + jj] = valid ? acc[jj] : 0.f;
                }
                if constexpr (KEEP) {
                    hraw[0].load(hrow, d.dim, gl);
                    rraw[0].load(rrow, d.dim, gl);
                }
                return;
            }
            int cc[NE];
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                cc[k] = R::col(k, d.dim, gl);
                q[0].v[k] = 0.f;
            }
#pragma unroll 2
            for (int i = 0; i < d.dim; ++i) {
                const float hi = hrow[i];
#pragma unroll
                for (int k = 0; k < NE; ++k) {
                    int idx = (cc[k] < 0 ? 0 : cc[k]) - i;
                    idx += idx < 0 ? d.dim : 0;
                    q[0].v[k] = fmaf(hi, rrow[idx], q[0].v[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < NE; ++k) q[0].v[k] = cc[k] >= 0 ? q[0].v[k] : 0.f;
            if constexpr (KEEP) {
                hraw[0].load(hrow, d.dim, gl);
                rraw[0].load(rrow, d.dim, gl);
            }
            return;
        }
        if constexpr (SPLIT_BUILD) {
            R hh[MT::NH], rr[MT::NR];
            issue(d, h, r, gl, hh, rr);
            fold<KEEP>(d.dim, gl, hh, rr, hraw, rraw);
        }
    }

    // build() in two halves for the models whose query is an element-wise fold of the head and relation rows: issue() only
    // starts the row loads, fold() consumes them.  A kernel with other memory round trips ahead of its tail loop (the fused
    // sampler's key -> slice -> ids chain) puts those between the two, so that the rows travel under them.
    __device__ __forceinline__ void issue(const hw_kge_desc& d, int h, int r, int gl, R (&hh)[MT::NH], R (&rr)[MT::NR]) {
        const size_t ho = (size_t)h * d.dim, ro = (size_t)r * d.dim;
#pragma unroll
        for (int k = 0; k < MT::NH; ++k) hh[k].load_raw(d.head[k] + ho, d.dim, gl);
#pragma unroll
        for (int k = 0; k < MT::NR; ++k) rr[k].load_raw(d.rel[k] + ro, d.dim, gl);
    }
    template <bool KEEP>
    __device__ __forceinline__ void fold(int dim, int gl, R (&hh)[MT::NH], R (&rr)[MT::NR], R* hraw, R* rraw) {
        // the zeroing half of Row::load: padding slots of the raw loads hold copies of valid data
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const bool valid = R::slot_valid(i, dim, gl);
#pragma unroll
            for (int k = 0; k < MT::NH; ++k) hh[k].v[i] = valid ? hh[k].v[i] : 0.f;
#pragma unroll
            for (int k = 0; k < MT::NR; ++k) rr[k].v[i] = valid ? rr[k].v[i] : 0.f;
        }
        float red = 0.f;  // TransH: sum of the norm vector; TransD: h . h_vec
        if constexpr (MODEL == HW_TRANSH || MODEL == HW_TRANSD) {
#pragma unroll
            for (int i = 0; i < NE; ++i) red = MODEL == HW_TRANSH ? red + rr[1].v[i] : fmaf(hh[0].v[i], hh[1].v[i], red);
            red = group_sum<G>(red);
        }
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            if constexpr (MODEL == HW_TORUSE) {
                q[0].v[i] = fracf_(hh[0].v[i]) + fracf_(rr[0].v[i]);
            } else if constexpr (MODEL == HW_TRANSH) {
                const float c = 1.f - red * rr[1].v[i];
                q[0].v[i] = fmaf(hh[0].v[i], c, rr[0].v[i]);
                q[1].v[i] = c;
            } else if constexpr (MODEL == HW_TRANSD) {
                q[0].v[i] = (hh[0].v[i] + rr[1].v[i] * red) + rr[0].v[i];
                q[1].v[i] = rr[1].v[i];
            } else if constexpr (MODEL == HW_ANALOGY) {
                q[0].v[i] = hh[0].v[i] * rr[0].v[i];
                q[1].v[i] = hh[1].v[i] * rr[1].v[i] - hh[2].v[i] * rr[2].v[i];
                q[2].v[i] = hh[1].v[i] * rr[2].v[i] + hh[2].v[i] * rr[1].v[i];
            } else if constexpr (MODEL == HW_TRANSE) {
                q[0].v[i] = hh[0].v[i] + rr[0].v[i];
            } else if constexpr (MODEL == HW_DISTMULT) {
                q[0].v[i] = hh[0].v[i] * rr[0].v[i];
            } else if constexpr (MODEL == HW_ROTATE) {
                float sn, cs;
                __sincosf(rr[0].v[i], &sn, &cs);  // v_sin_f32 / v_cos_f32: abs error ~5e-7 on a unit-modulus factor
                q[0].v[i] = cs * hh[0].v[i] - sn * hh[1].v[i];
                q[1].v[i] = cs * hh[1].v[i] + sn * hh[0].v[i];
                if constexpr (KEEP) {  // hand back (cos, sin) instead of theta: what the backward consumes
                    rraw[0].v[i] = cs;
                    rraw[1].v[i] = sn;
                }
            } else {  // ComplEx
                q[0].v[i] = hh[0].v[i] * rr[0].v[i];
                q[1].v[i] = hh[1].v[i] * rr[0].v[i] + hh[0].v[i] * rr[1].v[i] - hh[1].v[i] * rr[1].v[i];
            }
        }
        if constexpr (KEEP) {
#pragma unroll
            for (int k = 0; k < MT::NH; ++k) hraw[k] = hh[k];
            if constexpr (MODEL != HW_ROTATE) {
#pragma unroll
                for (int k = 0; k < MT::NR; ++k) rraw[k] = rr[k];
            }
        }
    }

    // This lane's share of sum_d f(Q_d, t_d).  Query padding slots are 0.  With RAW = false the tail rows are
    // zero-padded too (Row::load); with RAW = true they were loaded with Row::load_raw and padding slots hold
    // junk, so models whose f(0, t) != 0 (TransE, RotatE) mask those slots here -- one select per load unit.
    // TWO_PASS models: this lane's share of the tail-only first reduction (TransD: t . t_vec); the kernels sum it over
    // the group and hand the total back to partial() as `aux`.
    template <bool RAW = false>
    __device__ __forceinline__ float pre(const R* t, int dim, int gl) const {
        float s = 0.f;
        if constexpr (MODEL == HW_TRANSD) {
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                const float v = t[0].v[i] * t[1].v[i];
                s += (!RAW || R::slot_valid(i, dim, gl)) ? v : 0.f;
            }
        }
        return s;
    }

    template <bool RAW = false>
    __device__ __forceinline__ float partial(const R* t, float eps, int dim, int gl, float aux = 0.f) const {
        float s = 0.f;
        constexpr int UNIT = VEC ? 4 : 1;
#pragma unroll
        for (int c = 0; c < NE / UNIT; ++c) {
            float sc = 0.f;
#pragma unroll
            for (int k = 0; k < UNIT; ++k) {
                const int i = c * UNIT + k;
                if constexpr (MODEL == HW_TRANSE) {
                    const float df = (q[0].v[i] - t[0].v[i]) + eps;
                    sc = fmaf(df, df, sc);
                } else if constexpr (MODEL == HW_TORUSE) {
                    const float x = q[0].v[i] - fracf_(t[0].v[i]);
                    const float x2 = x * x;
                    sc += fminf(x2, 1.f - x2);
                } else if constexpr (MODEL == HW_TRANSH) {
                    const float df = (q[0].v[i] - t[0].v[i] * q[1].v[i]) + eps;
                    sc = fmaf(df, df, sc);
                } else if constexpr (MODEL == HW_TRANSD) {
                    const float df = ((q[0].v[i] - t[0].v[i]) - q[1].v[i] * aux) + eps;
                    sc = fmaf(df, df, sc);
                } else if constexpr (MODEL == HW_ANALOGY) {
                    sc = fmaf(q[0].v[i], t[0].v[i], sc);
                    sc = fmaf(q[1].v[i], t[1].v[i], sc);
                    sc = fmaf(q[2].v[i], t[2].v[i], sc);
                } else if constexpr (MODEL == HW_DISTMULT || MODEL == HW_RESCAL || MODEL == HW_HOLE) {
                    sc = fmaf(q[0].v[i], t[0].v[i], sc);
                } else if constexpr (MODEL == HW_ROTATE) {
                    const float a = q[0].v[i] - t[0].v[i];
                    const float b = q[1].v[i] - t[1].v[i];
                    sc = fmaf(a, a, sc);
                    sc = fmaf(b, b, sc);
                } else {
                    sc = fmaf(q[0].v[i], t[0].v[i], sc);
                    sc = fmaf(q[1].v[i], t[1].v[i], sc);
                }
            }
            if constexpr (MODEL == HW_TRANSE || MODEL == HW_ROTATE || MODEL == HW_TORUSE || MODEL == HW_TRANSH ||
                          MODEL == HW_TRANSD) {
                // padding: f(0, junk) != 0 when RAW; f(0, 0) = eps^2 != 0 for the eps models even when zero-padded
                if (RAW || MODEL == HW_TRANSE || MODEL == HW_TRANSH || MODEL == HW_TRANSD)
                    sc = R::slot_valid(c * UNIT, dim, gl) ? sc : 0.f;
            }
            s += sc;
        }
        return s;
    }

    __device__ __forceinline__ static float finalize(float s, float margin) {
        // v_sqrt_f32 (1 ulp) instead of the correctly rounded libm sequence: ~15 instructions fewer per store
        if constexpr (MODEL == HW_TRANSE || MODEL == HW_TRANSH || MODEL == HW_TRANSD) return -__builtin_amdgcn_sqrtf(s);
        if constexpr (MODEL == HW_TORUSE) return -(4.f * s);
        if constexpr (MODEL == HW_ROTATE) return margin - __builtin_amdgcn_sqrtf(s);
        return s;
    }
};

}  // namespace hw
