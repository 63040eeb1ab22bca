/* Plain-C restatement of the grouped KGE scorers (TEST INFRASTRUCTURE + bench.py cpu_baseline ONLY).
 *
 * Same arithmetic as oracle/kge.py:score_rows, which is pinned against the reference by
 * tests/golden/kge_*.npz; tests/test_oracle_c.py checks this file against oracle/kge.py.
 * Follows /root/reference/hopwise/model/knowledge_graph_embedding_recommender/
 *   transe.py:55-57, rotate.py:61-69, complex.py:53-62, distmult.py:50-51.
 * OpenMP over (h, r) groups; `hw_oracle_threads()` reports the thread count used.
 *
 * Build: gcc -O3 -march=native -fopenmp -shared -fPIC oracle/kge_ref.c -o oracle/libhw_oracle.so -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { TRANSE = 0, ROTATE = 1, COMPLEX = 2, DISTMULT = 3 };

int hw_oracle_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* score[b*n_tail + j] for tail t_idx[b*n_tail + j]; r_idx may be NULL (then r_fixed). */
int hw_oracle_score_grouped(int model, int dim, const float* h0, const float* h1, const float* r0, const float* r1,
                            const float* t0, const float* t1, float margin, float eps, const int64_t* h_idx,
                            const int64_t* r_idx, int64_t r_fixed, const int64_t* t_idx, int64_t n_groups,
                            int64_t n_tail, float* out) {
    if (model < 0 || model > 3 || dim < 1) return -1;
    int err = 0;
#pragma omp parallel
    {
        float* q0 = (float*)malloc(sizeof(float) * (size_t)dim * 2);
        float* q1 = q0 + dim;
        if (!q0) {
#pragma omp atomic write
            err = 1;
        } else {
#pragma omp for schedule(static)
            for (int64_t b = 0; b < n_groups; ++b) {
                const size_t ho = (size_t)h_idx[b] * dim;
                const size_t ro = (size_t)(r_idx ? r_idx[b] : r_fixed) * dim;
                for (int k = 0; k < dim; ++k) {
                    switch (model) {
                        case TRANSE: q0[k] = h0[ho + k] + r0[ro + k]; break;
                        case DISTMULT: q0[k] = h0[ho + k] * r0[ro + k]; break;
                        case ROTATE: {
                            const float c = cosf(r0[ro + k]), s = sinf(r0[ro + k]);
                            q0[k] = c * h0[ho + k] - s * h1[ho + k];
                            q1[k] = c * h1[ho + k] + s * h0[ho + k];
                        } break;
                        default:
                            q0[k] = h0[ho + k] * r0[ro + k];
                            q1[k] = h1[ho + k] * r0[ro + k] + h0[ho + k] * r1[ro + k] - h1[ho + k] * r1[ro + k];
                    }
                }
                for (int64_t j = 0; j < n_tail; ++j) {
                    const size_t to = (size_t)t_idx[b * n_tail + j] * dim;
                    float s = 0.f;
                    switch (model) {
                        case TRANSE:
                            for (int k = 0; k < dim; ++k) {
                                const float d = (q0[k] - t0[to + k]) + eps;
                                s += d * d;
                            }
                            s = -sqrtf(s);
                            break;
                        case DISTMULT:
                            for (int k = 0; k < dim; ++k) s += q0[k] * t0[to + k];
                            break;
                        case ROTATE:
                            for (int k = 0; k < dim; ++k) {
                                const float a = q0[k] - t0[to + k], c = q1[k] - t1[to + k];
                                s += a * a + c * c;
                            }
                            s = margin - sqrtf(s);
                            break;
                        default:
                            for (int k = 0; k < dim; ++k) s += q0[k] * t0[to + k] + q1[k] * t1[to + k];
                    }
                    out[b * n_tail + j] = s;
                }
            }
            free(q0);
        }
    }
    return err ? -2 : 0;
}
