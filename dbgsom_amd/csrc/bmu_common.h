// Pieces shared by the two BMU kernels (bmu.hip: register-staged, any shape; bmu_dma.hip:
// LDS-DMA ring for 16-byte aligned rows with d % 16 == 0).
#pragma once
#include <math.h>

#include "common.h"

namespace dbgsom {

typedef double d4_t __attribute__((ext_vector_type(4)));

constexpr int BI = 128;     // samples per workgroup
constexpr int BJ = 128;     // prototypes per sweep chunk
constexpr int KT = 16;      // feature depth of one LDS tile
constexpr int NT = 256;

__device__ __forceinline__ bool lex_lt(double a, int ja, double b, int jb) {
    return a < b || (a == b && ja < jb);
}

template <int K>
struct Best {
    double v[K];
    int j[K];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int t = 0; t < K; ++t) { v[t] = INFINITY; j[t] = 0x7fffffff; }
    }
    // candidates arrive with ascending index inside one lane: strict '<' keeps the lowest index
    __device__ __forceinline__ void push(double r, int idx) {
        if constexpr (K == 1) {
            if (r < v[0]) { v[0] = r; j[0] = idx; }
        } else {
            if (r < v[0]) { v[1] = v[0]; j[1] = j[0]; v[0] = r; j[0] = idx; }
            else if (r < v[1]) { v[1] = r; j[1] = idx; }
        }
    }
    // merge with another sorted list (lexicographic on (value, index))
    __device__ __forceinline__ void merge(const double (&ov)[K], const int (&oj)[K]) {
        if constexpr (K == 1) {
            if (lex_lt(ov[0], oj[0], v[0], j[0])) { v[0] = ov[0]; j[0] = oj[0]; }
        } else {
            double n0, n1; int m0, m1;
            if (lex_lt(ov[0], oj[0], v[0], j[0])) {
                n0 = ov[0]; m0 = oj[0];
                if (lex_lt(v[0], j[0], ov[1], oj[1])) { n1 = v[0]; m1 = j[0]; }
                else { n1 = ov[1]; m1 = oj[1]; }
            } else {
                n0 = v[0]; m0 = j[0];
                if (lex_lt(ov[0], oj[0], v[1], j[1])) { n1 = ov[0]; m1 = oj[0]; }
                else { n1 = v[1]; m1 = j[1]; }
            }
            v[0] = n0; j[0] = m0; v[1] = n1; j[1] = m1;
        }
    }
};


int launch_bmu_dma(const void *X, int x_dtype, int64_t N, int64_t d, int64_t ldx, const double *xx,
                   const double *W, int64_t M, const double *ww, int k, int round_f32,
                   int64_t *idx, double *dist, hipStream_t s);
bool bmu_dma_usable(const void *X, int x_dtype, int64_t d, int64_t ldx, const void *W, int64_t M);

}  // namespace dbgsom
