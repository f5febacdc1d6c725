// Best-matching-unit search on gfx950 (MI355X): float64 expanded-L2 distances on the f64 matrix
// cores + per-sample arg-k-min with wavefront shuffles.
//
// Replaces BaseSom._get_winning_neurons (reference dbgsom/BaseSom.py:446-464), i.e. sklearn's
// brute NearestNeighbors engine, whose arithmetic is float64 even for float32 samples:
//     r_ij = (|x_i|^2 + (-2 <x_i,w_j>)) + |w_j|^2 ; clamp 0 ; argmin_j (ties -> lowest j) ; sqrt.
//
// Design (one 256-thread workgroup = 4 wavefronts of 64):
//   * a workgroup owns BI=128 samples and sweeps all prototypes in chunks of BJ=128; the 2x2
//     wavefronts each own a 64x64 (prototype x sample) block = 4x4 tiles of
//     v_mfma_f64_16x16x4_f64, accumulated over the whole feature dimension in ONE chain per
//     (sample, prototype) pair: the dot product is the sequential fma chain k = 0..d-1, the same
//     order as oracle/bmu_chain.c, so winners and distances compare bit for bit.
//   * prototypes are the A operand (tile rows), samples the B operand (tile columns): a lane then
//     holds ONE sample (column = lane&15) against 4 prototypes per tile, so the running
//     (min, argmin) lives in registers and never leaves the lane until the last chunk.
//   * X (f32 or f64) and W (f64) tiles are staged global -> registers -> LDS (converted to f64 on
//     the way) with the next tile's loads in flight under the current tile's MFMAs; LDS rows are
//     padded to 18 doubles so the ds_read_b64 fragment reads are bank-conflict free.
//   * epilogue: cross-lane merge with __shfl_xor (lanes l, l^16, l^32, l^48 share a sample),
//     then the two wavefronts that share a sample merge through LDS.
#include <math.h>
#include <stdlib.h>

#include "bmu_common.h"

namespace dbgsom {

constexpr int LS = KT + 2;  // LDS row stride (doubles)

template <typename T>
__device__ __forceinline__ void load8(const T *__restrict__ base, int64_t row, int64_t nrows,
                                      int64_t ld, int k, int d, int vec_ok, T (&v)[8]) {
    if (row < nrows && k < d) {
        const T *p = base + row * ld + k;
        if (vec_ok && k + 8 <= d) {
            if constexpr (sizeof(T) == 2) {
                const uint4 a = *reinterpret_cast<const uint4 *>(p);
                const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e].bits = (uint16_t)(w[e] & 0xffffu);
                    v[2 * e + 1].bits = (uint16_t)(w[e] >> 16);
                }
            } else if constexpr (sizeof(T) == 4) {
                const float4 a = *reinterpret_cast<const float4 *>(p);
                const float4 b = *reinterpret_cast<const float4 *>(p + 4);
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
                v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const double2 a = *reinterpret_cast<const double2 *>(p + 2 * e);
                    v[2 * e] = a.x;
                    v[2 * e + 1] = a.y;
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (k + e < d) ? p[e] : T(0);
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = T(0);
    }
}

template <typename XT, int K>
__global__ __launch_bounds__(NT, 2) void bmu_kernel(
    const XT *__restrict__ X, int64_t N, int d, int64_t ldx, const double *__restrict__ xx,
    const double *__restrict__ W, int M, const double *__restrict__ ww, int round_f32, int xvec,
    int wvec, int64_t *__restrict__ idx_out, double *__restrict__ dist_out) {
    // double-buffered operand tiles: tile t+1 is written while tile t feeds the MFMAs, one
    // barrier per tile
    __shared__ __attribute__((aligned(16))) double xs[2][BI * LS];
    __shared__ __attribute__((aligned(16))) double wsm[2][BJ * LS];
    __shared__ double yy_s[2][BJ];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;  // 2 x 2 wavefronts: sample half, prototype half
    const int lr = lane & 15, lq = lane >> 4;
    const int64_t i0 = (int64_t)blockIdx.x * BI;

    double xi[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int64_t i = i0 + wi * 64 + it * 16 + lr;
        xi[it] = (i < N) ? xx[i] : 0.0;
    }
    Best<K> best[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) best[it].init();

    const int lrow = tid >> 1, lk = (tid & 1) * 8;  // staging: 2 threads per tile row, 8 values each
    const int nkt = (d + KT - 1) / KT;
    const int nchunk = (M + BJ - 1) / BJ;
    const int ntile = nkt * nchunk;  // flat (chunk, k-tile) sequence: the pipeline never drains
    XT xr[8];
    double wr[8];

    auto stage_store = [&](int buf) {
        double *xd = &xs[buf][lrow * LS + lk];
        double *wd = &wsm[buf][lrow * LS + lk];
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            *reinterpret_cast<double2 *>(xd + e) =
                double2{widen(xr[e]), widen(xr[e + 1])};
            *reinterpret_cast<double2 *>(wd + e) = double2{wr[e], wr[e + 1]};
        }
    };

    d4_t acc[4][4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int it = 0; it < 4; ++it) acc[jt][it] = d4_t{0.0, 0.0, 0.0, 0.0};

    load8<XT>(X, i0 + lrow, N, ldx, lk, d, xvec, xr);
    load8<double>(W, (int64_t)lrow, M, d, lk, d, wvec, wr);
    if (tid < BJ) yy_s[0][tid] = (tid < M) ? ww[tid] : 0.0;
    stage_store(0);
    __syncthreads();

    int kt = 0, jc = 0, parity = 0;
    for (int t = 0; t < ntile; ++t) {
        const int cur = t & 1;
        // next tile in the flat sequence
        int kt_n = kt + 1, jc_n = jc;
        if (kt_n == nkt) { kt_n = 0; jc_n = jc + BJ; }
        const bool more = (t + 1 < ntile);
        if (more) {  // its global loads fly under this tile's MFMAs
            const int kn = kt_n * KT + lk;
            load8<XT>(X, i0 + lrow, N, ldx, kn, d, xvec, xr);
            load8<double>(W, (int64_t)jc_n + lrow, M, d, kn, d, wvec, wr);
            if (kt_n == 0 && tid < BJ)
                yy_s[parity ^ 1][tid] = (jc_n + tid < M) ? ww[jc_n + tid] : 0.0;
        }
#pragma unroll
        for (int ks = 0; ks < KT / 4; ++ks) {
            if (ks == KT / 8 && more) stage_store(cur ^ 1);  // half-way: loads have landed
            double a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a[u] = wsm[cur][(wj * 64 + u * 16 + lr) * LS + ks * 4 + lq];
                b[u] = xs[cur][(wi * 64 + u * 16 + lr) * LS + ks * 4 + lq];
            }
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int it = 0; it < 4; ++it)
                    acc[jt][it] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[jt], b[it], acc[jt][it],
                                                                       0, 0, 0);
        }
        if (kt == nkt - 1) {
            // chunk epilogue: expanded L2 + running arg-k-min (prototype index ascends per lane)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int jl = wj * 64 + jt * 16 + 4 * r + lq;
                    const int j = jc + jl;
                    const double y = yy_s[parity][jl];
                    if (j < M) {
#pragma unroll
                        for (int it = 0; it < 4; ++it) {
                            double rv = (xi[it] + (-2.0 * acc[jt][it][r])) + y;
                            if (!(rv > 0.0)) rv = (rv != rv) ? rv : 0.0;  // max(r, 0), NaN kept
                            best[it].push(rv, j);
                        }
                    }
                }
            }
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int it = 0; it < 4; ++it) acc[jt][it] = d4_t{0.0, 0.0, 0.0, 0.0};
            parity ^= 1;
        }
        __syncthreads();  // tile t+1 is complete in LDS; tile t's buffer may be overwritten next
        kt = kt_n;
        jc = jc_n;
    }

    // lanes l, l^16, l^32, l^48 hold the same sample against different prototypes
#pragma unroll
    for (int it = 0; it < 4; ++it) {
#pragma unroll
        for (int m = 16; m <= 32; m <<= 1) {
            double ov[K];
            int oj[K];
#pragma unroll
            for (int t = 0; t < K; ++t) {
                ov[t] = __shfl_xor(best[it].v[t], m, 64);
                oj[t] = __shfl_xor(best[it].j[t], m, 64);
            }
            best[it].merge(ov, oj);
        }
    }

    // the two wavefronts with the same `wi` hold the two prototype halves of the same samples
    double *mv = &xs[0][0];                            // [2][BI][K]
    int *mj = reinterpret_cast<int *>(&wsm[0][0]);     // [2][BI][K]
    if (lq == 0) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int s = wi * 64 + it * 16 + lr;
#pragma unroll
            for (int t = 0; t < K; ++t) {
                mv[(wj * BI + s) * K + t] = best[it].v[t];
                mj[(wj * BI + s) * K + t] = best[it].j[t];
            }
        }
    }
    __syncthreads();
    if (tid < BI) {
        const int64_t i = i0 + tid;
        if (i < N) {
            Best<K> b;
            double ov[K];
            int oj[K];
#pragma unroll
            for (int t = 0; t < K; ++t) {
                b.v[t] = mv[(0 * BI + tid) * K + t];
                b.j[t] = mj[(0 * BI + tid) * K + t];
                ov[t] = mv[(1 * BI + tid) * K + t];
                oj[t] = mj[(1 * BI + tid) * K + t];
            }
            b.merge(ov, oj);
#pragma unroll
            for (int t = 0; t < K; ++t) {
                double dv = sqrt(b.v[t]);
                if (round_f32) dv = (double)(float)dv;
                idx_out[i * K + t] = (b.j[t] == 0x7fffffff) ? (int64_t)-1 : (int64_t)b.j[t];
                dist_out[i * K + t] = dv;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// squared row norms, sequential fma chain per row (same order as the oracle); rows are staged
// through LDS so the global reads stay coalesced while each thread walks its own row.
// ---------------------------------------------------------------------------------------------
// Two shapes of the same kernel: 128 rows x 32 features per staged tile, 128 threads, for the
// sample matrix (one thread per row walks the chain); 8 rows x 1024 features, 512 threads, for the
// small prototype matrix, which is latency-bound at M ~ 1000: more workgroups, and a row of up to
// 1024 features is staged in ONE round of loads that are all in flight at once (the 16 x 256
// shape before it spent 38 us per epoch in four rounds of four dependent load batches).
template <typename T, int NR, int NK, int NTH>
__global__ __launch_bounds__(NTH) void row_sqnorms_kernel(const T *__restrict__ A,
                                                          int64_t rows, int d, int64_t ld,
                                                          double *__restrict__ out) {
    constexpr int NORM_THREADS = NTH;
    constexpr int LOAD_UNROLL = NR * NK / NTH < 16 ? NR * NK / NTH : 16;
    __shared__ double tile[NR][NK + 1];
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * NR;
    double acc = 0.0;
    for (int k0 = 0; k0 < d; k0 += NK) {
        __syncthreads();
#pragma clang loop unroll_count(LOAD_UNROLL)
        for (int e = tid; e < NR * NK; e += NORM_THREADS) {
            const int r = e / NK, c = e % NK;  // consecutive threads -> consecutive features
            const int64_t row = r0 + r;
            const int k = k0 + c;
            tile[r][c] = (row < rows && k < d) ? widen(A[row * ld + k]) : 0.0;
        }
        __syncthreads();
        if (tid < NR) {  // one sequential fma chain per row (the order is the specification)
            const int kmax = min(NK, d - k0);
            if (kmax == NK) {  // whole tile: LDS reads batched 16 ahead of the dependent chain
#pragma unroll 16
                for (int kk = 0; kk < NK; ++kk) {
                    const double v = tile[tid][kk];
                    acc = fma(v, v, acc);
                }
            } else {
                for (int kk = 0; kk < kmax; ++kk) {
                    const double v = tile[tid][kk];
                    acc = fma(v, v, acc);
                }
            }
        }
    }
    if (tid < NR && r0 + tid < rows) out[r0 + tid] = acc;
}

__global__ void exp_similarity_kernel(const double *__restrict__ dist, int64_t N, double gamma,
                                      double *__restrict__ kw) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride) {
        const double dd = dist[i];
        kw[i] = 1.0 - sqrt(1.0 - exp(-gamma * (dd * dd)));
    }
}

// ---------------------------------------------------------------------------------------------
int launch_row_sqnorms(const void *A, int dtype, int64_t rows, int64_t d, int64_t ld, double *out,
                       hipStream_t s) {
    DBGSOM_REQUIRE(valid_dtype(dtype), "dtype must be DBGSOM_F32/F64/BF16");
    DBGSOM_REQUIRE(rows >= 0 && d >= 1 && ld >= d && d <= 0x7fffffff, "bad shape");
    if (rows == 0) return DBGSOM_OK;
    DBGSOM_REQUIRE(A && out, "null pointer");
    const bool small = rows <= 16384;
    const int nr = small ? 8 : 128;
    const int64_t nb = (rows + nr - 1) / nr;
    DBGSOM_REQUIRE(nb <= 0x7fffffff, "too many rows");
    dim3 grid((unsigned)nb);
#define DBGSOM_NORMS(T)                                                                            \
    do {                                                                                           \
        if (small) hipLaunchKernelGGL((row_sqnorms_kernel<T, 8, 1024, 512>), grid, dim3(512), 0, s, (const T *)A, rows, (int)d, ld, out); \
        else hipLaunchKernelGGL((row_sqnorms_kernel<T, 128, 32, 128>), grid, dim3(128), 0, s, (const T *)A, rows, (int)d, ld, out); \
    } while (0)
    if (dtype == DBGSOM_F32) DBGSOM_NORMS(float);
    else if (dtype == DBGSOM_F64) DBGSOM_NORMS(double);
    else DBGSOM_NORMS(bf16_t);
#undef DBGSOM_NORMS
    return launch_status("row_sqnorms_kernel");
}

int launch_bmu(const void *X, int x_dtype, int64_t N, int64_t d, int64_t ldx, const double *xx,
               const double *W, int64_t M, const double *ww, int k, int round_f32, int64_t *idx,
               double *dist, hipStream_t s) {
    DBGSOM_REQUIRE(valid_dtype(x_dtype), "x_dtype must be DBGSOM_F32/F64/BF16");
    DBGSOM_REQUIRE(k == 1 || k == 2, "k must be 1 or 2");
    DBGSOM_REQUIRE(N >= 0 && d >= 1 && ldx >= d && d <= 0x7fffffff, "bad sample shape");
    DBGSOM_REQUIRE(M >= k && M <= 0x7fffff00, "need k <= M < 2^31");
    if (N == 0) return DBGSOM_OK;
    DBGSOM_REQUIRE(X && xx && W && ww && idx && dist, "null pointer");
    const int64_t nb = (N + BI - 1) / BI;
    DBGSOM_REQUIRE(nb <= 0x7fffffff, "too many samples for one launch");
    static const bool force_generic = []() {
        const char *e = getenv("DBGSOM_BMU_PATH");  // "generic" forces the register-staged kernel
        return e && e[0] == 'g';
    }();
    if (!force_generic && bmu_dma_usable(X, x_dtype, d, ldx, W, M))
        return launch_bmu_dma(X, x_dtype, N, d, ldx, xx, W, M, ww, k, round_f32, idx, dist, s);
    const size_t xe = dtype_size(x_dtype);
    const int xvec = is_aligned(X, 16) && ((ldx * xe) % 16 == 0);
    const int wvec = is_aligned(W, 16) && ((d * 8) % 16 == 0);
    dim3 grid((unsigned)nb), block(NT);
#define DBGSOM_BMU_LAUNCH(XT, KK)                                                              \
    hipLaunchKernelGGL((bmu_kernel<XT, KK>), grid, block, 0, s, (const XT *)X, N, (int)d, ldx, \
                       xx, W, (int)M, ww, round_f32, xvec, wvec, idx, dist)
    if (x_dtype == DBGSOM_F32) {
        if (k == 1) DBGSOM_BMU_LAUNCH(float, 1); else DBGSOM_BMU_LAUNCH(float, 2);
    } else if (x_dtype == DBGSOM_F64) {
        if (k == 1) DBGSOM_BMU_LAUNCH(double, 1); else DBGSOM_BMU_LAUNCH(double, 2);
    } else {
        if (k == 1) DBGSOM_BMU_LAUNCH(bf16_t, 1); else DBGSOM_BMU_LAUNCH(bf16_t, 2);
    }
#undef DBGSOM_BMU_LAUNCH
    return launch_status("bmu_kernel");
}

int launch_exp_similarity(const double *dist, int64_t N, double gamma, double *kw, hipStream_t s) {
    DBGSOM_REQUIRE(N >= 0, "bad N");
    if (N == 0) return DBGSOM_OK;
    DBGSOM_REQUIRE(dist && kw, "null pointer");
    const int64_t nb = (N + 255) / 256;
    hipLaunchKernelGGL(exp_similarity_kernel, dim3((unsigned)(nb > 4096 ? 4096 : nb)), dim3(256),
                       0, s, dist, N, gamma, kw);
    return launch_status("exp_similarity_kernel");
}

}  // namespace dbgsom
