// Post-fit consumers of the BMU step as device reductions: the N-sized
// winners / distances never leave HBM.
//
// Replaces host loops of the reference (dbgsom/BaseSom.py):
//   calculate_quantization_error   :904-922   mean BMU distance          -> sum_f64
//   _calculate_topographic_error   :924-953   Python loop over N samples -> topographic_count
//   _calculate_node_statistics     :181-211   O(N*M) boolean masks       -> density_terms + accumulate
//   entropy criterion / _label_prototypes  :547-551, SomClassifier.py:130-152 -> class_histogram
#include <math.h>

#include "common.h"

namespace dbgsom {

constexpr int RB = 1024;  // partial sums (fixed -> the reduction tree does not depend on N)

__global__ __launch_bounds__(256) void sum_partial_kernel(const double *__restrict__ v, int64_t n,
                                                          double *__restrict__ part) {
    __shared__ double red[256];
    const int t = threadIdx.x;
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + t; i < n; i += (int64_t)RB * 256) s += v[i];
    red[t] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (t < w) red[t] += red[t + w];
        __syncthreads();
    }
    if (t == 0) part[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(RB) void sum_final_kernel(const double *__restrict__ part,
                                                       double *__restrict__ out) {
    __shared__ double red[RB];
    const int t = threadIdx.x;
    red[t] = part[t];
    __syncthreads();
    for (int w = RB / 2; w > 0; w >>= 1) {
        if (t < w) red[t] += red[t + w];
        __syncthreads();
    }
    if (t == 0) out[0] = red[0];
}

// number of samples whose two best matching units are further than 1.5 apart on the lattice
__global__ __launch_bounds__(256) void topographic_kernel(const int64_t *__restrict__ idx2,
                                                          int64_t n, const int32_t *__restrict__ xy,
                                                          int M, unsigned long long *count) {
    unsigned int local = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * 256) {
        const int64_t a = idx2[2 * i], b = idx2[2 * i + 1];
        if (a >= 0 && a < M && b >= 0 && b < M) {
            const double dx = (double)(xy[2 * a] - xy[2 * b]);
            const double dy = (double)(xy[2 * a + 1] - xy[2 * b + 1]);
            local += (sqrt(dx * dx + dy * dy) > 1.5) ? 1u : 0u;
        }
    }
    // integer adds: exact and order independent
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(count, (unsigned long long)local);
}

__global__ void density_kernel(const double *__restrict__ dist, int64_t n, double sigma,
                               double *__restrict__ out) {
    const double two_s2 = 2.0 * (sigma * sigma);
    const double norm = sigma * sqrt(2.0 * M_PI);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const double dd = dist[i];
        out[i] = exp(-(dd * dd) / two_s2) / norm;
    }
}

__global__ __launch_bounds__(256) void class_hist_kernel(const int64_t *__restrict__ win,
                                                         const int32_t *__restrict__ y, int64_t n,
                                                         int M, int C,
                                                         unsigned long long *__restrict__ hist) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * 256) {
        const int64_t j = win[i];
        const int c = y[i];
        if (j >= 0 && j < M && c >= 0 && c < C) atomicAdd(&hist[(size_t)j * C + c], 1ull);
    }
}

static unsigned grid_for(int64_t n) {
    const int64_t nb = (n + 255) / 256;
    return (unsigned)(nb < 1 ? 1 : (nb > 4096 ? 4096 : nb));
}

// Column sums in NumPy's own arithmetic for a reduction over axis 0 of a C-ordered array: every
// column is summed SEQUENTIALLY over the rows, in the array's dtype, no fused multiply-add --
// np.sum(X, axis=0) and, with `mean`, np.sum((X - mean) ** 2, axis=0) bit for bit (checked against
// NumPy in tests/test_gpu_parity.py).  The chain of a column's adds is one lane's; what feeds it is not: all
// four wavefronts of a workgroup bring a tile of 1024 rows x 16 columns in (64 loads in flight per thread) and leave it
// transposed in LDS, ONE wavefront adds the tile's rows in order while the next tile's loads are under way.
// (Round 3: one thread per column loading 32 rows at a time -- every batch a round trip to HBM with 13 wavefronts
//  on the whole chip: 24 ms per pass at 1e6 x 784, 5 % of a fit.)
template <typename T> __device__ __forceinline__ T add_rn(T a, T b);
template <> __device__ __forceinline__ float add_rn<float>(float a, float b) { return __fadd_rn(a, b); }
template <> __device__ __forceinline__ double add_rn<double>(double a, double b) { return __dadd_rn(a, b); }
template <typename T> __device__ __forceinline__ T mul_rn(T a, T b);
template <> __device__ __forceinline__ float mul_rn<float>(float a, float b) { return __fmul_rn(a, b); }
template <> __device__ __forceinline__ double mul_rn<double>(double a, double b) { return __dmul_rn(a, b); }

constexpr int CS_CW = 16;   // columns per workgroup: few, so that many workgroups stream (49 at d = 784) and a tile is long
template <typename T>
__global__ __launch_bounds__(256) void column_sums_kernel(const T *__restrict__ X, int64_t N, int d,
                                                          int64_t ld, const T *__restrict__ mean,
                                                          T *__restrict__ out) {
    constexpr int TR = 4096 / (int)sizeof(T);      // rows per tile: 1024 (float32) / 512 (float64)
    constexpr int P = TR + 16 / (int)sizeof(T);    // LDS pitch of a column (16 bytes of padding: aligned 16-byte reads)
    constexpr int RL = 256 / CS_CW;                // row lanes of the loaders
    constexpr int PER = TR / RL;                   // rows a thread loads per tile
    constexpr int V = 16 / (int)sizeof(T);         // values per 16-byte LDS read
    extern __shared__ __attribute__((aligned(16))) char cs_lds[];
    T *const tiles = reinterpret_cast<T *>(cs_lds);   // two tiles of CS_CW columns x P
    const int t = threadIdx.x, col = t % CS_CW, r0 = t / CS_CW;
    const int j = blockIdx.x * CS_CW + col;
    const bool live = j < d;
    const T m = (mean && live) ? mean[j] : (T)0;
    const bool centred = mean != nullptr;
    const T *p = X + (live ? j : 0);
    const int64_t ntile = (N + TR - 1) / TR;
    T v[PER];
    auto load = [&](int64_t tl) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {   // (unconditional loads from clamped rows: a load under a condition waits
            const int64_t r = tl * TR + r0 + RL * u;   //  for itself before the next one is issued)
            v[u] = p[(r < N ? r : N - 1) * ld];
        }
    };
    auto store = [&](int b) {
#pragma unroll
        for (int u = 0; u < PER; ++u) tiles[b * CS_CW * P + col * P + r0 + RL * u] = v[u];
    };
    T acc = (T)0;
    if (ntile > 0) { load(0); store(0); }
    __syncthreads();
    for (int64_t tl = 0; tl < ntile; ++tl) {
        if (tl + 1 < ntile) load(tl + 1);          // (in flight under the adds below)
        if (t < CS_CW) {                           // the chain: lane = column, rows in order
            const int64_t left = N - tl * TR;
            const int rows = left < TR ? (int)left : TR;
            const T *src = tiles + (int)(tl & 1) * CS_CW * P + col * P;
            int r = 0;
            for (; r + V <= rows; r += V) {
                T x[V];
                if constexpr (sizeof(T) == 4) {
                    const float4 q = *reinterpret_cast<const float4 *>(src + r);
                    x[0] = q.x; x[1] = q.y; x[2] = q.z; x[3] = q.w;
                } else {
                    const double2 q = *reinterpret_cast<const double2 *>(src + r);
                    x[0] = q.x; x[1] = q.y;
                }
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    T y = x[e];
                    if (centred) { y = add_rn<T>(y, -m); y = mul_rn<T>(y, y); }
                    acc = add_rn<T>(acc, y);
                }
            }
            for (; r < rows; ++r) {
                T y = src[r];
                if (centred) { y = add_rn<T>(y, -m); y = mul_rn<T>(y, y); }
                acc = add_rn<T>(acc, y);
            }
        }
        if (tl + 1 < ntile) store((int)((tl + 1) & 1));
        __syncthreads();
    }
    if (t < CS_CW && live) out[j] = acc;
}

}  // namespace dbgsom

using namespace dbgsom;

extern "C" {

size_t dbgsom_sum_workspace_bytes(void) { return (size_t)RB * sizeof(double); }

int dbgsom_sum_f64(const double *v_dev, int64_t n, double *out_dev, void *workspace_dev,
                   size_t workspace_bytes, void *stream) {
    DBGSOM_REQUIRE(n >= 0 && out_dev && workspace_dev, "bad arguments");
    DBGSOM_REQUIRE(n == 0 || v_dev, "null input");
    if (workspace_bytes < (size_t)RB * sizeof(double)) {
        set_error("dbgsom_sum_f64: workspace too small");
        return DBGSOM_ENOMEM;
    }
    hipStream_t s = (hipStream_t)stream;
    double *part = (double *)workspace_dev;
    hipLaunchKernelGGL(sum_partial_kernel, dim3(RB), dim3(256), 0, s, v_dev, n, part);
    hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(RB), 0, s, part, out_dev);
    return launch_status("sum kernels");
}

int dbgsom_topographic_count(const int64_t *idx2_dev, int64_t n, const int32_t *xy_dev, int64_t M,
                             uint64_t *count_dev, void *stream) {
    DBGSOM_REQUIRE(n >= 0 && M >= 1 && M <= 0x7fffffff && count_dev, "bad arguments");
    hipStream_t s = (hipStream_t)stream;
    DBGSOM_HIP_CHECK(hipMemsetAsync(count_dev, 0, sizeof(uint64_t), s));
    if (n == 0) return DBGSOM_OK;
    DBGSOM_REQUIRE(idx2_dev && xy_dev, "null pointer");
    hipLaunchKernelGGL(topographic_kernel, dim3(grid_for(n)), dim3(256), 0, s, idx2_dev, n, xy_dev,
                       (int)M, (unsigned long long *)count_dev);
    return launch_status("topographic_kernel");
}

int dbgsom_density_terms(const double *dist_dev, int64_t n, double sigma, double *out_dev,
                         void *stream) {
    DBGSOM_REQUIRE(n >= 0 && sigma > 0.0, "bad arguments");
    if (n == 0) return DBGSOM_OK;
    DBGSOM_REQUIRE(dist_dev && out_dev, "null pointer");
    hipLaunchKernelGGL(density_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream,
                       dist_dev, n, sigma, out_dev);
    return launch_status("density_kernel");
}

int dbgsom_class_histogram(const int64_t *idx_dev, const int32_t *y_dev, int64_t n, int64_t M,
                           int64_t n_classes, uint64_t *hist_dev, void *stream) {
    DBGSOM_REQUIRE(n >= 0 && M >= 1 && M <= 0x7fffffff && n_classes >= 1 &&
                       n_classes <= 0x7fffffff && hist_dev, "bad arguments");
    hipStream_t s = (hipStream_t)stream;
    DBGSOM_HIP_CHECK(hipMemsetAsync(hist_dev, 0, (size_t)M * n_classes * sizeof(uint64_t), s));
    if (n == 0) return DBGSOM_OK;
    DBGSOM_REQUIRE(idx_dev && y_dev, "null pointer");
    hipLaunchKernelGGL(class_hist_kernel, dim3(grid_for(n)), dim3(256), 0, s, idx_dev, y_dev, n,
                       (int)M, (int)n_classes, (unsigned long long *)hist_dev);
    return launch_status("class_hist_kernel");
}

int dbgsom_column_sums(const void *X_dev, int x_dtype, int64_t N, int64_t d, int64_t ldx,
                       const void *mean_dev, void *out_dev, void *stream) {
    DBGSOM_REQUIRE(x_dtype == DBGSOM_F32 || x_dtype == DBGSOM_F64, "float32 / float64 samples only");
    DBGSOM_REQUIRE(N >= 0 && d >= 1 && d <= 0x7fffffff && ldx >= d && out_dev, "bad arguments");
    DBGSOM_REQUIRE(N == 0 || X_dev, "null input");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((d + CS_CW - 1) / CS_CW)), block(256);
    if (x_dtype == DBGSOM_F32) {
        constexpr size_t lds = 2 * CS_CW * (1024 + 4) * 4;   // (two transposed tiles: 132 KB)
        static bool attr = false;
        if (!attr) {
            DBGSOM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&column_sums_kernel<float>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr = true;
        }
        hipLaunchKernelGGL(column_sums_kernel<float>, grid, block, lds, s, (const float *)X_dev, N, (int)d,
                           ldx, (const float *)mean_dev, (float *)out_dev);
    } else {
        constexpr size_t lds = 2 * CS_CW * (512 + 2) * 8;
        static bool attr = false;
        if (!attr) {
            DBGSOM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&column_sums_kernel<double>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr = true;
        }
        hipLaunchKernelGGL(column_sums_kernel<double>, grid, block, lds, s, (const double *)X_dev, N, (int)d,
                           ldx, (const double *)mean_dev, (double *)out_dev);
    }
    return launch_status("column_sums_kernel");
}

}  // extern "C"
