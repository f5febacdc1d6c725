// Shared helpers of the gfx950 batch-SOM library (internal; the public ABI is include/dbgsom_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/dbgsom_hip.h"

namespace dbgsom {

void set_error(const char *fmt, ...);

#define DBGSOM_HIP_CHECK(expr)                                                           \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            ::dbgsom::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),   \
                                __FILE__, __LINE__);                                     \
            return DBGSOM_EHIP;                                                          \
        }                                                                                \
    } while (0)

#define DBGSOM_REQUIRE(cond, msg)                                   \
    do {                                                            \
        if (!(cond)) {                                              \
            ::dbgsom::set_error("%s: %s", __func__, msg);           \
            return DBGSOM_EINVAL;                                   \
        }                                                           \
    } while (0)

inline int launch_status(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("launch of %s failed: %s", what, hipGetErrorString(e));
        return DBGSOM_EHIP;
    }
    return DBGSOM_OK;
}

// bfloat16 storage element: the upper 16 bits of a float32; widening is exact
struct bf16_t {
    uint16_t bits;
    __host__ __device__ bf16_t() = default;
    __host__ __device__ explicit bf16_t(int) : bits(0) {}
    __device__ __forceinline__ operator float() const { return __uint_as_float(((uint32_t)bits) << 16); }
};

// exact widening of a stored sample element to float64
__device__ __forceinline__ double widen(double v) { return v; }
__device__ __forceinline__ double widen(float v) { return (double)v; }
__device__ __forceinline__ double widen(bf16_t v) { return (double)(float)v; }

inline bool valid_dtype(int dt) { return dt == DBGSOM_F32 || dt == DBGSOM_F64 || dt == DBGSOM_BF16; }
inline size_t dtype_size(int dt) { return dt == DBGSOM_F64 ? 8 : (dt == DBGSOM_F32 ? 4 : 2); }

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
inline bool is_aligned(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// internal launchers (one per .hip file)
int launch_row_sqnorms(const void *A, int dtype, int64_t rows, int64_t d, int64_t ld, double *out,
                       hipStream_t s);
int launch_bmu(const void *X, int x_dtype, int64_t N, int64_t d, int64_t ldx, const double *xx,
               const double *W, int64_t M, const double *ww, int k, int round_f32, int64_t *idx,
               double *dist, hipStream_t s);
int launch_exp_similarity(const double *dist, int64_t N, double gamma, double *kw, hipStream_t s);
size_t accumulate_workspace_bytes(int64_t N, int64_t d, int64_t M);
int launch_accumulate(const void *X, int x_dtype, int64_t N, int64_t d, int64_t ldx,
                      const int64_t *idx, const double *kw, const double *dist, int64_t M,
                      double *sums, int32_t *status, void *ws, size_t ws_bytes, hipStream_t s);
size_t bucket_sort_workspace_bytes(int64_t N, int64_t M);
int launch_bucket_sort(const int64_t *idx, int64_t N, int64_t M, int32_t *order, void *ws,
                       hipStream_t s);
size_t smooth_workspace_bytes(int64_t M, int64_t d);
int launch_smooth(const double *sums, int64_t M, int64_t d, const float *hop, double sigma,
                  int layout, const double *W_old, double *W_new, double *change_total, void *ws,
                  size_t ws_bytes, hipStream_t s);

}  // namespace dbgsom
