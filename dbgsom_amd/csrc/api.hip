// Device-level C-ABI entry points (include/dbgsom_hip.h) and the error channel.  The context-level
// API lives in engine.hip.
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace dbgsom {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

}  // namespace dbgsom

using namespace dbgsom;

extern "C" {

int dbgsom_abi_version(void) { return DBGSOM_ABI_VERSION; }
const char *dbgsom_last_error(void) { return g_err; }

int dbgsom_device_count(int *count) {
    DBGSOM_REQUIRE(count, "null pointer");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); n = 0; }
    *count = n;
    return DBGSOM_OK;
}

int dbgsom_row_sqnorms(const void *A, int dtype, int64_t rows, int64_t d, int64_t ld, double *out,
                       void *stream) {
    return launch_row_sqnorms(A, dtype, rows, d, ld, out, (hipStream_t)stream);
}

int dbgsom_bmu(const void *X, int x_dtype, int64_t N, int64_t d, int64_t ldx, const double *xx,
               const double *W, int64_t M, const double *ww, int k, int round_f32, int64_t *idx,
               double *dist, void *stream) {
    return launch_bmu(X, x_dtype, N, d, ldx, xx, W, M, ww, k, round_f32, idx, dist,
                      (hipStream_t)stream);
}

int dbgsom_exp_similarity(const double *dist, int64_t N, double gamma, double *kw, void *stream) {
    return launch_exp_similarity(dist, N, gamma, kw, (hipStream_t)stream);
}

size_t dbgsom_accumulate_workspace_bytes(int64_t N, int64_t d, int64_t M) {
    return accumulate_workspace_bytes(N, d, M);
}

int dbgsom_accumulate(const void *X, int x_dtype, int64_t N, int64_t d, int64_t ldx,
                      const int64_t *idx, const double *kw, const double *dist, int64_t M,
                      double *sums, int32_t *status, void *ws, size_t ws_bytes, void *stream) {
    return launch_accumulate(X, x_dtype, N, d, ldx, idx, kw, dist, M, sums, status, ws, ws_bytes,
                             (hipStream_t)stream);
}

size_t dbgsom_smooth_workspace_bytes(int64_t M, int64_t d) { return smooth_workspace_bytes(M, d); }

int dbgsom_smooth(const double *sums, int64_t M, int64_t d, const float *hop, double sigma,
                  int layout, const double *W_old, double *W_new, double *change_total, void *ws,
                  size_t ws_bytes, void *stream) {
    return launch_smooth(sums, M, d, hop, sigma, layout, W_old, W_new, change_total, ws, ws_bytes,
                         (hipStream_t)stream);
}

}  // extern "C"
