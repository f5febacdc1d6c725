// C-ABI entry points (include/dbgsom_hip.h): argument checks, error capture, and the
// context-level host API that keeps X resident in HBM across epochs.
#include <stdarg.h>
#include <string.h>

#include <new>
#include <vector>

#include "common.h"

namespace dbgsom {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// a device allocation that grows on demand and is reused across calls
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return DBGSOM_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        const size_t want = align_up(bytes + bytes / 8, 1 << 20);
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            p = nullptr;
            set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
            return DBGSOM_ENOMEM;
        }
        cap = want;
        return DBGSOM_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

}  // namespace dbgsom

using namespace dbgsom;

struct dbgsom_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // resident samples
    int x_dtype = -1;
    int64_t N = 0, d = 0;
    DevBuf X, xx;
    // topology
    int64_t topoM = 0;
    DevBuf hop;
    // per-epoch device state
    DevBuf W, Wn, ww, idx, dist, kw, sums, acc_ws, sm_ws, scal;
    // query scratch (predict on other samples)
    DevBuf Xq, xxq;
    std::vector<float> hop_stage;
};

#define CTX_CHECK(c)                                        \
    do {                                                    \
        if (!(c)) { set_error("%s: null context", __func__); return DBGSOM_EINVAL; } \
        DBGSOM_HIP_CHECK(hipSetDevice((c)->device));        \
    } while (0)
#define TRY(expr) do { int _rc = (expr); if (_rc != DBGSOM_OK) return _rc; } while (0)

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

// ------------------------------------------------------------------------------------------
// context level
// ------------------------------------------------------------------------------------------
int dbgsom_ctx_create(int device, dbgsom_ctx **out) {
    DBGSOM_REQUIRE(out, "null pointer");
    *out = nullptr;
    int n = 0;
    DBGSOM_HIP_CHECK(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) {
        set_error("dbgsom_ctx_create: device %d not available (%d visible)", device, n);
        return DBGSOM_EINVAL;
    }
    DBGSOM_HIP_CHECK(hipSetDevice(device));
    dbgsom_ctx *c = new (std::nothrow) dbgsom_ctx();
    if (!c) { set_error("out of host memory"); return DBGSOM_ENOMEM; }
    c->device = device;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
        delete c;
        return DBGSOM_EHIP;
    }
    *out = c;
    return DBGSOM_OK;
}

int dbgsom_ctx_destroy(dbgsom_ctx *c) {
    if (!c) return DBGSOM_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    DevBuf *bufs[] = {&c->X, &c->xx, &c->hop, &c->W, &c->Wn, &c->ww, &c->idx, &c->dist, &c->kw,
                      &c->sums, &c->acc_ws, &c->sm_ws, &c->scal, &c->Xq, &c->xxq};
    for (DevBuf *b : bufs) b->release();
    delete c;
    return DBGSOM_OK;
}

int dbgsom_ctx_load(dbgsom_ctx *c, const void *X_host, int x_dtype, int64_t N, int64_t d) {
    CTX_CHECK(c);
    DBGSOM_REQUIRE(valid_dtype(x_dtype), "x_dtype must be DBGSOM_F32/F64/BF16");
    DBGSOM_REQUIRE(X_host && N >= 1 && d >= 1, "bad samples");
    const size_t es = dtype_size(x_dtype);
    TRY(c->X.reserve((size_t)N * d * es));
    TRY(c->xx.reserve((size_t)N * 8));
    DBGSOM_HIP_CHECK(hipMemcpyAsync(c->X.p, X_host, (size_t)N * d * es, hipMemcpyHostToDevice,
                                    c->stream));
    c->x_dtype = x_dtype; c->N = N; c->d = d;
    TRY(launch_row_sqnorms(c->X.p, x_dtype, N, d, d, c->xx.as<double>(), c->stream));
    DBGSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
    return DBGSOM_OK;
}

int dbgsom_ctx_set_topology(dbgsom_ctx *c, const double *hop_host, int64_t M) {
    CTX_CHECK(c);
    DBGSOM_REQUIRE(hop_host && M >= 1 && M <= DBGSOM_MAX_PROTOTYPES, "bad topology");
    c->hop_stage.resize((size_t)M * M);
    for (size_t e = 0; e < (size_t)M * M; ++e) c->hop_stage[e] = (float)hop_host[e];
    TRY(c->hop.reserve((size_t)M * M * 4));
    DBGSOM_HIP_CHECK(hipMemcpyAsync(c->hop.p, c->hop_stage.data(), (size_t)M * M * 4,
                                    hipMemcpyHostToDevice, c->stream));
    DBGSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
    c->topoM = M;
    return DBGSOM_OK;
}

static int upload_W(dbgsom_ctx *c, const double *W_host, int64_t M) {
    DBGSOM_REQUIRE(W_host && M >= 1, "bad prototypes");
    TRY(c->W.reserve((size_t)M * c->d * 8));
    TRY(c->ww.reserve((size_t)M * 8));
    DBGSOM_HIP_CHECK(hipMemcpyAsync(c->W.p, W_host, (size_t)M * c->d * 8, hipMemcpyHostToDevice,
                                    c->stream));
    return launch_row_sqnorms(c->W.p, DBGSOM_F64, M, c->d, c->d, c->ww.as<double>(), c->stream);
}

int dbgsom_ctx_bmu(dbgsom_ctx *c, const double *W_host, int64_t M, int k, int round_f32,
                   int64_t *idx_host, double *dist_host) {
    CTX_CHECK(c);
    if (c->x_dtype < 0) { set_error("dbgsom_ctx_bmu: no samples loaded"); return DBGSOM_ESTATE; }
    DBGSOM_REQUIRE(idx_host && dist_host && (k == 1 || k == 2), "bad arguments");
    TRY(upload_W(c, W_host, M));
    TRY(c->idx.reserve((size_t)c->N * 2 * 8));
    TRY(c->dist.reserve((size_t)c->N * 2 * 8));
    TRY(launch_bmu(c->X.p, c->x_dtype, c->N, c->d, c->d, c->xx.as<double>(), c->W.as<double>(), M,
                   c->ww.as<double>(), k, round_f32, c->idx.as<int64_t>(), c->dist.as<double>(),
                   c->stream));
    DBGSOM_HIP_CHECK(hipMemcpyAsync(idx_host, c->idx.p, (size_t)c->N * k * 8,
                                    hipMemcpyDeviceToHost, c->stream));
    DBGSOM_HIP_CHECK(hipMemcpyAsync(dist_host, c->dist.p, (size_t)c->N * k * 8,
                                    hipMemcpyDeviceToHost, c->stream));
    DBGSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
    return DBGSOM_OK;
}

int dbgsom_ctx_bmu_query(dbgsom_ctx *c, const void *Xq_host, int x_dtype, int64_t Nq, int64_t d,
                         const double *W_host, int64_t M, int k, int round_f32, int64_t *idx_host,
                         double *dist_host) {
    CTX_CHECK(c);
    DBGSOM_REQUIRE(valid_dtype(x_dtype), "x_dtype must be DBGSOM_F32/F64/BF16");
    DBGSOM_REQUIRE(Xq_host && W_host && idx_host && dist_host && Nq >= 0 && d >= 1 && M >= 1 &&
                       (k == 1 || k == 2), "bad arguments");
    if (Nq == 0) return DBGSOM_OK;
    const size_t es = dtype_size(x_dtype);
    DevBuf Wq, wwq, iq, dq;  // query-sized scratch; independent of the training state
    int rc = DBGSOM_OK;
    do {
        if ((rc = c->Xq.reserve((size_t)Nq * d * es))) break;
        if ((rc = c->xxq.reserve((size_t)Nq * 8))) break;
        if ((rc = Wq.reserve((size_t)M * d * 8))) break;
        if ((rc = wwq.reserve((size_t)M * 8))) break;
        if ((rc = iq.reserve((size_t)Nq * k * 8))) break;
        if ((rc = dq.reserve((size_t)Nq * k * 8))) break;
        hipError_t e = hipMemcpyAsync(c->Xq.p, Xq_host, (size_t)Nq * d * es, hipMemcpyHostToDevice,
                                      c->stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync(Wq.p, W_host, (size_t)M * d * 8, hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) { set_error("H2D copy failed: %s", hipGetErrorString(e)); rc = DBGSOM_EHIP; break; }
        if ((rc = launch_row_sqnorms(c->Xq.p, x_dtype, Nq, d, d, c->xxq.as<double>(), c->stream))) break;
        if ((rc = launch_row_sqnorms(Wq.p, DBGSOM_F64, M, d, d, wwq.as<double>(), c->stream))) break;
        if ((rc = launch_bmu(c->Xq.p, x_dtype, Nq, d, d, c->xxq.as<double>(), Wq.as<double>(), M,
                             wwq.as<double>(), k, round_f32, iq.as<int64_t>(), dq.as<double>(),
                             c->stream))) break;
        e = hipMemcpyAsync(idx_host, iq.p, (size_t)Nq * k * 8, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync(dist_host, dq.p, (size_t)Nq * k * 8, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { set_error("D2H copy failed: %s", hipGetErrorString(e)); rc = DBGSOM_EHIP; }
    } while (0);
    if (rc != DBGSOM_OK) (void)hipStreamSynchronize(c->stream);
    Wq.release(); wwq.release(); iq.release(); dq.release();
    return rc;
}

int dbgsom_ctx_epoch(dbgsom_ctx *c, const double *W_host, int64_t M, int round_f32, double gamma,
                     double sigma, int layout, double *W_new_host, double *change_total_host,
                     double *errors_host, double *activations_host, int64_t *idx_host,
                     double *dist_host) {
    CTX_CHECK(c);
    if (c->x_dtype < 0) { set_error("dbgsom_ctx_epoch: no samples loaded"); return DBGSOM_ESTATE; }
    if (c->topoM != M) {
        set_error("dbgsom_ctx_epoch: topology holds %lld neurons, weights %lld (call "
                  "dbgsom_ctx_set_topology after growth)", (long long)c->topoM, (long long)M);
        return DBGSOM_ESTATE;
    }
    DBGSOM_REQUIRE(W_new_host && change_total_host && errors_host && activations_host,
                   "null output");
    const int64_t N = c->N, d = c->d;
    TRY(upload_W(c, W_host, M));
    TRY(c->Wn.reserve((size_t)M * d * 8));
    TRY(c->idx.reserve((size_t)N * 2 * 8));
    TRY(c->dist.reserve((size_t)N * 2 * 8));
    TRY(c->kw.reserve((size_t)N * 8));
    TRY(c->sums.reserve((size_t)M * (d + 3) * 8));
    TRY(c->acc_ws.reserve(accumulate_workspace_bytes(N, d, M)));
    TRY(c->sm_ws.reserve(smooth_workspace_bytes(M, d)));
    TRY(c->scal.reserve(256));
    double *chg = c->scal.as<double>();
    int32_t *status = reinterpret_cast<int32_t *>(c->scal.as<char>() + 64);

    TRY(launch_bmu(c->X.p, c->x_dtype, N, d, d, c->xx.as<double>(), c->W.as<double>(), M,
                   c->ww.as<double>(), 1, round_f32, c->idx.as<int64_t>(), c->dist.as<double>(),
                   c->stream));
    TRY(launch_exp_similarity(c->dist.as<double>(), N, gamma, c->kw.as<double>(), c->stream));
    TRY(launch_accumulate(c->X.p, c->x_dtype, N, d, d, c->idx.as<int64_t>(), c->kw.as<double>(),
                          c->dist.as<double>(), M, c->sums.as<double>(), status, c->acc_ws.p,
                          c->acc_ws.cap, c->stream));
    TRY(launch_smooth(c->sums.as<double>(), M, d, c->hop.as<float>(), sigma, layout,
                      c->W.as<double>(), c->Wn.as<double>(), chg, c->sm_ws.p, c->sm_ws.cap,
                      c->stream));
    const double *sums = c->sums.as<double>();
    int32_t st = 0;
    DBGSOM_HIP_CHECK(hipMemcpyAsync(W_new_host, c->Wn.p, (size_t)M * d * 8, hipMemcpyDeviceToHost, c->stream));
    DBGSOM_HIP_CHECK(hipMemcpyAsync(change_total_host, chg, 8, hipMemcpyDeviceToHost, c->stream));
    DBGSOM_HIP_CHECK(hipMemcpyAsync(activations_host, sums + (size_t)M * d + M, (size_t)M * 8, hipMemcpyDeviceToHost, c->stream));
    DBGSOM_HIP_CHECK(hipMemcpyAsync(errors_host, sums + (size_t)M * d + 2 * M, (size_t)M * 8, hipMemcpyDeviceToHost, c->stream));
    DBGSOM_HIP_CHECK(hipMemcpyAsync(&st, status, 4, hipMemcpyDeviceToHost, c->stream));
    if (idx_host) DBGSOM_HIP_CHECK(hipMemcpyAsync(idx_host, c->idx.p, (size_t)N * 8, hipMemcpyDeviceToHost, c->stream));
    if (dist_host) DBGSOM_HIP_CHECK(hipMemcpyAsync(dist_host, c->dist.p, (size_t)N * 8, hipMemcpyDeviceToHost, c->stream));
    DBGSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (st != 0) { set_error("dbgsom_ctx_epoch: winner index out of range"); return DBGSOM_ERANGE; }
    return DBGSOM_OK;
}

}  // extern "C"
