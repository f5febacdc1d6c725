// Context-level C ABI (include/dbgsom_hip.h, "Context-level entry points"): the host engine of the
// batch-SOM hot path.  A context owns one GPU's share of the job -- the resident samples (padded,
// optionally bfloat16), their norms and int8 digit planes, the prototypes, every workspace -- and
// runs the epoch body of BaseSom._grow_som (reference dbgsom/BaseSom.py:403-407) as one blocking
// call.  What used to be Python policy lives here: feature padding, the choice between the
// all-pairs and the filtered BMU search ("auto" / back-off), the adaptive number of digit planes,
// previous winners as seeds, device-resident prototypes between epochs, the one all-reduce per
// epoch (through the caller's callback).  No kernels of the hot path in this file: it drives the
// launchers of bmu*.hip, filter.hip, accumulate.hip, smooth.hip, stats.hip.
#include <dlfcn.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <new>
#include <vector>

#include "common.h"

namespace dbgsom {

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
            (void)hipGetLastError();
            set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
            return DBGSOM_ENOMEM;
        }
        cap = want;
        return DBGSOM_OK;
    }
    // (re)allocations come zero-filled: the "last workgroup" tickets at the front of the filter
    // workspace must be 0 before their first use (they reset themselves afterwards)
    int reserve_zeroed(size_t bytes, hipStream_t s) {
        if (bytes <= cap) return DBGSOM_OK;
        const int rc = reserve(bytes);
        if (rc != DBGSOM_OK) return rc;
        hipError_t e = hipMemsetAsync(p, 0, cap, s);
        if (e != hipSuccess) { set_error("hipMemsetAsync failed: %s", hipGetErrorString(e)); return DBGSOM_EHIP; }
        return DBGSOM_OK;
    }
    // grow and keep the first `keep` bytes (ordered on `s`)
    int reserve_keep(size_t bytes, size_t keep, hipStream_t s) {
        if (bytes <= cap) return DBGSOM_OK;
        void *old = p;
        const size_t old_cap = cap;
        p = nullptr;
        cap = 0;
        const int rc = reserve(bytes * 2);
        if (rc != DBGSOM_OK) { p = old; cap = old_cap; return rc; }
        if (old && keep) {
            hipError_t e = hipMemcpyAsync(p, old, keep, hipMemcpyDeviceToDevice, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) { set_error("device copy failed: %s", hipGetErrorString(e)); (void)hipFree(old); return DBGSOM_EHIP; }
        }
        if (old) (void)hipFree(old);
        return DBGSOM_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

// page-locked host staging for the small per-epoch results
struct PinBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return DBGSOM_OK;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        const size_t want = align_up(bytes * 2, 4096);
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocMapped);
        if (e == hipSuccess) e = hipHostGetDevicePointer(&dev, p, 0);
        if (e != hipSuccess) { if (p) (void)hipHostFree(p); p = nullptr; dev = nullptr; (void)hipGetLastError(); set_error("hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e)); return DBGSOM_ENOMEM; }
        cap = want;
        return DBGSOM_OK;
    }
    void *dev = nullptr;  // the same memory as the GPU addresses it (kernels write results straight into it)
    void release() { if (p) (void)hipHostFree(p); p = nullptr; dev = nullptr; cap = 0; }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

// ---- small helper kernels ----------------------------------------------------------------------
// float32 -> bfloat16 (round to nearest even, NaN stays NaN) and back: the rows that stay resident
// and the exactly widened copy the BMU kernels read
__global__ void round_bf16_kernel(float *__restrict__ x, uint16_t *__restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t u = __float_as_uint(x[i]);
        uint16_t b;
        if ((u & 0x7fffffffu) > 0x7f800000u) b = 0x7fc0;
        else b = (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
        out[i] = b;
        x[i] = __uint_as_float(((uint32_t)b) << 16);
    }
}
__global__ void widen_bf16_kernel(const uint16_t *__restrict__ in, float *__restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = __uint_as_float(((uint32_t)in[i]) << 16);
}
__global__ void f64_to_f32_kernel(const double *__restrict__ in, float *__restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (float)in[i];
}
__global__ void u64_to_f64_kernel(const unsigned long long *__restrict__ in, double *__restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (double)in[i];
}
// The epoch's small results in ONE kernel straight into page-locked host memory (no copy engine
// round trips behind the last kernel: four queued D2H copies cost ~30 us per epoch):
// out = [a (M) | E (M) | change_total | status | sum of candidate-list lengths | the same of a pruning probe |
//        workgroups of the pruning form with long lists]
__global__ void pack_results_kernel(const double *__restrict__ aE, int64_t M, const double *__restrict__ chg,
                                    const double *__restrict__ status, const unsigned long long *__restrict__ list_sum,
                                    double *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * M; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = aE[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out[2 * M] = chg[0];
        out[2 * M + 1] = status[0];
        out[2 * M + 2] = list_sum ? (double)list_sum[0] : 0.0;
        out[2 * M + 3] = list_sum ? (double)list_sum[1] : 0.0;  // (of a counting-only pruning launch)
        out[2 * M + 4] = list_sum ? (double)list_sum[2] : 0.0;  // (workgroups whose pruned lists came out long)
    }
    __threadfence_system();
}
// shift[p] >= |a_p - b_p| (Euclidean, rows of two M x ld matrices) for p < rows, +inf beyond: how far a
// prototype has moved since the distances of the hint were measured (filter.hip 2c); one wavefront per row
__global__ __launch_bounds__(64) void row_shift_kernel(const double *__restrict__ A, const double *__restrict__ B,
                                                       int64_t ld, int64_t d, int64_t rows, int64_t M,
                                                       double *__restrict__ shift) {
    const int64_t p = blockIdx.x;
    if (p >= M) return;
    if (p >= rows) { if (threadIdx.x == 0) shift[p] = INFINITY; return; }
    double acc = 0.0;
    if (A != B)
        for (int64_t k = threadIdx.x; k < d; k += 64) {
            const double t = A[p * ld + k] - B[p * ld + k];
            acc = fma(t, t, acc);
        }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    // (rounding of d squares and sums: relative d 2^-52 at most; NaN rows give NaN -- "no bound")
    if (threadIdx.x == 0) shift[p] = sqrt(acc) * (1.0 + 1e-9);
}
__global__ void status_to_f64_kernel(const int32_t *__restrict__ status, double *__restrict__ out) {
    out[0] = status[0] ? 1.0 : 0.0;
}
// rows of a (rows x ld) matrix of element size ES gathered by index, 16 bytes per thread where aligned
template <typename T>
__global__ void gather_rows_kernel(const T *__restrict__ src, int64_t ld, const int32_t *__restrict__ ids,
                                   int64_t first, int64_t n, int64_t cols, T *__restrict__ dst) {
    const int64_t r = blockIdx.x;
    if (r >= n) return;
    const T *s = src + (int64_t)ids[first + r] * ld;
    T *d = dst + r * cols;
    for (int64_t c = threadIdx.x; c < cols; c += blockDim.x) d[c] = s[c];
}
__global__ void gather_i32_kernel(const int32_t *__restrict__ src, const int32_t *__restrict__ ids, int64_t first,
                                  int64_t n, int32_t *__restrict__ dst) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = src[ids[first + i]];
}
template <typename T>
__global__ void rows_to_f64_kernel(const T *__restrict__ src, int64_t ld, const int64_t *__restrict__ ids, int64_t n,
                                   int64_t cols, double *__restrict__ dst) {
    const int64_t r = blockIdx.x;
    if (r >= n) return;
    const T *s = src + ids[r] * ld;
    for (int64_t c = threadIdx.x; c < cols; c += blockDim.x) dst[r * cols + c] = widen(s[c]);
}
__global__ void seg_counts_kernel(const uint32_t *__restrict__ seg_start, int64_t M, int64_t N, int64_t *__restrict__ counts) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < M; j += (int64_t)gridDim.x * blockDim.x)
        counts[j] = (int64_t)((j + 1 < M ? seg_start[j + 1] : (uint32_t)N) - seg_start[j]);
}

static unsigned grid1d(int64_t n, int block = 256) {
    const int64_t nb = (n + block - 1) / block;
    return (unsigned)(nb < 1 ? 1 : (nb > 8192 ? 8192 : nb));
}

}  // namespace dbgsom

using namespace dbgsom;

// the device-level ABI the engine drives (defined in filter.hip / stats.hip)
extern "C" {
const unsigned long long *dbgsom_filter_count_sum_ptr(const void *workspace_dev, int64_t N, int64_t d, int64_t M);
size_t dbgsom_filter_planes_bytes(int64_t rows, int64_t d);
size_t dbgsom_bmu_filtered_workspace_bytes(int64_t N, int64_t d, int64_t M);
}

namespace {

constexpr int64_t FILTER_MIN_PROTOTYPES = 129;  // at or below 128 one chunk of the all-pairs kernel is cheaper (measured)
constexpr int64_t FILTER_MAX_FEATURES = 43690;  // int32 digit-product accumulators: 3 x 128 x 128 x d < 2^31
constexpr int FILTER_BACKOFF = 8;
// epochs an arm is kept before its alternatives get another look: a map in training changes (early,
// nearly collapsed maps want a fine sweep, organised ones the coarse one), and only running an arm
// tells how long its lists are; a look costs one epoch of an arm that could at best be cheaper
constexpr int PLANES_REPROBE = 16;
// cost model of the candidate sweep (per prototype, in units of the three-product sweep) against a
// list entry of the exact stage -- measured at C4 with this build's kernels: one product 1.00 ms,
// three 2.87 ms per 1024 prototypes; exact stage 1.16 ms per 33 list entries
const double SWEEP_COST[4] = {0.0, 0.35, 1.0, 1.96};
constexpr double LIST_COST = 12.5;
// arm 0 of the policy: no sweep, candidates from the triangle inequality (filter.hip 2c).  In the same
// units: one pass over the X plane ~ 170 prototypes of the one-product sweep (C4: 0.17 of 1.0 ms per
// 1024), plus the M x M gap matrix (three products, a third of the sweep's rate: ~ 9 M / N sweeps)
constexpr double PRUNE_PASS_COST = 60.0;
constexpr int64_t PRUNE_MAX_M = 8192;

struct Samples {  // one resident sample set (training samples, or a query batch)
    int dtype = -1;            // storage dtype in HBM
    int64_t N = 0, d = 0, dp = 0;
    const void *X = nullptr;   // N x dp, storage dtype (own.p or borrowed)
    const void *Xb = nullptr;  // what the BMU kernels read: X, or the float32 copy of bfloat16 rows
    int bdtype = -1;
    DevBuf own, x32, xx, planes;
    bool planes_ready = false;
    void release() { own.release(); x32.release(); xx.release(); planes.release(); planes_ready = false; dtype = -1; N = 0; X = Xb = nullptr; }
};

}  // namespace

struct dbgsom_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // options
    int algorithm = DBGSOM_ALG_AUTO;
    int sweep_planes = 0;
    int seed_stride = 0;
    int timing = 0;
    int use_graph = 0;
    // per-sample refinement in front of the exact stage (filter.hip 2d): 0 = off, 1 = on, 2 = by measurement
    // (per arm of the search policy, once it has settled on the arm: the first training epochs are timed with and
    // without it -- wall clock of the whole blocking epoch call, best of two -- and the faster form kept;
    // measured again when the arm's lists change by a quarter)
    int refine = 2;
    struct RefineTimes {
        double ms[2] = {NAN, NAN};  // epoch without / with the refinement: best of two
        int n[2] = {0, 0};
        double mean_ref = NAN;      // the lists these were measured on
    };
    RefineTimes rf[12];             // [4 seeds + planes]
    int rf_arm = 0;                 // the arm of the running call
    int defer = 0;                  // with the refinement: distances of decided samples inside the sums kernel
                                    // (experimental, off: one chain wavefront per CU cannot keep up -- NOTES.md)
    bool last_deferred = false;
    int64_t defer_epochs = 0;       // epochs whose sums kernel also evaluated the distances the refinement left open
    int last_round_f32 = 0;
    int64_t rf_M = -1;
    int rf_measuring = -1;          // the form the running call is timing (-1: none)
    bool last_refined = false;
    bool last_k2_filtered = false;  // the last k = 2 search went through the pruning form
    int64_t filter_min_query_rows = 32768;
    int64_t max_mean_candidates = 320;
    // samples
    Samples xs, xq;
    DevBuf y;
    bool has_labels = false;
    // topology
    int64_t topoM = 0;
    DevBuf hop, hop_stage;
    // prototypes: two M x dp float64 buffers
    DevBuf Wb[2];
    int cur = 0;
    int64_t M = 0;       // rows of the resident prototypes (Wb[cur])
    int64_t otherM = 0;  // rows of Wb[cur ^ 1]
    // per-epoch device state
    DevBuf ww, idx[2], dist, kw, sums, acc_ws, sm_ws, filt_ws, scal, qidx, qdist, red, hist, stage_dev;
    int icur = 0;            // idx[icur]: winners of the last epoch (the hint)
    bool hint_valid = false;
    int64_t hintM = 0;
    bool last_idx_valid = false;  // idx[icur] holds the winners of the last epoch / partition
    int64_t sumsM = 0;
    // partition (vertical growth)
    DevBuf part_order, part_ws, part_counts;
    int64_t partM = 0;
    bool part_valid = false;
    // policy
    int filter_backoff = 0, filter_fail = 0;
    int planes_next = 1, planes_used = 1;   // 1 .. 3 digit planes of the sweep; 0 = no sweep (triangle pruning)
    bool probe_next = false, last_probed = false;  // a counting-only pruning launch beside the sweep
    bool last_retry = false;
    bool exploring_next = false;   // the next epoch runs an arm that has never run on this map (adapt_arms)
    bool last_guarded = false;     // the last filtered search stopped at its lists and ran all pairs instead
    int64_t guarded_calls = 0;
    bool prune_retry = false;  // the last pruning launch met workgroups with poor seeds: re-seed those (DBGSOM_PRUNE_RETRY)
    // `dist` holds the exact distances of the resident samples to the rows idx[icur] of Wb[distW_buf]
    // (distW_M rows) as the last epoch's search left them: the hinted pruning bound (filter.hip 2c)
    bool dist_bound_valid = false;
    int distW_buf = 0;
    int64_t distW_M = 0;
    DevBuf shiftb;
    double last_probe_mean = NAN;
    int64_t planeM = -1;
    double arm_known[3][4] = {{NAN, NAN, NAN, NAN}, {NAN, NAN, NAN, NAN}, {NAN, NAN, NAN, NAN}};  // [seeds][planes]
    double arm_seen[3][4] = {{NAN, NAN, NAN, NAN}, {NAN, NAN, NAN, NAN}, {NAN, NAN, NAN, NAN}};   // last result ever
    int arm_age[3][4] = {};   // updates of the map since the arm last ran
    // measured: wall clock (ms) of the blocking epoch call when the arm last ran WITHOUT anything riding along (a
    // counting-only launch, the refinement's own measurements, results copied to the host).  The re-seeding passes
    // of the pruning arm are NOT "riding along": `prune_retry` stays on for as long as the arm's cheap seeds leave
    // workgroups with long lists, so they are what the arm costs on this data (round 3's advisor asked to drop such
    // epochs: with the flag sticky no arm would ever be timed again -- tests/test_gpu_parity.py
    // test_search_arms_that_have_been_timed_...); sweep arms do not look at the flag at all.
    // two arms that both have one are compared by it, the cost model only prices arms that have none
    double arm_ms[3][4] = {{NAN, NAN, NAN, NAN}, {NAN, NAN, NAN, NAN}, {NAN, NAN, NAN, NAN}};
    double last_epoch_ms = NAN;     // of the epoch update_policy is looking at (NaN: not a clean measurement)
    int arm_duels[3][4] = {};       // clean epochs an arm was given only to be timed
    int arm_wait[3][4] = {{16, 16, 16, 16}, {16, 16, 16, 16}, {16, 16, 16, 16}};
    bool last_frozen = false;
    int plane_hold = 0;
    int seed_mode = 0;   // stateless seeds: 0 = the cheap pre-pass, 1 = the full one
    double best_mean = NAN;  // list length of the cheapest known arm (what the back-off looks at)
    bool last_seed_full = false;
    // last epoch
    bool last_filtered = false, last_hinted = false;
    double last_mean = NAN;
    int64_t last_filter_M = 0, last_filter_N = 0, last_filter_d = 0;
    const void *last_filter_ws = nullptr;
    // staging
    PinBuf tail, counts;
    // collective: the caller's callback, or RCCL driven from here
    dbgsom_allreduce_fn allreduce = nullptr;
    void *allreduce_user = nullptr;
    void *rccl_comm = nullptr;
    bool rccl_owned = false;
    // smoothing sharded over the ranks (columns of W'): the epoch's collective is then a reduce-scatter of column
    // blocks of the sums and an all-gather of the W' blocks (smooth.hip).  shard_smooth: 0 = never, 1 = whenever
    // the collective can do it, 2 = when the smoothing GEMM is large (default).  Needs rank / nranks: from the
    // RCCL communicator, or from dbgsom_ctx_set_collectives.
    dbgsom_collective_fn coll = nullptr;
    void *coll_user = nullptr;
    int coll_rank = 0, coll_nranks = 1;
    int shard_smooth = 2;
    bool sums_sharded = false;   // the last accumulate step left the reduced sums as this rank's block
    int64_t shard_epochs = 0;    // epochs smoothed that way (diagnostics)
    DevBuf shard_send, shard_gather;
    // traffic of the prototypes across PCIe (f-4 evidence: whole matrices only at the first epoch, at
    // growth steps and at the end of a fit)
    int64_t w_up_calls = 0, w_up_bytes = 0, w_down_calls = 0, w_down_bytes = 0, w_row_writes = 0, w_row_reads = 0;
    // timing
    hipEvent_t ev[4] = {};
    bool ev_created = false, ev_valid = false;
    double filter_ms[5] = {0, 0, 0, 0, 0};
    bool filter_ms_valid = false;
    FilterAux faux;   // this context's stage timer and side streams of the filtered search (FilteredCall::aux)
};

namespace {

#define CTX_CHECK(c)                                                                 \
    do {                                                                             \
        if (!(c)) { set_error("%s: null context", __func__); return DBGSOM_EINVAL; } \
        DBGSOM_HIP_CHECK(hipSetDevice((c)->device));                                 \
    } while (0)
#define TRY(expr) do { int _rc = (expr); if (_rc != DBGSOM_OK) return _rc; } while (0)

inline int64_t pad16(int64_t d) { return (d + 15) / 16 * 16; }

int sync(dbgsom_ctx *c) {
    DBGSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
    return DBGSOM_OK;
}

// host (rows x d, element size es) -> device (rows x dp), zeros behind column d
int upload_padded(dbgsom_ctx *c, void *dst, const void *src, int64_t rows, int64_t d, int64_t dp, size_t es) {
    if (rows == 0) return DBGSOM_OK;
    if (d == dp) {
        DBGSOM_HIP_CHECK(hipMemcpyAsync(dst, src, (size_t)rows * d * es, hipMemcpyHostToDevice, c->stream));
    } else {
        DBGSOM_HIP_CHECK(hipMemsetAsync(dst, 0, (size_t)rows * dp * es, c->stream));
        DBGSOM_HIP_CHECK(hipMemcpy2DAsync(dst, (size_t)dp * es, src, (size_t)d * es, (size_t)d * es, (size_t)rows,
                                          hipMemcpyHostToDevice, c->stream));
    }
    return DBGSOM_OK;
}

int download_unpadded(dbgsom_ctx *c, void *dst, const void *src, int64_t rows, int64_t d, int64_t dp, size_t es) {
    if (rows == 0) return DBGSOM_OK;
    if (d == dp)
        DBGSOM_HIP_CHECK(hipMemcpyAsync(dst, src, (size_t)rows * d * es, hipMemcpyDeviceToHost, c->stream));
    else
        DBGSOM_HIP_CHECK(hipMemcpy2DAsync(dst, (size_t)d * es, src, (size_t)dp * es, (size_t)d * es, (size_t)rows,
                                          hipMemcpyDeviceToHost, c->stream));
    return DBGSOM_OK;
}

// norms + BMU view of a freshly placed sample set (s.X, s.dtype, s.N, s.d, s.dp set)
int finish_samples(dbgsom_ctx *c, Samples &s, bool x32_is_widened) {
    if (s.dtype == DBGSOM_BF16) {
        if (!x32_is_widened) {
            TRY(s.x32.reserve((size_t)s.N * s.dp * 4));
            hipLaunchKernelGGL(widen_bf16_kernel, dim3(grid1d(s.N * s.dp)), dim3(256), 0, c->stream,
                               (const uint16_t *)s.X, s.x32.as<float>(), s.N * s.dp);
            TRY(launch_status("widen_bf16_kernel"));
        }
        s.Xb = s.x32.p;
        s.bdtype = DBGSOM_F32;
    } else {
        s.Xb = s.X;
        s.bdtype = s.dtype;
    }
    TRY(s.xx.reserve((size_t)s.N * 8));
    TRY(launch_row_sqnorms(s.Xb, s.bdtype, s.N, s.dp, s.dp, s.xx.as<double>(), c->stream));
    s.planes_ready = false;
    return DBGSOM_OK;
}

int place_host_samples(dbgsom_ctx *c, Samples &s, const void *X_host, int x_dtype, int64_t N, int64_t d, int storage) {
    DBGSOM_REQUIRE(valid_dtype(x_dtype) && valid_dtype(storage), "dtype must be DBGSOM_F32/F64/BF16");
    DBGSOM_REQUIRE(X_host && N >= 1 && d >= 1 && N < 0x7fffffff, "bad samples");
    DBGSOM_REQUIRE(storage == x_dtype || (storage == DBGSOM_BF16 && x_dtype == DBGSOM_F32),
                   "storage must be the input dtype, or DBGSOM_BF16 for float32 input");
    const int64_t dp = pad16(d);
    s.N = N; s.d = d; s.dp = dp; s.dtype = storage;
    bool widened = false;
    if (storage == DBGSOM_BF16 && x_dtype == DBGSOM_F32) {
        TRY(s.x32.reserve((size_t)N * dp * 4));
        TRY(s.own.reserve((size_t)N * dp * 2));
        TRY(upload_padded(c, s.x32.p, X_host, N, d, dp, 4));
        hipLaunchKernelGGL(round_bf16_kernel, dim3(grid1d(N * dp)), dim3(256), 0, c->stream, s.x32.as<float>(),
                           s.own.as<uint16_t>(), N * dp);
        TRY(launch_status("round_bf16_kernel"));
        widened = true;
    } else {
        const size_t es = dtype_size(x_dtype);
        TRY(s.own.reserve((size_t)N * dp * es));
        TRY(upload_padded(c, s.own.p, X_host, N, d, dp, es));
    }
    s.X = s.own.p;
    return finish_samples(c, s, widened);
}

int ensure_planes(dbgsom_ctx *c, Samples &s) {
    if (s.planes_ready) return DBGSOM_OK;
    const size_t nbytes = dbgsom_filter_planes_bytes(s.N, s.dp);
    TRY(s.planes.reserve(nbytes));
    TRY(dbgsom_filter_prepare(s.Xb, s.bdtype, s.N, s.dp, s.dp, s.planes.p, s.planes.cap, c->stream));
    s.planes_ready = true;
    return DBGSOM_OK;
}

bool filter_shape_ok(const Samples &s, int64_t M) {
    return M >= FILTER_MIN_PROTOTYPES && M <= DBGSOM_MAX_PROTOTYPES && s.dp <= FILTER_MAX_FEATURES && s.N >= 1;
}

bool filter_applies(const dbgsom_ctx *c, int64_t M) {
    if (c->algorithm == DBGSOM_ALG_EXACT) return false;
    if (c->algorithm == DBGSOM_ALG_AUTO && c->filter_backoff > 0) return false;
    return filter_shape_ok(c->xs, M);
}

// what the next filtered search runs: 1 .. 3 digit planes, 0 = triangle pruning (option value 4)
int planes_for_call(const dbgsom_ctx *c) {
    return c->sweep_planes ? (c->sweep_planes == 4 ? 0 : c->sweep_planes) : c->planes_next;
}
// the (seed_stride, sweep_planes) arguments of dbgsom_bmu_filtered for arm `planes`
void filter_call_args(int planes, bool probe, bool retry, int64_t M, int *stride, int *planes_arg) {
    *planes_arg = planes ? planes : 1;
    if (!planes && M <= PRUNE_MAX_M) *stride |= DBGSOM_PRUNE | (retry ? DBGSOM_PRUNE_RETRY : 0);
    else if (probe && M <= PRUNE_MAX_M) *stride |= DBGSOM_PRUNE_PROBE | (retry ? DBGSOM_PRUNE_RETRY : 0);
}

// prototypes: make W (host, or the resident ones) the consumed matrix Wb[cur]; norms into ww
int stage_weights(dbgsom_ctx *c, const double *W_host, int64_t M, int64_t d, int64_t dp) {
    DBGSOM_REQUIRE(M >= 1 && M <= 0x7fffff00, "bad prototype count");
    if (W_host) {
        TRY(c->Wb[c->cur].reserve((size_t)M * dp * 8));
        TRY(upload_padded(c, c->Wb[c->cur].p, W_host, M, d, dp, 8));
        if (c->distW_buf == c->cur) c->dist_bound_valid = false;  // the matrix the hint's distances refer to is gone
        c->M = M;
        ++c->w_up_calls; c->w_up_bytes += M * d * 8;
    } else if (c->M != M) {
        set_error("no resident prototypes of %lld rows (resident: %lld); pass W_host or call dbgsom_ctx_set_weights",
                  (long long)M, (long long)c->M);
        return DBGSOM_ESTATE;
    }
    TRY(c->ww.reserve((size_t)M * 8));
    return launch_row_sqnorms(c->Wb[c->cur].p, DBGSOM_F64, M, dp, dp, c->ww.as<double>(), c->stream);
}

double bearable_mean(const dbgsom_ctx *c, int64_t M);

// k = 1 search through the int8 filter; seeds = previous winners when `hinted`
int run_filtered(dbgsom_ctx *c, Samples &s, DevBuf &ws, const double *W, int64_t M, int round_f32,
                 const int64_t *prev_idx, const int32_t *order, int64_t *idx, double *dist, bool may_probe = false,
                 bool allow_defer = false) {
    TRY(ensure_planes(c, s));
    TRY(ws.reserve_zeroed(dbgsom_bmu_filtered_workspace_bytes(s.N, s.dp, M), c->stream));
    c->planes_used = planes_for_call(c);
    if (c->planes_used == 0 && M > PRUNE_MAX_M) c->planes_used = 1;
    c->last_seed_full = !prev_idx && c->seed_mode == 1 && c->seed_stride == 0;
    int stride = c->last_seed_full ? DBGSOM_SEED_FULL : c->seed_stride, planes_arg = 1;
    // (a probe is read by the policy after a training epoch; the first epoch of a map size always has one)
    c->last_probed = may_probe && c->sweep_planes == 0 && (c->probe_next || c->planeM != M) && c->planes_used != 0 &&
                     M <= PRUNE_MAX_M;
    if (may_probe) c->probe_next = false;
    // seeds = the last epoch's winners, and their exact distances are still around: the pruning
    // bound need not read X for samples whose prototype has hardly moved
    FilteredCall call;
    call.aux = &c->faux;
    if (may_probe && prev_idx && c->dist_bound_valid && dist == c->dist.as<double>() && M <= PRUNE_MAX_M &&
        (c->planes_used == 0 || c->last_probed) && c->Wb[c->distW_buf].p) {
        TRY(c->shiftb.reserve((size_t)M * 8));
        const int64_t rows = c->distW_M < M ? c->distW_M : M;
        hipLaunchKernelGGL(row_shift_kernel, dim3((unsigned)M), dim3(64), 0, c->stream, W,
                           c->Wb[c->distW_buf].as<double>(), s.dp, s.dp, rows, M, c->shiftb.as<double>());
        TRY(launch_status("row_shift_kernel"));
        call.hint_dist = c->dist.as<double>();
        call.hint_shift = c->shiftb.as<double>();
    }
    c->last_retry = c->prune_retry;
    filter_call_args(c->planes_used, c->last_probed, c->prune_retry, M, &stride, &planes_arg);
    call.X = s.Xb; call.x_dtype = s.bdtype; call.N = s.N; call.d = s.dp; call.ldx = s.dp;
    call.xx = s.xx.as<double>(); call.xplanes = s.planes.p; call.W = W; call.M = M; call.ww = c->ww.as<double>();
    call.prev_idx = prev_idx; call.order = order; call.seed_stride = stride; call.sweep_planes = planes_arg;
    call.round_f32 = round_f32; call.idx = idx; call.dist = dist; call.ws = ws.p; call.ws_bytes = ws.cap;
    call.stream = c->stream;
    // the tile of the refinement's first list-length class: what the lists were last time, with some room
    // (longer lists go to its largest tile, beyond that to the matrix-core stage)
    int rf_rows = 64;
    const bool mean_known = c->last_mean == c->last_mean && c->last_filter_M == M;
    if (mean_known) rf_rows = (int)(c->last_mean * 1.25 + 8.0);
    bool use_refine = c->refine == 1;
    c->rf_measuring = -1;
    c->rf_arm = 4 * (prev_idx ? 2 : (c->last_seed_full ? 1 : 0)) + c->planes_used;
    if (c->refine == 2 && mean_known) {
        if (c->rf_M > 0 && llabs((long long)(M - c->rf_M)) * 4 <= (long long)c->rf_M) c->rf_M = M;   // (a growth step: see adapt_arms)
        if (c->rf_M != M) {
            c->rf_M = M;
            for (auto &r : c->rf) r = dbgsom_ctx::RefineTimes();
        }
        dbgsom_ctx::RefineTimes &r = c->rf[c->rf_arm];
        // (the lists of THIS arm: what it left the last time it ran, else what the last epoch had)
        const double arm_mean = c->arm_known[c->rf_arm >> 2][c->rf_arm & 3];
        const double lists = arm_mean == arm_mean ? arm_mean : c->last_mean;
        if (!(fabs(lists - r.mean_ref) <= 0.25 * r.mean_ref)) { r = dbgsom_ctx::RefineTimes(); r.mean_ref = lists; }
        // prior (what has not been measured is not tried blind): the refinement reads two digit planes and the
        // rows once more whatever the lists are -- short lists, few features or a few workgroups never pay;
        // lists beyond twice its largest tile stay the matrix-core stage's anyway
        const bool eligible = lists >= 24.0 && lists <= 400.0 && s.dp >= 256 && s.N >= 65536;
        // timed only on an arm the policy has settled on (or the caller fixed): two forms of the SAME search
        const bool settled = c->sweep_planes != 0 || c->plane_hold > 0;
        if (!eligible) use_refine = false;
        else if (may_probe && settled && r.n[0] < 2) { use_refine = false; c->rf_measuring = 0; }
        else if (may_probe && settled && r.n[1] < 2) { use_refine = true; c->rf_measuring = 1; }
        else use_refine = r.n[0] >= 2 && r.n[1] >= 2 && r.ms[1] < r.ms[0];
    }
    call.refine_rows = use_refine ? rf_rows : 0;
    c->last_refined = use_refine;
    // an epoch's accumulate step can evaluate the distances of the samples the refinement decided without
    // looking at their float rows: one pass over those rows for the distance AND the sums
    call.defer_dist = use_refine && allow_defer && c->defer && M < 0xffff &&
                      accumulate_can_fill_distances(s.dtype, s.dp);
    c->last_deferred = call.defer_dist;
    c->last_round_f32 = round_f32;
    if (s.dtype == DBGSOM_BF16) { call.X_store = s.X; call.store_dtype = DBGSOM_BF16; call.ld_store = s.dp; }
    // An arm on trial (`auto`, a training epoch, nothing known of this arm on this map): the call stops at lists
    // that average more than the policy bears and the all-pairs kernel finds the winners instead -- the policy
    // still learns what the arm leaves, for the price of its sweep instead of an exact stage over the whole map.
    c->last_guarded = false;
    const int arm_row = prev_idx ? 2 : (c->last_seed_full ? 1 : 0);
    if (may_probe && c->algorithm == DBGSOM_ALG_AUTO && call.k == 1 &&
        (c->planeM != M || isnan(c->arm_seen[arm_row][c->planes_used])))
        call.guard_mean = bearable_mean(c, M);
    const int rc_f = launch_bmu_filtered(call);
    c->last_filter_M = M; c->last_filter_N = s.N; c->last_filter_d = s.dp; c->last_filter_ws = ws.p;
    if (rc_f == DBGSOM_LISTS_LONG) {
        c->last_guarded = true;
        ++c->guarded_calls;
        c->last_refined = false;
        c->last_deferred = false;
        c->rf_measuring = -1;
        return launch_bmu(s.Xb, s.bdtype, s.N, s.dp, s.dp, s.xx.as<double>(), W, M, c->ww.as<double>(), 1, round_f32, idx,
                          dist, c->stream);
    }
    TRY(rc_f);
    return DBGSOM_OK;
}

// librccl, resolved at run time (no link-time dependency: single-GPU callers never load it)
struct RcclApi {
    int (*get_unique_id)(void *id) = nullptr;                                        // ncclGetUniqueId
    int (*comm_init_rank)(void **comm, int nranks, /* ncclUniqueId by value */ ...) = nullptr;
    int (*all_reduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*reduce_scatter)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;   // (optional)
    int (*all_gather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;            // (optional)
    int (*comm_count)(void *, int *) = nullptr;                                                     // (optional)
    int (*comm_user_rank)(void *, int *) = nullptr;                                                 // (optional)
    int (*comm_destroy)(void *) = nullptr;
    const char *(*error_string)(int) = nullptr;
    bool tried = false, ok = false;
};
struct NcclUniqueId { char internal[128]; };   // (rccl.h: ncclUniqueId)
typedef int (*nccl_comm_init_rank_fn)(void **comm, int nranks, NcclUniqueId id, int rank);
RcclApi g_rccl;
nccl_comm_init_rank_fn g_rccl_init = nullptr;

int rccl_load() {
    if (g_rccl.tried) {
        if (!g_rccl.ok) { set_error("librccl could not be loaded (see the first failure)"); return DBGSOM_ESTATE; }
        return DBGSOM_OK;
    }
    g_rccl.tried = true;
    void *h = nullptr;
    // a copy already in the process (PyTorch's) shares the HIP runtime that is in use: take it.  RTLD_DEFAULT is a
    // null pointer on glibc, so "found in the process" is a flag of its own, not a non-null handle
    const bool in_process = dlsym(RTLD_DEFAULT, "ncclAllReduce") != nullptr;
    if (in_process) h = RTLD_DEFAULT;
    const char *env = getenv("DBGSOM_RCCL_LIB");
    const char *names[] = {env, "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char *n : names) {
        if (in_process || h) break;
        if (n && *n) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!h && !in_process) { set_error("librccl not found (set DBGSOM_RCCL_LIB): %s", dlerror()); return DBGSOM_ESTATE; }
    g_rccl.get_unique_id = (int (*)(void *))dlsym(h, "ncclGetUniqueId");
    g_rccl_init = (nccl_comm_init_rank_fn)dlsym(h, "ncclCommInitRank");
    g_rccl.all_reduce = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(h, "ncclAllReduce");
    g_rccl.comm_destroy = (int (*)(void *))dlsym(h, "ncclCommDestroy");
    g_rccl.reduce_scatter = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(h, "ncclReduceScatter");
    g_rccl.all_gather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))dlsym(h, "ncclAllGather");
    g_rccl.comm_count = (int (*)(void *, int *))dlsym(h, "ncclCommCount");
    g_rccl.comm_user_rank = (int (*)(void *, int *))dlsym(h, "ncclCommUserRank");
    g_rccl.error_string = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
    g_rccl.ok = g_rccl.get_unique_id && g_rccl_init && g_rccl.all_reduce && g_rccl.comm_destroy;
    if (!g_rccl.ok) { set_error("librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy"); return DBGSOM_ESTATE; }
    return DBGSOM_OK;
}
const char *rccl_err(int rc) { return g_rccl.error_string ? g_rccl.error_string(rc) : "?"; }
void drop_rccl(dbgsom_ctx *c) {
    if (c->rccl_comm && c->rccl_owned && g_rccl.ok) {
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        (void)g_rccl.comm_destroy(c->rccl_comm);
    }
    if (c->rccl_comm) { c->coll_rank = 0; c->coll_nranks = 1; }
    c->rccl_comm = nullptr;
    c->rccl_owned = false;
}

int run_allreduce(dbgsom_ctx *c, double *buf, int64_t count) {
    if (c->rccl_comm) {  // ncclDouble = 8, ncclSum = 0 (rccl.h), in place, on the context's stream
        const int rc = g_rccl.all_reduce(buf, buf, (size_t)count, 8, 0, c->rccl_comm, c->stream);
        if (rc != 0) {
            (void)hipStreamSynchronize(c->stream);
            set_error("ncclAllReduce failed (%d: %s)", rc, rccl_err(rc));
            return DBGSOM_ECALLBACK;
        }
        return DBGSOM_OK;
    }
    if (c->coll) {
        const int rc = c->coll(c->coll_user, DBGSOM_COLL_ALLREDUCE, buf, count, (void *)c->stream);
        if (rc != 0) {
            (void)hipStreamSynchronize(c->stream);
            set_error("collective callback failed (all-reduce, %d)", rc);
            return DBGSOM_ECALLBACK;
        }
        return DBGSOM_OK;
    }
    if (!c->allreduce) return DBGSOM_OK;
    const int rc = c->allreduce(c->allreduce_user, buf, count, (void *)c->stream);
    if (rc != 0) {
        (void)hipStreamSynchronize(c->stream);
        set_error("all-reduce callback failed (%d)", rc);
        return DBGSOM_ECALLBACK;
    }
    return DBGSOM_OK;
}

// in place over nranks blocks of `per` values: op = DBGSOM_COLL_REDUCE_SCATTER leaves the SUM of block `rank` in
// block `rank`; DBGSOM_COLL_ALLGATHER fills every block from its owner's
int run_block_collective(dbgsom_ctx *c, int op, double *buf, int64_t per) {
    if (c->rccl_comm) {
        double *mine = buf + (size_t)c->coll_rank * per;
        const int rc = op == DBGSOM_COLL_REDUCE_SCATTER
                           ? g_rccl.reduce_scatter(buf, mine, (size_t)per, 8, 0, c->rccl_comm, c->stream)
                           : g_rccl.all_gather(mine, buf, (size_t)per, 8, c->rccl_comm, c->stream);
        if (rc != 0) {
            (void)hipStreamSynchronize(c->stream);
            set_error("%s failed (%d: %s)", op == DBGSOM_COLL_REDUCE_SCATTER ? "ncclReduceScatter" : "ncclAllGather", rc, rccl_err(rc));
            return DBGSOM_ECALLBACK;
        }
        return DBGSOM_OK;
    }
    const int rc = c->coll ? c->coll(c->coll_user, op, buf, per, (void *)c->stream) : 1;
    if (rc != 0) {
        (void)hipStreamSynchronize(c->stream);
        set_error("collective callback failed (op %d, %d)", op, rc);
        return DBGSOM_ECALLBACK;
    }
    return DBGSOM_OK;
}

// does this epoch smooth column blocks?  (a function of the options, the collective and the shape alone:
// every rank decides the same)
bool shard_smoothing(const dbgsom_ctx *c, int64_t M, int64_t dp) {
    if (c->shard_smooth == 0 || (c->coll_nranks < 2 && c->shard_smooth != 1)) return false;   // (one rank: only when forced -- tests)
    const bool can = c->rccl_comm ? (g_rccl.reduce_scatter && g_rccl.all_gather) : c->coll != nullptr;
    if (!can) return false;
    // by default only where the replicated GEMM is worth two more launches and a second collective: from
    // ~8 GFLOP (the C5 map: 68.7; the C4 map, 1.6 GFLOP in 48 us, is bound by its chain of k-tiles, not by the
    // products, and would gain nothing)
    return c->shard_smooth == 1 || 2.0 * (double)M * (double)M * (double)dp >= 8e9;
}

// The mean list length `auto` bears before it goes back to the all-pairs kernel: the option (320), and never more
// than half the map -- in the cost model's units an arm costs (its sweep) x M + 12.5 x (mean list) against the
// all-pairs kernel's 8.4 x M, so lists beyond ~0.5 .. 0.65 M lose to it whatever the arm.  (A young, still collapsed
// map of a growing fit: at M = 130 .. 260 the lists were 0.75 M and a filtered epoch cost 8 .. 14 ms against 6 .. 7
// of all pairs -- profiles/r04_fit_trace_before.txt.)
double bearable_mean(const dbgsom_ctx *c, int64_t M) {
    return fmin((double)c->max_mean_candidates, 0.5 * (double)M);
}

// What the next filtered search runs: an ARM = (seeds, digit planes).  Seeds: 0 = the cheap stateless
// pre-pass (every 4th .. 64th prototype on three 64-feature blocks), 1 = the full one (every
// prototype, every feature: one more sweep), 2 = the previous epoch's winners (not a choice: whenever
// the caller's algorithm allows them and they exist).  Digit planes 1 .. 3 of the candidate sweep.
// Cost model per arm: seeds {1.15, 2, 1} x SWEEP_COST[planes] x M + LIST_COST x mean list length.
// The lists of an arm are only known once it has run, so the policy explores: from the cheapest
// known arm it tries an unknown arm when even EMPTY lists would make it cheaper -- its neighbours
// (one coordinate changed) first, and, when the lists are long (weakly clustered data: a seed that
// is not nearly the winner, or a bound as wide as the spread of the distances, leaves most of the
// map a candidate), the strong corner (full seeds, three products) directly: on isotropic data no
// single step leads there (full seeds alone: 1024 candidates, finer planes alone: 981, both: 17).
// When nothing is left to try it stays for PLANES_REPROBE epochs, then forgets the alternatives.
// Results never depend on any of this.
void adapt_arms(dbgsom_ctx *c, double mean, int64_t M, int64_t N) {
    c->exploring_next = false;
    const int row = c->last_hinted ? 2 : (c->last_seed_full ? 1 : 0);
    const int p = c->planes_used;
    // A growing map changes its size by a few neurons at a time: what the arms left on a map within a quarter of
    // this one's size stays the best guess there is (lists and times move with it by a few per cent, and every arm
    // is looked at again as it ages) -- forgetting it at every growth step made every step pay for the
    // exploration again, on unclustered data an epoch or two at ten times the settled cost.
    if (c->planeM > 0 && c->planeM != M && llabs((long long)(M - c->planeM)) * 4 <= (long long)c->planeM) c->planeM = M;
    if (c->planeM != M) {  // another map size: what was learnt no longer applies
        c->planeM = M;
        for (auto &r : c->arm_known) for (double &k : r) k = NAN;
        for (auto &r : c->arm_seen) for (double &k : r) k = NAN;
        for (auto &r : c->arm_ms) for (double &k : r) k = NAN;
        for (auto &r : c->arm_duels) for (int &k : r) k = 0;
        for (auto &r : c->arm_wait) for (int &k : r) k = 16;
        c->plane_hold = 0;
    }
    static const double SEED_COST[3] = {1.15, 2.0, 1.0};
    // (arm 0: the one-product pre-pass, one pass over the X plane and the gap matrix)
    auto fixed = [&](int s_, int q) {
        if (q == 0)  // (+ two short dependent launches, ~25 us: what decides on small sample sets)
            return (SEED_COST[s_] - 1.0) * SWEEP_COST[1] * (double)M + PRUNE_PASS_COST +
                   SWEEP_COST[1] * (double)M * 9.0 * (double)M / (double)(N > 0 ? N : 1) +
                   25.0 / (2.8 * ((double)(N > 0 ? N : 1) * (double)c->xs.dp) / (1.0e6 * 784.0));
        return SEED_COST[s_] * SWEEP_COST[q] * (double)M;
    };
    // an arm's age = how often the map has been UPDATED since it ran (a frozen map -- the bench, a
    // series of queries -- does not age what is known about it)
    if (!c->last_frozen)
        for (auto &r : c->arm_age) for (int &a : r) ++a;
    const bool remeasured = !isnan(c->arm_seen[row][p]);
    c->arm_known[row][p] = c->arm_seen[row][p] = mean;
    c->arm_age[row][p] = 0;
    // (an arm that left more than the policy bears gets its next look late: a look at it costs an all-pairs epoch)
    if (mean > bearable_mean(c, M)) c->arm_wait[row][p] = 128;
    if (!isnan(c->last_epoch_ms))   // (the mean of the last two looks: one epoch's clock jitters by a few per cent)
        c->arm_ms[row][p] = isnan(c->arm_ms[row][p]) ? c->last_epoch_ms : 0.5 * (c->arm_ms[row][p] + c->last_epoch_ms);
    if (c->last_probed) {  // what arm 0 would have produced from the same seeds
        c->arm_known[row][0] = c->arm_seen[row][0] = c->last_probe_mean;
        c->arm_age[row][0] = 0;
    }
    auto allowed = [&](int s_, int q) {
        if (c->sweep_planes && q != (c->sweep_planes == 4 ? 0 : c->sweep_planes)) return false;  // fixed by the caller
        if (q == 0 && M > PRUNE_MAX_M) return false;
        if (row == 2) return s_ == 2;                                      // hinted: only the planes vary
        return s_ == 0 || (s_ == 1 && c->seed_stride == 0);               // a caller's stride: cheap seeds only
    };
    if (c->plane_hold > 0) {
        c->best_mean = mean;
        // A contender the model prices within a factor of two of this arm and that has never run clean: one epoch
        // of it, on its own, and the clock decides between the two (once per arm until it ages out).
        if (!isnan(c->arm_ms[row][p])) {
            int ds = -1, dq = -1;
            double dc = 2.0 * (fixed(row, p) + LIST_COST * mean);
            for (int s_ = 0; s_ < 3; ++s_)
                for (int q = 0; q <= 3; ++q)
                    if (allowed(s_, q) && !(s_ == row && q == p) && !isnan(c->arm_known[s_][q]) &&
                        isnan(c->arm_ms[s_][q]) && c->arm_duels[s_][q] < 1) {
                        const double cst = fixed(s_, q) + LIST_COST * c->arm_known[s_][q];
                        if (cst < dc) { dc = cst; ds = s_; dq = q; }
                    }
            if (ds >= 0) {
                ++c->arm_duels[ds][dq];
                c->seed_mode = ds == 1 ? 1 : 0;
                c->planes_next = dq;
                c->plane_hold = 0;
                c->exploring_next = true;
                return;
            }
        }
        if (--c->plane_hold == 0) {
            // The alternatives get another look once the map has moved on: an arm whose sweep /
            // pre-pass costs LESS than the current one after arm_wait (16, doubling up to 128 every
            // time the look does not pay) updates of the map -- one epoch that can only be dearer by
            // its lists; a dearer arm after 128 (a look at the full pre-pass is a whole extra sweep).
            for (int s_ = 0; s_ < 3; ++s_)
                for (int q = 0; q <= 3; ++q) {
                    if ((s_ == row && q == p) || isnan(c->arm_known[s_][q])) continue;
                    const int wait = fixed(s_, q) < fixed(row, p) ? c->arm_wait[s_][q] : 128;
                    if (c->arm_age[s_][q] >= wait) { c->arm_known[s_][q] = c->arm_ms[s_][q] = NAN; c->arm_duels[s_][q] = 0; }
                }
        }
        return;
    }
    // An arm that has been timed costs what it took; the model prices the others.  Both in the model's units: the
    // timed arms give the units per millisecond (geometric mean of model cost / time over them).
    double log_sum = 0.0;
    int n_timed = 0;
    for (int s_ = 0; s_ < 3; ++s_)
        for (int q = 0; q <= 3; ++q)
            if (!isnan(c->arm_known[s_][q]) && c->arm_ms[s_][q] > 0.0) {
                log_sum += log((fixed(s_, q) + LIST_COST * c->arm_known[s_][q]) / c->arm_ms[s_][q]);
                ++n_timed;
            }
    const double per_ms = n_timed ? exp(log_sum / n_timed) : NAN;
    auto priced = [&](int s_, int q, double lists) {
        return c->arm_ms[s_][q] > 0.0 ? c->arm_ms[s_][q] * per_ms : fixed(s_, q) + LIST_COST * lists;
    };
    int bs = row, bp = p;
    double bc = priced(row, p, mean);
    for (int s_ = 0; s_ < 3; ++s_)
        for (int q = 0; q <= 3; ++q)
            if (allowed(s_, q) && !isnan(c->arm_known[s_][q]) && !(s_ == row && q == p)) {
                const double cst = priced(s_, q, c->arm_known[s_][q]);
                if (cst < bc) { bc = cst; bs = s_; bp = q; }
            }
    // unknown arms worth a look, cheapest optimistic cost first
    int es = -1, ep = -1;
    double ec = bc;
    auto consider = [&](int s_, int q) {
        if (s_ < 0 || s_ > 2 || q < 0 || q > 3 || !allowed(s_, q) || !isnan(c->arm_known[s_][q])) return;
        // (no list is cheaper than one step of the exact stage: 16 entries)
        const double opt = fixed(s_, q) + LIST_COST * fmin(16.0, (double)M);
        if (opt < ec) { ec = opt; es = s_; ep = q; }
    };
    const double best_mean = (bs == row && bp == p) ? mean : c->arm_known[bs][bp];
    c->best_mean = best_mean;
    if (best_mean > fmax(96.0, (double)M / 8.0)) {   // long lists: the strong corner first
        consider(bs == 2 ? 2 : 1, 2);
        if (es < 0) consider(bs == 2 ? 2 : 1, 3);
    }
    if (es < 0) {
        consider(bs, 0);  // (never run blind: see below)
        consider(bs, bp + 1); consider(bs, bp - 1);
        if (bs != 2) consider(1 - bs, bp);
    }
    // arm 0 is looked at by a counting-only launch beside an arm whose lists are known to be
    // bearable (isotropic data: the whole map survives the triangle inequality -- an exact stage
    // over such lists would cost ten ordinary epochs)
    // (the launch counts from the seeds of the call it rides on: that call uses the seeds of the arm
    //  being looked at, with a sweep whose cost is known or about to be)
    if (es >= 0 && ep == 0) {
        c->probe_next = true;
        ep = bp ? bp : 1;
    }
    if (remeasured)  // a second look at this arm: did it pay?
        c->arm_wait[row][p] = (bs == row && bp == p) ? 16 : (c->arm_wait[row][p] >= 64 ? 128 : 2 * c->arm_wait[row][p]);
    if (es >= 0) {
        c->seed_mode = es == 1 ? 1 : 0;
        c->planes_next = ep;
        c->exploring_next = true;
    } else {
        c->seed_mode = bs == 1 ? 1 : 0;
        c->planes_next = bp;
        c->plane_hold = PLANES_REPROBE;
    }
}

// after an epoch has completed: look at how long the candidate lists were, decide what comes next
void update_policy(dbgsom_ctx *c, double list_sum, double probe_sum, double retry_groups, int64_t nb, int64_t M) {
    if (!c->last_filtered) { c->last_mean = NAN; return; }
    // Workgroups of the pruning form whose lists came out long (poor cheap seeds) while the re-seeding
    // passes were off: what this call measured of arm 0 is not what the arm costs.  Turn them on and
    // measure again -- the same arm once more, or another counting-only launch.
    bool again = false;
    if (c->planes_used == 0 || c->last_probed) {
        const bool need = retry_groups > 0.0;
        if (need && !c->last_retry && !c->last_hinted && !c->last_seed_full) {   // (cheap seeds: re-seeding can help)
            c->prune_retry = true;
            if (c->planes_used == 0) {
                c->last_mean = nb ? list_sum / (double)nb : 0.0;
                return;
            }
            c->last_probed = false;
            again = true;
        } else if (!c->last_hinted && !c->last_seed_full) {
            c->prune_retry = need;
        }
    }
    const double mean = nb ? list_sum / (double)nb : 0.0;
    c->last_mean = mean;
    c->last_probe_mean = c->last_probed && nb ? probe_sum / (double)nb : NAN;
    adapt_arms(c, mean, M, nb * 128);
    if (again) c->probe_next = true;
    if (c->algorithm == DBGSOM_ALG_AUTO) {
        // (the cheapest arm known so far, not an arm that is only being looked at)
        // (not while an arm that has never run on this map is up next: on unclustered data the strong corner --
        //  good seeds and a finer sweep -- is what works, and eight all-pairs epochs in front of its first try
        //  cost forty settled ones)
        if (c->best_mean > bearable_mean(c, M) && !c->exploring_next) {  // exponential back-off, capped
            c->filter_fail = c->filter_fail < 6 ? c->filter_fail + 1 : 6;
            c->filter_backoff = FILTER_BACKOFF << (c->filter_fail - 1);
        } else if (c->best_mean > bearable_mean(c, M)) {
            // (exploring)
        } else {
            c->filter_fail = 0;
        }
    }
}

void mark(dbgsom_ctx *c, int k) {
    if (!c->timing) return;
    if (!c->ev_created) { for (auto &e : c->ev) (void)hipEventCreate(&e); c->ev_created = true; }
    (void)hipEventRecord(c->ev[k], c->stream);
}

// BMU (k = 1) of the resident samples under Wb[cur] into idx[icur ^ 1] / dist, by policy
int epoch_bmu(dbgsom_ctx *c, int64_t M, int round_f32) {
    Samples &s = c->xs;
    TRY(c->idx[0].reserve((size_t)s.N * 8));
    TRY(c->idx[1].reserve((size_t)s.N * 8));
    TRY(c->dist.reserve((size_t)s.N * 8));
    int64_t *out = c->idx[c->icur ^ 1].as<int64_t>();
    const double *W = c->Wb[c->cur].as<double>();
    c->last_hinted = false;
    c->last_deferred = false;
    if (filter_applies(c, M)) {
        const bool hint = (c->algorithm == DBGSOM_ALG_AUTO || c->algorithm == DBGSOM_ALG_FILTERED_HINT) &&
                          c->hint_valid && c->hintM <= M;
        c->last_filtered = true;
        c->last_hinted = hint;
        TRY(run_filtered(c, s, c->filt_ws, W, M, round_f32, hint ? c->idx[c->icur].as<int64_t>() : nullptr,
                         hint ? c->acc_ws.as<int32_t>() : nullptr, out, c->dist.as<double>(), true, true));
    } else {
        c->last_filtered = false;
        if (c->filter_backoff > 0) --c->filter_backoff;
        TRY(launch_bmu(s.Xb, s.bdtype, s.N, s.dp, s.dp, s.xx.as<double>(), W, M, c->ww.as<double>(), 1, round_f32,
                       out, c->dist.as<double>(), c->stream));
    }
    c->icur ^= 1;
    c->hint_valid = false;  // re-established by the accumulate step that follows
    c->last_idx_valid = true;
    c->dist_bound_valid = true; c->distW_buf = c->cur; c->distW_M = M;
    return DBGSOM_OK;
}

// sums = [S | K | a | E | status] of the resident samples for winners idx / distances dist and the
// sample weights kw (kw == nullptr: the sample kernel with `gamma`, computed inside the sums kernel)
int accumulate_and_reduce(dbgsom_ctx *c, const int64_t *idx, const double *kw, double gamma, const double *dist,
                          int64_t M) {
    Samples &s = c->xs;
    DBGSOM_REQUIRE(M <= DBGSOM_MAX_PROTOTYPES, "M exceeds DBGSOM_MAX_PROTOTYPES");
    const int64_t count = M * (s.dp + 3);
    TRY(c->sums.reserve((size_t)(count + 1) * 8));
    TRY(c->acc_ws.reserve(accumulate_workspace_bytes(s.N, s.dp, M)));
    TRY(c->scal.reserve(256));
    int32_t *status = reinterpret_cast<int32_t *>(c->scal.as<char>() + 64);
    c->part_valid = false;
    if (kw) {
        TRY(launch_accumulate(s.X, s.dtype, s.N, s.dp, s.dp, idx, kw, dist, M, c->sums.as<double>(), status,
                              c->acc_ws.p, c->acc_ws.cap, c->stream));
        hipLaunchKernelGGL(status_to_f64_kernel, dim3(1), dim3(1), 0, c->stream, status, c->sums.as<double>() + count);
        TRY(launch_status("status_to_f64_kernel"));
    } else {
        DistFill fill;
        fill.W = c->Wb[c->cur].as<double>(); fill.ww = c->ww.as<double>(); fill.xx = s.xx.as<double>();
        fill.round_f32 = c->last_round_f32;
        TRY(launch_accumulate_epoch(s.X, s.dtype, s.N, s.dp, s.dp, idx, gamma, dist, M, c->sums.as<double>(), status,
                                    c->acc_ws.p, c->acc_ws.cap, c->stream, c->last_deferred ? &fill : nullptr));
        if (c->last_deferred) ++c->defer_epochs;
        c->last_deferred = false;
    }
    c->sumsM = M;
    c->sums_sharded = false;
    if (shard_smoothing(c, M, s.dp)) {
        // column blocks of S -> reduce-scatter: this rank's block holds the sums of all ranks for the columns it
        // smooths.  The small vectors [K | a | E | status] behind S go through an ordinary all-reduce where they lie:
        // growth and convergence are decided from them on every rank, so they must be the same bits everywhere (a
        // reduce-scatter sums each block along its own chain of ranks)
        const int G = c->coll_nranks;
        const int64_t blk = smooth_block_elems(M, s.dp, G);
        TRY(c->shard_send.reserve((size_t)G * blk * 8));
        double *send = c->shard_send.as<double>();
        TRY(launch_pack_blocks(c->sums.as<double>(), M, s.dp, G, send, c->stream));
        TRY(run_allreduce(c, c->sums.as<double>() + (size_t)M * s.dp, 3 * M + 1));
        TRY(run_block_collective(c, DBGSOM_COLL_REDUCE_SCATTER, send, blk));
        c->sums_sharded = true;
        c->sumsM = 0;   // (the S part of `sums` is this rank's share, not the reduced sums: nothing to read back)
        return DBGSOM_OK;
    }
    return run_allreduce(c, c->sums.as<double>(), count + 1);
}

// smoothing of the reduced sums: Wb[cur] -> Wb[cur ^ 1]; queues the small results D2H and waits
int smooth_and_fetch(dbgsom_ctx *c, int64_t M, double sigma, int layout, int flags, double *W_new_host,
                     double *change_total_host, double *errors_host, double *activations_host,
                     const int64_t *idx_dev, int64_t *idx_host, double *dist_host) {
    Samples &s = c->xs;
    const int64_t dp = s.dp, d = s.d;
    if (c->topoM != M) {
        set_error("topology holds %lld neurons, prototypes %lld (call dbgsom_ctx_set_topology after growth)",
                  (long long)c->topoM, (long long)M);
        return DBGSOM_ESTATE;
    }
    const int nxt = c->cur ^ 1;
    TRY(c->Wb[nxt].reserve((size_t)M * dp * 8));
    TRY(c->sm_ws.reserve(smooth_workspace_bytes(M, dp)));
    double *chg = c->scal.as<double>();
    double *sums = c->sums.as<double>();
    if (c->sums_sharded) {
        const int G = c->coll_nranks, r = c->coll_rank;
        const int64_t cb = smooth_block_cols(dp, G), blk = smooth_block_elems(M, dp, G);
        const double *mine = c->shard_send.as<double>() + (size_t)r * blk;
        TRY(c->shard_gather.reserve((size_t)G * M * cb * 8));
        double *gather = c->shard_gather.as<double>();
        TRY(launch_smooth_block(mine, sums + (size_t)M * dp, sums + (size_t)M * dp + M, M, cb, dp, c->hop.as<float>(), sigma,
                                layout, gather + (size_t)r * M * cb, c->sm_ws.p, c->sm_ws.cap, c->stream));
        TRY(run_block_collective(c, DBGSOM_COLL_ALLGATHER, gather, M * cb));
        TRY(launch_rowchange_blocks(gather, M, dp, cb, c->Wb[c->cur].as<double>(), c->Wb[nxt].as<double>(), chg, c->sm_ws.p,
                                    c->stream));
        c->sums_sharded = false;
        ++c->shard_epochs;
    } else {
        TRY(launch_smooth(sums, M, dp, c->hop.as<float>(), sigma, layout, c->Wb[c->cur].as<double>(),
                          c->Wb[nxt].as<double>(), chg, c->sm_ws.p, c->sm_ws.cap, c->stream));
    }
    mark(c, 3);
    // the epoch's small results: one kernel writes them into mapped page-locked memory, one stream
    // synchronisation
    TRY(c->tail.reserve((size_t)(2 * M + 5) * 8));
    double *tail = c->tail.as<double>();
    const unsigned long long *list_sum =
        c->last_filtered ? dbgsom_filter_count_sum_ptr(c->filt_ws.p, s.N, dp, M) : nullptr;
    hipLaunchKernelGGL(pack_results_kernel, dim3((unsigned)((2 * M + 255) / 256 < 64 ? (2 * M + 255) / 256 : 64)), dim3(256), 0,
                       c->stream, sums + (size_t)M * dp + M, M, chg, sums + (size_t)M * (dp + 3), list_sum,
                       reinterpret_cast<double *>(c->tail.dev));
    TRY(launch_status("pack_results_kernel"));
    if (W_new_host) {
        TRY(download_unpadded(c, W_new_host, c->Wb[nxt].p, M, d, dp, 8));
        ++c->w_down_calls; c->w_down_bytes += M * d * 8;
    }
    if (idx_host) DBGSOM_HIP_CHECK(hipMemcpyAsync(idx_host, idx_dev, (size_t)s.N * 8, hipMemcpyDeviceToHost, c->stream));
    if (dist_host) DBGSOM_HIP_CHECK(hipMemcpyAsync(dist_host, c->dist.p, (size_t)s.N * 8, hipMemcpyDeviceToHost, c->stream));
    TRY(sync(c));
    c->ev_valid = c->timing != 0;
    if (c->timing && c->last_filtered) {
        c->filter_ms_valid = filter_stage_ms(c->faux, c->filter_ms) == DBGSOM_OK;
    } else {
        c->filter_ms_valid = false;
    }
    memcpy(activations_host, tail, (size_t)M * 8);
    memcpy(errors_host, tail + M, (size_t)M * 8);
    change_total_host[0] = tail[2 * M];
    c->otherM = M;
    if (!(flags & DBGSOM_EPOCH_FROZEN)) c->cur = nxt;  // W' becomes the resident matrix, W the "previous" one
    if (tail[2 * M + 1] != 0.0) { set_error("winner index out of range"); return DBGSOM_ERANGE; }
    return DBGSOM_OK;
}

int loaded(const dbgsom_ctx *c, const char *fn) {
    if (c->xs.dtype < 0) { set_error("%s: no samples loaded", fn); return DBGSOM_ESTATE; }
    return DBGSOM_OK;
}

// BMU of the resident samples for the reductions around the path: k = 1 by policy, k = 2 all-pairs.
// Results in qidx / qdist (N x k); the training hint (idx[icur], bucket order) is left alone.
int resident_bmu(dbgsom_ctx *c, const double *W_host, int64_t M, int k, int round_f32) {
    Samples &s = c->xs;
    TRY(stage_weights(c, W_host, M, s.d, s.dp));
    TRY(c->qidx.reserve((size_t)s.N * k * 8));
    TRY(c->qdist.reserve((size_t)s.N * k * 8));
    const double *W = c->Wb[c->cur].as<double>();
    if (k == 1 && filter_applies(c, M)) {
        const bool hint = (c->algorithm == DBGSOM_ALG_AUTO || c->algorithm == DBGSOM_ALG_FILTERED_HINT) &&
                          c->hint_valid && c->hintM <= M;
        return run_filtered(c, s, c->filt_ws, W, M, round_f32, hint ? c->idx[c->icur].as<int64_t>() : nullptr,
                            hint ? c->acc_ws.as<int32_t>() : nullptr, c->qidx.as<int64_t>(), c->qdist.as<double>());
    }
    // k = 2 (topographic error, BaseSom.py:945): through the pruning form of the filtered search when the
    // training epochs have shown that it works on this data -- arm 0 of the policy has run (or been counted)
    // on this map size and left lists a fraction of the map (clustered data); otherwise all pairs
    double lists0 = NAN;
    if (c->planeM == M)
        for (int r = 0; r < 3; ++r) {
            const double v = c->arm_seen[r][0];
            if (v == v && !(lists0 <= v)) lists0 = v;
        }
    if (k == 2 && filter_applies(c, M) && M <= PRUNE_MAX_M && M >= 2 && c->last_filter_M == M && lists0 == lists0 &&
        lists0 <= bearable_mean(c, M)) {
        const bool hint = (c->algorithm == DBGSOM_ALG_AUTO || c->algorithm == DBGSOM_ALG_FILTERED_HINT) &&
                          c->hint_valid && c->hintM <= M;
        TRY(ensure_planes(c, s));
        TRY(c->filt_ws.reserve_zeroed(dbgsom_bmu_filtered_workspace_bytes(s.N, s.dp, M), c->stream));
        FilteredCall call;
        call.aux = &c->faux;
        call.X = s.Xb; call.x_dtype = s.bdtype; call.N = s.N; call.d = s.dp; call.ldx = s.dp;
        call.xx = s.xx.as<double>(); call.xplanes = s.planes.p; call.W = W; call.M = M; call.ww = c->ww.as<double>();
        call.prev_idx = hint ? c->idx[c->icur].as<int64_t>() : nullptr;
        call.order = hint ? c->acc_ws.as<int32_t>() : nullptr;
        call.seed_stride = c->seed_stride | DBGSOM_PRUNE | (c->prune_retry && !hint ? DBGSOM_PRUNE_RETRY : 0);
        call.sweep_planes = 1; call.round_f32 = round_f32; call.k = 2;
        call.idx = c->qidx.as<int64_t>(); call.dist = c->qdist.as<double>();
        call.ws = c->filt_ws.p; call.ws_bytes = c->filt_ws.cap; call.stream = c->stream;
        c->last_k2_filtered = true;
        return launch_bmu_filtered(call);
    }
    c->last_k2_filtered = false;
    return launch_bmu(s.Xb, s.bdtype, s.N, s.dp, s.dp, s.xx.as<double>(), W, M, c->ww.as<double>(), k, round_f32,
                      c->qidx.as<int64_t>(), c->qdist.as<double>(), c->stream);
}

}  // namespace

extern "C" {

// ------------------------------------------------------------------------------------------
// life cycle, options
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
    const char *e = getenv("DBGSOM_SWEEP_PLANES");  // diagnostic: fixes the digit planes of every context
    if (e) { const int v = atoi(e); if (v >= 0 && v <= 4) c->sweep_planes = v; }
    // (highest priority: the side streams of the filtered search -- a few long chains off the critical
    //  path -- must not starve the short dependent kernels of this one)
    int prio_low = 0, prio_high = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
    hipError_t err = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_high);
    if (err != hipSuccess) { (void)hipGetLastError(); err = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking); }
    if (err != hipSuccess) {
        set_error("hipStreamCreate failed: %s", hipGetErrorString(err));
        delete c;
        return DBGSOM_EHIP;
    }
    *out = c;
    return DBGSOM_OK;
}

// every DevBuf of a context, for dbgsom_ctx_destroy and the "device_bytes" option alike
#define CTX_DEVBUFS(c)                                                                                             \
    {&(c)->y, &(c)->hop, &(c)->hop_stage, &(c)->Wb[0], &(c)->Wb[1], &(c)->ww, &(c)->idx[0], &(c)->idx[1], &(c)->dist,  \
     &(c)->kw, &(c)->sums, &(c)->acc_ws, &(c)->sm_ws, &(c)->filt_ws, &(c)->scal, &(c)->qidx, &(c)->qdist, &(c)->red,  \
     &(c)->hist, &(c)->stage_dev, &(c)->part_order, &(c)->part_ws, &(c)->part_counts, &(c)->shiftb, &(c)->shard_send,    \
     &(c)->shard_gather}

int dbgsom_ctx_destroy(dbgsom_ctx *c) {
    if (!c) return DBGSOM_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    drop_rccl(c);
    c->faux.destroy();
    c->xs.release(); c->xq.release();
    DevBuf *bufs[] = CTX_DEVBUFS(c);
    for (DevBuf *b : bufs) b->release();
    c->tail.release(); c->counts.release();
    if (c->ev_created) for (auto &e : c->ev) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return DBGSOM_OK;
}

int dbgsom_ctx_set_option(dbgsom_ctx *c, const char *name, int64_t v) {
    CTX_CHECK(c);
    DBGSOM_REQUIRE(name, "null option name");
    if (!strcmp(name, "algorithm")) {
        DBGSOM_REQUIRE(v >= DBGSOM_ALG_AUTO && v <= DBGSOM_ALG_FILTERED_HINT, "algorithm must be a DBGSOM_ALG_* value");
        c->algorithm = (int)v;
    } else if (!strcmp(name, "sweep_planes")) {
        DBGSOM_REQUIRE(v >= 0 && v <= 4, "sweep_planes must be 0 .. 4 (4 = no sweep: triangle pruning)");
        c->sweep_planes = (int)v;
    } else if (!strcmp(name, "seed_stride")) {
        DBGSOM_REQUIRE(v >= 0 && v <= 64, "seed_stride outside [0, 64]");
        c->seed_stride = (int)v;
    } else if (!strcmp(name, "timing")) {
        c->timing = v != 0;
        c->ev_valid = false;
        c->faux.timer.enabled = c->timing != 0;
        c->faux.timer.valid = false;
    } else if (!strcmp(name, "graph")) {
        c->use_graph = v != 0;
    } else if (!strcmp(name, "defer")) {
        c->defer = v != 0;
    } else if (!strcmp(name, "shard_smooth")) {
        DBGSOM_REQUIRE(v >= 0 && v <= 2, "shard_smooth must be 0 (never), 1 (whenever the collective can) or 2 (large maps)");
        c->shard_smooth = (int)v;
    } else if (!strcmp(name, "refine")) {
        DBGSOM_REQUIRE(v >= 0 && v <= 2, "refine must be 0 (off), 1 (on) or 2 (by measurement)");
        c->refine = (int)v;
        for (auto &r : c->rf) r = dbgsom_ctx::RefineTimes();
    } else if (!strcmp(name, "filter_min_query_rows")) {
        DBGSOM_REQUIRE(v >= 0, "filter_min_query_rows must be >= 0");
        c->filter_min_query_rows = v;
    } else if (!strcmp(name, "max_mean_candidates")) {
        DBGSOM_REQUIRE(v >= 1, "max_mean_candidates must be >= 1");
        c->max_mean_candidates = v;
    } else {
        set_error("dbgsom_ctx_set_option: unknown option '%s'", name);
        return DBGSOM_EINVAL;
    }
    return DBGSOM_OK;
}

int dbgsom_ctx_get_option(dbgsom_ctx *c, const char *name, int64_t *v) {
    CTX_CHECK(c);
    DBGSOM_REQUIRE(name && v, "null pointer");
    if (!strcmp(name, "algorithm")) *v = c->algorithm;
    else if (!strcmp(name, "sweep_planes")) *v = c->sweep_planes;
    else if (!strcmp(name, "seed_stride")) *v = c->seed_stride;
    else if (!strcmp(name, "timing")) *v = c->timing;
    else if (!strcmp(name, "graph")) *v = c->use_graph;
    else if (!strcmp(name, "refine")) *v = c->refine;
    else if (!strcmp(name, "refined")) *v = c->last_refined ? 1 : 0;
    else if (!strcmp(name, "defer")) *v = c->defer;
    else if (!strcmp(name, "shard_smooth")) *v = c->shard_smooth;
    else if (!strcmp(name, "shard_epochs")) *v = c->shard_epochs;
    else if (!strcmp(name, "defer_epochs")) *v = c->defer_epochs;
    else if (!strcmp(name, "guarded_calls")) *v = c->guarded_calls;
    else if (!strcmp(name, "collective_rank")) *v = c->coll_rank;
    else if (!strcmp(name, "collective_ranks")) *v = c->coll_nranks;
    else if (!strcmp(name, "k2_filtered")) *v = c->last_k2_filtered ? 1 : 0;
    else if (!strcmp(name, "filter_min_query_rows")) *v = c->filter_min_query_rows;
    else if (!strcmp(name, "max_mean_candidates")) *v = c->max_mean_candidates;
    else if (!strcmp(name, "n_samples")) *v = c->xs.dtype < 0 ? 0 : c->xs.N;
    else if (!strcmp(name, "features")) *v = c->xs.dtype < 0 ? 0 : c->xs.d;
    else if (!strcmp(name, "padded_features")) *v = c->xs.dtype < 0 ? 0 : c->xs.dp;
    else if (!strcmp(name, "storage")) *v = c->xs.dtype;
    else if (!strcmp(name, "prototypes")) *v = c->M;
    else if (!strcmp(name, "planes_cached")) *v = c->xs.planes_ready ? 1 : 0;
    else if (!strcmp(name, "planes_used")) *v = c->planes_used;
    else if (!strcmp(name, "planes_next")) *v = planes_for_call(c);
    else if (!strcmp(name, "hint_valid")) *v = c->hint_valid ? 1 : 0;
    else if (!strcmp(name, "filter_backoff")) *v = c->filter_backoff;
    else if (!strcmp(name, "plane_hold")) *v = c->plane_hold;
    else if (!strcmp(name, "seed_mode")) *v = c->seed_mode;
    else if (!strcmp(name, "prune_retry")) *v = c->prune_retry ? 1 : 0;
    else if (!strcmp(name, "w_upload_calls")) *v = c->w_up_calls;
    else if (!strcmp(name, "w_upload_bytes")) *v = c->w_up_bytes;
    else if (!strcmp(name, "w_download_calls")) *v = c->w_down_calls;
    else if (!strcmp(name, "w_download_bytes")) *v = c->w_down_bytes;
    else if (!strcmp(name, "w_row_writes")) *v = c->w_row_writes;
    else if (!strcmp(name, "w_row_reads")) *v = c->w_row_reads;
    else if (!strcmp(name, "device_bytes")) {
        size_t tot = c->xs.own.cap + c->xs.x32.cap + c->xs.xx.cap + c->xs.planes.cap + c->xq.own.cap + c->xq.x32.cap +
                     c->xq.xx.cap + c->xq.planes.cap;
        DevBuf *bufs[] = CTX_DEVBUFS(c);
        for (DevBuf *b : bufs) tot += b->cap;
        *v = (int64_t)tot;
    } else {
        set_error("dbgsom_ctx_get_option: unknown option '%s'", name);
        return DBGSOM_EINVAL;
    }
    return DBGSOM_OK;
}

int dbgsom_ctx_stream(dbgsom_ctx *c, void **stream) {
    CTX_CHECK(c);
    DBGSOM_REQUIRE(stream, "null pointer");
    *stream = (void *)c->stream;
    return DBGSOM_OK;
}

// ------------------------------------------------------------------------------------------
// residency
// ------------------------------------------------------------------------------------------
static void reset_training_state(dbgsom_ctx *c) {
    c->hint_valid = c->last_idx_valid = c->part_valid = false;
    c->has_labels = false;
    c->filter_backoff = c->filter_fail = 0;
    c->planes_next = 1;
    c->planeM = -1;
    c->plane_hold = 0;
    c->seed_mode = 0;
    c->probe_next = c->last_probed = false;
    c->prune_retry = false;
    c->dist_bound_valid = false;
    c->last_filtered = false;
    c->last_mean = NAN;
    c->sumsM = 0;
    // the resident prototypes were laid out for the old samples' padded row length: gone with them
    c->M = c->otherM = 0;
    c->rf_M = -1;
    for (auto &r : c->rf) r = dbgsom_ctx::RefineTimes();
}

int dbgsom_ctx_load(dbgsom_ctx *c, const void *X_host, int x_dtype, int64_t N, int64_t d, int storage) {
    CTX_CHECK(c);
    reset_training_state(c);
    c->xs.dtype = -1;
    const int rc = place_host_samples(c, c->xs, X_host, x_dtype, N, d, storage);
    if (rc != DBGSOM_OK) { (void)hipStreamSynchronize(c->stream); c->xs.dtype = -1; return rc; }
    return sync(c);
}

int dbgsom_ctx_load_device(dbgsom_ctx *c, const void *X_dev, int x_dtype, int64_t N, int64_t d, int64_t ldx) {
    CTX_CHECK(c);
    DBGSOM_REQUIRE(valid_dtype(x_dtype), "x_dtype must be DBGSOM_F32/F64/BF16");
    DBGSOM_REQUIRE(X_dev && N >= 1 && d >= 1 && ldx >= d && N < 0x7fffffff, "bad samples");
    reset_training_state(c);
    Samples &s = c->xs;
    const size_t es = dtype_size(x_dtype);
    const int64_t dp = pad16(d);
    s.dtype = -1;
    s.N = N; s.d = d; s.dp = dp;
    if (ldx == dp && d == dp && is_aligned(X_dev, 16)) {
        s.own.release();
        s.X = X_dev;  // borrowed
    } else {
        TRY(s.own.reserve((size_t)N * dp * es));
        if (d != dp) DBGSOM_HIP_CHECK(hipMemsetAsync(s.own.p, 0, (size_t)N * dp * es, c->stream));
        DBGSOM_HIP_CHECK(hipMemcpy2DAsync(s.own.p, (size_t)dp * es, X_dev, (size_t)ldx * es, (size_t)d * es, (size_t)N,
                                          hipMemcpyDeviceToDevice, c->stream));
        s.X = s.own.p;
    }
    s.dtype = x_dtype;
    const int rc = finish_samples(c, s, false);
    if (rc != DBGSOM_OK) { (void)hipStreamSynchronize(c->stream); s.dtype = -1; return rc; }
    return sync(c);
}

int dbgsom_ctx_read_samples(dbgsom_ctx *c, const int64_t *rows_host, int64_t n, double *out_host) {
    CTX_CHECK(c);
    TRY(loaded(c, __func__));
    DBGSOM_REQUIRE(n >= 0 && (n == 0 || (rows_host && out_host)), "bad arguments");
    if (n == 0) return DBGSOM_OK;
    Samples &s = c->xs;
    for (int64_t r = 0; r < n; ++r) DBGSOM_REQUIRE(rows_host[r] >= 0 && rows_host[r] < s.N, "row index out of range");
    TRY(c->stage_dev.reserve((size_t)n * 8 + (size_t)n * s.d * 8 + 256));
    int64_t *ids = c->stage_dev.as<int64_t>();
    double *rows = reinterpret_cast<double *>(c->stage_dev.as<char>() + align_up((size_t)n * 8));
    DBGSOM_HIP_CHECK(hipMemcpyAsync(ids, rows_host, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    if (s.dtype == DBGSOM_F32)
        hipLaunchKernelGGL(rows_to_f64_kernel<float>, dim3((unsigned)n), dim3(256), 0, c->stream, (const float *)s.X, s.dp, ids, n, s.d, rows);
    else if (s.dtype == DBGSOM_F64)
        hipLaunchKernelGGL(rows_to_f64_kernel<double>, dim3((unsigned)n), dim3(256), 0, c->stream, (const double *)s.X, s.dp, ids, n, s.d, rows);
    else
        hipLaunchKernelGGL(rows_to_f64_kernel<bf16_t>, dim3((unsigned)n), dim3(256), 0, c->stream, (const bf16_t *)s.X, s.dp, ids, n, s.d, rows);
    TRY(launch_status("rows_to_f64_kernel"));
    DBGSOM_HIP_CHECK(hipMemcpyAsync(out_host, rows, (size_t)n * s.d * 8, hipMemcpyDeviceToHost, c->stream));
    return sync(c);
}

int dbgsom_ctx_set_labels(dbgsom_ctx *c, const int32_t *y_host, int64_t N) {
    CTX_CHECK(c);
    TRY(loaded(c, __func__));
    if (!y_host) { c->has_labels = false; return DBGSOM_OK; }
    DBGSOM_REQUIRE(N == c->xs.N, "one label per resident sample");
    TRY(c->y.reserve((size_t)N * 4));
    DBGSOM_HIP_CHECK(hipMemcpyAsync(c->y.p, y_host, (size_t)N * 4, hipMemcpyHostToDevice, c->stream));
    c->has_labels = true;
    return sync(c);
}

int dbgsom_ctx_set_topology(dbgsom_ctx *c, const double *hop_host, int64_t M) {
    CTX_CHECK(c);
    DBGSOM_REQUIRE(hop_host && M >= 1 && M <= DBGSOM_MAX_PROTOTYPES, "bad topology");
    const int64_t n = M * M;
    TRY(c->hop_stage.reserve((size_t)n * 8));
    TRY(c->hop.reserve((size_t)n * 4));
    DBGSOM_HIP_CHECK(hipMemcpyAsync(c->hop_stage.p, hop_host, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(f64_to_f32_kernel, dim3(grid1d(n)), dim3(256), 0, c->stream, c->hop_stage.as<double>(),
                       c->hop.as<float>(), n);
    TRY(launch_status("f64_to_f32_kernel"));
    TRY(sync(c));
    c->topoM = M;
    return DBGSOM_OK;
}

int dbgsom_ctx_set_collectives(dbgsom_ctx *c, dbgsom_collective_fn fn, void *user, int rank, int nranks) {
    CTX_CHECK(c);
    DBGSOM_REQUIRE(!fn || (nranks >= 1 && rank >= 0 && rank < nranks), "rank outside [0, nranks)");
    drop_rccl(c);
    c->allreduce = nullptr;
    c->allreduce_user = nullptr;
    c->coll = fn;
    c->coll_user = user;
    c->coll_rank = fn ? rank : 0;
    c->coll_nranks = fn ? nranks : 1;
    return DBGSOM_OK;
}

int dbgsom_ctx_set_allreduce(dbgsom_ctx *c, dbgsom_allreduce_fn fn, void *user) {
    CTX_CHECK(c);
    drop_rccl(c);
    c->allreduce = fn;
    c->allreduce_user = user;
    c->coll = nullptr; c->coll_user = nullptr; c->coll_rank = 0; c->coll_nranks = 1;
    return DBGSOM_OK;
}

int dbgsom_rccl_unique_id(char *id128) {
    DBGSOM_REQUIRE(id128, "null pointer");
    TRY(rccl_load());
    NcclUniqueId id;
    const int rc = g_rccl.get_unique_id(&id);
    if (rc != 0) { set_error("ncclGetUniqueId failed (%d: %s)", rc, rccl_err(rc)); return DBGSOM_ECALLBACK; }
    memcpy(id128, id.internal, 128);
    return DBGSOM_OK;
}

int dbgsom_rccl_comm_init(const char *id128, int nranks, int rank, void **comm_out) {
    DBGSOM_REQUIRE(id128 && comm_out && nranks >= 1 && rank >= 0 && rank < nranks, "bad communicator arguments");
    TRY(rccl_load());
    NcclUniqueId id;
    memcpy(id.internal, id128, 128);
    void *comm = nullptr;
    const int rc = g_rccl_init(&comm, nranks, id, rank);
    if (rc != 0 || !comm) { set_error("ncclCommInitRank failed (%d: %s)", rc, rccl_err(rc)); return DBGSOM_ECALLBACK; }
    *comm_out = comm;
    return DBGSOM_OK;
}

int dbgsom_rccl_comm_destroy(void *comm) {
    if (!comm) return DBGSOM_OK;
    TRY(rccl_load());
    const int rc = g_rccl.comm_destroy(comm);
    if (rc != 0) { set_error("ncclCommDestroy failed (%d: %s)", rc, rccl_err(rc)); return DBGSOM_ECALLBACK; }
    return DBGSOM_OK;
}

int dbgsom_ctx_allreduce_host(dbgsom_ctx *c, double *vals_host, int64_t n) {
    CTX_CHECK(c);
    DBGSOM_REQUIRE(vals_host && n >= 1, "bad arguments");
    if (!c->rccl_comm && !c->allreduce && !c->coll) return DBGSOM_OK;
    TRY(c->red.reserve((size_t)n * 8));
    DBGSOM_HIP_CHECK(hipMemcpyAsync(c->red.p, vals_host, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    TRY(run_allreduce(c, c->red.as<double>(), n));
    DBGSOM_HIP_CHECK(hipMemcpyAsync(vals_host, c->red.p, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    return sync(c);
}

int dbgsom_ctx_set_rccl(dbgsom_ctx *c, void *nccl_comm) {
    CTX_CHECK(c);
    if (nccl_comm) TRY(rccl_load());
    drop_rccl(c);
    c->rccl_comm = nccl_comm;
    c->rccl_owned = false;
    c->coll_rank = 0; c->coll_nranks = 1;
    if (nccl_comm) {
        c->allreduce = nullptr; c->allreduce_user = nullptr;
        c->coll = nullptr; c->coll_user = nullptr;
        // rank and size of the communicator: what the sharded smoothing needs (absent symbols: it stays off)
        int n = 1, r = 0;
        if (g_rccl.comm_count && g_rccl.comm_user_rank && g_rccl.comm_count(nccl_comm, &n) == 0 &&
            g_rccl.comm_user_rank(nccl_comm, &r) == 0 && n >= 1 && r >= 0 && r < n) {
            c->coll_nranks = n;
            c->coll_rank = r;
        }
    }
    return DBGSOM_OK;
}

// ------------------------------------------------------------------------------------------
// prototypes
// ------------------------------------------------------------------------------------------
int dbgsom_ctx_set_weights(dbgsom_ctx *c, const double *W_host, int64_t M) {
    CTX_CHECK(c);
    TRY(loaded(c, __func__));
    DBGSOM_REQUIRE(W_host && M >= 1, "bad prototypes");
    TRY(c->Wb[c->cur].reserve((size_t)M * c->xs.dp * 8));
    TRY(upload_padded(c, c->Wb[c->cur].p, W_host, M, c->xs.d, c->xs.dp, 8));
    if (c->distW_buf == c->cur) c->dist_bound_valid = false;
    c->M = M;
    ++c->w_up_calls; c->w_up_bytes += M * c->xs.d * 8;
    return sync(c);
}

int dbgsom_ctx_get_weights(dbgsom_ctx *c, int which, double *W_host, int64_t M) {
    CTX_CHECK(c);
    TRY(loaded(c, __func__));
    DBGSOM_REQUIRE(W_host && (which == 0 || which == 1), "bad arguments");
    const int b = which ? c->cur ^ 1 : c->cur;
    const int64_t have = which ? c->otherM : c->M;
    if (have != M || M < 1) {
        set_error("dbgsom_ctx_get_weights: buffer %d holds %lld rows, %lld asked for", which, (long long)have, (long long)M);
        return DBGSOM_ESTATE;
    }
    TRY(download_unpadded(c, W_host, c->Wb[b].p, M, c->xs.d, c->xs.dp, 8));
    ++c->w_down_calls; c->w_down_bytes += M * c->xs.d * 8;
    return sync(c);
}

int dbgsom_ctx_read_weight_rows(dbgsom_ctx *c, int which, const int64_t *rows_host, int64_t n, double *out_host) {
    CTX_CHECK(c);
    TRY(loaded(c, __func__));
    DBGSOM_REQUIRE((which == 0 || which == 1) && n >= 0 && (n == 0 || (rows_host && out_host)), "bad arguments");
    const int b = which ? c->cur ^ 1 : c->cur;
    const int64_t have = which ? c->otherM : c->M;
    const int64_t d = c->xs.d, dp = c->xs.dp;
    c->w_row_reads += n;
    for (int64_t r = 0; r < n; ++r) {
        DBGSOM_REQUIRE(rows_host[r] >= 0 && rows_host[r] < have, "row index out of range");
        DBGSOM_HIP_CHECK(hipMemcpyAsync(out_host + r * d, c->Wb[b].as<double>() + rows_host[r] * dp, (size_t)d * 8,
                                        hipMemcpyDeviceToHost, c->stream));
    }
    return sync(c);
}

int dbgsom_ctx_write_weight_rows(dbgsom_ctx *c, int64_t row0, int64_t n, const double *rows_host) {
    CTX_CHECK(c);
    TRY(loaded(c, __func__));
    DBGSOM_REQUIRE(n >= 0 && row0 >= 0 && row0 <= c->M && (n == 0 || rows_host), "rows must lie in [0, M] (row0 == M appends)");
    if (n == 0) return DBGSOM_OK;
    const int64_t d = c->xs.d, dp = c->xs.dp;
    const int64_t newM = row0 + n > c->M ? row0 + n : c->M;
    TRY(c->Wb[c->cur].reserve_keep((size_t)newM * dp * 8, (size_t)c->M * dp * 8, c->stream));
    double *dst = c->Wb[c->cur].as<double>() + row0 * dp;
    TRY(upload_padded(c, dst, rows_host, n, d, dp, 8));
    if (c->distW_buf == c->cur) c->dist_bound_valid = false;
    c->M = newM;
    c->w_row_writes += n;
    return sync(c);
}

// ------------------------------------------------------------------------------------------
// BMU
// ------------------------------------------------------------------------------------------
int dbgsom_ctx_bmu(dbgsom_ctx *c, const double *W_host, int64_t M, int k, int round_f32, int64_t *idx_host,
                   double *dist_host) {
    CTX_CHECK(c);
    TRY(loaded(c, __func__));
    DBGSOM_REQUIRE(idx_host && dist_host && (k == 1 || k == 2), "bad arguments");
    DBGSOM_REQUIRE(M >= k, "need k <= M");
    TRY(resident_bmu(c, W_host, M, k, round_f32));
    DBGSOM_HIP_CHECK(hipMemcpyAsync(idx_host, c->qidx.p, (size_t)c->xs.N * k * 8, hipMemcpyDeviceToHost, c->stream));
    DBGSOM_HIP_CHECK(hipMemcpyAsync(dist_host, c->qdist.p, (size_t)c->xs.N * k * 8, hipMemcpyDeviceToHost, c->stream));
    return sync(c);
}

int dbgsom_ctx_bmu_query(dbgsom_ctx *c, const void *Xq_host, int x_dtype, int64_t Nq, int64_t d, const double *W_host,
                         int64_t M, int k, int round_f32, int64_t *idx_host, double *dist_host) {
    CTX_CHECK(c);
    DBGSOM_REQUIRE(valid_dtype(x_dtype), "x_dtype must be DBGSOM_F32/F64/BF16");
    DBGSOM_REQUIRE(Xq_host && W_host && idx_host && dist_host && Nq >= 0 && d >= 1 && M >= 1 && (k == 1 || k == 2) && M >= k,
                   "bad arguments");
    if (Nq == 0) return DBGSOM_OK;
    Samples &s = c->xq;
    DevBuf Wq, wwq, iq, dq, fws;  // query-sized scratch; independent of the training state
    int rc = DBGSOM_OK;
    const int64_t dp = pad16(d);
    do {
        if ((rc = place_host_samples(c, s, Xq_host, x_dtype, Nq, d, x_dtype))) break;
        if ((rc = Wq.reserve((size_t)M * dp * 8))) break;
        if ((rc = wwq.reserve((size_t)M * 8))) break;
        if ((rc = iq.reserve((size_t)Nq * k * 8))) break;
        if ((rc = dq.reserve((size_t)Nq * k * 8))) break;
        if ((rc = upload_padded(c, Wq.p, W_host, M, d, dp, 8))) break;
        if ((rc = launch_row_sqnorms(Wq.p, DBGSOM_F64, M, dp, dp, wwq.as<double>(), c->stream))) break;
        // large k = 1 queries go through the filter (the digit planes of a one-off X cost a pass over it)
        const bool filt = k == 1 && c->algorithm != DBGSOM_ALG_EXACT && Nq >= c->filter_min_query_rows &&
                          filter_shape_ok(s, M);
        if (filt) {
            if ((rc = ensure_planes(c, s))) break;
            if ((rc = fws.reserve_zeroed(dbgsom_bmu_filtered_workspace_bytes(Nq, dp, M), c->stream))) break;
            int stride = c->seed_stride, planes = 1;
            filter_call_args(planes_for_call(c), false, c->prune_retry, M, &stride, &planes);
            rc = dbgsom_bmu_filtered(s.Xb, s.bdtype, Nq, dp, dp, s.xx.as<double>(), s.planes.p, Wq.as<double>(), M,
                                     wwq.as<double>(), nullptr, nullptr, stride, planes, round_f32,
                                     iq.as<int64_t>(), dq.as<double>(), fws.p, fws.cap, c->stream);
        } else {
            rc = launch_bmu(s.Xb, s.bdtype, Nq, dp, dp, s.xx.as<double>(), Wq.as<double>(), M, wwq.as<double>(), k,
                            round_f32, iq.as<int64_t>(), dq.as<double>(), c->stream);
        }
        if (rc) break;
        hipError_t e = hipMemcpyAsync(idx_host, iq.p, (size_t)Nq * k * 8, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(dist_host, dq.p, (size_t)Nq * k * 8, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { set_error("D2H copy failed: %s", hipGetErrorString(e)); rc = DBGSOM_EHIP; }
    } while (0);
    if (rc != DBGSOM_OK) (void)hipStreamSynchronize(c->stream);
    Wq.release(); wwq.release(); iq.release(); dq.release(); fws.release();
    if (Nq * dp * (int64_t)dtype_size(x_dtype) > ((int64_t)256 << 20)) s.release();  // do not sit on a large one-off batch
    return rc;
}

int dbgsom_ctx_exp_similarity(dbgsom_ctx *c, const double *dist_host, int64_t n, double gamma, double *kw_host) {
    CTX_CHECK(c);
    DBGSOM_REQUIRE(n >= 0 && (n == 0 || (dist_host && kw_host)), "bad arguments");
    if (n == 0) return DBGSOM_OK;
    TRY(c->stage_dev.reserve((size_t)2 * n * 8));
    double *din = c->stage_dev.as<double>(), *dout = din + n;
    DBGSOM_HIP_CHECK(hipMemcpyAsync(din, dist_host, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    TRY(launch_exp_similarity(din, n, gamma, dout, c->stream));
    DBGSOM_HIP_CHECK(hipMemcpyAsync(kw_host, dout, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    return sync(c);
}

// ------------------------------------------------------------------------------------------
// the epoch
// ------------------------------------------------------------------------------------------
int dbgsom_ctx_epoch(dbgsom_ctx *c, const double *W_host, int64_t M, int round_f32, double gamma, double sigma,
                     int layout, int flags, double *W_new_host, double *change_total_host, double *errors_host,
                     double *activations_host, int64_t *idx_host, double *dist_host) {
    CTX_CHECK(c);
    TRY(loaded(c, __func__));
    DBGSOM_REQUIRE(change_total_host && errors_host && activations_host, "null output");
    DBGSOM_REQUIRE(layout == DBGSOM_CENTRES_COMPACT || layout == DBGSOM_CENTRES_ALIGNED, "bad layout");
    if (c->topoM != M) {
        set_error("dbgsom_ctx_epoch: topology holds %lld neurons, weights %lld (call "
                  "dbgsom_ctx_set_topology after growth)", (long long)c->topoM, (long long)M);
        return DBGSOM_ESTATE;
    }
    Samples &s = c->xs;
    int rc = DBGSOM_OK;
    do {
        if ((rc = stage_weights(c, W_host, M, s.d, s.dp))) break;
        mark(c, 0);
        const auto t_begin = std::chrono::steady_clock::now();
        if ((rc = epoch_bmu(c, M, round_f32))) break;
        mark(c, 1);
        const int measuring = c->rf_measuring;   // (the refinement's policy: this epoch times one of the two forms)
        c->rf_measuring = -1;
        const int64_t *idx = c->idx[c->icur].as<int64_t>();
        if ((rc = accumulate_and_reduce(c, idx, nullptr, gamma, c->dist.as<double>(), M))) break;
        // the next epoch's filter visits the samples bucketed by this epoch's winners: the stable
        // counting sort the accumulate step just did (first N int32 of its workspace)
        c->hint_valid = true;
        c->hintM = M;
        mark(c, 2);
        rc = smooth_and_fetch(c, M, sigma, layout, flags, W_new_host, change_total_host, errors_host, activations_host,
                              idx, idx_host, dist_host);
        if (rc != DBGSOM_OK && rc != DBGSOM_ERANGE) break;
        if (measuring >= 0) {   // wall clock of the blocking call behind the upload of W: BMU + sums + smoothing
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
            dbgsom_ctx::RefineTimes &r = c->rf[c->rf_arm];
            r.ms[measuring] = r.n[measuring] == 0 ? ms : fmin(ms, r.ms[measuring]);
            ++r.n[measuring];
        }
        c->last_frozen = (flags & DBGSOM_EPOCH_FROZEN) != 0;
        c->last_epoch_ms = (c->last_filtered && !c->last_probed && !c->last_guarded && measuring < 0 && !W_new_host && !idx_host && !dist_host)
                               ? std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count()
                               : NAN;
        update_policy(c, c->tail.as<double>()[2 * M + 2], c->tail.as<double>()[2 * M + 3], c->tail.as<double>()[2 * M + 4],
                      (s.N + 127) / 128, M);
    } while (0);
    if (rc != DBGSOM_OK && rc != DBGSOM_ERANGE) { (void)hipStreamSynchronize(c->stream); c->hint_valid = false; }
    return rc;
}

int dbgsom_ctx_update(dbgsom_ctx *c, const double *W_host, int64_t M, const int64_t *idx_host, const double *kw_host,
                      const double *dist_host, double sigma, int layout, double *W_new_host,
                      double *change_total_host, double *errors_host, double *activations_host) {
    CTX_CHECK(c);
    TRY(loaded(c, __func__));
    DBGSOM_REQUIRE(idx_host && kw_host && dist_host && change_total_host && errors_host && activations_host, "null pointer");
    DBGSOM_REQUIRE(layout == DBGSOM_CENTRES_COMPACT || layout == DBGSOM_CENTRES_ALIGNED, "bad layout");
    Samples &s = c->xs;
    int rc = DBGSOM_OK;
    do {
        if ((rc = stage_weights(c, W_host, M, s.d, s.dp))) break;
        if ((rc = c->idx[0].reserve((size_t)s.N * 8))) break;
        if ((rc = c->idx[1].reserve((size_t)s.N * 8))) break;
        if ((rc = c->dist.reserve((size_t)s.N * 8))) break;
        if ((rc = c->kw.reserve((size_t)s.N * 8))) break;
        c->hint_valid = false;
        c->dist_bound_valid = false;  // (the caller's distances: nothing a bound may rest on)
        c->icur ^= 1;
        int64_t *idx = c->idx[c->icur].as<int64_t>();
        hipError_t e = hipMemcpyAsync(idx, idx_host, (size_t)s.N * 8, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(c->kw.p, kw_host, (size_t)s.N * 8, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(c->dist.p, dist_host, (size_t)s.N * 8, hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) { set_error("H2D copy failed: %s", hipGetErrorString(e)); rc = DBGSOM_EHIP; break; }
        c->last_filtered = false;
        c->last_idx_valid = true;
        if ((rc = accumulate_and_reduce(c, idx, c->kw.as<double>(), 0.0, c->dist.as<double>(), M))) break;
        rc = smooth_and_fetch(c, M, sigma, layout, 0, W_new_host, change_total_host, errors_host, activations_host, idx,
                              nullptr, nullptr);
    } while (0);
    if (rc != DBGSOM_OK && rc != DBGSOM_ERANGE) (void)hipStreamSynchronize(c->stream);
    return rc;
}

int dbgsom_ctx_set_hint(dbgsom_ctx *c, const int64_t *idx_host, int64_t M) {
    CTX_CHECK(c);
    TRY(loaded(c, __func__));
    DBGSOM_REQUIRE(idx_host && M >= 1 && M <= DBGSOM_MAX_PROTOTYPES, "bad arguments");
    Samples &s = c->xs;
    for (int64_t i = 0; i < s.N; ++i) DBGSOM_REQUIRE(idx_host[i] >= 0 && idx_host[i] < M, "seed index out of range");
    TRY(c->idx[0].reserve((size_t)s.N * 8));
    TRY(c->idx[1].reserve((size_t)s.N * 8));
    // the bucket order lives where the accumulate step leaves it: the first N int32 of its workspace
    TRY(c->acc_ws.reserve(accumulate_workspace_bytes(s.N, s.dp, M)));
    TRY(c->part_ws.reserve(bucket_sort_workspace_bytes(s.N, M)));
    int64_t *idx = c->idx[c->icur].as<int64_t>();
    DBGSOM_HIP_CHECK(hipMemcpyAsync(idx, idx_host, (size_t)s.N * 8, hipMemcpyHostToDevice, c->stream));
    TRY(launch_bucket_sort(idx, s.N, M, c->acc_ws.as<int32_t>(), c->part_ws.p, c->stream));
    TRY(sync(c));
    c->hint_valid = true;
    c->dist_bound_valid = false;
    c->hintM = M;
    c->part_valid = false;
    return DBGSOM_OK;
}

int dbgsom_ctx_read_sums(dbgsom_ctx *c, double *sums_host, int64_t M) {
    CTX_CHECK(c);
    TRY(loaded(c, __func__));
    DBGSOM_REQUIRE(sums_host, "null pointer");
    if (c->sumsM != M || M < 1) { set_error("dbgsom_ctx_read_sums: no sums of %lld neurons", (long long)M); return DBGSOM_ESTATE; }
    const int64_t d = c->xs.d, dp = c->xs.dp;
    TRY(download_unpadded(c, sums_host, c->sums.p, M, d, dp, 8));
    DBGSOM_HIP_CHECK(hipMemcpyAsync(sums_host + M * d, c->sums.as<double>() + M * dp, (size_t)3 * M * 8,
                                    hipMemcpyDeviceToHost, c->stream));
    return sync(c);
}

// ------------------------------------------------------------------------------------------
// reductions around the path
// ------------------------------------------------------------------------------------------
int dbgsom_ctx_column_sums(dbgsom_ctx *c, const void *mean_host, void *out_host) {
    CTX_CHECK(c);
    TRY(loaded(c, __func__));
    DBGSOM_REQUIRE(out_host, "null pointer");
    Samples &s = c->xs;
    DBGSOM_REQUIRE(s.dtype == DBGSOM_F32 || s.dtype == DBGSOM_F64, "float32 / float64 resident samples only");
    const size_t es = dtype_size(s.dtype);
    TRY(c->stage_dev.reserve((size_t)2 * s.dp * es + 512));
    char *mean_dev = c->stage_dev.as<char>();
    char *out_dev = mean_dev + align_up((size_t)s.dp * es);
    if (mean_host) {
        DBGSOM_HIP_CHECK(hipMemsetAsync(mean_dev, 0, (size_t)s.dp * es, c->stream));
        DBGSOM_HIP_CHECK(hipMemcpyAsync(mean_dev, mean_host, (size_t)s.d * es, hipMemcpyHostToDevice, c->stream));
    }
    TRY(dbgsom_column_sums(s.X, s.dtype, s.N, s.dp, s.dp, mean_host ? mean_dev : nullptr, out_dev, c->stream));
    DBGSOM_HIP_CHECK(hipMemcpyAsync(out_host, out_dev, (size_t)s.d * es, hipMemcpyDeviceToHost, c->stream));
    return sync(c);
}

static int reduce_small(dbgsom_ctx *c, double *buf_dev, int64_t n, double *out_host) {
    TRY(run_allreduce(c, buf_dev, n));
    DBGSOM_HIP_CHECK(hipMemcpyAsync(out_host, buf_dev, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    return sync(c);
}

int dbgsom_ctx_quantization_error(dbgsom_ctx *c, const double *W_host, int64_t M, int round_f32, double *out2) {
    CTX_CHECK(c);
    TRY(loaded(c, __func__));
    DBGSOM_REQUIRE(out2, "null pointer");
    TRY(resident_bmu(c, W_host, M, 1, round_f32));
    TRY(c->red.reserve(256 + dbgsom_sum_workspace_bytes()));
    double *r = c->red.as<double>();
    TRY(dbgsom_sum_f64(c->qdist.as<double>(), c->xs.N, r, c->red.as<char>() + 256, c->red.cap - 256, c->stream));
    const double n = (double)c->xs.N;
    DBGSOM_HIP_CHECK(hipMemcpyAsync(r + 1, &n, 8, hipMemcpyHostToDevice, c->stream));
    return reduce_small(c, r, 2, out2);
}

int dbgsom_ctx_topographic_count(dbgsom_ctx *c, const double *W_host, int64_t M, int round_f32, const int32_t *xy_host,
                                 double *count_host) {
    CTX_CHECK(c);
    TRY(loaded(c, __func__));
    DBGSOM_REQUIRE(xy_host && count_host && M >= 2, "bad arguments");
    TRY(resident_bmu(c, W_host, M, 2, round_f32));
    TRY(c->red.reserve(256 + (size_t)M * 8));
    int32_t *xy = reinterpret_cast<int32_t *>(c->red.as<char>() + 256);
    DBGSOM_HIP_CHECK(hipMemcpyAsync(xy, xy_host, (size_t)M * 8, hipMemcpyHostToDevice, c->stream));
    uint64_t *cnt = reinterpret_cast<uint64_t *>(c->red.as<char>() + 64);
    TRY(dbgsom_topographic_count(c->qidx.as<int64_t>(), c->xs.N, xy, M, cnt, c->stream));
    double *r = c->red.as<double>();
    hipLaunchKernelGGL(u64_to_f64_kernel, dim3(1), dim3(64), 0, c->stream, (const unsigned long long *)cnt, r, (int64_t)1);
    TRY(launch_status("u64_to_f64_kernel"));
    return reduce_small(c, r, 1, count_host);
}

int dbgsom_ctx_node_statistics(dbgsom_ctx *c, const double *W_host, int64_t M, int round_f32, double sigma,
                               double *hits_host, double *density_host) {
    CTX_CHECK(c);
    TRY(loaded(c, __func__));
    DBGSOM_REQUIRE(hits_host && density_host && sigma > 0.0, "bad arguments");
    Samples &s = c->xs;
    TRY(resident_bmu(c, W_host, M, 1, round_f32));
    TRY(c->kw.reserve((size_t)s.N * 8));
    TRY(dbgsom_density_terms(c->qdist.as<double>(), s.N, sigma, c->kw.as<double>(), c->stream));
    // K = density sums, a = hit counts of the fused buffer; the bucket order of the training hint is rewritten
    c->hint_valid = false;
    DBGSOM_REQUIRE(M <= DBGSOM_MAX_PROTOTYPES, "M exceeds DBGSOM_MAX_PROTOTYPES");
    const int64_t count = M * (s.dp + 3);
    TRY(c->sums.reserve((size_t)(count + 1) * 8));
    TRY(c->acc_ws.reserve(accumulate_workspace_bytes(s.N, s.dp, M)));
    c->part_valid = false;
    c->sumsM = 0;
    TRY(launch_accumulate(s.X, s.dtype, s.N, s.dp, s.dp, c->qidx.as<int64_t>(), c->kw.as<double>(), c->qdist.as<double>(), M,
                          c->sums.as<double>(), nullptr, c->acc_ws.p, c->acc_ws.cap, c->stream));
    double *tail = c->sums.as<double>() + M * s.dp;  // [K | a]
    TRY(run_allreduce(c, tail, 2 * M));
    DBGSOM_HIP_CHECK(hipMemcpyAsync(density_host, tail, (size_t)M * 8, hipMemcpyDeviceToHost, c->stream));
    DBGSOM_HIP_CHECK(hipMemcpyAsync(hits_host, tail + M, (size_t)M * 8, hipMemcpyDeviceToHost, c->stream));
    return sync(c);
}

int dbgsom_ctx_class_histogram(dbgsom_ctx *c, const int64_t *idx_host, int64_t n_classes, int64_t M, int64_t *hist_host) {
    CTX_CHECK(c);
    TRY(loaded(c, __func__));
    DBGSOM_REQUIRE(hist_host && n_classes >= 1 && M >= 1, "bad arguments");
    if (!c->has_labels) { set_error("class histogram requested but no labels attached (dbgsom_ctx_set_labels)"); return DBGSOM_ESTATE; }
    Samples &s = c->xs;
    const int64_t *idx = nullptr;
    if (idx_host) {
        TRY(c->qidx.reserve((size_t)s.N * 8));
        DBGSOM_HIP_CHECK(hipMemcpyAsync(c->qidx.p, idx_host, (size_t)s.N * 8, hipMemcpyHostToDevice, c->stream));
        idx = c->qidx.as<int64_t>();
    } else {
        if (!c->last_idx_valid) { set_error("no winners of a previous epoch in HBM"); return DBGSOM_ESTATE; }
        idx = c->idx[c->icur].as<int64_t>();
    }
    const int64_t n = M * n_classes;
    TRY(c->hist.reserve((size_t)2 * n * 8));
    uint64_t *h = c->hist.as<uint64_t>();
    double *hd = c->hist.as<double>() + n;
    TRY(dbgsom_class_histogram(idx, c->y.as<int32_t>(), s.N, M, n_classes, h, c->stream));
    hipLaunchKernelGGL(u64_to_f64_kernel, dim3(grid1d(n)), dim3(256), 0, c->stream, (const unsigned long long *)h, hd, n);
    TRY(launch_status("u64_to_f64_kernel"));
    TRY(run_allreduce(c, hd, n));
    std::vector<double> tmp;
    try { tmp.resize((size_t)n); } catch (...) { set_error("out of host memory"); return DBGSOM_ENOMEM; }
    DBGSOM_HIP_CHECK(hipMemcpyAsync(tmp.data(), hd, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    TRY(sync(c));
    for (int64_t e = 0; e < n; ++e) hist_host[e] = (int64_t)llround(tmp[(size_t)e]);
    return DBGSOM_OK;
}

// ------------------------------------------------------------------------------------------
// vertical growth
// ------------------------------------------------------------------------------------------
int dbgsom_ctx_partition(dbgsom_ctx *c, const double *W_host, int64_t M, int round_f32, int64_t *counts_host,
                         int64_t *idx_host) {
    CTX_CHECK(c);
    TRY(loaded(c, __func__));
    DBGSOM_REQUIRE(counts_host && M >= 1 && M <= DBGSOM_MAX_PROTOTYPES, "bad arguments");
    Samples &s = c->xs;
    TRY(resident_bmu(c, W_host, M, 1, round_f32));
    TRY(c->part_order.reserve((size_t)s.N * 4));
    TRY(c->part_ws.reserve(bucket_sort_workspace_bytes(s.N, M)));
    TRY(c->part_counts.reserve((size_t)(M + 1) * 8));
    TRY(launch_bucket_sort(c->qidx.as<int64_t>(), s.N, M, c->part_order.as<int32_t>(), c->part_ws.p, c->stream));
    const uint32_t *seg_start = bucket_sort_seg_start(c->part_ws.p, s.N, M);
    hipLaunchKernelGGL(seg_counts_kernel, dim3(grid1d(M)), dim3(256), 0, c->stream, seg_start, M, s.N, c->part_counts.as<int64_t>());
    TRY(launch_status("seg_counts_kernel"));
    DBGSOM_HIP_CHECK(hipMemcpyAsync(counts_host, c->part_counts.p, (size_t)M * 8, hipMemcpyDeviceToHost, c->stream));
    if (idx_host) DBGSOM_HIP_CHECK(hipMemcpyAsync(idx_host, c->qidx.p, (size_t)s.N * 8, hipMemcpyDeviceToHost, c->stream));
    TRY(sync(c));
    c->partM = M;
    c->part_valid = true;
    return DBGSOM_OK;
}

int dbgsom_ctx_subset_create(dbgsom_ctx *c, int64_t neuron, dbgsom_ctx **child_out) {
    CTX_CHECK(c);
    TRY(loaded(c, __func__));
    DBGSOM_REQUIRE(child_out, "null pointer");
    *child_out = nullptr;
    if (!c->part_valid) { set_error("dbgsom_ctx_subset_create: call dbgsom_ctx_partition first"); return DBGSOM_ESTATE; }
    DBGSOM_REQUIRE(neuron >= 0 && neuron < c->partM, "neuron out of range");
    Samples &s = c->xs;
    const uint32_t *seg_start = bucket_sort_seg_start(c->part_ws.p, s.N, c->partM);
    uint32_t seg[2] = {0, 0};
    DBGSOM_HIP_CHECK(hipMemcpyAsync(&seg[0], seg_start + neuron, 4, hipMemcpyDeviceToHost, c->stream));
    if (neuron + 1 < c->partM)
        DBGSOM_HIP_CHECK(hipMemcpyAsync(&seg[1], seg_start + neuron + 1, 4, hipMemcpyDeviceToHost, c->stream));
    TRY(sync(c));
    if (neuron + 1 >= c->partM) seg[1] = (uint32_t)s.N;
    const int64_t first = seg[0], n = (int64_t)seg[1] - (int64_t)seg[0];
    if (n < 1) { set_error("dbgsom_ctx_subset_create: neuron %lld has no samples", (long long)neuron); return DBGSOM_EINVAL; }
    dbgsom_ctx *k = nullptr;
    TRY(dbgsom_ctx_create(c->device, &k));
    k->algorithm = c->algorithm; k->sweep_planes = c->sweep_planes; k->seed_stride = c->seed_stride;
    k->filter_min_query_rows = c->filter_min_query_rows; k->max_mean_candidates = c->max_mean_candidates;
    k->allreduce = nullptr;  // a child map is fitted on this rank's rows alone
    Samples &t = k->xs;
    int rc = DBGSOM_OK;
    do {
        const size_t es = dtype_size(s.dtype);
        if ((rc = t.own.reserve((size_t)n * s.dp * es))) break;
        t.N = n; t.d = s.d; t.dp = s.dp; t.dtype = s.dtype; t.X = t.own.p;
        const int32_t *order = c->part_order.as<int32_t>();
        // the gather runs on the parent's stream (it reads the parent's buffers), then both are idle
        if (s.dtype == DBGSOM_F32)
            hipLaunchKernelGGL(gather_rows_kernel<float>, dim3((unsigned)n), dim3(256), 0, c->stream, (const float *)s.X, s.dp, order, first, n, s.dp, (float *)t.own.p);
        else if (s.dtype == DBGSOM_F64)
            hipLaunchKernelGGL(gather_rows_kernel<double>, dim3((unsigned)n), dim3(256), 0, c->stream, (const double *)s.X, s.dp, order, first, n, s.dp, (double *)t.own.p);
        else
            hipLaunchKernelGGL(gather_rows_kernel<uint16_t>, dim3((unsigned)n), dim3(256), 0, c->stream, (const uint16_t *)s.X, s.dp, order, first, n, s.dp, (uint16_t *)t.own.p);
        if ((rc = launch_status("gather_rows_kernel"))) break;
        if (c->has_labels) {
            if ((rc = k->y.reserve((size_t)n * 4))) break;
            hipLaunchKernelGGL(gather_i32_kernel, dim3(grid1d(n)), dim3(256), 0, c->stream, c->y.as<int32_t>(), order, first, n, k->y.as<int32_t>());
            if ((rc = launch_status("gather_i32_kernel"))) break;
            k->has_labels = true;
        }
        if ((rc = sync(c))) break;
        if ((rc = finish_samples(k, t, false))) break;
        rc = sync(k);
    } while (0);
    if (rc != DBGSOM_OK) { (void)hipStreamSynchronize(c->stream); dbgsom_ctx_destroy(k); return rc; }
    *child_out = k;
    return DBGSOM_OK;
}

// ------------------------------------------------------------------------------------------
// diagnostics
// ------------------------------------------------------------------------------------------
int dbgsom_ctx_epoch_info(dbgsom_ctx *c, double *info8) {
    CTX_CHECK(c);
    DBGSOM_REQUIRE(info8, "null pointer");
    info8[0] = c->last_filtered ? 1.0 : 0.0;
    info8[1] = c->last_mean;
    info8[2] = c->last_filtered ? (double)c->planes_used : 0.0;
    info8[3] = c->last_hinted ? 1.0 : 0.0;
    info8[4] = (double)c->filter_backoff;
    info8[5] = (double)c->plane_hold;
    info8[6] = c->last_filtered && c->last_probed ? c->last_probe_mean : NAN;
    info8[7] = c->last_filtered && c->last_seed_full ? 1.0 : 0.0;
    return DBGSOM_OK;
}

int dbgsom_ctx_arm_ms(dbgsom_ctx *c, double *ms12) {
    CTX_CHECK(c);
    DBGSOM_REQUIRE(ms12, "null pointer");
    for (int s_ = 0; s_ < 3; ++s_)
        for (int q = 0; q < 4; ++q) ms12[4 * s_ + q] = c->arm_ms[s_][q];
    return DBGSOM_OK;
}

int dbgsom_ctx_filter_counts(dbgsom_ctx *c, uint32_t *counts_host, int64_t n) {
    CTX_CHECK(c);
    DBGSOM_REQUIRE(counts_host, "null pointer");
    if (!c->last_filter_ws) { set_error("dbgsom_ctx_filter_counts: no filtered search has run"); return DBGSOM_ESTATE; }
    return dbgsom_bmu_filtered_counts(c->last_filter_ws, c->last_filter_N, c->last_filter_d, c->last_filter_M, counts_host, n,
                                      c->stream);
}

int dbgsom_ctx_refine_counts(dbgsom_ctx *c, uint64_t *out4) {
    CTX_CHECK(c);
    DBGSOM_REQUIRE(out4, "null pointer");
    if (!c->last_filter_ws) { set_error("dbgsom_ctx_refine_counts: no filtered search has run"); return DBGSOM_ESTATE; }
    return dbgsom_bmu_filtered_refine_counts(c->last_filter_ws, c->last_filter_N, c->last_filter_d, c->last_filter_M, out4,
                                             c->stream);
}

int dbgsom_ctx_phase_ms(dbgsom_ctx *c, double *ms8) {
    CTX_CHECK(c);
    DBGSOM_REQUIRE(ms8, "null pointer");
    if (!c->timing || !c->ev_valid) { set_error("dbgsom_ctx_phase_ms: no timed epoch (set option \"timing\")"); return DBGSOM_ESTATE; }
    for (int k = 0; k < 3; ++k) {
        float ms = 0.f;
        DBGSOM_HIP_CHECK(hipEventElapsedTime(&ms, c->ev[k], c->ev[k + 1]));
        ms8[k] = ms;
    }
    for (int k = 0; k < 5; ++k) ms8[3 + k] = c->filter_ms_valid ? c->filter_ms[k] : 0.0;
    return DBGSOM_OK;
}

}  // extern "C"
