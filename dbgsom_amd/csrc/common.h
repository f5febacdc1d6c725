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

// "The last workgroup to finish does the follow-up step": folds a tiny dependent kernel (a scan
// over M entries, a final reduction) into the launch that produces its input -- a 4-5 us launch
// less per use.  Every workgroup calls this after its last global store; it returns true (in all
// threads) in exactly one workgroup, the one whose ticket is the last, and by then every other
// workgroup's stores are visible to it: plain stores -> every wave's vmcnt(0) -> barrier ->
// agent-scope release -> ticket (relaxed agent atomic); the last arriver: agent-scope acquire
// (invalidates this CU's L1) -> vmcnt(0) -> barrier -> plain loads (MI355X_MICROARCH.md,
// "Workgroup dispatch, XCD placement & inter-workgroup visibility").  The ticket counter must be 0
// before the launch; the last workgroup leaves it at 0 again.
__device__ __forceinline__ bool last_workgroup_done(uint32_t *ticket, uint32_t n_workgroups) {
    __shared__ int is_last_;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0 && threadIdx.z == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (t == n_workgroups - 1u);
        if (last) {
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        is_last_ = last;
    }
    __syncthreads();
    return is_last_ != 0;
}

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
// the same with the sample kernel computed on the fly (kw_i = 1 - sqrt(1 - exp(-gamma dist_i^2)),
// the arithmetic of dbgsom_exp_similarity) and the status flag also left as a float64 behind the
// sums (sums[M (d + 3)]: it rides in the all-reduce buffer): two launches less per epoch
// `fill`: rows whose distance is still -1 (the filtered search knew their winner without evaluating it:
// FilteredCall::defer_dist) get it inside the sums kernel -- the chain against their winner, W / ww = the
// epoch's prototypes and their norms, xx = the samples' norms; `dist` is then written too
struct DistFill {
    const double *W = nullptr, *ww = nullptr, *xx = nullptr;
    int round_f32 = 0;
};
bool accumulate_can_fill_distances(int x_dtype, int64_t d);
int launch_accumulate_epoch(const void *X, int x_dtype, int64_t N, int64_t d, int64_t ldx,
                            const int64_t *idx, double gamma, const double *dist, int64_t M,
                            double *sums, int32_t *status, void *ws, size_t ws_bytes, hipStream_t s,
                            const DistFill *fill = nullptr);
size_t bucket_sort_workspace_bytes(int64_t N, int64_t M);
int launch_bucket_sort(const int64_t *idx, int64_t N, int64_t M, int32_t *order, void *ws,
                       hipStream_t s);
const uint32_t *bucket_sort_seg_start(const void *ws, int64_t N, int64_t M);
// the filtered search with everything the engine may add to the public dbgsom_bmu_filtered call
// optional per-stage timing of dbgsom_bmu_filtered (bench / profiling): events on the caller's stream
struct StageTimer {
    bool enabled = false, valid = false;
    hipEvent_t ev[6] = {};
    bool created = false;
    void mark(int k, hipStream_t s) {
        if (!enabled) return;
        if (!created) { for (auto &e : ev) (void)hipEventCreate(&e); created = true; }
        (void)hipEventRecord(ev[k], s);
    }
    void destroy() {
        if (created) for (auto &e : ev) (void)hipEventDestroy(e);
        created = false; valid = false;
    }
};

// a second stream for launches that may overlap (per FilterAux, created on first use; if it cannot be
// created the work simply stays on the caller's stream)
struct SideStream {
    hipStream_t stream = nullptr, stream2 = nullptr;
    hipEvent_t forked = nullptr, joined = nullptr, joined2 = nullptr, mid = nullptr;
    hipEvent_t gap_fork = nullptr, gap_done = nullptr;   // the prototype gaps beside the seed pre-pass
    int state = 0;  // 0 untried, 1 ready, -1 unavailable
    int device = -1;
    bool ready() {
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess) return false;
        if (state == 0) {
            device = dev;
            // (lowest priority: what runs here is off the critical path -- a few long chains beside the
            //  caller's stream -- and must not starve the short dependent kernels there: at equal priority the
            //  bucket sort between the refinement and the pair kernel took 0.5 ms instead of 0.06)
            int prio_low = 0, prio_high = 0;
            (void)hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
            state = (hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, prio_low) == hipSuccess &&
                     hipStreamCreateWithPriority(&stream2, hipStreamNonBlocking, prio_low) == hipSuccess &&
                     hipEventCreateWithFlags(&forked, hipEventDisableTiming) == hipSuccess &&
                     hipEventCreateWithFlags(&joined, hipEventDisableTiming) == hipSuccess &&
                     hipEventCreateWithFlags(&joined2, hipEventDisableTiming) == hipSuccess &&
                     hipEventCreateWithFlags(&mid, hipEventDisableTiming) == hipSuccess &&
                     hipEventCreateWithFlags(&gap_fork, hipEventDisableTiming) == hipSuccess &&
                     hipEventCreateWithFlags(&gap_done, hipEventDisableTiming) == hipSuccess) ? 1 : -1;
        }
        return state == 1 && dev == device;  // (another device current than the one the streams live on: no fork)
    }
    void destroy() {
        if (state == 1) {
            (void)hipStreamDestroy(stream); (void)hipStreamDestroy(stream2);
            for (hipEvent_t e : {forked, joined, joined2, mid, gap_fork, gap_done}) (void)hipEventDestroy(e);
        }
        state = 0;
    }
};

struct FilterAux {
    StageTimer timer;
    SideStream side;
    void destroy() { timer.destroy(); side.destroy(); }
};
int filter_stage_ms(FilterAux &aux, double *ms5);   // [0] slice W + tables, [1] pre-pass, [2] bucket sort, [3] sweep, [4] exact stage

struct FilteredCall {
    const void *X = nullptr;
    int x_dtype = 0;
    int64_t N = 0, d = 0, ldx = 0;
    const double *xx = nullptr;
    const void *xplanes = nullptr;
    const double *W = nullptr;
    int64_t M = 0;
    const double *ww = nullptr;
    const int64_t *prev_idx = nullptr;
    const int32_t *order = nullptr;
    int seed_stride = 0, sweep_planes = 0, round_f32 = 0;
    int k = 1;   // nearest prototypes per sample: 1, or 2 (idx / dist then hold N x 2; the pruning form only)
    int64_t *idx = nullptr;
    double *dist = nullptr;
    void *ws = nullptr;
    size_t ws_bytes = 0;
    hipStream_t stream = nullptr;
    // hinted pruning bound (filter.hip 2c): the previous epoch's exact distances and how far each
    // prototype has moved since
    const double *hint_dist = nullptr, *hint_shift = nullptr;
    // per-sample refinement of the candidate lists (2d): 0 = off, else the longest list it should take
    int refine_rows = 0;
    // (with the refinement) a sample whose candidates narrow down to ONE gets its winner and dist = -1: the
    // caller evaluates that distance itself, on its own pass over the rows (launch_accumulate_epoch's `fill`)
    bool defer_dist = false;
    // the stored rows when they are not what X points to (bfloat16 storage behind a float32 copy): the
    // pair kernel of the refinement streams these
    const void *X_store = nullptr;
    int store_dtype = -1;
    int64_t ld_store = 0;
    // > 0: the call waits for its candidate lists and, when they average more than this, returns
    // DBGSOM_LISTS_LONG without the exact stage (idx / dist untouched): an arm of the search policy that has
    // never run on this map may leave the whole map a candidate, and the exact stage over such lists costs
    // twice the all-pairs search the caller can run instead
    double guard_mean = 0.0;
    // stage timer and side streams of the caller (a context's own); nullptr: this thread's
    FilterAux *aux = nullptr;
};
constexpr int DBGSOM_LISTS_LONG = 1000;   // (internal status of launch_bmu_filtered, never crosses the ABI)
int launch_bmu_filtered(const FilteredCall &call);
size_t smooth_workspace_bytes(int64_t M, int64_t d);
int launch_smooth(const double *sums, int64_t M, int64_t d, const float *hop, double sigma,
                  int layout, const double *W_old, double *W_new, double *change_total, void *ws,
                  size_t ws_bytes, hipStream_t s);
// smoothing sharded over the ranks (smooth.hip): column blocks of cb = smooth_block_cols columns; one block
// of the reduce-scatter buffer = S block (M x cb) = smooth_block_elems values; [K | a | E | status] stay in `sums`
int64_t smooth_block_cols(int64_t d, int nranks);
int64_t smooth_block_elems(int64_t M, int64_t d, int nranks);
int launch_pack_blocks(const double *sums, int64_t M, int64_t d, int nranks, double *out, hipStream_t s);
int launch_smooth_block(const double *S_b, const double *K, const double *a, int64_t M, int64_t cb, int64_t d_full,
                        const float *hop, double sigma, int layout, double *Wb, void *ws, size_t ws_bytes,
                        hipStream_t s);
int launch_rowchange_blocks(const double *blocks, int64_t M, int64_t d, int64_t cb, const double *W_old, double *W_new,
                            double *change_total, void *ws, hipStream_t s);

}  // namespace dbgsom
