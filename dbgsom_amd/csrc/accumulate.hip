// Per-neuron sums of one epoch on gfx950: S_j = sum kw_i x_i, K_j = sum kw_i, a_j = hits,
// E_j = sum dist_i, id-indexed and bitwise reproducible.
//
// Replaces (reference dbgsom/BaseSom.py)
//   :488-489    argsort(winners) / unique(return_index)          -> stable counting sort
//   :1028-1055  numba_voronoi_set_centers (numerator/denominator) -> ordered segmented sums
//   :500-503    neuron_activations                                -> histogram counts
//   :1058-1073  numba_quantization_error (serial semantics)       -> ordered segmented sum
//
// HBM-bound: X is streamed exactly once (row gathers of whole contiguous rows, 16 B per lane),
// everything else is O(N) integers or O(chunks * d).  No floating-point atomics: samples are
// bucketed by winner with a stable counting sort (per-workgroup LDS histograms + a column scan),
// each neuron's list is cut into chunks of <= CH rows, one workgroup sums one chunk in list
// order, and a second pass adds a neuron's chunk partials in chunk order.
#include "common.h"

namespace dbgsom {

constexpr int AT = 256;          // threads per workgroup
// samples per histogram / scatter workgroup: the scatter walks its samples 64 per round, one
// dependent round after the other -- 2048 samples are 32 rounds (fine when there are thousands of
// workgroups), a small sample set (C2, a rank's share in strong scaling) gets 512 = 8 rounds and
// four times the workgroups (scatter_kernel 19 -> see DESIGN.md).  A function of N alone.
static int hs_for(int64_t N) { return N <= 300000 ? 512 : 2048; }
constexpr int CH = 128;          // rows per segmented-sum chunk

struct AccWs {
    int32_t *order;       // N          sample ids, bucketed by winner, stable
    uint32_t *blk;        // nb * M     per-workgroup histograms -> exclusive block offsets
    uint32_t *count;      // M
    uint32_t *seg_start;  // M + 1
    uint32_t *chunk_pre;  // M + 1      exclusive scan of ceil(count / CH)
    double *slab;         // maxchunks * (d + 2)   [partial S | partial K | partial E]
    double *gslab;        // M * finalize_groups(M) * (d + 2): second level of the ordered sum
    uint32_t *ticket;     // "last workgroup" ticket of the fused column / segment scan
    int64_t nb, maxchunks;
};

// A small map has few neurons with very many chunk partials each: their ordered sum is split into
// NG consecutive groups summed side by side (then added in group order) so that it fills the
// chip; NG depends on M alone, the grouping on the counts alone -> still bitwise reproducible.
static int finalize_groups(int64_t M) {
    const int64_t g = 512 / (M > 0 ? M : 1);
    return (int)(g < 1 ? 1 : (g > 32 ? 32 : g));
}

static size_t carve(AccWs *w, char *base, int64_t N, int64_t d, int64_t M) {
    const int HS = hs_for(N);
    const int64_t nb = (N + HS - 1) / HS;
    const int64_t maxchunks = (N + CH - 1) / CH + M;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return o; };
    const size_t o_order = take((size_t)N * 4);
    const size_t o_blk = take((size_t)nb * M * 4);
    const size_t o_count = take((size_t)M * 4);
    const size_t o_seg = take((size_t)(M + 1) * 4);
    const size_t o_chunk = take((size_t)(M + 1) * 4);
    const size_t o_slab = take((size_t)maxchunks * (d + 2) * 8);
    const size_t o_gslab = take((size_t)M * finalize_groups(M) * (d + 2) * 8);
    const size_t o_ticket = take(256);
    if (w) {
        w->ticket = (uint32_t *)(base + o_ticket);
        w->order = (int32_t *)(base + o_order);
        w->blk = (uint32_t *)(base + o_blk);
        w->count = (uint32_t *)(base + o_count);
        w->seg_start = (uint32_t *)(base + o_seg);
        w->chunk_pre = (uint32_t *)(base + o_chunk);
        w->slab = (double *)(base + o_slab);
        w->gslab = (double *)(base + o_gslab);
        w->nb = nb;
        w->maxchunks = maxchunks;
    }
    return off;
}

size_t accumulate_workspace_bytes(int64_t N, int64_t d, int64_t M) {
    if (N < 0 || d < 1 || M < 1) return 0;
    return carve(nullptr, nullptr, N, d, M);
}

// ---- 1. per-workgroup histogram of winners ---------------------------------------------------
__global__ __launch_bounds__(AT) void hist_kernel(const int64_t *__restrict__ win, int64_t N,
                                                  int M, uint32_t *__restrict__ blk,
                                                  int32_t *__restrict__ status,
                                                  uint32_t *__restrict__ ticket, int HS) {
    extern __shared__ uint32_t h[];
    if (blockIdx.x == 0 && threadIdx.x == 0) *ticket = 0u;  // of the scan kernel that follows
    for (int j = threadIdx.x; j < M; j += AT) h[j] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * HS;
    for (int t = threadIdx.x; t < HS; t += AT) {
        const int64_t i = base + t;
        if (i < N) {
            const int64_t j = win[i];
            if (j >= 0 && j < M) atomicAdd(&h[j], 1u);
            else if (status) atomicOr(status, 1);
        }
    }
    __syncthreads();
    uint32_t *dst = blk + (size_t)blockIdx.x * M;
    for (int j = threadIdx.x; j < M; j += AT) dst[j] = h[j];
}

// ---- 2. column scan over workgroups: blk -> exclusive offsets, count[j] ----------------------
// A column is nb = N / HS entries long and its prefix is serial: CS_GROUPS threads share it (sum of
// a group of consecutive workgroups each, prefix over the groups in LDS, then every thread rewrites
// its group) -- 8 x fewer dependent load rounds than one thread per column.
constexpr int CS_COLS = 32, CS_GROUPS = 8;
__device__ __forceinline__ void colscan_body(uint32_t *__restrict__ blk, int64_t nb, int M,
                                             uint32_t *__restrict__ count) {
    __shared__ uint32_t part[CS_GROUPS][CS_COLS];
    const int c = threadIdx.x % CS_COLS, g = threadIdx.x / CS_COLS;
    const int j = blockIdx.x * CS_COLS + c;
    const int64_t len = (nb + CS_GROUPS - 1) / CS_GROUPS;
    const int64_t b0 = min(nb, (int64_t)g * len), b1 = min(nb, b0 + len);
    uint32_t sum = 0;
    if (j < M) {
        int64_t b = b0;
        for (; b + 8 <= b1; b += 8) {  // 8 independent loads in flight
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = blk[(size_t)(b + u) * M + j];
#pragma unroll
            for (int u = 0; u < 8; ++u) sum += v[u];
        }
        for (; b < b1; ++b) sum += blk[(size_t)b * M + j];
    }
    part[g][c] = sum;
    __syncthreads();
    uint32_t run = 0;
    for (int u = 0; u < g; ++u) run += part[u][c];
    if (j >= M) return;
    if (g == CS_GROUPS - 1) count[j] = run + sum;
    int64_t b = b0;
    for (; b + 8 <= b1; b += 8) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = blk[(size_t)(b + u) * M + j];
#pragma unroll
        for (int u = 0; u < 8; ++u) { blk[(size_t)(b + u) * M + j] = run; run += v[u]; }
    }
    for (; b < b1; ++b) {
        const uint32_t v = blk[(size_t)b * M + j];
        blk[(size_t)b * M + j] = run;
        run += v;
    }
}

// ---- 3. exclusive scans over the M neurons (one workgroup of NTHR threads) ----------------------
template <int NTHR>
__device__ __forceinline__ void segscan_body(const uint32_t *__restrict__ count, int M,
                                             uint32_t *__restrict__ seg_start,
                                             uint32_t *__restrict__ chunk_pre) {
    constexpr int NWAVE = NTHR / 64;
    __shared__ uint32_t wa[NWAVE], wb[NWAVE];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int per = (M + NTHR - 1) / NTHR;
    const int lo = min(M, t * per), hi = min(M, lo + per);
    uint32_t a = 0, b = 0;
    for (int j = lo; j < hi; ++j) { a += count[j]; b += (count[j] + CH - 1) / CH; }
    uint32_t ia = a, ib = b;  // inclusive scan over the wavefront, then over the wavefronts
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t va = __shfl_up(ia, off, 64), vb = __shfl_up(ib, off, 64);
        if (lane >= off) { ia += va; ib += vb; }
    }
    if (lane == 63) { wa[wv] = ia; wb[wv] = ib; }
    __syncthreads();
    uint32_t pa = 0, pb = 0;
    for (int u = 0; u < wv; ++u) { pa += wa[u]; pb += wb[u]; }
    if (t == NTHR - 1) { seg_start[M] = pa + ia; chunk_pre[M] = pb + ib; }
    a = pa + ia - a; b = pb + ib - b;  // exclusive prefix of this thread's range
    for (int j = lo; j < hi; ++j) {
        seg_start[j] = a; chunk_pre[j] = b;
        a += count[j]; b += (count[j] + CH - 1) / CH;
    }
}

// column scan (2.) and, in the workgroup that finishes last, the scans over the neurons (3.): they
// need every column's count, and a kernel of their own is one 5 us launch more per sort
__global__ __launch_bounds__(CS_COLS * CS_GROUPS) void scan_kernel(uint32_t *__restrict__ blk, int64_t nb, int M,
                                                                   uint32_t *__restrict__ count,
                                                                   uint32_t *__restrict__ seg_start,
                                                                   uint32_t *__restrict__ chunk_pre,
                                                                   uint32_t *__restrict__ ticket) {
    colscan_body(blk, nb, M, count);
    if (last_workgroup_done(ticket, gridDim.x)) segscan_body<CS_COLS * CS_GROUPS>(count, M, seg_start, chunk_pre);
}

// ---- 4. stable scatter of sample ids into their neuron's segment -----------------------------
// One wavefront per workgroup of HS samples, 64 per round in sample order.  The rank of a sample
// among the lanes holding the same winner comes from ballots over the bits of the key (peers =
// lanes that agree in every bit), the first of them moves the neuron's write position on: no
// search through the round's keys, no barrier between wavefronts.
constexpr int SCW = 64;
template <int HS>
__global__ __launch_bounds__(SCW) void scatter_kernel(const int64_t *__restrict__ win, int64_t N,
                                                      int M, int nbits,
                                                      const uint32_t *__restrict__ blk,
                                                      const uint32_t *__restrict__ seg_start,
                                                      int32_t *__restrict__ order) {
    extern __shared__ uint32_t cnt[];  // M running write positions of this workgroup
    constexpr int R = HS / SCW;
    const int lane = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * HS;
    int keys[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {  // all of the workgroup's keys are in flight at once
        const int64_t i = base + r * SCW + lane;
        int64_t j = -1;
        if (i < N) j = win[i];
        keys[r] = (j >= 0 && j < M) ? (int)j : -1;
    }
    const uint32_t *off = blk + (size_t)blockIdx.x * M;
#pragma unroll 8
    for (int j = lane; j < M; j += SCW) cnt[j] = seg_start[j] + off[j];
    __syncthreads();
    const uint64_t lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int key = keys[r];
        uint64_t peers = __builtin_amdgcn_ballot_w64(key >= 0);
        for (int b = 0; b < nbits; ++b) {
            const bool bit = (key >> b) & 1;
            const uint64_t m = __builtin_amdgcn_ballot_w64(bit);
            peers &= bit ? m : ~m;
        }
        const uint64_t before = peers & lt;
        if (key >= 0) order[cnt[key] + (uint32_t)__popcll(before)] = (int32_t)(base + r * SCW + lane);
        __syncthreads();  // every position of this round has been read from cnt
        if (key >= 0 && before == 0) cnt[key] += (uint32_t)__popcll(peers);  // one lane per key
        __syncthreads();
    }
}

template <typename XT, int VEC>
__device__ __forceinline__ void load_vec(const XT *__restrict__ src, double (&v)[VEC]) {
    if constexpr (sizeof(XT) == 4 && VEC == 4) {
        const float4 t4 = *reinterpret_cast<const float4 *>(src);
        v[0] = t4.x; v[1] = t4.y; v[2] = t4.z; v[3] = t4.w;
    } else if constexpr (sizeof(XT) == 8 && VEC == 2) {
        const double2 t2 = *reinterpret_cast<const double2 *>(src);
        v[0] = t2.x; v[1] = t2.y;
    } else if constexpr (sizeof(XT) == 2 && VEC == 8) {
        const uint4 a = *reinterpret_cast<const uint4 *>(src);
        const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] = (double)__uint_as_float(w[e] << 16);
            v[2 * e + 1] = (double)__uint_as_float(w[e] & 0xffff0000u);
        }
    } else {
        static_assert(VEC == 1, "unsupported vector width");
        v[0] = widen(src[0]);
    }
}

// ---- 5. one workgroup sums one chunk (<= CH rows of one neuron) in list order ----------------
template <typename XT, int VEC>
__global__ __launch_bounds__(AT) void segsum_kernel(
    const XT *__restrict__ X, int d, int64_t ldx, const int32_t *__restrict__ order,
    const double *__restrict__ kw, double gamma, const double *__restrict__ dist,
    const uint32_t *__restrict__ seg_start, const uint32_t *__restrict__ count,
    const uint32_t *__restrict__ chunk_pre, int M, double *__restrict__ slab) {
    __shared__ int32_t rows_s[CH];
    __shared__ double kw_s[CH];
    __shared__ double dist_s[CH];
    __shared__ double red[AT * VEC];
    __shared__ uint32_t info[3];
    const int tid = threadIdx.x;
    const uint32_t c = blockIdx.x;
    if (c >= chunk_pre[M]) return;  // uniform per workgroup
    if (tid == 0) {
        int lo = 0, hi = M;  // last j with chunk_pre[j] <= c
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (chunk_pre[mid] <= c) lo = mid; else hi = mid;
        }
        const uint32_t begin = seg_start[lo] + (c - chunk_pre[lo]) * CH;
        const uint32_t end = min(begin + (uint32_t)CH, seg_start[lo] + count[lo]);
        info[0] = begin; info[1] = end - begin;
    }
    __syncthreads();
    const uint32_t begin = info[0];
    const int n = (int)info[1];
    if (tid < n) {
        const int32_t r = order[begin + tid];
        rows_s[tid] = r;
        const double dd = dist[r];
        // kw == nullptr: the sample kernel of BaseSom._calculate_exp_similarity (BaseSom.py:533-538)
        // on the fly, the arithmetic of exp_similarity_kernel (bmu.hip)
        kw_s[tid] = kw ? kw[r] : 1.0 - sqrt(1.0 - exp(-gamma * (dd * dd)));
        dist_s[tid] = dd;
    }
    __syncthreads();
    double *out = slab + (size_t)c * (d + 2);
    if (tid == AT - 1) {  // the scalar partials, in list order
        double sk = 0.0, se = 0.0;
        for (int p = 0; p < n; ++p) { sk += kw_s[p]; se += dist_s[p]; }
        out[d] = sk;
        out[d + 1] = se;
    }
    const int Q = d / VEC;  // column groups (VEC divides d by construction)
    if (Q >= AT) {
        for (int q = tid; q < Q; q += AT) {
            double acc[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] = 0.0;
#pragma unroll 4
            for (int p = 0; p < n; ++p) {
                const XT *src = X + (int64_t)rows_s[p] * ldx + (int64_t)q * VEC;
                const double w = kw_s[p];
                double v[VEC];
                load_vec<XT, VEC>(src, v);
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] += w * v[e];
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) out[q * VEC + e] = acc[e];
        }
    } else {
        const int RL = AT / Q;  // row lanes working side by side on the same column group
        const int rl = tid / Q, q = tid - rl * Q;
        double acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.0;
        if (rl < RL) {
#pragma unroll 4
            for (int p = rl; p < n; p += RL) {
                const XT *src = X + (int64_t)rows_s[p] * ldx + (int64_t)q * VEC;
                const double w = kw_s[p];
                double v[VEC];
                load_vec<XT, VEC>(src, v);
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] += w * v[e];
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) red[(rl * Q + q) * VEC + e] = acc[e];
        }
        __syncthreads();
        if (rl == 0) {  // row lanes are added in lane order
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                double s = red[q * VEC + e];
                for (int u = 1; u < RL; ++u) s += red[(u * Q + q) * VEC + e];
                out[q * VEC + e] = s;
            }
        }
    }
}

// ---- 5b. the same chunk, with the distances of the rows that do not have one yet ----------------------
// The refinement of the filtered search (filter.hip 2d) knows the winner of a sample whose candidates it
// could narrow down to ONE without ever touching the sample's float rows; its distance is the float64 chain
// against that one prototype -- against THIS chunk's prototype.  Such rows arrive with dist == -1: the
// chunk's rows are brought into LDS in sub-blocks of SR whole rows (LDS-DMA, two buffers), one lane per row
// runs the chain acc = fma(w_k, x_k, acc), k ascending (the arithmetic of every exact BMU kernel: same
// bits), then all threads form the weighted sums from the same LDS image -- the float rows are streamed
// ONCE for the distance and the sums (pair kernel + segsum: twice).  Sums, their order of additions and
// the scalar partials are those of segsum_kernel bit for bit.
typedef __attribute__((address_space(3))) void *acc_lds_ptr_t;
typedef const __attribute__((address_space(1))) void *acc_gbl_ptr_t;
__device__ __forceinline__ void acc_dma16(const void *src, void *lds_dst) {
    __builtin_amdgcn_global_load_lds((acc_gbl_ptr_t)src, (acc_lds_ptr_t)lds_dst, 16, 0, 0);
}
constexpr int SR = 16;   // rows per sub-block (lanes of the chain)
constexpr int SD_QG = 4; // column groups a thread keeps across the sub-blocks of a chunk (rows of <= 4 AT groups)

struct SegDistLds {   // byte offsets inside the one dynamic LDS object
    int buf_bytes, o_w, o_rows, o_kw, o_dist, o_red, o_info, total;
};
static SegDistLds segdist_lds(int64_t d, size_t es, int vec) {
    SegDistLds l;
    l.buf_bytes = (int)(((size_t)SR * d * es + 1023) / 1024 * 1024);
    l.o_w = 2 * l.buf_bytes;
    l.o_rows = l.o_w + (int)d * 8;
    l.o_kw = l.o_rows + CH * 4;
    l.o_dist = l.o_kw + CH * 8;
    l.o_red = l.o_dist + CH * 8;
    l.o_info = l.o_red + AT * vec * 8;
    l.total = l.o_info + 16;
    return l;
}

template <typename XT, int VEC>
__device__ __forceinline__ void lds_vec(const char *src, double (&v)[VEC]) {
    if constexpr (sizeof(XT) == 4 && VEC == 4) {
        typedef float f4_t __attribute__((ext_vector_type(4)));
        const f4_t t4 = *reinterpret_cast<const f4_t *>(src);
        v[0] = t4.x; v[1] = t4.y; v[2] = t4.z; v[3] = t4.w;
    } else if constexpr (sizeof(XT) == 8 && VEC == 2) {
        typedef double d2v_t __attribute__((ext_vector_type(2)));
        const d2v_t t2 = *reinterpret_cast<const d2v_t *>(src);
        v[0] = t2.x; v[1] = t2.y;
    } else {
        static_assert(sizeof(XT) == 2 && VEC == 8, "unsupported vector width");
        typedef unsigned u4_t __attribute__((ext_vector_type(4)));
        const u4_t a = *reinterpret_cast<const u4_t *>(src);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] = (double)__uint_as_float(a[e] << 16);
            v[2 * e + 1] = (double)__uint_as_float(a[e] & 0xffff0000u);
        }
    }
}

template <typename XT, int VEC>
__global__ __launch_bounds__(AT) void segsum_dist_kernel(
    const XT *__restrict__ X, int d, int64_t ldx, const int32_t *__restrict__ order, double gamma,
    double *__restrict__ dist, const uint32_t *__restrict__ seg_start, const uint32_t *__restrict__ count,
    const uint32_t *__restrict__ chunk_pre, int M, double *__restrict__ slab, const double *__restrict__ W,
    const double *__restrict__ ww, const double *__restrict__ xx, int round_f32, SegDistLds L) {
    extern __shared__ __attribute__((aligned(16))) char dyn[];   // ONE LDS object (see refine.h)
    double *w_s = reinterpret_cast<double *>(dyn + L.o_w);
    int32_t *rows_s = reinterpret_cast<int32_t *>(dyn + L.o_rows);
    double *kw_s = reinterpret_cast<double *>(dyn + L.o_kw), *dist_s = reinterpret_cast<double *>(dyn + L.o_dist);
    double *red = reinterpret_cast<double *>(dyn + L.o_red);
    uint32_t *info = reinterpret_cast<uint32_t *>(dyn + L.o_info);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t c = blockIdx.x;
    if (c >= chunk_pre[M]) return;  // uniform per workgroup
    if (tid == 0) {
        int lo = 0, hi = M;  // last j with chunk_pre[j] <= c
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (chunk_pre[mid] <= c) lo = mid; else hi = mid;
        }
        const uint32_t begin = seg_start[lo] + (c - chunk_pre[lo]) * CH;
        const uint32_t end = min(begin + (uint32_t)CH, seg_start[lo] + count[lo]);
        info[0] = begin; info[1] = end - begin; info[2] = (uint32_t)lo;
    }
    __syncthreads();
    const uint32_t begin = info[0];
    const int n = (int)info[1], j = (int)info[2];
    if (tid < n) {
        const int32_t r = order[begin + tid];
        rows_s[tid] = r;
        dist_s[tid] = dist[r];   // (-1: to be computed here)
    }
    for (int k = tid; k < d; k += AT) w_s[k] = W[(size_t)j * d + k];
    const double ww_j = ww[j];
    __syncthreads();
    constexpr int ES = (int)sizeof(XT);
    const int rowbytes = d * ES;
    const int ninstr = L.buf_bytes / 1024;
    // sub-block s into buffer s & 1: instruction i covers LDS bytes [1024 i, +1024) = 16-byte pieces of whole rows
    auto issue = [&](int sb) {
        char *buf = dyn + (sb & 1) * L.buf_bytes;
        const int nrow = min(SR, n - sb * SR);
        for (int i = wave; i < ninstr; i += AT / 64) {
            const int a = i * 1024 + lane * 16;
            int row = a / rowbytes;
            const int off = a - row * rowbytes;
            row = row < nrow ? row : nrow - 1;   // (behind the sub-block's rows: any valid address, never read)
            acc_dma16(reinterpret_cast<const char *>(X + (int64_t)rows_s[sb * SR + row] * ldx) + off, buf + i * 1024);
        }
    };
    const int nsub = (n + SR - 1) / SR;
    const int Q = d / VEC;  // column groups (VEC divides d by construction)
    const bool wide = Q >= AT;
    const int RL = wide ? 1 : AT / Q;  // row lanes working side by side on the same column group (segsum_kernel)
    const int rl = wide ? 0 : tid / Q, q0 = wide ? tid : tid - rl * Q;
    // (narrow rows: acc[0] = this thread's column group for its row lane; wide rows: up to SD_QG column groups
    //  q = tid + g AT per thread, kept in registers across the sub-blocks -- the launcher checks Q <= SD_QG AT)
    double acc[SD_QG][VEC];
#pragma unroll
    for (int g = 0; g < SD_QG; ++g)
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[g][e] = 0.0;
    double *out = slab + (size_t)c * (d + 2);
    issue(0);
    for (int sb = 0; sb < nsub; ++sb) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // sub-block sb has landed; buffer (sb + 1) & 1 has been summed
        if (sb + 1 < nsub) issue(sb + 1);
        const char *buf = dyn + (sb & 1) * L.buf_bytes;
        const int nrow = min(SR, n - sb * SR);
        if (tid < SR) {   // ---- the chains: lane = row
            const int p = sb * SR + tid;
            const bool need = tid < nrow && dist_s[p] == -1.0;
            if (__builtin_amdgcn_ballot_w64(need) != 0ull) {
                const char *xr = buf + (tid < nrow ? tid : 0) * rowbytes;
                double a = 0.0;
                // 16 features per block, the NEXT block's LDS reads in flight under this block's 16 dependent
                // fmas (sched_barriers: left alone the compiler reuses one register quad for every read and
                // waits for each -- 24 exposed LDS round trips per block, 7 x the time of the chain itself)
                // (raw 16-byte pieces are kept as they come and widened in the fma phase: the wait for a block's
                //  reads then sits in front of ITS fmas, one block later)
                typedef unsigned raw4_t __attribute__((ext_vector_type(4)));
                constexpr int XR = ES;            // 16-byte pieces of 16 features of the row: 4 (f32), 8 (f64), 2 (bf16)
                struct Blk { raw4_t x[XR]; raw4_t w[8]; };
                auto load_block = [&](int k, Blk &b) {
#pragma unroll
                    for (int u = 0; u < XR; ++u) b.x[u] = *reinterpret_cast<const raw4_t *>(xr + k * ES + 16 * u);
#pragma unroll
                    for (int u = 0; u < 8; ++u) b.w[u] = *reinterpret_cast<const raw4_t *>(reinterpret_cast<const char *>(w_s + k) + 16 * u);
                };
                auto chain_block = [&](const Blk &b) {
                    // (the prototype is the first factor, as the A operand of the matrix instruction)
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const raw4_t wq = b.w[u >> 1];
                        const double wv = __hiloint2double((int)wq[2 * (u & 1) + 1], (int)wq[2 * (u & 1)]);
                        double xv;
                        if constexpr (ES == 4) {
                            xv = (double)__uint_as_float(b.x[u >> 2][u & 3]);
                        } else if constexpr (ES == 8) {
                            const raw4_t xq = b.x[u >> 1];
                            xv = __hiloint2double((int)xq[2 * (u & 1) + 1], (int)xq[2 * (u & 1)]);
                        } else {
                            const unsigned wd = b.x[u >> 3][(u & 7) >> 1];
                            xv = (double)__uint_as_float((u & 1) ? (wd & 0xffff0000u) : (wd << 16));
                        }
                        a = fma(wv, xv, a);
                    }
                };
                Blk ba, bb;
                load_block(0, ba);
                for (int k = 0; k < d; k += 32) {   // (d % 16 == 0: the engine's padded rows)
                    __builtin_amdgcn_sched_barrier(0);
                    if (k + 16 < d) load_block(k + 16, bb);
                    __builtin_amdgcn_sched_barrier(0);
                    chain_block(ba);
                    if (k + 16 >= d) break;
                    __builtin_amdgcn_sched_barrier(0);
                    if (k + 32 < d) load_block(k + 32, ba);
                    __builtin_amdgcn_sched_barrier(0);
                    chain_block(bb);
                }
                if (need) {
                    const int32_t r = rows_s[p];
                    double rv = (xx[r] + (-2.0 * a)) + ww_j;
                    if (!(rv > 0.0)) rv = (rv != rv) ? rv : 0.0;
                    double dv = sqrt(rv);
                    if (round_f32) dv = (double)(float)dv;
                    dist_s[p] = dv;
                    dist[r] = dv;
                }
            }
            if (tid < nrow) {
                const double dd = dist_s[p];
                // the sample kernel of BaseSom._calculate_exp_similarity (BaseSom.py:533-538), the arithmetic
                // of exp_similarity_kernel (bmu.hip) and of segsum_kernel
                kw_s[p] = 1.0 - sqrt(1.0 - exp(-gamma * (dd * dd)));
            }
        }
        __syncthreads();
        // ---- the weighted sums of this sub-block's rows, in list order (segsum_kernel's order)
        if (wide) {
#pragma unroll
            for (int g = 0; g < SD_QG; ++g) {
                const int q = tid + g * AT;
                if (q < Q)
                    for (int t = 0; t < nrow; ++t) {
                        const double w = kw_s[sb * SR + t];
                        double v[VEC];
                        lds_vec<XT, VEC>(buf + t * rowbytes + q * VEC * ES, v);
#pragma unroll
                        for (int e = 0; e < VEC; ++e) acc[g][e] += w * v[e];
                    }
            }
        } else if (rl < RL) {
            // (row lane rl takes the chunk's rows p = rl, rl + RL, ...)
            int t = (rl - (sb * SR) % RL + RL) % RL;
            for (; t < nrow; t += RL) {
                const double w = kw_s[sb * SR + t];
                double v[VEC];
                lds_vec<XT, VEC>(buf + t * rowbytes + q0 * VEC * ES, v);
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[0][e] += w * v[e];
            }
        }
    }
    __syncthreads();
    if (tid == AT - 1) {  // the scalar partials, in list order
        double sk = 0.0, se = 0.0;
        for (int p = 0; p < n; ++p) { sk += kw_s[p]; se += dist_s[p]; }
        out[d] = sk;
        out[d + 1] = se;
    }
    if (wide) {
#pragma unroll
        for (int g = 0; g < SD_QG; ++g) {
            const int q = tid + g * AT;
            if (q < Q)
#pragma unroll
                for (int e = 0; e < VEC; ++e) out[q * VEC + e] = acc[g][e];
        }
    } else {
        if (rl < RL) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) red[(rl * Q + q0) * VEC + e] = acc[0][e];
        }
        __syncthreads();
        if (rl == 0) {  // row lanes are added in lane order
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                double s = red[q0 * VEC + e];
                for (int u = 1; u < RL; ++u) s += red[(u * Q + q0) * VEC + e];
                out[q0 * VEC + e] = s;
            }
        }
    }
}

// ---- 6. add each neuron's chunk partials in chunk order --------------------------------------
__global__ __launch_bounds__(AT) void finalize_kernel(const double *__restrict__ slab, int d,
                                                      int M, const uint32_t *__restrict__ count,
                                                      const uint32_t *__restrict__ chunk_pre,
                                                      int NG, double *__restrict__ gslab,
                                                      double *__restrict__ sums,
                                                      const int32_t *__restrict__ status,
                                                      double *__restrict__ status_f64) {
    const int j = blockIdx.x, g = blockIdx.y;
    if (status_f64 && j == 0 && g == 0 && blockIdx.z == 0 && threadIdx.x == 0)
        status_f64[0] = status[0] ? 1.0 : 0.0;   // the flag rides behind the sums in the all-reduce buffer
    const uint32_t b0 = chunk_pre[j], b1 = chunk_pre[j + 1];
    const uint32_t per = (b1 - b0 + NG - 1) / NG;  // chunks per group
    const uint32_t c0 = min(b1, b0 + g * per), c1 = min(b1, c0 + per);
    double *S = sums + (size_t)j * d;
    double *Kp = sums + (size_t)M * d, *ap = Kp + M, *Ep = ap + M;
    // (the column blocks of a neuron are separate workgroups, gridDim.z of them: one short chain
    // per thread instead of four one after the other)
    for (int col = threadIdx.x + AT * blockIdx.z; col < d + 2; col += AT * gridDim.z) {
        double s = 0.0;
        uint32_t c = c0;
        for (; c + 8 <= c1; c += 8) {  // loads batched, additions still in chunk order
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = slab[(size_t)(c + u) * (d + 2) + col];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; c < c1; ++c) s += slab[(size_t)c * (d + 2) + col];
        if (NG > 1) gslab[((size_t)j * NG + g) * (d + 2) + col] = s;
        else if (col < d) S[col] = s;
        else if (col == d) Kp[j] = s;
        else Ep[j] = s;
    }
    if (threadIdx.x == 0 && g == 0 && blockIdx.z == 0) ap[j] = (double)count[j];
}

// second level (NG > 1): the NG group sums of a neuron in group order
__global__ __launch_bounds__(AT) void finalize_groups_kernel(const double *__restrict__ gslab, int d,
                                                             int M, int NG,
                                                             double *__restrict__ sums) {
    const int j = blockIdx.x;
    double *S = sums + (size_t)j * d;
    double *Kp = sums + (size_t)M * d, *Ep = Kp + 2 * (size_t)M;
    for (int col = threadIdx.x; col < d + 2; col += AT) {
        double s = 0.0;
        for (int g = 0; g < NG; ++g) s += gslab[((size_t)j * NG + g) * (d + 2) + col];
        if (col < d) S[col] = s;
        else if (col == d) Kp[j] = s;
        else Ep[j] = s;
    }
}

static int key_bits(int64_t M) {  // bits that tell the keys 0 .. M - 1 apart
    int b = 0;
    while (((int64_t)1 << b) < M) ++b;
    return b;
}

// Stable bucket order of the samples by winner, on its own (the filtered BMU search visits the
// samples in this order).  `ws` needs bucket_sort_workspace_bytes(N, M); `order` gets N int32.
size_t bucket_sort_workspace_bytes(int64_t N, int64_t M) {
    const int HS = hs_for(N);
    const int64_t nb = (N + HS - 1) / HS;
    return align_up((size_t)nb * M * 4) + align_up((size_t)M * 4) + 2 * align_up((size_t)(M + 1) * 4) + 256;
}

// where launch_bucket_sort leaves the exclusive segment starts (M + 1 entries) in its workspace
const uint32_t *bucket_sort_seg_start(const void *ws, int64_t N, int64_t M) {
    const int HS = hs_for(N);
    const int64_t nb = (N + HS - 1) / HS;
    return reinterpret_cast<const uint32_t *>((const char *)ws + align_up((size_t)nb * M * 4) + align_up((size_t)M * 4));
}

static void launch_scatter(const int64_t *idx, int64_t N, int Mi, int64_t nb, const uint32_t *blk,
                           const uint32_t *seg_start, int32_t *order, hipStream_t s) {
    if (hs_for(N) == 512)
        hipLaunchKernelGGL(scatter_kernel<512>, dim3((unsigned)nb), dim3(SCW), (size_t)Mi * 4, s, idx, N, Mi,
                           key_bits(Mi), blk, seg_start, order);
    else
        hipLaunchKernelGGL(scatter_kernel<2048>, dim3((unsigned)nb), dim3(SCW), (size_t)Mi * 4, s, idx, N, Mi,
                           key_bits(Mi), blk, seg_start, order);
}

int launch_bucket_sort(const int64_t *idx, int64_t N, int64_t M, int32_t *order, void *ws,
                       hipStream_t s) {
    const int HS = hs_for(N);
    const int64_t nb = (N + HS - 1) / HS;
    char *base = (char *)ws;
    uint32_t *blk = (uint32_t *)base;
    base += align_up((size_t)nb * M * 4);
    uint32_t *count = (uint32_t *)base;
    base += align_up((size_t)M * 4);
    uint32_t *seg_start = (uint32_t *)base;
    base += align_up((size_t)(M + 1) * 4);
    uint32_t *chunk_pre = (uint32_t *)base;
    base += align_up((size_t)(M + 1) * 4);
    uint32_t *ticket = (uint32_t *)base;
    const int Mi = (int)M;
    hipLaunchKernelGGL(hist_kernel, dim3((unsigned)nb), dim3(AT), (size_t)M * 4, s, idx, N, Mi, blk,
                       (int32_t *)nullptr, ticket, HS);
    hipLaunchKernelGGL(scan_kernel, dim3((unsigned)((M + CS_COLS - 1) / CS_COLS)),
                       dim3(CS_COLS * CS_GROUPS), 0, s, blk, nb, Mi, count, seg_start, chunk_pre, ticket);
    launch_scatter(idx, N, Mi, nb, blk, seg_start, order, s);
    return launch_status("bucket sort kernels");
}

// ---------------------------------------------------------------------------------------------
bool accumulate_can_fill_distances(int x_dtype, int64_t d) {
    if (d % 16 != 0) return false;
    const int vec = x_dtype == DBGSOM_F32 ? 4 : (x_dtype == DBGSOM_F64 ? 2 : 8);
    if (d / vec > (int64_t)SD_QG * AT) return false;
    return segdist_lds(d, dtype_size(x_dtype), vec).total <= 160 * 1024;
}

static int accumulate_impl(const void *X, int x_dtype, int64_t N, int64_t d, int64_t ldx,
                           const int64_t *idx, const double *kw, double gamma, const double *dist, int64_t M,
                           double *sums, int32_t *status, bool status_behind_sums, void *ws,
                           size_t ws_bytes, hipStream_t s, const DistFill *fill = nullptr) {
    DBGSOM_REQUIRE(valid_dtype(x_dtype), "x_dtype must be DBGSOM_F32/F64/BF16");
    DBGSOM_REQUIRE(N >= 0 && N < 0x7fffffff && d >= 1 && d <= 0x7ffffff0 && ldx >= d, "bad sample shape");
    DBGSOM_REQUIRE(M >= 1 && M <= DBGSOM_MAX_PROTOTYPES, "M outside [1, DBGSOM_MAX_PROTOTYPES]");
    DBGSOM_REQUIRE(sums, "null sums");
    DBGSOM_REQUIRE(!status_behind_sums || status, "the status flag is needed behind the sums");
    if (status) DBGSOM_HIP_CHECK(hipMemsetAsync(status, 0, sizeof(int32_t), s));
    if (N == 0) {
        DBGSOM_HIP_CHECK(hipMemsetAsync(sums, 0, (size_t)(M * (d + 3) + (status_behind_sums ? 1 : 0)) * sizeof(double), s));
        return DBGSOM_OK;
    }
    DBGSOM_REQUIRE(X && idx && dist && ws, "null pointer");
    DBGSOM_REQUIRE(is_aligned(ws, 256), "workspace must be 256-byte aligned");
    if (ws_bytes < accumulate_workspace_bytes(N, d, M)) {
        set_error("dbgsom_accumulate: workspace too small (%zu < %zu)", ws_bytes,
                  accumulate_workspace_bytes(N, d, M));
        return DBGSOM_ENOMEM;
    }
    AccWs w;
    carve(&w, (char *)ws, N, d, M);
    const int Mi = (int)M, di = (int)d;

    hipLaunchKernelGGL(hist_kernel, dim3((unsigned)w.nb), dim3(AT), (size_t)M * 4, s, idx, N, Mi,
                       w.blk, status, w.ticket, hs_for(N));
    hipLaunchKernelGGL(scan_kernel, dim3((unsigned)((M + CS_COLS - 1) / CS_COLS)),
                       dim3(CS_COLS * CS_GROUPS), 0, s, w.blk, w.nb, Mi, w.count, w.seg_start, w.chunk_pre,
                       w.ticket);
    launch_scatter(idx, N, Mi, w.nb, w.blk, w.seg_start, w.order, s);

    const size_t xe = dtype_size(x_dtype);
    const bool al16 = is_aligned(X, 16) && ((ldx * xe) % 16 == 0);
    dim3 grid((unsigned)w.maxchunks), block(AT);
#define DBGSOM_SEGSUM(XT, V)                                                                    \
    hipLaunchKernelGGL((segsum_kernel<XT, V>), grid, block, 0, s, (const XT *)X, di, ldx, w.order, \
                       kw, gamma, dist, w.seg_start, w.count, w.chunk_pre, Mi, w.slab)
    if (fill) {
        // rows with dist == -1 get their distance (to their winner: this chunk's prototype) on the way
        DBGSOM_REQUIRE(!kw && al16 && accumulate_can_fill_distances(x_dtype, d) && fill->W && fill->ww && fill->xx,
                       "distances cannot be filled in for this shape");
        const int vec = x_dtype == DBGSOM_F32 ? 4 : (x_dtype == DBGSOM_F64 ? 2 : 8);
        const SegDistLds L = segdist_lds(d, xe, vec);
#define DBGSOM_SEGDIST(XT, V)                                                                             \
    do {                                                                                                  \
        static int attr_set = 0;                                                                          \
        if (attr_set < L.total) {                                                                         \
            DBGSOM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&segsum_dist_kernel<XT, V>), \
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            attr_set = 160 * 1024;                                                                        \
        }                                                                                                 \
        hipLaunchKernelGGL((segsum_dist_kernel<XT, V>), grid, block, (size_t)L.total, s, (const XT *)X, di, ldx, \
                           w.order, gamma, const_cast<double *>(dist), w.seg_start, w.count, w.chunk_pre, Mi, \
                           w.slab, fill->W, fill->ww, fill->xx, fill->round_f32, L);                      \
    } while (0)
        if (x_dtype == DBGSOM_F32) DBGSOM_SEGDIST(float, 4);
        else if (x_dtype == DBGSOM_F64) DBGSOM_SEGDIST(double, 2);
        else DBGSOM_SEGDIST(bf16_t, 8);
#undef DBGSOM_SEGDIST
    } else if (x_dtype == DBGSOM_F32) {
        if (al16 && d % 4 == 0) DBGSOM_SEGSUM(float, 4); else DBGSOM_SEGSUM(float, 1);
    } else if (x_dtype == DBGSOM_F64) {
        if (al16 && d % 2 == 0) DBGSOM_SEGSUM(double, 2); else DBGSOM_SEGSUM(double, 1);
    } else {
        if (al16 && d % 8 == 0) DBGSOM_SEGSUM(bf16_t, 8); else DBGSOM_SEGSUM(bf16_t, 1);
    }
#undef DBGSOM_SEGSUM
    const int NG = finalize_groups(M);
    const unsigned col_blocks = (unsigned)((d + 2 + AT - 1) / AT < 8 ? (d + 2 + AT - 1) / AT : 8);
    hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)M, (unsigned)NG, col_blocks), dim3(AT), 0, s, w.slab, di, Mi,
                       w.count, w.chunk_pre, NG, w.gslab, sums, (const int32_t *)status,
                       status_behind_sums ? sums + (size_t)M * (d + 3) : (double *)nullptr);
    if (NG > 1)
        hipLaunchKernelGGL(finalize_groups_kernel, dim3((unsigned)M), dim3(AT), 0, s, w.gslab, di, Mi, NG,
                           sums);
    return launch_status("accumulate kernels");
}

int launch_accumulate(const void *X, int x_dtype, int64_t N, int64_t d, int64_t ldx,
                      const int64_t *idx, const double *kw, const double *dist, int64_t M,
                      double *sums, int32_t *status, void *ws, size_t ws_bytes, hipStream_t s) {
    DBGSOM_REQUIRE(N == 0 || kw, "null sample weights");
    return accumulate_impl(X, x_dtype, N, d, ldx, idx, kw, 0.0, dist, M, sums, status, false, ws, ws_bytes, s);
}

int launch_accumulate_epoch(const void *X, int x_dtype, int64_t N, int64_t d, int64_t ldx,
                            const int64_t *idx, double gamma, const double *dist, int64_t M,
                            double *sums, int32_t *status, void *ws, size_t ws_bytes, hipStream_t s,
                            const DistFill *fill) {
    return accumulate_impl(X, x_dtype, N, d, ldx, idx, nullptr, gamma, dist, M, sums, status, true, ws, ws_bytes, s, fill);
}

}  // namespace dbgsom
