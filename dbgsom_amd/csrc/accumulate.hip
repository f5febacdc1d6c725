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
#include <type_traits>

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

// (non-temporal: every row is read exactly once by this kernel, 16 bytes per lane and whole cache lines per
//  wavefront instruction -- streamed past the caches, segsum_kernel's 3.4 GB at C4 take 0.57 instead of 0.64 ms)
template <typename XT, int VEC>
__device__ __forceinline__ void load_vec(const XT *__restrict__ src, double (&v)[VEC]) {
    if constexpr (sizeof(XT) == 4 && VEC == 4) {
        typedef float f4_t __attribute__((ext_vector_type(4)));
        const f4_t t4 = __builtin_nontemporal_load(reinterpret_cast<const f4_t *>(src));
        v[0] = t4.x; v[1] = t4.y; v[2] = t4.z; v[3] = t4.w;
    } else if constexpr (sizeof(XT) == 8 && VEC == 2) {
        typedef double d2_t __attribute__((ext_vector_type(2)));
        const d2_t t2 = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(src));
        v[0] = t2.x; v[1] = t2.y;
    } else if constexpr (sizeof(XT) == 2 && VEC == 8) {
        typedef unsigned u4_t __attribute__((ext_vector_type(4)));
        const u4_t a = __builtin_nontemporal_load(reinterpret_cast<const u4_t *>(src));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] = (double)__uint_as_float(a[e] << 16);
            v[2 * e + 1] = (double)__uint_as_float(a[e] & 0xffff0000u);
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
// against that one prototype -- against THIS chunk's prototype.  Such rows arrive with dist == -1.  The float
// rows are streamed ONCE for the distance and the sums (pair kernel + segsum_kernel: twice):
//
//  * a chunk is walked in slices of SR = 16 rows; every thread loads its column group (16 bytes) of the
//    slice's rows into REGISTERS -- as in segsum_kernel, 16 loads in flight per thread -- and keeps them;
//  * the chain acc = fma(w_k, x_k, acc), k ascending, runs on the matrix cores of ONE wavefront:
//    v_mfma_f64_4x4x4_4b (four 4x4x4 blocks, 16 B-columns per instruction = the slice's 16 rows against the
//    chunk's prototype in every A row) is that chain bit for bit, like the 16x16x4 form of the exact kernels,
//    at a quarter of its pipe time (tools/probe_mfma_f64_4x4.hip: 17.6 cycles per instruction, 43.6 cycles from
//    one dependent instruction to the next -- 26 without the loop branch, 51 with B read from LDS and widened in
//    front of it, tools/probe_mfma_chain.hip).  The operands reach it through LDS in ranges of 64 column groups:
//    the threads of range r + 1 write their registers (raw 16-byte pieces, row pitch 65 pieces: no bank
//    conflicts on either side) while the chain walks range r; one barrier per range.  (Ranges of values already
//    widened to float64 by their owners were measured: twice the LDS bytes and a second conversion of every
//    value -- slower, C4 1.27 against 1.12 ms, the C5 shard 2.33 against 1.38.);
//  * the distances and the sample kernel follow, then all threads form the weighted sums FROM THEIR REGISTERS
//    in list order.  Sums, their order of additions and the scalar partials are those of segsum_kernel bit
//    for bit (the two search forms must leave bit-identical prototypes).
// A slice none of whose rows needs a distance skips the chain altogether.
// Stamps and experiment switches of segsum_chain_kernel live in experiments.h, which only the experiment builds of
// tools/build_variant.sh compile (-DDBGSOM_EXPERIMENTS ...); the shipped kernel carries four empty macros.
#ifdef DBGSOM_EXPERIMENTS
#include "experiments.h"
#else
#define CSTAMP_DECL
#define CSTAMP(k)
#define CSTAMP_LOADS_LANDED
#define CSTAMP_FLUSH(dist, rows_s, n)
#define CHAIN_PAD 0
#endif
#ifndef CHAIN_CB
#define CHAIN_CB 4        // k-steps per batch of the chain's operand ring
#endif
#ifndef CHAIN_CW
#define CHAIN_CW ((c >> 8) & 3u)
#endif
#ifndef CHAIN_OCC
#define CHAIN_OCC 3       // wavefronts per SIMD the one-group kernel is compiled for
#endif
constexpr int SR = 16;     // rows per slice = B-columns of one matrix instruction
constexpr int SRG = 64;    // column groups (16-byte pieces) per range
constexpr int SRP = (SRG + 1) * 16;   // bytes per row of a range buffer (one piece of padding)

struct ChainLds {   // byte offsets inside the one dynamic LDS object (see refine.h: one object, no alias waits)
    int o_w, o_rows, o_kw, o_dist, o_xx, o_info, total;   // (the two range buffers sit at 0; `red` reuses them)
};
static ChainLds chain_lds(int64_t d) {
    ChainLds l;
    l.o_w = 2 * SR * SRP;
    l.o_rows = l.o_w + (int)d * 8;
    l.o_kw = l.o_rows + CH * 4;
    l.o_dist = l.o_kw + CH * 8;
    l.o_xx = l.o_dist + CH * 8;
    l.o_info = l.o_xx + CH * 8;
    l.total = l.o_info + 32;
    return l;
}

typedef unsigned raw4_t __attribute__((ext_vector_type(4)));
template <typename XT, int VEC>
__device__ __forceinline__ void raw_vec(const raw4_t r, double (&v)[VEC]) {
    if constexpr (sizeof(XT) == 4 && VEC == 4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (double)__uint_as_float(r[e]);
    } else if constexpr (sizeof(XT) == 8 && VEC == 2) {
        v[0] = __hiloint2double((int)r[1], (int)r[0]);
        v[1] = __hiloint2double((int)r[3], (int)r[2]);
    } else {
        static_assert(sizeof(XT) == 2 && VEC == 8, "unsupported vector width");
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] = (double)__uint_as_float(r[e] << 16);
            v[2 * e + 1] = (double)__uint_as_float(r[e] & 0xffff0000u);
        }
    }
}

// G = column groups per thread (rows of more than AT groups: q = tid + g AT)
template <typename XT, int VEC, int G>
__global__ __launch_bounds__(AT, G == 1 ? CHAIN_OCC : 2) void segsum_chain_kernel(
    const XT *__restrict__ X, int d, int64_t ldx, const int32_t *__restrict__ order, double gamma,
    double *__restrict__ dist, const uint32_t *__restrict__ seg_start, const uint32_t *__restrict__ count,
    const uint32_t *__restrict__ chunk_pre, int M, double *__restrict__ slab, const double *__restrict__ W,
    const double *__restrict__ ww, const double *__restrict__ xx, int round_f32, ChainLds L) {
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    double *w_s = reinterpret_cast<double *>(dyn + L.o_w);
    int32_t *rows_s = reinterpret_cast<int32_t *>(dyn + L.o_rows);
    double *kw_s = reinterpret_cast<double *>(dyn + L.o_kw), *dist_s = reinterpret_cast<double *>(dyn + L.o_dist);
    double *xx_s = reinterpret_cast<double *>(dyn + L.o_xx);
    double *red = reinterpret_cast<double *>(dyn);
    uint32_t *info = reinterpret_cast<uint32_t *>(dyn + L.o_info);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t c = blockIdx.x;
    if (c >= chunk_pre[M]) return;  // uniform per workgroup
    CSTAMP_DECL;
    {   // the chunk's neuron: the one j with chunk_pre[j] <= c < chunk_pre[j + 1] -- every thread looks at its share of
        // the table at once (one round trip instead of the ten of a binary search by one thread)
        const int per = (M + AT - 1) / AT;
        for (int u = 0; u < per; ++u) {
            const int jj = tid * per + u;
            if (jj < M && chunk_pre[jj] <= c && c < chunk_pre[jj + 1]) {
                const uint32_t begin = seg_start[jj] + (c - chunk_pre[jj]) * CH;
                const uint32_t end = min(begin + (uint32_t)CH, seg_start[jj] + count[jj]);
                info[0] = begin; info[1] = end - begin; info[2] = (uint32_t)jj;
            }
        }
        if (tid == 0) info[3] = 0u;
    }
    __syncthreads();
    const uint32_t begin = info[0];
    const int n = (int)info[1], j = (int)info[2];
    if (tid < n) {
        const int32_t r = order[begin + tid];
        rows_s[tid] = r;
        const double dd = dist[r];   // (-1: to be computed here)
        dist_s[tid] = dd;
        xx_s[tid] = xx[r];
        // the sample kernel of BaseSom._calculate_exp_similarity (BaseSom.py:533-538), the arithmetic of
        // exp_similarity_kernel (bmu.hip) and of segsum_kernel
        kw_s[tid] = 1.0 - sqrt(1.0 - exp(-gamma * (dd * dd)));
        if (dd == -1.0) atomicOr(&info[3], 1u << (tid / SR));   // slices with a row that needs its distance
    }
    for (int k = tid; k < d; k += AT) w_s[k] = W[(size_t)j * d + k];
    const double ww_j = ww[j];
    __syncthreads();
    const uint32_t need_mask = info[3];
    const int Q = d / VEC;                       // column groups (VEC divides d by construction)
    const bool wide = Q >= AT;
    const int RL = wide ? 1 : AT / Q;            // row lanes side by side on the same column group (segsum_kernel)
    const int rl = wide ? 0 : tid / Q, q0 = wide ? tid : tid - rl * Q;
    const bool active = wide || rl < RL;
    const int NR = (Q + SRG - 1) / SRG;          // ranges of 64 column groups
    const int cw = (int)(CHAIN_CW);             // the chain's wavefront: spread over the SIMDs (chunks resident on one CU
                                                 // share c mod 256 -- XCD, then CU, round robin -- so the bits above decide)
    const int lr = lane & 15, lq = lane >> 4;    // B operand: column (row of the slice) lr, k = lq
    double acc[G][VEC];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[g][e] = 0.0;
    double *out = slab + (size_t)c * (d + 2);
    int my_range[G], my_off[G];   // where this thread's column groups go in the range buffers
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int q = q0 + g * AT;
        my_range[g] = (active && q < Q) ? q / SRG : -1;
        my_off[g] = (q % SRG) * 16;
    }
    CSTAMP(0);

    for (int s0 = 0; s0 < n; s0 += SR) {
        const int nrow = min(SR, n - s0);
        // ---- the slice's rows into registers (a thread's row lane: rows p = rl, rl + RL, ... of the chunk)
        raw4_t raw[SR][G];
        // bit i: row s0 + i is one of this thread's (its row lane: rows p = rl, rl + RL, ... of the chunk)
        uint32_t mine = 0u;
        if (active) {
            if (RL == 1) mine = 0xffffu;
            else for (int i = (rl - s0 % RL + RL) % RL; i < SR; i += RL) mine |= 1u << i;
            mine &= (1u << nrow) - 1u;
        }
#pragma unroll
        for (int i = 0; i < SR; ++i) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                raw[i][g] = raw4_t{0u, 0u, 0u, 0u};
                if (((mine >> i) & 1u) && my_range[g] >= 0)
                    raw[i][g] = *reinterpret_cast<const raw4_t *>(X + (int64_t)rows_s[s0 + i] * ldx + (int64_t)(q0 + g * AT) * VEC);
            }
        }
        CSTAMP(1);
        CSTAMP_LOADS_LANDED;
        if ((need_mask >> (s0 / SR)) & 1u) {   // (uniform)
            // ---- the chain: ranges of 64 column groups through two LDS buffers
            // (rows behind the slice's end: zeros, written by row lane 0 -- never garbage in a B column)
            const uint32_t wmask = mine | ((active && rl == 0) ? (0xffffu & ~((1u << nrow) - 1u)) : 0u);
            auto write_range = [&](int r) {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    if (my_range[g] == r) {
                        char *dst = dyn + (r & 1) * (SR * SRP) + my_off[g];
#pragma unroll
                        for (int i = 0; i < SR; ++i) {
                            if ((wmask >> i) & 1u) *reinterpret_cast<raw4_t *>(dst + i * SRP) = raw[i][g];
                        }
                    }
                }
            };
            double a4 = 0.0;   // the chain's accumulator: lanes 0 .. 15 = rows of the slice
            write_range(0);
            __syncthreads();
            CSTAMP(3);
            for (int r = 0; r < NR; ++r) {
                if (r + 1 < NR) write_range(r + 1);
                if (wave == cw) {
                    const char *buf = dyn + (r & 1) * (SR * SRP) + lr * SRP + lq * (int)sizeof(XT);   // B: row lr, feature k0 + lq
                    const int nstep = min(SRG, Q - r * SRG) * VEC / 4;                 // k-steps of 4 features in this range
                    const double *wk = w_s + (size_t)r * SRG * VEC + lq;               // A: w[k0 + lq] in every A row
                    auto ld_a = [&](int st) { return wk[4 * st]; };
                    // (B stays as it was read until its matrix instruction: widened at the point of the read, the
                    //  conversion -- and with it the wait for the NEXT batch's reads -- would sit in front of this
                    //  batch's instructions)
                    typedef typename std::conditional<sizeof(XT) == 8, double, unsigned>::type braw_t;
                    auto ld_b = [&](int st) -> braw_t {
                        if constexpr (sizeof(XT) == 4) return *reinterpret_cast<const unsigned *>(buf + st * 16);
                        else if constexpr (sizeof(XT) == 8) return *reinterpret_cast<const double *>(buf + st * 32);
                        else return (unsigned)*reinterpret_cast<const unsigned short *>(buf + st * 8);
                    };
                    auto wid = [&](braw_t v) -> double {
                        if constexpr (sizeof(XT) == 4) return (double)__uint_as_float(v);
                        else if constexpr (sizeof(XT) == 8) return v;
                        else return (double)__uint_as_float(v << 16);
                    };
                    // double batches of 2 CB steps: the next batch's LDS reads are in flight under this batch's dependent
                    // matrix instructions.  Every read of the loop is unconditional (the last one runs a batch past the
                    // range: still inside this workgroup's LDS, never used): with reads under a condition the compiler's
                    // wait counts merge the two paths and every batch waits for the reads just issued (71 cycles per
                    // step instead of ~55)
                    constexpr int CB = CHAIN_CB;
                    const int npair = nstep / (2 * CB);
                    if (npair) {
                        double a0[CB], a1[CB];
                        braw_t b0[CB], b1[CB];
#pragma unroll
                        for (int u = 0; u < CB; ++u) { a0[u] = ld_a(u); b0[u] = ld_b(u); }
                        for (int it = 0, st = 0; it < npair; ++it, st += 2 * CB) {
#pragma unroll
                            for (int u = 0; u < CB; ++u) { a1[u] = ld_a(st + CB + u); b1[u] = ld_b(st + CB + u); }
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int u = 0; u < CB; ++u) a4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a0[u], wid(b0[u]), a4, 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int u = 0; u < CB; ++u) { a0[u] = ld_a(st + 2 * CB + u); b0[u] = ld_b(st + 2 * CB + u); }
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int u = 0; u < CB; ++u) a4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a1[u], wid(b1[u]), a4, 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    for (int st = npair * 2 * CB; st < nstep; ++st)
                        a4 = __builtin_amdgcn_mfma_f64_4x4x4f64(ld_a(st), wid(ld_b(st)), a4, 0, 0, 0);
                    CSTAMP(8);   // (the chain wave's own view: its matrix loop ...)
                }
                __syncthreads();
                if (wave == cw) { CSTAMP(9); }   // (... and its wait at the range's barrier, writes of the next range included)
            }
            CSTAMP(4);
            if (wave == cw && lane < nrow) {
                const int p = s0 + lane;
                if (dist_s[p] == -1.0) {
                    double rv = (xx_s[p] + (-2.0 * a4)) + ww_j;
                    if (!(rv > 0.0)) rv = (rv != rv) ? rv : 0.0;
                    double dv = sqrt(rv);
                    if (round_f32) dv = (double)(float)dv;
                    dist_s[p] = dv;
                    dist[rows_s[p]] = dv;
                    kw_s[p] = 1.0 - sqrt(1.0 - exp(-gamma * (dv * dv)));
                }
            }
            __syncthreads();
            CSTAMP(5);
        }
        // ---- the weighted sums of the slice's rows, in list order (segsum_kernel's order), from the registers
        if (mine) {
#pragma unroll
            for (int i = 0; i < SR; ++i) {
                if ((mine >> i) & 1u) {
                    const double w = kw_s[s0 + i];
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        double v[VEC];
                        // (laundered: the compiler must not keep the values widened for the range buffers alive until
                        //  here -- 16 rows of them are 128 registers -- but widen the raw pieces once more)
                        unsigned r0 = raw[i][g][0], r1 = raw[i][g][1], r2 = raw[i][g][2], r3 = raw[i][g][3];
                        asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));
                        raw_vec<XT, VEC>(raw4_t{r0, r1, r2, r3}, v);
#pragma unroll
                        for (int e = 0; e < VEC; ++e) acc[g][e] += w * v[e];
                    }
                }
            }
        }
        CSTAMP(6);
    }
    __syncthreads();
    CSTAMP_FLUSH(dist, rows_s, n);
    if (tid == AT - 1) {  // the scalar partials, in list order
        double sk = 0.0, se = 0.0;
        for (int p = 0; p < n; ++p) { sk += kw_s[p]; se += dist_s[p]; }
        out[d] = sk;
        out[d + 1] = se;
    }
    if (wide) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
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
    if (d / vec > 2 * (int64_t)AT) return false;              // (two column groups per thread at most)
    return chain_lds(d).total <= 96 * 1024 && (size_t)AT * vec * 8 <= (size_t)2 * SR * SRP;   // (`red` reuses the range buffers)
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
        const ChainLds L = chain_lds(d);
        const bool two = d / (x_dtype == DBGSOM_F32 ? 4 : (x_dtype == DBGSOM_F64 ? 2 : 8)) > AT;
#define DBGSOM_SEGDIST(XT, V, G_)                                                                         \
    do {                                                                                                  \
        static int attr_set = 0;                                                                          \
        if (attr_set < L.total) {                                                                         \
            DBGSOM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&segsum_chain_kernel<XT, V, G_>), \
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024 + CHAIN_PAD)); \
            attr_set = 96 * 1024;                                                                         \
        }                                                                                                 \
        hipLaunchKernelGGL((segsum_chain_kernel<XT, V, G_>), grid, block, (size_t)L.total + CHAIN_PAD, s, (const XT *)X, di, ldx, \
                           w.order, gamma, const_cast<double *>(dist), w.seg_start, w.count, w.chunk_pre, Mi, \
                           w.slab, fill->W, fill->ww, fill->xx, fill->round_f32, L);                      \
    } while (0)
        if (x_dtype == DBGSOM_F32) { if (two) DBGSOM_SEGDIST(float, 4, 2); else DBGSOM_SEGDIST(float, 4, 1); }
        else if (x_dtype == DBGSOM_F64) { if (two) DBGSOM_SEGDIST(double, 2, 2); else DBGSOM_SEGDIST(double, 2, 1); }
        else { if (two) DBGSOM_SEGDIST(bf16_t, 8, 2); else DBGSOM_SEGDIST(bf16_t, 8, 1); }
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
