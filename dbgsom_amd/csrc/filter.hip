// Filtered BMU search: an int8-MFMA sweep decides, with a rigorous error bound, which prototypes
// can possibly be a sample's best matching unit; only those pairs are then evaluated exactly
// (float64 chain, same arithmetic as bmu.hip / oracle/bmu_chain.c).  Results are IDENTICAL to the
// all-pairs float64 search -- the filter only removes pairs that provably cannot win.
//
// Replaces the same reference step as bmu.hip: BaseSom._get_winning_neurons (BaseSom.py:446-464).
//
// 1. slice_rows_kernel: every row a (samples once per fit, prototypes once per epoch) is scaled by
//    s = max|a_k| and quantised to Q_k = rint(a_k / s * F), F = 127 * 2^16, then split into three
//    balanced base-256 digits Q = D0 * 2^16 + D1 * 2^8 + D2, D in [-128, 127], stored as int8
//    planes.  |a_k - s Q_k / F| <= s / (2F).
// 2. sweep_i8_kernel: v_mfma_i32_32x32x32_i8 forms the six digit products with a + b <= 2,
//    accumulated EXACTLY in int32 per level L = a + b; T = P0 2^16 + P1 2^8 + P2 and
//       r~_ij = (|x_i|^2 + |w_j|^2) - 2 s_i t_j 2^16 T / F^2 ,   |r~_ij - r_ij| <= eps_i
//    with eps_i from the quantisation and the dropped (a + b >= 3) products (formula at
//    filter_eps below).  Samples are visited in the order of their PREVIOUS winner (the stable
//    bucket order dbgsom_accumulate produced), so a 128-sample workgroup shares its candidates;
//    prototype j is marked for the workgroup when r~_ij <= thr_i = r~_{i,prev(i)} + 2 eps_i for any
//    of its samples i (every j with r_ij <= r_{i,prev(i)} satisfies this, in particular the winner
//    and everything tied with it).
// 3. subset_exact_kernel (bmu_dma-style f64 MFMA on gathered rows): exact arg-min of each sample
//    over its workgroup's marked prototypes = exact arg-min over all prototypes.
#include <math.h>
#include <type_traits>

#include "bmu_common.h"

// (in-kernel stamps of the experiment builds: csrc/experiments.h, tools/build_variant.sh; empty in the library)
#ifdef DBGSOM_EXPERIMENTS
#include "experiments.h"
#else
#define PM_STAMP(k)
#define XT_DECL
#define XT_MARK(k)
#define XT_MARK_ONCE(k)
#define XT_FLUSH(JTL_, cnt_, dist_, isamp0_, Kk_)
#endif

namespace dbgsom {

constexpr double FQ = 8323072.0;  // 127 * 2^16
constexpr double F16 = 32512.0;   // 127 * 2^8: scale of the top two digit planes taken as one 16-bit digit
constexpr int FKT = 64;           // bytes (= features) per plane row per LDS stage
constexpr int FNT = 512;          // threads per sweep workgroup (8 wavefronts)
constexpr int FSTAGES = 3;
constexpr int PREPASS_KTILES = 3;  // k-tiles the seed pre-pass samples (tile_select_kernel picks them)
constexpr int SCHED_BINS = 16;     // launch-order bins of the exact stage (section 2b)
constexpr int SCHED_RETRY = 2 * SCHED_BINS + 10;  // u64: workgroups of the pruning form whose lists came out long (re-seeded, 2c)
constexpr int SCHED_CTR = 2 * SCHED_BINS + 12;  // bin counts | cursors | [start, n] of classes 3, 2, 1 | sum of list lengths (u64) | the same of a counting-only pruning launch (u64)
constexpr int SCHED_SUM = 2 * SCHED_BINS + 6;  // (8-byte aligned: the counters sit on a 256-byte boundary)
constexpr int RF_CTR = 4;  // u64 counters of the refinement (2d): pairs | refined workgroups | left to the MFMA stage | -
static_assert(SCHED_CTR % 2 == 0, "the 64-bit counters of the refinement sit behind the schedule's");
constexpr int SW_MAX_KT = 1024;   // k-tiles that selection handles (d <= 65536)

// plane rows are padded to whole k-tiles, at least two of them (the sweep's ring runs three tiles
// ahead and keeps three chunk tables)
inline int64_t filter_dpad(int64_t d) {
    // (a 128-byte pitch -- whole cache lines per row -- was measured in round 3: prune_mark_kernel fetched
    //  the same 1.25 GB at d = 784 either way, and the seventh part more plane cost the pre-pass and the
    //  pruning pass 8 % each)
    const int64_t p = (d + FKT - 1) / FKT * FKT;
    return p < 2 * FKT ? 2 * FKT : p;
}

// ---- 1. digit planes --------------------------------------------------------------------------
// Upper bound of |a - a16| (Euclidean) from the float64 sum r2 of the squared residuals as evaluated:
// each residual a_k - s q_k / F16 is formed with an absolute error of at most 3 u max|a| (product,
// quotient, difference), the d squares and their sum with a relative one of (d + 2) u; a row with a
// NaN or an infinity gives a NaN ("no bound": every candidate is kept)
__device__ __forceinline__ double plane16_residual(double r2, double s, double amax, int d) {
    (void)s;
    return sqrt(r2) * (1.0 + 1e-9 + 2.3e-16 * (double)d) + sqrt((double)d) * amax * 4e-16;
}

// planes: int8 [3][rows][dpad]; scale[rows] = s; l1[rows] = sum |a_k|
template <typename T>
__global__ __launch_bounds__(256) void slice_rows_kernel(const T *__restrict__ A, int64_t rows,
                                                         int d, int64_t ld, int dpad,
                                                         int8_t *__restrict__ planes,
                                                         double *__restrict__ scale,
                                                         double *__restrict__ l1,
                                                         double *__restrict__ res16) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T *a = A + row * ld;
    double m = 0.0, s1 = 0.0;
    for (int k = lane; k < d; k += 64) {
        const double v = fabs(widen(a[k]));
        m = fmax(m, v);
        s1 += v;
    }
    for (int off = 32; off > 0; off >>= 1) {
        m = fmax(m, __shfl_xor(m, off, 64));
        s1 += __shfl_xor(s1, off, 64);
    }
    const double s = (m > 0.0) ? m : 1.0;
    if (lane == 0) { scale[row] = s; l1[row] = s1; }
    const size_t plane_stride = (size_t)rows * dpad;
    int8_t *p0 = planes + (size_t)row * dpad;
    double r2 = 0.0;  // |a - a16|^2, a16 = s (256 D0 + D1) / F16: the row the top TWO planes stand for (2d)
    for (int k = lane; k < dpad; k += 64) {
        int q = 0;
        const double av = k < d ? widen(a[k]) : 0.0;
        if (k < d) q = (int)rint(av / s * FQ);
        const int d2 = ((q + 128) & 255) - 128;
        const int q1 = (q - d2) >> 8;
        const int d1 = ((q1 + 128) & 255) - 128;
        const int d0 = (q1 - d1) >> 8;
        p0[k] = (int8_t)d0;
        p0[plane_stride + k] = (int8_t)d1;
        p0[2 * plane_stride + k] = (int8_t)d2;
        const double t = av - s * (double)q1 / F16;
        r2 = fma(t, t, r2);
    }
    for (int off = 32; off > 0; off >>= 1) r2 += __shfl_xor(r2, off, 64);
    if (lane == 0) res16[row] = plane16_residual(r2, s, m, d);
}

// Which k-tiles (64 features each) the seed pre-pass looks at: the nkt_used tiles in which the
// prototypes differ most, score = sum over the tile's features of the variance across prototypes
// (the tile's share of the expected squared distance between two prototypes).  On isotropic data
// every choice is as good; on data whose information sits in a few feature ranges (images with
// constant borders, one-hot blocks) evenly spaced tiles can miss it and the seeds degrade to
// noise -- still exact, but with candidate lists as long as the map.
constexpr int TS_RB = 32;  // row blocks of the two-stage column sums below
__global__ __launch_bounds__(256) void tile_partial_kernel(const double *__restrict__ W, int M, int d,
                                                           int dpad, double *__restrict__ part) {
    // part[rb][0][k] = sum_j w_jk, part[rb][1][k] = sum_j w_jk^2 over the rows of block rb
    __shared__ double s1[256], s2[256];
    const int kt = blockIdx.x, rb = blockIdx.y, f = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int k = kt * FKT + f;
    const int per = (M + TS_RB - 1) / TS_RB, j0 = rb * per, j1 = min(M, j0 + per);
    double a = 0.0, b = 0.0;
    if (k < d)
        for (int j = j0 + g; j < j1; j += 4) {
            const double v = W[(size_t)j * d + k];
            a += v;
            b += v * v;
        }
    s1[threadIdx.x] = a;
    s2[threadIdx.x] = b;
    __syncthreads();
    if (g == 0) {
        part[((size_t)rb * 2 + 0) * dpad + k] = (s1[f] + s1[f + 64]) + (s1[f + 128] + s1[f + 192]);
        part[((size_t)rb * 2 + 1) * dpad + k] = (s2[f] + s2[f + 64]) + (s2[f + 128] + s2[f + 192]);
    }
}

// score[kt] per k-tile, then -- in the workgroup that finishes last -- kt_sel[0 .. nkt_used) = the
// nkt_used best tiles, ascending (ties: lower tile first); one wavefront per tile
__global__ __launch_bounds__(64) void tile_score_select_kernel(const double *__restrict__ part, int M, int dpad,
                                                               double *__restrict__ score, int nkt,
                                                               int nkt_used, int32_t *__restrict__ kt_sel,
                                                               uint32_t *__restrict__ ticket) {
    const int kt = blockIdx.x, k = kt * FKT + threadIdx.x;
    double a = 0.0, b = 0.0;
    for (int rb = 0; rb < TS_RB; ++rb) {
        a += part[((size_t)rb * 2 + 0) * dpad + k];
        b += part[((size_t)rb * 2 + 1) * dpad + k];
    }
    double var = b - a * a / (double)M;  // M x variance of feature k over the prototypes
    var = var > 0.0 ? var : 0.0;
    for (int off = 32; off > 0; off >>= 1) var += __shfl_xor(var, off, 64);
    if (threadIdx.x == 0) score[kt] = var;
    if (!last_workgroup_done(ticket, (uint32_t)nkt)) return;
    __shared__ int taken[SW_MAX_KT];
    const int lane = threadIdx.x;
    for (int t = lane; t < nkt; t += 64) taken[t] = 0;
    __syncthreads();
    for (int u = 0; u < nkt_used; ++u) {
        double best = -1.0;
        int bt = 0x7fffffff;
        for (int t = lane; t < nkt; t += 64)
            if (!taken[t] && (score[t] > best)) { best = score[t]; bt = t; }
        for (int off = 32; off > 0; off >>= 1) {
            const double ob = __shfl_xor(best, off, 64);
            const int ot = __shfl_xor(bt, off, 64);
            if (ob > best || (ob == best && ot < bt)) { best = ob; bt = ot; }
        }
        if (lane == 0) taken[bt] = 1;
        __syncthreads();
    }
    if (lane == 0) {
        int u = 0;
        for (int t = 0; t < nkt; ++t)
            if (taken[t]) kt_sel[u++] = t;
    }
}

// per-prototype tables of the sweep, padded with zeros to Mpad (a multiple of 128) entries:
// ctab_j = 2 t_j 2^16 / F^2 (so r~ = (xx+yy) - s_i ctab_j T), yypad_j = |w_j|^2;
// summary[0] = max_j l1_j, summary[1] = max_j t_j, summary[2] = max_j |w_j|^2
// (ctab_sub, yy_sub: the same for every `stride`-th prototype -- the seed pre-pass;
//  ictab, yctab: 1 / (ctab tscale) and |w|^2 / (ctab tscale), the form the marking test uses)
// float32 image of a table entry for the float32 chunk epilogue: exact up to a relative 2^-24 when
// the value is a normal float32 (or 0); anything else has no bounded relative error and the entry
// is made "always marked" by the caller
__device__ __forceinline__ bool f32_representable(double v) {
    const double a = fabs(v);
    return v == 0.0 || (a >= 1.1754943508222875e-38 && a <= 3.4028234663852886e+38);
}

struct WTables {
    const double *ww;
    float *tab32, *chk32;
    double *ctab, *yypad, *ctab_sub, *yy_sub, *ictab, *yctab, *summary;
    uint32_t *sched_ctr;
    double tscale;
};
template <int NTHR>
__device__ __forceinline__ void wtables_body(const double *__restrict__ tw, const double *__restrict__ l1w,
                                             const double *__restrict__ yy_part, const double *__restrict__ res16,
                                             int M, int Mpad, int stride, const WTables &o) {
    __shared__ double r0[NTHR], r1[NTHR], r2[NTHR], r3[NTHR];
    const int t = threadIdx.x;
    if (t < SCHED_CTR + 2 * RF_CTR + 4) o.sched_ctr[t] = 0u;  // bin counts / cursors / ranges of the exact stage's schedule, counters of 2d
    double a = 0.0, b = 0.0, c = 0.0, e = 0.0;
    for (int j = t; j < Mpad; j += NTHR) {
        const long js = (long)j * stride;  // j-th entry of the strided tables
        const bool in_sub = js < M;
        const double csub = in_sub ? 2.0 * tw[js] * 65536.0 / (FQ * FQ) : 0.0;
        o.ctab_sub[j] = csub;
        o.yy_sub[j] = in_sub ? yy_part[j] : 0.0;
        // (the seed pre-pass only needs SOME seed: plain float32, no guard)
        o.tab32[2 * (size_t)Mpad + j] = in_sub ? (float)yy_part[j] : 0.f;
        o.tab32[3 * (size_t)Mpad + j] = (float)(csub * 65536.0);
        // the same plain form for EVERY prototype and the whole row: the re-seed pass of section 2c
        o.tab32[4 * (size_t)Mpad + j] = j < M ? (float)o.ww[j] : 0.f;
        o.tab32[5 * (size_t)Mpad + j] = j < M ? (float)(2.0 * tw[j] * 65536.0 / (FQ * FQ) * 65536.0) : 0.f;
        if (j >= M) {
            o.ctab[j] = 0.0; o.yypad[j] = 0.0; o.ictab[j] = 0.0; o.yctab[j] = 0.0;
            o.tab32[j] = 0.f; o.tab32[(size_t)Mpad + j] = 0.f;
            continue;
        }
        const double cj = 2.0 * tw[j] * 65536.0 / (FQ * FQ);
        o.ctab[j] = cj;
        o.yypad[j] = o.ww[j];
        // the marking test r~ <= thr in the form (xx - thr) / c' + |w|^2 / c' <= s T', c' = c tscale
        // (T' = T / tscale is what the sweep's accumulators give without the last shift); a
        // prototype too small for a finite reciprocal is simply always marked
        const double cs = cj * o.tscale;
        const bool ok = cs > 1e-280;
        o.ictab[j] = ok ? 1.0 / cs : 0.0;
        o.yctab[j] = ok ? o.ww[j] / cs : -INFINITY;
        const bool ok32 = ok && f32_representable(1.0 / cs) && f32_representable(o.ww[j] / cs);
        o.tab32[j] = ok32 ? (float)(o.ww[j] / cs) : -INFINITY;
        o.tab32[(size_t)Mpad + j] = ok32 ? (float)(1.0 / cs) : 0.f;
        a = fmax(a, l1w[j]); b = fmax(b, tw[j]); c = fmax(c, o.ww[j]);
        // (fmax drops a NaN: a prototype row with a NaN or an infinity must void the bound instead)
        e = (e != e || res16[j] != res16[j]) ? NAN : fmax(e, res16[j]);
    }
    r0[t] = a; r1[t] = b; r2[t] = c; r3[t] = e;
    __syncthreads();
    for (int w = NTHR / 2; w > 0; w >>= 1) {
        if (t < w) {
            r0[t] = fmax(r0[t], r0[t + w]); r1[t] = fmax(r1[t], r1[t + w]); r2[t] = fmax(r2[t], r2[t + w]);
            r3[t] = (r3[t] != r3[t] || r3[t + w] != r3[t + w]) ? NAN : fmax(r3[t], r3[t + w]);
        }
        __syncthreads();
    }
    if (t == 0) { o.summary[0] = r0[0]; o.summary[1] = r1[0]; o.summary[2] = r2[0]; o.summary[3] = r3[0]; }
    // per 256-prototype chunk: the smallest |w|^2 / c' and the smallest / largest 1 / c' of its float32
    // table entries -- the coarse integer test of the two-per-CU sweep's epilogue (a lower bound of
    // the per-pair threshold over the whole chunk).  Round u of the loop above handled chunk u, one
    // entry per thread (NTHR == 256): wavefront reduction, then the four wavefronts' results.
    static_assert(NTHR == 256, "one chunk of 256 table entries per round");
    __shared__ float cst[64][4][3];
    __syncthreads();   // this workgroup's table entries are visible to it
    for (int ch = 0; ch < Mpad / 256; ++ch) {
        const int j = ch * 256 + t;
        float ymin = INFINITY, cmin = INFINITY, cmax = -INFINITY;
        if (j < M) { ymin = o.tab32[j]; cmin = cmax = o.tab32[(size_t)Mpad + j]; }
        for (int off = 32; off > 0; off >>= 1) {
            ymin = fminf(ymin, __shfl_xor(ymin, off, 64));
            cmin = fminf(cmin, __shfl_xor(cmin, off, 64));
            cmax = fmaxf(cmax, __shfl_xor(cmax, off, 64));
        }
        if ((t & 63) == 0 && ch < 64) { cst[ch][t >> 6][0] = ymin; cst[ch][t >> 6][1] = cmin; cst[ch][t >> 6][2] = cmax; }
    }
    __syncthreads();
    if (t < Mpad / 256 && t < 64) {
        float ymin = cst[t][0][0], cmin = cst[t][0][1], cmax = cst[t][0][2];
        for (int w = 1; w < 4; ++w) {
            ymin = fminf(ymin, cst[t][w][0]); cmin = fminf(cmin, cst[t][w][1]); cmax = fmaxf(cmax, cst[t][w][2]);
        }
        o.chk32[4 * t + 0] = ymin; o.chk32[4 * t + 1] = cmin; o.chk32[4 * t + 2] = cmax; o.chk32[4 * t + 3] = 0.f;
    }
}

// The prototypes' digit planes in the order the sweep's DMA wants them: int8 [3][dpad / 64][rows_pad]
// [64], i.e. k-tile-major, so that the 64-byte pieces of 16 consecutive rows are ONE contiguous
// KiB (whole 128-byte lines per DMA instruction instead of 16 half lines), with the 16-byte
// chunks of a piece already XOR-swizzled by (row >> 2) & 3 the way the LDS image is read.
// wt: all M prototypes (rows_pad = Mpad); wt_sub: every `stride`-th (rows_pad = Msubpad), the
// seed pre-pass.  Pad rows are never initialised: the sweep masks j >= M.
__device__ __forceinline__ void slice_w_row(const double *__restrict__ W, int M, int d, int dpad, int Mpad,
                                            int stride, int Msubpad, int nkt_used,
                                            const int32_t *__restrict__ kt_sel, int8_t *__restrict__ wt,
                                            int8_t *__restrict__ wt_sub, double *__restrict__ scale,
                                            double *__restrict__ l1, double *__restrict__ yy_part,
                                            double *__restrict__ nrm0, double *__restrict__ res16, int row, int lane) {
    const double *a = W + (size_t)row * d;
    double m = 0.0, s1 = 0.0;
    for (int k = lane; k < d; k += 64) {
        const double v = fabs(a[k]);
        m = fmax(m, v);
        s1 += v;
    }
    for (int off = 32; off > 0; off >>= 1) {
        m = fmax(m, __shfl_xor(m, off, 64));
        s1 += __shfl_xor(s1, off, 64);
    }
    const double s = (m > 0.0) ? m : 1.0;
    if (lane == 0) { scale[row] = s; l1[row] = s1; }
    const size_t plane_stride = (size_t)Mpad * dpad, sub_stride = (size_t)Msubpad * dpad;
    const bool in_sub = (row % stride) == 0;
    const int q = row / stride;
    if (in_sub) {  // |w|^2 over the k-tiles the seed pre-pass looks at (prepass_tile below)
        double p2 = 0.0;
        for (int u = 0; u < nkt_used; ++u) {
            const int k = (kt_sel ? kt_sel[u] : u) * FKT + lane;  // no selection: all tiles
            if (k < d) p2 += a[k] * a[k];
        }
        for (int off = 32; off > 0; off >>= 1) p2 += __shfl_xor(p2, off, 64);
        if (lane == 0) yy_part[q] = p2;
    }
    long long n0 = 0;  // sum of the squared top digits (exact)
    double r2 = 0.0;   // squared residual of the top two planes (slice_rows_kernel)
    for (int k = lane; k < dpad; k += 64) {
        int v = 0;
        const double av = k < d ? a[k] : 0.0;
        if (k < d) v = (int)rint(av / s * FQ);
        const int d2 = ((v + 128) & 255) - 128;
        const int v1 = (v - d2) >> 8;
        const int d1 = ((v1 + 128) & 255) - 128;
        const int d0 = (v1 - d1) >> 8;
        n0 += (long long)(d0 * d0);
        const double tr = av - s * (double)v1 / F16;
        r2 = fma(tr, tr, r2);
        const int kt = k >> 6, c = (k >> 4) & 3, b = k & 15;
        const size_t o = ((size_t)kt * Mpad + row) * FKT + ((c ^ ((row >> 2) & 3)) << 4) + b;
        wt[o] = (int8_t)d0;
        wt[plane_stride + o] = (int8_t)d1;
        wt[2 * plane_stride + o] = (int8_t)d2;
        if (in_sub) {
            const size_t oq = ((size_t)kt * Msubpad + q) * FKT + ((c ^ ((q >> 2) & 3)) << 4) + b;
            wt_sub[oq] = (int8_t)d0;
            wt_sub[sub_stride + oq] = (int8_t)d1;
            wt_sub[2 * sub_stride + oq] = (int8_t)d2;
        }
    }
    for (int off = 32; off > 0; off >>= 1) { n0 += __shfl_xor(n0, off, 64); r2 += __shfl_xor(r2, off, 64); }
    if (lane == 0) { nrm0[row] = (double)n0; res16[row] = plane16_residual(r2, s, m, d); }
}


__global__ __launch_bounds__(256) void slice_w_tiled_kernel(const double *__restrict__ W, int M, int d,
                                                            int dpad, int Mpad, int stride,
                                                            int Msubpad, int nkt_used,
                                                            const int32_t *__restrict__ kt_sel,
                                                            int8_t *__restrict__ wt,
                                                            int8_t *__restrict__ wt_sub,
                                                            double *__restrict__ scale,
                                                            double *__restrict__ l1,
                                                            double *__restrict__ yy_part,
                                                            double *__restrict__ nrm0,
                                                            double *__restrict__ res16,
                                                            uint32_t *__restrict__ ticket, WTables tables) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row < M) slice_w_row(W, M, d, dpad, Mpad, stride, Msubpad, nkt_used, kt_sel, wt, wt_sub, scale, l1, yy_part,
                             nrm0, res16, row, lane);
    // the per-prototype tables of the sweep need every row's scale / l1 / partial norm: the
    // workgroup that finishes last builds them (one launch less than a kernel of their own)
    if (last_workgroup_done(ticket, gridDim.x)) wtables_body<256>(scale, l1, yy_part, res16, M, Mpad, stride, tables);
}


// Error bound of r~ for sample i against ANY prototype of this epoch, measured against the value
// the exact kernel computes (r_chain):
//   quantisation   |a_k - s Q_k / F| <= (s / F)(1/2 + 3 u F)  (the division and the product by F
//                  are rounded before rint)  ->  |x.w - x^.w^| <= [(s |w|_1 + t |x|_1) / (2F)
//                  + d s t / (4 F^2)] (1 + 1e-7)
//   dropped digit products (a + b >= 3): <= d * 128 * 128 * (2 * 256 + 1) * s t / F^2
//   float64 rounding of forming r~ and of the exact kernel's own fma chain (d u |x||w| each):
//                  <= 4 (d + 16) 2^-53 (|x|^2 + max |w|^2)
//   eps = 2 (quantisation + dropped) + rounding.
// A prototype j with r_chain(i, j) <= r_chain(i, prev) has r~_j <= r~_prev + 2 eps: the test the
// sweep applies.
// With two digit planes (products a + b <= 1) the dropped set also holds the three level-2
// products: d * 128 * 128 * (3 * 65536 + 2 * 256 + 1) * s t / F^2.
// T = sum over the kept digit products of P_level 2^(16 - 8 level), from the per-level int32
// accumulators (exact: |T| < 2^53); sweep_T_scaled = T / sweep_tscale(planes): what the marking
// test uses (the scale is folded into the reciprocal tables)
template <int PLANES>
__device__ __forceinline__ double sweep_T_scaled(int a0, int a1, int a2) {
    if constexpr (PLANES == 1) return (double)a0;
    else if constexpr (PLANES == 2) return (double)a0 * 256.0 + (double)a1;
    else return ((double)a0 * 256.0 + (double)a1) * 256.0 + (double)a2;
}
template <int PLANES>
__device__ __forceinline__ double sweep_T(int a0, int a1, int a2) {
    constexpr double scale = PLANES == 1 ? 65536.0 : (PLANES == 2 ? 256.0 : 1.0);
    return sweep_T_scaled<PLANES>(a0, a1, a2) * scale;
}
static double sweep_tscale(int planes) { return planes == 1 ? 65536.0 : (planes == 2 ? 256.0 : 1.0); }

__device__ __forceinline__ double filter_eps(double s, double l1x, double xx, double l1w_max,
                                             double t_max, double yy_max, int d, int planes) {
    const double quant = (s * l1w_max + t_max * l1x) / (2.0 * FQ) + (double)d * s * t_max / (4.0 * FQ * FQ);
    // dropped digit products per feature, in units of 128 * 128: levels 3, 4 / 2 .. 4 / 1 .. 4
    const double per_k = 16384.0 * (planes >= 3 ? 513.0
                                    : planes == 2 ? (3.0 * 65536.0 + 513.0)
                                                  : (2.0 * 16777216.0 + 3.0 * 65536.0 + 513.0));
    const double dropped = (double)d * per_k * s * t_max / (FQ * FQ);
    const double rounding = 4.0 * (double)(d + 16) * 1.1102230246251565e-16 * (xx + yy_max);
    return 2.0 * (quant + dropped) * (1.0 + 1e-7) + rounding;
}

// launch-order bin of a workgroup of the exact stage by the length of its candidate list (section 2b)
__device__ __forceinline__ int sched_bin(uint32_t c) {
    if (c <= 16u) return 15;
    if (c <= 32u) return 14;
    const int steps = (int)((c + 47u) / 48u);
    return steps >= 14 ? 0 : 14 - steps;
}

// extra slack of thr_i for the float32 chunk epilogue of the two-per-CU sweep (derivation there):
// 2.5e-7 >= 4 2^-24 / (1 - 4 2^-24); the |A_i| part is applied where A_i is converted
__device__ __forceinline__ double epilogue32_slack(double xx, double yy_max) {
    return 2.5e-7 * (xx + 2.1 * yy_max);
}

// ---- 2. the int8 sweep -------------------------------------------------------------------------
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v16i_t __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *gbl_ptr_t;

__device__ __forceinline__ void fdma16(const void *src, void *lds_dst) {
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)lds_dst, 16, 0, 0);
}

// the same for bytes that are read once (streamed sample rows): non-temporal, so that they do not push the
// prototype tiles every workgroup shares out of the L2
__device__ __forceinline__ void fdma16_nt(const void *src, void *lds_dst) {
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)lds_dst, 16, 0, 2);
}

// the same with an immediate offset, which the instruction adds to BOTH the global and the LDS address
template <int OFF>
__device__ __forceinline__ void fdma16_off(const void *src, void *lds_dst) {
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)lds_dst, 16, OFF, 0);
}

constexpr int SW_PLANE = 128 * FKT;  // 8 KB: 128 rows of one digit plane per stage
constexpr int SW_MAX_M = 16000;

// LDS map of one sweep instantiation: 3 ring stages of NPL X planes (128 rows) + NPL W planes
// (128 JT rows), then 3 x (|w|^2 | ctab) chunk tables, the thresholds, the seeds and the bitmask
template <int PLANES, int JT>
struct SweepLds {
    static constexpr int BJ = 128 * JT;                         // prototypes per chunk
    static constexpr int W_PLANE = SW_PLANE * JT;
    static constexpr int STAGE = PLANES * (SW_PLANE + W_PLANE);  // 48 KB for (3,1) and (2,2)
    static constexpr int TAB = BJ * 8;
    static constexpr int OFF_TAB = FSTAGES * STAGE;
    static constexpr int OFF_THR = OFF_TAB + 3 * 2 * TAB;
    static constexpr int OFF_PREV = OFF_THR + 128 * 8;
    static constexpr int OFF_MASK = OFF_PREV + 128 * 4;
    static constexpr int OFF_MISC = OFF_MASK + (SW_MAX_M + 31) / 32 * 4;
    static constexpr int BYTES = OFF_MISC + 16;
};

// MODE 0 (mark): the candidate sweep described above; samples in bucket order of `prev`,
//   candidate prototypes of every 128-sample workgroup written to ulist / ucount.
// MODE 1 (seed): the pre-pass when there is no previous winner to start from: natural sample
//   order (order == nullptr), every jstride-th prototype (M = their number, ww / ctab = strided
//   tables), output seed[i] = arg-min of r~ -- any index is a valid starting point for MODE 0, a
//   near-minimal one keeps its candidate lists short.
// PLANES = digit planes per operand: 3 -> six digit products (a + b <= 2), 2 -> three (a + b <= 1).
// JT = 32-prototype tiles per wavefront: workgroup tile = 128 samples x 128 JT prototypes
//   (8 wavefronts as 2 x 4, each 64 x 32 JT); JT = 2 halves the number of passes over the X planes.
template <int MODE, int PLANES, int JT>
__global__ __launch_bounds__(FNT, 2) void sweep_i8_kernel(
    const int8_t *__restrict__ xplanes, const double *__restrict__ sx,
    const double *__restrict__ l1x, const double *__restrict__ xx, int64_t N, int d, int dpad,
    const int8_t *__restrict__ wplanes, const double *__restrict__ ww,
    const double *__restrict__ ctab, const double *__restrict__ yraw,
    const double *__restrict__ craw, const double *__restrict__ summary, int M,
    const int64_t *__restrict__ prev, const int32_t *__restrict__ order,
    uint16_t *__restrict__ ulist, int ulist_stride, uint32_t *__restrict__ ucount,
    int64_t *__restrict__ seed, int jstride, int w_rows, int nkt_used,
    const int32_t *__restrict__ kt_sel, uint32_t *__restrict__ sched_ctr) {
    using L = SweepLds<PLANES, JT>;
    constexpr int NPL = PLANES, NLV = PLANES, BJ = L::BJ;
    constexpr int DMA_TILE = NPL * (1 + JT);  // DMA instructions per wave per tile
    __shared__ __attribute__((aligned(16))) char smem[L::BYTES];
    double *thr_s = reinterpret_cast<double *>(smem + L::OFF_THR);
    int *prev_s = reinterpret_cast<int *>(smem + L::OFF_PREV);
    uint32_t *mask = reinterpret_cast<uint32_t *>(smem + L::OFF_MASK);
    int *misc = reinterpret_cast<int *>(smem + L::OFF_MISC);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 2, wj = wave & 3;  // 2 (samples) x 4 (prototypes) wavefronts
    const int lc = lane & 31, lh = lane >> 5;
    const int64_t p0 = (int64_t)blockIdx.x * 128;  // first sorted position of this workgroup
    const int nwords = (M + 31) / 32;

    auto sample_at = [&](int64_t p) -> int64_t { return order ? (int64_t)order[p] : p; };
    // Every thread's own loads first (sample indices, then the per-sample constants that hang on
    // them), BEFORE the first workgroup barrier: with one workgroup per CU nothing else hides the
    // prologue, and taken one after the other its order -> prev / order -> (s, |x|_1, |x|^2) /
    // order -> row chains are five or six HBM round trips before the first DMA can go out.
    const int dr = 16 * wave + (lane >> 2), dcp = lane & 3;  // DMA: wave w loads X rows 16w..16w+15
    const int64_t xpos = (p0 + dr < N) ? (p0 + dr) : (N - 1);
    const int64_t i_dr = sample_at(xpos);
    int64_t i_il[2];
    double s_i[2], l1_i[2], xx_i[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int64_t p = p0 + wi * 64 + it * 32 + lc;
        i_il[it] = sample_at(p < N ? p : N - 1);
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        s_i[it] = sx[i_il[it]];
        l1_i[it] = MODE == 0 ? l1x[i_il[it]] : 0.0;
        xx_i[it] = MODE == 0 ? xx[i_il[it]] : 0.0;
    }
    int jlo = 0, jhi = -1;
    if constexpr (MODE == 0) {
        for (int w = tid; w < nwords; w += FNT) mask[w] = 0u;
        if (tid < 128) {
            const int64_t p = p0 + tid;
            int pj = -1;
            if (p < N) pj = (int)prev[sample_at(p)];
            prev_s[tid] = (pj >= 0 && pj < M) ? pj : -1;
            thr_s[tid] = (p < N) ? -INFINITY : INFINITY;  // A = |x|^2 - thr: no bound yet / padding never marks
        }
        __syncthreads();
        if (wave == 0) {  // range of the seeds: the sweep starts at the chunk holding the lowest
            const int a = prev_s[lane], b = prev_s[lane + 64];
            int lo = min(a >= 0 ? a : 0x7fffffff, b >= 0 ? b : 0x7fffffff), hi = max(a, b);
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) {
                lo = min(lo, __shfl_xor(lo, m, 64));
                hi = max(hi, __shfl_xor(hi, m, 64));
            }
            if (lane == 0) {
                misc[0] = (hi >= 0) ? lo : 0;
                misc[1] = hi;
            }
        }
        __syncthreads();
        jlo = __builtin_amdgcn_readfirstlane(misc[0]);
        jhi = __builtin_amdgcn_readfirstlane(misc[1]);
        // The seeds of a workgroup sit in one chunk, except where the sorted order crosses a chunk
        // border.  There the samples whose seed lies in a later chunk would sweep the first chunk(s)
        // without a bound (and mark all of them): give those samples their bound up front, from
        // the same exact integer products by v_dot4 (4 threads per sample, K split by 16-byte
        // chunks).  Rare path: a handful of workgroups per launch.
        if (jlo / BJ != jhi / BJ) {
            const int il = tid >> 2, q = tid & 3;
            const int64_t p = p0 + il;
            const int pj = prev_s[il];
            const bool need = p < N && pj >= 0 && pj / BJ != jlo / BJ;
            int a0 = 0, a1 = 0, a2 = 0;
            const int64_t i = sample_at(p < N ? p : N - 1);
            if (need) {
                const size_t xps = (size_t)N * dpad, wps = (size_t)w_rows * dpad;
                const int8_t *xr = xplanes + (size_t)i * dpad;
                const int wsw = (pj >> 2) & 3;
                for (int ch = q; ch < dpad / 16; ch += 4) {
                    const int8_t *wr = wplanes + ((size_t)(ch >> 2) * w_rows + pj) * FKT + (((ch & 3) ^ wsw) << 4);
                    v4i_t xv[PLANES], wv[PLANES];
#pragma unroll
                    for (int pl = 0; pl < PLANES; ++pl) {
                        xv[pl] = *reinterpret_cast<const v4i_t *>(xr + pl * xps + ch * 16);
                        wv[pl] = *reinterpret_cast<const v4i_t *>(wr + pl * wps);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        a0 = __builtin_amdgcn_sdot4(xv[0][e], wv[0][e], a0, false);
                        if constexpr (PLANES >= 2) {
                            a1 = __builtin_amdgcn_sdot4(xv[0][e], wv[PLANES >= 2 ? 1 : 0][e], a1, false);
                            a1 = __builtin_amdgcn_sdot4(xv[PLANES >= 2 ? 1 : 0][e], wv[0][e], a1, false);
                        }
                        if constexpr (PLANES == 3) {
                            constexpr int P1 = PLANES >= 2 ? 1 : 0, P2 = PLANES >= 3 ? 2 : 0;
                            a2 = __builtin_amdgcn_sdot4(xv[0][e], wv[P2][e], a2, false);
                            a2 = __builtin_amdgcn_sdot4(xv[P1][e], wv[P1][e], a2, false);
                            a2 = __builtin_amdgcn_sdot4(xv[P2][e], wv[0][e], a2, false);
                        }
                    }
                }
            }
#pragma unroll
            for (int m = 1; m <= 2; m <<= 1) {
                a0 += __shfl_xor(a0, m, 64);
                a1 += __shfl_xor(a1, m, 64);
                a2 += __shfl_xor(a2, m, 64);
            }
            if (need && q == 0) {
                const double T = sweep_T<PLANES>(a0, a1, a2);
                const double sv = sx[i], xv2 = xx[i];
                const double e2 = 2.0 * filter_eps(sv, l1x[i], xv2, summary[0], summary[1], summary[2], d, PLANES);
                thr_s[il] = (sv * (craw[pj] * T) - yraw[pj]) - e2;  // = |x|^2 - (r~_seed + 2 eps)
            }
            __syncthreads();
        }
    }

    // per-lane sample constants (2 samples: one per 32-column tile)
    // A_i = |x_i|^2 - thr_i with thr_i = r~(i, seed_i) + 2 eps_i the bound from the seed: the marking
    // test r~_ij <= thr_i only needs A_i, and A_i = s c T - |w_seed|^2 - 2 eps needs no |x_i|^2
    double eps2_i[2], A_i[2];
    const double l1w_max = summary[0], t_max = summary[1], yy_max = summary[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int il = wi * 64 + it * 32 + lc;
        if constexpr (MODE == 0) {
            eps2_i[it] = 2.0 * filter_eps(s_i[it], l1_i[it], xx_i[it], l1w_max, t_max, yy_max, d, PLANES);
            A_i[it] = thr_s[il];
        } else {
            eps2_i[it] = 0.0; A_i[it] = 0.0;
        }
    }
    double bestv[2] = {INFINITY, INFINITY};  // MODE 1: running arg-min of r~
    int bestj[2] = {0, 0};

    // ---- DMA sources: per plane, wave w loads X rows 16w..16w+15 and W rows 16w + 128u ..+15 ----
    const int dc = dcp ^ ((dr >> 2) & 3);  // source chunk for the linear LDS chunk (swizzle)
    const size_t xplane_stride = (size_t)N * dpad;
    const size_t wplane_stride = (size_t)w_rows * dpad;  // w_rows: padded rows of one W plane
    const int8_t *xsrc = xplanes + (size_t)i_dr * dpad + dc * 16;
    // MODE 1 may look at a sample of the k-tiles only (the nkt_used tiles of kt_sel): seeds need
    // not be good, only cheap -- see dbgsom_bmu_filtered
    const int nkt_full = dpad / FKT;  // >= 2 (filter_dpad)
    const int nkt = (MODE == 1 && nkt_used >= 2 && nkt_used < nkt_full) ? nkt_used : nkt_full;
    auto tile_of = [&](int kt) { return (MODE == 1 && nkt < nkt_full) ? (int)kt_sel[kt] : kt; };
    const int nchunk = (M + BJ - 1) / BJ;
    const int ntile = nkt * nchunk;
    const int c0 = jlo / BJ;  // the sweep starts at the chunk holding the seeds

    // issue side of the ring: tile i_t = (chunk i_chunk, k-tile i_kt) goes to stage offset i_stage;
    // the chunk tables ride with the first k-tile of their chunk (ring of 3, sequence i_cseq)
    int i_kt = 0, i_chunk = c0, i_cseq = 0, i_stage = 0;
    // the DMA_TILE = NPL (1 + JT) LDS-DMA instructions of a tile: X plane p (op p), then W rows
    // 128 u .. of plane p (op NPL + u NPL + p); ops [lo, hi) are issued, hi == DMA_TILE also
    // sends the chunk tables (first k-tile of a chunk) and advances the issue counters
    const uint32_t lane_off = 1024u * JT * wave + 16u * lane;  // this lane's 16 bytes of the wave's JT contiguous KiB
    auto issue_ops = [&](int lo, int hi) {
        char *stage = smem + i_stage;
        const int i_tile = tile_of(i_kt);
        const int k0 = i_tile * FKT, jc_t = i_chunk * BJ;
        // (the plane strides are made opaque here: hoisted out of the tile loop, "row + p stride"
        // per plane and operand costs the registers the fragments need)
        size_t xps = xplane_stride, wps = wplane_stride;
        asm volatile("" : "+s"(xps), "+s"(wps));
#pragma unroll
        for (int p = 0; p < NPL; ++p)
            if (p >= lo && p < hi) fdma16(xsrc + p * xps + k0, stage + p * SW_PLANE + 1024 * wave);
        // k-tile-major planes: 16 rows of a k-tile are one contiguous KiB; wave w takes the JT
        // consecutive KiB 16 (JT w + u) .. of the chunk -- one address and one LDS base per plane,
        // the KiB steps in the instruction's immediate offset (global and LDS side alike)
        const int8_t *wsrc = wplanes + ((size_t)i_tile * w_rows + jc_t) * FKT + lane_off;
#pragma unroll
        for (int u = 0; u < JT; ++u) {
#pragma unroll
            for (int p = 0; p < NPL; ++p)
                if (NPL + u * NPL + p >= lo && NPL + u * NPL + p < hi) {
                    const int8_t *src = wsrc + p * wps;
                    char *dst = stage + NPL * SW_PLANE + p * L::W_PLANE + 1024 * JT * wave;
                    if (u == 0) fdma16(src, dst);
                    else if (u == 1) fdma16_off<1024>(src, dst);
                    else if (u == 2) fdma16_off<2048>(src, dst);
                    else fdma16_off<3072>(src, dst);
                }
        }
        if (hi == DMA_TILE) {
            if (i_kt == 0) {  // this chunk's tables (1 KB pieces; the 8 waves cover them, twice for JT = 1)
                constexpr int PIECES = 2 * JT;  // |w|^2 pieces, then ctab pieces
                const int piece = wave % PIECES, half = piece % JT;
                const int j2 = jc_t + 128 * half + 2 * lane;  // tables are padded to a multiple of 256 entries
                char *tab = smem + L::OFF_TAB + i_cseq * 2 * L::TAB;
                if (piece >= JT) fdma16(ctab + j2, tab + L::TAB + 1024 * half);
                else fdma16(ww + j2, tab + 1024 * half);
            }
            i_stage = (i_stage == (FSTAGES - 1) * L::STAGE) ? 0 : i_stage + L::STAGE;
            if (++i_kt == nkt) {
                i_kt = 0;
                i_chunk = (i_chunk + 1 == nchunk) ? 0 : i_chunk + 1;
                i_cseq = (i_cseq == 2) ? 0 : i_cseq + 1;
            }
        }
    };

    // fragment read offsets inside a stage (bytes); chunk (2 ks + lh) ^ swz = (2 ks) ^ (lh ^ swz):
    // k-step 1 flips byte 32
    int xoff[2], woff[JT];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int r = wi * 64 + it * 32 + lc;
        xoff[it] = r * FKT + ((lh ^ ((r >> 2) & 3)) * 16);
    }
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
        const int r = wj * 32 * JT + jt * 32 + lc;
        woff[jt] = NPL * SW_PLANE + r * FKT + ((lh ^ ((r >> 2) & 3)) * 16);
    }
    struct Frags { v4i_t x[2][NPL], w[JT][NPL]; };
    auto load_frags = [&](int stage_off, int ks, Frags &f) {
        const char *stage = smem + stage_off;
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
                f.w[jt][p] = *reinterpret_cast<const v4i_t *>(stage + p * L::W_PLANE + (woff[jt] ^ (ks * 32)));
#pragma unroll
            for (int it = 0; it < 2; ++it)
                f.x[it][p] = *reinterpret_cast<const v4i_t *>(stage + p * SW_PLANE + (xoff[it] ^ (ks * 32)));
        }
    };
    auto touch_frags = [&](const Frags &f) {
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) asm volatile("" ::"v"(f.w[jt][p]));
#pragma unroll
            for (int it = 0; it < 2; ++it) asm volatile("" ::"v"(f.x[it][p]));
        }
    };
    v16i_t acc[JT][2][NLV];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int lv = 0; lv < NLV; ++lv)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[jt][it][lv][r] = 0;
    // digit products of one k-step, level = plane(x) + plane(w); prototypes = rows (A), samples =
    // cols (B).  `between(g)` runs after the g-th group of 2 JT products (DMA issue slots).
    auto products = [&](const Frags &f, auto between) {
        int g = 0;
#pragma unroll
        for (int lv = 0; lv < NLV; ++lv)
#pragma unroll
            for (int px = 0; px <= lv; ++px)
#pragma unroll
            for (int jh = 0; jh < JT; jh += 2) {  // groups of (at most) two prototype tiles x two sample tiles
#pragma unroll
                for (int jt = jh; jt < jh + 2 && jt < JT; ++jt)
#pragma unroll
                    for (int it = 0; it < 2; ++it)
                        acc[jt][it][lv] = __builtin_amdgcn_mfma_i32_32x32x32_i8(
                            f.w[jt][lv - px], f.x[it][px], acc[jt][it][lv], 0, 0, 0);
                between(g++);
            }
    };
    auto wait_vm = [&](int n) {  // s_waitcnt vmcnt needs an immediate
        if (n == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (n == DMA_TILE) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_TILE) : "memory");
        else if (n == DMA_TILE + 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_TILE + 1) : "memory");
        else if (n == 2 * DMA_TILE) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DMA_TILE) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DMA_TILE + 1) : "memory");
    };

    // ---- pipeline -------------------------------------------------------------------------------
    // tile t lives in stage t % 3.  Per tile: [read k-step 1 of t] [products of k-step 0] -- own
    // DMAs of t + 1 landed, own reads of t retired, barrier -- [DMA t + 3 into the stage of t]
    // [read k-step 0 of t + 1] [products of k-step 1].  Every read has 2 JT (1 + .. + NLV)
    // products of the other k-step in front of it, the barrier is the only point the matrix pipe
    // drains.
    const int n_pre = ntile < 3 ? ntile : 3;
    for (int u = 0; u < n_pre; ++u) issue_ops(0, DMA_TILE);
    {   // groups 1 and 2 may stay in flight (group 2 opens a chunk iff nkt == 2)
        const int g2 = DMA_TILE + (nkt == 2 ? 1 : 0);
        wait_vm(ntile > 2 ? DMA_TILE + g2 : (ntile > 1 ? DMA_TILE : 0));
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    Frags f0, f1;
    load_frags(0, 0, f0);
    // (measured, no effect beyond run-to-run noise: s_setprio for the later-slot wavefronts, slots
    // mixed across the SIMDs instead of by wavefront half)
    constexpr int N_GROUPS = NLV * (NLV + 1) / 2 * ((JT + 1) / 2);  // product groups per k-step
    const int dma_slot = (wave >= 4 && N_GROUPS > 1) ? 1 : 0;  // group behind which this wave issues DMAs
    int r_kt = 0, r_cseq = 0, r_chunk = c0, r_stage = 0;
    for (int t = 0; t < ntile; ++t) {
        const int r_next = (r_stage == (FSTAGES - 1) * L::STAGE) ? 0 : r_stage + L::STAGE;
        // DMA issue: a wave stalls ~100+ cycles per LDS-DMA instruction and issues in order, so
        // the matrix pipe only stays fed while its other wave is NOT stalled in the same place.
        // Every wave issues half of a tile's DMAs in each half tile (front half of tile t + 3
        // behind the barrier, back half in the first half of the next tile); waves 0-3 (one per
        // SIMD) do so behind the first product group, their SIMD mates 4-7 behind the second.
        const bool back_now = t >= 1 && t + 2 < ntile;
        products(f0, [&](int g) {  // the reads of k-step 1 go behind the first product group
            if (g == 0) {
                __builtin_amdgcn_sched_barrier(0);
                load_frags(r_stage, 1, f1);
                touch_frags(f0);
            }
            if (g == dma_slot && back_now) issue_ops(DMA_TILE / 2, DMA_TILE);
            __builtin_amdgcn_sched_barrier(0);
        });
        __builtin_amdgcn_sched_barrier(0);
        // (after the last tile this block is a no-op on stale data: no branch, so that the
        // compiler's LDS wait counting sees one path)
        if (t + 2 < ntile) wait_vm(DMA_TILE + ((r_kt + 2 == nkt) ? 1 : 0));
        else wait_vm(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const bool front_now = t + 3 < ntile;
        products(f1, [&](int g) {
            if (g == 0) {
                __builtin_amdgcn_sched_barrier(0);
                // (before a chunk epilogue the next tile's fragments are fetched after it: they
                // would only sit in 32 registers the epilogue needs)
                if (r_kt != nkt - 1) load_frags(r_next, 0, f0);
                // the compiler counts an LDS-DMA as a pending LDS event of unknown order and
                // answers the next fragment use with lgkmcnt(0); retire f1 in ITS books here,
                // before the DMAs, where the counted wait it emits is already satisfied
                touch_frags(f1);
            }
            if (g == dma_slot && front_now) issue_ops(0, DMA_TILE / 2);
            __builtin_amdgcn_sched_barrier(0);
        });

        if (r_kt == nkt - 1) {
            const int jc = r_chunk * BJ;
            const double *ytab = reinterpret_cast<const double *>(smem + L::OFF_TAB + r_cseq * 2 * L::TAB);
            const double *ctb = ytab + BJ;
            const bool has_prev = (jc <= jhi) && (jc + BJ - 1 >= jlo);
            (void)has_prev;
            // everything derived from the lane's prototype offset is rebuilt here: hoisted out of
            // the tile loop those 64 bit masks / table addresses cost the fragment registers
            int jl0 = wj * 32 * JT + 4 * lh;
            asm volatile("" : "+v"(jl0));
            auto combine = [&](int a0, int a1, int a2) -> double { return sweep_T<PLANES>(a0, a1, a2); };
            constexpr int L1 = NLV >= 2 ? 1 : 0, L2 = NLV >= 3 ? 2 : 0;  // level indices that exist
            if constexpr (MODE == 0) {
                if (has_prev) {  // bound from the seed: thr_i = r~(i, seed_i) + 2 eps_i
                    // the seed's own products are picked by selects, not 64 divergent branches
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        int a0s = 0, a1s = 0, a2s = 0, jls = -1;
                        const int want = prev_s[wi * 64 + it * 32 + lc] - jc;
#pragma unroll
                        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int jl = jl0 + jt * 32 + (r & 3) + 8 * (r >> 2);
                                const bool sel = jl == want;
                                a0s = sel ? acc[jt][it][0][r] : a0s;
                                if constexpr (PLANES >= 2) a1s = sel ? acc[jt][it][L1][r] : a1s;
                                if constexpr (PLANES == 3) a2s = sel ? acc[jt][it][L2][r] : a2s;
                                jls = sel ? jl : jls;
                            }
                        if (jls >= 0)
                            thr_s[wi * 64 + it * 32 + lc] =
                                (s_i[it] * (craw[jc + jls] * combine(a0s, a1s, a2s)) - yraw[jc + jls]) - eps2_i[it];
                    }
                    __syncthreads();
#pragma unroll
                    for (int it = 0; it < 2; ++it) A_i[it] = thr_s[wi * 64 + it * 32 + lc];
                }
            }
            // all table reads and compares of a 32-prototype tile first (bits), the LDS atomics
            // after them: a possible atomic between two elements pins every later table read
            // behind it and exposes one LDS round trip per element.
            // MODE 0 tests r~ <= thr as (xx - thr) ic_j + yc_j <= s T' with the tables in reciprocal
            // form (6 float64 operations per pair instead of 9); MODE 1 needs r~ itself.
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                // The 32 prototypes of (wavefront column wj, tile jt) are ONE word of the mask:
                // bit 4 lh + 8 g + i.  A compare leaves its result as a 64-lane mask in scalar
                // registers anyway, so the word is put together there (did any sample in lane
                // half lh pass?) and the wavefront issues a single LDS atomic for it.
                uint32_t word = 0;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    double y4[4], c4[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        y4[i] = ytab[jl0 + jt * 32 + 8 * g + i];
                        c4[i] = ctb[jl0 + jt * 32 + 8 * g + i];
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = 4 * g + i;
                        const int j = jc + jl0 + jt * 32 + 8 * g + i;
                        uint64_t pass = 0;  // lanes with a sample that passes (both tiles OR-ed, no
                                            // short circuit: a branch per pair costs more than the test)
#pragma unroll
                        for (int it = 0; it < 2; ++it) {
                            if constexpr (MODE == 0) {
                                const double Tp = sweep_T_scaled<PLANES>(acc[jt][it][0][r], acc[jt][it][L1][r],
                                                                         acc[jt][it][L2][r]);
                                // (a NaN marks: a seed that is a NaN row of W gives no bound, not an empty list)
                                pass |= __builtin_amdgcn_ballot_w64(!((A_i[it] * c4[i] + y4[i]) > s_i[it] * Tp));
                            } else {
                                const double T = combine(acc[jt][it][0][r], acc[jt][it][L1][r], acc[jt][it][L2][r]);
                                const double rv = y4[i] - s_i[it] * (c4[i] * T);  // r~ - |x_i|^2
                                if (j < M && rv < bestv[it]) { bestv[it] = rv; bestj[it] = j * jstride; }
                            }
                        }
                        if constexpr (MODE == 0) {
                            const uint64_t b = pass;
                            word |= (uint32_t)((uint32_t)b != 0u) << (8 * g + i);
                            word |= (uint32_t)((uint32_t)(b >> 32) != 0u) << (4 + 8 * g + i);
                        }
                    }
                }
                if constexpr (MODE == 0) {
                    const int wbase = jc + wj * 32 * JT + jt * 32;  // multiple of 32
                    if (wbase < M) {
                        if (M - wbase < 32) word &= (1u << (M - wbase)) - 1u;  // prototypes >= M: padding
                        if (word != 0u && lane == 0) atomicOr(&mask[wbase >> 5], word);
                    }
                }
            }
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                for (int it = 0; it < 2; ++it)
#pragma unroll
                    for (int lv = 0; lv < NLV; ++lv)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[jt][it][lv][r] = 0;
            r_chunk = (r_chunk + 1 == nchunk) ? 0 : r_chunk + 1;
            r_cseq = (r_cseq == 2) ? 0 : r_cseq + 1;
            load_frags(r_next, 0, f0);
        }
        r_kt = (r_kt == nkt - 1) ? 0 : r_kt + 1;  // a select: see sweep4_i8_kernel
        r_stage = r_next;
    }

    if constexpr (MODE == 1) {
        // seed = arg-min of r~ over the 2 lane halves and the 4 prototype wavefronts
        __syncthreads();
        double *sv = reinterpret_cast<double *>(smem);          // [4][128]
        int *sj = reinterpret_cast<int *>(smem + 4 * 128 * 8);  // [4][128]
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const double ov = __shfl_xor(bestv[it], 32, 64);
            const int oj = __shfl_xor(bestj[it], 32, 64);
            if (ov < bestv[it] || (ov == bestv[it] && oj < bestj[it])) { bestv[it] = ov; bestj[it] = oj; }
            if (lh == 0) {
                sv[wj * 128 + wi * 64 + it * 32 + lc] = bestv[it];
                sj[wj * 128 + wi * 64 + it * 32 + lc] = bestj[it];
            }
        }
        __syncthreads();
        if (tid < 128 && p0 + tid < N) {
            double bv = sv[tid];
            int bj = sj[tid];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const double ov = sv[w * 128 + tid];
                const int oj = sj[w * 128 + tid];
                if (ov < bv || (ov == bv && oj < bj)) { bv = ov; bj = oj; }
            }
            seed[sample_at(p0 + tid)] = (int64_t)bj;
        }
        return;
    }

    // ---- compact the marked prototypes, ascending ------------------------------------------------
    __syncthreads();
    if (wave == 0) {
        uint32_t base = 0;
        uint16_t *out = ulist + (size_t)blockIdx.x * ulist_stride;
        for (int w0 = 0; w0 < nwords; w0 += 64) {
            const int w = w0 + lane;
            uint32_t bits = (w < nwords) ? mask[w] : 0u;
            const uint32_t cnt = __popc(bits);
            uint32_t pre = cnt;  // inclusive scan over the wavefront
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t v = __shfl_up(pre, off, 64);
                if (lane >= off) pre += v;
            }
            uint32_t pos = base + pre - cnt;
            while (bits) {
                const int b = __ffs(bits) - 1;
                bits &= bits - 1;
                out[pos++] = (uint16_t)(w * 32 + b);
            }
            base += __shfl(pre, 63, 64);
        }
        if (lane == 0) {
            ucount[blockIdx.x] = base;
            atomicAdd(&sched_ctr[sched_bin(base)], 1u);  // bin counts of the exact stage's schedule (2b)
            // sum of the list lengths (what the engine's policy looks at: 8 bytes D2H instead of nb x 4)
            atomicAdd(reinterpret_cast<unsigned long long *>(sched_ctr + SCHED_SUM), (unsigned long long)base);
        }
    }
}

// ---- 2a'. the one-product candidate sweep, two workgroups per CU -------------------------------
// Same arithmetic, bound, marking rule and outputs as sweep_i8_kernel<0, 1, JT>; another shape of
// the same work: workgroup tile 128 samples x 256 prototypes, 24 KB ring stages and ONE set of chunk
// tables, 79.5 KB of LDS in all, so that TWO workgroups share a CU: the wavefronts of a SIMD belong
// to different workgroups -- they are not tied to the same barrier, and one workgroup's prologue,
// chunk epilogues and list compaction run under the other's products.  NW = 4 wavefronts as 2 x 2
// (wavefront tile 64 x 128, two wavefronts per SIMD) or NW = 8 as 2 x 4 (64 x 64 tiles, 64
// accumulator registers, <= 128 VGPRs: FOUR wavefronts per SIMD -- the default).
constexpr int S4_NT = 256;
#ifndef S4_SPLIT_ISSUE
#define S4_SPLIT_ISSUE 0  // 1: half of a tile's DMAs behind the barrier, half in the next tile's first half
#endif
struct Sweep4Lds {
    static constexpr int JT = 4, BJ = 256;
    static constexpr int X_BYTES = 128 * FKT, W_BYTES = BJ * FKT, STAGE = X_BYTES + W_BYTES;  // 8 + 16 KB
    static constexpr int TAB = BJ * 8;
    static constexpr int OFF_TAB = FSTAGES * STAGE;
    static constexpr int OFF_THR = OFF_TAB + 2 * TAB;
    static constexpr int OFF_PREV = OFF_THR + 128 * 8;
    static constexpr int OFF_EPS = OFF_PREV + 128 * 4;   // 2 eps_i per sample (MODE 0)
    static constexpr int OFF_MASK = OFF_EPS + 128 * 8;
    static constexpr int MAX_M = 8192;                   // bitmask of the marked prototypes: 1 KB
    static constexpr int OFF_MISC = OFF_MASK + MAX_M / 8;
    static constexpr int BYTES = OFF_MISC + 16;
};
static_assert(Sweep4Lds::BYTES <= 81920, "two workgroups per CU");

template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64, NW / 2) void sweep4_i8_kernel(  // (HIP: the second figure is waves per SIMD)
    const int8_t *__restrict__ xplanes, const double *__restrict__ sx,
    const double *__restrict__ l1x, const double *__restrict__ xx, int64_t N, int d, int dpad,
    const int8_t *__restrict__ wplanes, const float *__restrict__ ytab_g,
    const float *__restrict__ ctab_g, const double *__restrict__ yraw,
    const double *__restrict__ craw, const double *__restrict__ summary, int M,
    const int64_t *__restrict__ prev, const int32_t *__restrict__ order,
    uint16_t *__restrict__ ulist, int ulist_stride, uint32_t *__restrict__ ucount, int w_rows,
    int64_t *__restrict__ seed, int jstride, int nkt_used, const int32_t *__restrict__ kt_sel,
    uint32_t *__restrict__ sched_ctr, const float *__restrict__ chk_g,
    const int32_t *__restrict__ retry_groups, const unsigned long long *__restrict__ retry_len) {
    using L = Sweep4Lds;
    // NW wavefronts as 2 (samples) x WJ (prototypes), wavefront tile 64 x 32 JT: 4 -> 64 x 128,
    // 8 -> 64 x 64 (64 accumulator registers: <= 128 VGPRs, four wavefronts per SIMD)
    constexpr int NT = NW * 64, WJ = NW / 2, JT = 8 / WJ, BJ = L::BJ, PLANES = 1;
    static_assert(NW == 4 || NW == 8, "4 or 8 wavefronts");
    constexpr int XI = 8 / NW, WI = 16 / NW;  // LDS-DMA instructions per wave and tile: X rows, W rows
    constexpr int DMA_TILE = XI + WI;
    __shared__ __attribute__((aligned(16))) char smem[L::BYTES];
    double *thr_s = reinterpret_cast<double *>(smem + L::OFF_THR);
    int *prev_s = reinterpret_cast<int *>(smem + L::OFF_PREV);
    uint32_t *mask = reinterpret_cast<uint32_t *>(smem + L::OFF_MASK);
    int *misc = reinterpret_cast<int *>(smem + L::OFF_MISC);
    // chunk tables of the epilogue, float32 (see "float32 chunk epilogue" below): |w|^2 / c' and 1 / c'
    // (MODE 0) or |w|^2 and c 2^16 (MODE 1) of the chunk's BJ prototypes
    const float *ytab = reinterpret_cast<const float *>(smem + L::OFF_TAB);
    const float *ctb = ytab + BJ;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave / WJ, wj = wave % WJ;
    const int lc = lane & 31, lh = lane >> 5;
    // (MODE 1 re-seeding the listed 128-sample workgroups of a bucket order: section 2c)
    int64_t group = blockIdx.x;
    if (retry_groups) {
        if ((unsigned long long)blockIdx.x >= *retry_len) return;
        group = retry_groups[blockIdx.x];
    }
    const int64_t p0 = group * 128;
    const int nwords = (M + 31) / 32;

    auto sample_at = [&](int64_t p) -> int64_t { return (MODE == 0 || order) ? (int64_t)order[p] : p; };
    // every thread's own loads first (see sweep_i8_kernel)
    int64_t i_dr[XI];
    int dc[XI];
#pragma unroll
    for (int u = 0; u < XI; ++u) {  // DMA: wave w loads X rows 16 XI w .., 16 per instruction
        const int r = 16 * (XI * wave + u) + (lane >> 2);
        const int64_t xpos = (p0 + r < N) ? (p0 + r) : (N - 1);
        i_dr[u] = sample_at(xpos);
        dc[u] = (lane & 3) ^ ((r >> 2) & 3);
    }
    int64_t i_il[2];
    double s_i[2], l1_i[2], xx_i[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int64_t p = p0 + wi * 64 + it * 32 + lc;
        i_il[it] = sample_at(p < N ? p : N - 1);
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        s_i[it] = sx[i_il[it]];
        l1_i[it] = MODE == 0 ? l1x[i_il[it]] : 0.0;
        xx_i[it] = MODE == 0 ? xx[i_il[it]] : 0.0;
    }
    int jlo = 0, jhi = -1;
    if constexpr (MODE == 0) {
    for (int w = tid; w < nwords; w += NT) mask[w] = 0u;
    if (tid < 128) {
        const int64_t p = p0 + tid;
        int pj = -1;
        if (p < N) pj = (int)prev[sample_at(p)];
        prev_s[tid] = (pj >= 0 && pj < M) ? pj : -1;
        thr_s[tid] = (p < N) ? -INFINITY : INFINITY;
    }
    __syncthreads();
    if (wave == 0) {
        const int a = prev_s[lane], b = prev_s[lane + 64];
        int lo = min(a >= 0 ? a : 0x7fffffff, b >= 0 ? b : 0x7fffffff), hi = max(a, b);
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            lo = min(lo, __shfl_xor(lo, m, 64));
            hi = max(hi, __shfl_xor(hi, m, 64));
        }
        if (lane == 0) {
            misc[0] = (hi >= 0) ? lo : 0;
            misc[1] = hi;
        }
    }
    __syncthreads();
    jlo = __builtin_amdgcn_readfirstlane(misc[0]);
    jhi = __builtin_amdgcn_readfirstlane(misc[1]);
    if (jlo / BJ != jhi / BJ) {  // seeds in a later chunk: their bound up front (TPS threads per sample)
        constexpr int TPS = NT / 128;
        const int il = tid / TPS, q = tid % TPS;
        const int64_t p = p0 + il;
        const int pj = prev_s[il];
        const bool need = p < N && pj >= 0 && pj / BJ != jlo / BJ;
        int a0 = 0;
        const int64_t i = sample_at(p < N ? p : N - 1);
        if (need) {
            const int8_t *xr = xplanes + (size_t)i * dpad;
            const int wsw = (pj >> 2) & 3;
            for (int ch = q; ch < dpad / 16; ch += TPS) {
                const int8_t *wr = wplanes + ((size_t)(ch >> 2) * w_rows + pj) * FKT + (((ch & 3) ^ wsw) << 4);
                const v4i_t xv = *reinterpret_cast<const v4i_t *>(xr + ch * 16);
                const v4i_t wv = *reinterpret_cast<const v4i_t *>(wr);
#pragma unroll
                for (int e = 0; e < 4; ++e) a0 = __builtin_amdgcn_sdot4(xv[e], wv[e], a0, false);
            }
        }
#pragma unroll
        for (int m = 1; m < TPS; m <<= 1) a0 += __shfl_xor(a0, m, 64);
        if (need && q == 0) {
            const double T = sweep_T<PLANES>(a0, 0, 0);
            const double sv = sx[i], xv2 = xx[i];
            const double e2 = 2.0 * filter_eps(sv, l1x[i], xv2, summary[0], summary[1], summary[2], d, PLANES) +
                              epilogue32_slack(xv2, summary[2]);
            thr_s[il] = (sv * (craw[pj] * T) - yraw[pj]) - e2;
        }
        __syncthreads();
    }
    }  // MODE == 0

    // 2 eps_i and A_i = |x_i|^2 - thr_i live in LDS between the chunk epilogues (eight registers
    // the 8-wavefront shape does not have)
    double *eps_s = reinterpret_cast<double *>(smem + L::OFF_EPS);
    if constexpr (MODE == 0) {
        const double l1w_max = summary[0], t_max = summary[1], yy_max = summary[2];
        if (wj == 0 && lh == 0) {
#pragma unroll
            for (int it = 0; it < 2; ++it)
                eps_s[wi * 64 + it * 32 + lc] =
                    2.0 * filter_eps(s_i[it], l1_i[it], xx_i[it], l1w_max, t_max, yy_max, d, PLANES) +
                    epilogue32_slack(xx_i[it], yy_max);
        }
    }
    float bestv[2] = {INFINITY, INFINITY};  // MODE 1: running arg-min of r~ (float32: any seed will do)
    int bestj[2] = {0, 0};

    const int8_t *xsrc[XI];
#pragma unroll
    for (int u = 0; u < XI; ++u) xsrc[u] = xplanes + (size_t)i_dr[u] * dpad + dc[u] * 16;
    if constexpr (XI == 2) xsrc[XI - 1] -= 1024;  // its DMA carries the immediate offset 1024 (for the LDS side)
    // MODE 1 may look at a sample of the k-tiles only (kt_sel, see dbgsom_bmu_filtered)
    const int nkt_full = dpad / FKT;  // >= 2 (filter_dpad)
    const int nkt = (MODE == 1 && nkt_used >= 2 && nkt_used < nkt_full) ? nkt_used : nkt_full;
    auto tile_of = [&](int kt) { return (MODE == 1 && nkt < nkt_full) ? (int)kt_sel[kt] : kt; };
    const int nchunk = (M + BJ - 1) / BJ;
    const int ntile = nkt * nchunk;
    const int c0 = jlo / BJ;

    int i_kt = 0, i_chunk = c0, i_stage = 0;
    // The 6 LDS-DMA instructions of a tile: X rows 16 (2 w + u) .. (ops 0, 1), then the wave's four
    // consecutive KiB of the chunk's W rows (ops 2 + v: rows 16 (4 w + v) ..) -- ONE address and one
    // LDS base for the four, the KiB steps sit in the instruction's immediate offset (it moves the
    // global and the LDS address alike).  Ops [lo, hi) are issued, hi == DMA_TILE advances.
    const int8_t *wlane = wplanes + 16u * lane + 1024u * WI * wave;  // this lane's 16 bytes of the wave's WI KiB
    const size_t w_tile_step = (size_t)w_rows * FKT;
    size_t i_woff = (size_t)c0 * BJ * FKT;  // (i_kt w_rows + i_chunk BJ) FKT, kept incrementally
    auto issue_ops = [&](int lo, int hi) {
        char *stage = smem + i_stage;
        const int i_tile = tile_of(i_kt);
        const int k0 = i_tile * FKT;
        if constexpr (MODE == 1) i_woff = ((size_t)i_tile * w_rows + (size_t)i_chunk * BJ) * FKT;
        if (0 >= lo && 0 < hi) fdma16(xsrc[0] + k0, stage + 1024 * XI * wave);
        if constexpr (XI == 2)
            if (1 >= lo && 1 < hi) fdma16_off<1024>(xsrc[XI - 1] + k0, stage + 1024 * XI * wave);
        const int8_t *wsrc = wlane + i_woff;
        char *wdst = stage + L::X_BYTES + 1024 * WI * wave;
        if (XI + 0 >= lo && XI + 0 < hi) fdma16(wsrc, wdst);
        if (XI + 1 >= lo && XI + 1 < hi) fdma16_off<1024>(wsrc, wdst);
        if constexpr (WI == 4) {
            if (XI + 2 >= lo && XI + 2 < hi) fdma16_off<2048>(wsrc, wdst);
            if (XI + 3 >= lo && XI + 3 < hi) fdma16_off<3072>(wsrc, wdst);
        }
        if (hi == DMA_TILE) {
            i_stage = (i_stage == (FSTAGES - 1) * L::STAGE) ? 0 : i_stage + L::STAGE;
            i_woff += w_tile_step;
            if (++i_kt == nkt) {
                i_kt = 0;
                i_chunk = (i_chunk + 1 == nchunk) ? 0 : i_chunk + 1;
                i_woff = (size_t)i_chunk * BJ * FKT;
            }
        }
    };
    // the chunk tables (2 pieces of 1 KiB = 256 float32: |w|^2 / c', 1 / c'), one piece per wave 0, 1
    const bool tab_wave = wave < 2;
    auto issue_tables = [&](int chunk) {
        if (!tab_wave) return;
        const int j4 = chunk * BJ + 4 * lane;  // tables are padded to whole 512-entry chunks
        char *tab = smem + L::OFF_TAB;
        if (wave == 1) fdma16(ctab_g + j4, tab + 1024);
        else fdma16(ytab_g + j4, tab);
    };

    int xoff[2], woff[JT];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int r = wi * 64 + it * 32 + lc;
        xoff[it] = r * FKT + ((lh ^ ((r >> 2) & 3)) * 16);
    }
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
        const int r = wj * 32 * JT + jt * 32 + lc;
        woff[jt] = L::X_BYTES + r * FKT + ((lh ^ ((r >> 2) & 3)) * 16);
    }
    struct Frags { v4i_t x[2], w[JT]; };
    auto load_frags = [&](int stage_off, int ks, Frags &f) {
        const char *stage = smem + stage_off;
#pragma unroll
        for (int jt = 0; jt < JT; ++jt)
            f.w[jt] = *reinterpret_cast<const v4i_t *>(stage + (woff[jt] ^ (ks * 32)));
#pragma unroll
        for (int it = 0; it < 2; ++it)
            f.x[it] = *reinterpret_cast<const v4i_t *>(stage + (xoff[it] ^ (ks * 32)));
    };
    auto touch_frags = [&](const Frags &f) {
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) asm volatile("" ::"v"(f.w[jt]));
#pragma unroll
        for (int it = 0; it < 2; ++it) asm volatile("" ::"v"(f.x[it]));
    };
    v16i_t acc[JT][2];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[jt][it][r] = 0;
    auto products = [&](const Frags &f, auto between) {
#pragma unroll
        for (int jh = 0; jh < JT; jh += 2) {
#pragma unroll
            for (int jt = jh; jt < jh + 2 && jt < JT; ++jt)
#pragma unroll
                for (int it = 0; it < 2; ++it)
                    acc[jt][it] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.w[jt], f.x[it], acc[jt][it], 0, 0, 0);
            between(jh >> 1);
        }
    };
    auto wait_vm = [&](int n) {  // s_waitcnt vmcnt needs an immediate
        if (n == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (n == DMA_TILE) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_TILE) : "memory");
        else if (n == DMA_TILE + 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_TILE + 1) : "memory");
        else if (n == 2 * DMA_TILE) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DMA_TILE) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DMA_TILE + 1) : "memory");
    };

    // ---- pipeline: as sweep_i8_kernel (tile t in stage t % 3, one barrier per tile) ---------------
    issue_tables(c0);  // first in the queue: complete with the first tile
    const int n_pre = ntile < 3 ? ntile : 3;
    for (int u = 0; u < n_pre; ++u) issue_ops(0, DMA_TILE);
    wait_vm(ntile > 2 ? 2 * DMA_TILE : (ntile > 1 ? DMA_TILE : 0));
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    Frags f0, f1;
    load_frags(0, 0, f0);
    int r_kt = 0, r_chunk = c0, r_stage = 0;
    bool tab_pending = false;  // a table DMA was issued in the previous tile's second half
    for (int t = 0; t < ntile; ++t) {
        const int r_next = (r_stage == (FSTAGES - 1) * L::STAGE) ? 0 : r_stage + L::STAGE;
        const bool back_now = t >= 1 && t + 2 < ntile;
        products(f0, [&](int g) {
            if (g == 0) {
                __builtin_amdgcn_sched_barrier(0);
                load_frags(r_stage, 1, f1);
                touch_frags(f0);
                if (S4_SPLIT_ISSUE && back_now) issue_ops(DMA_TILE / 2, DMA_TILE);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        __builtin_amdgcn_sched_barrier(0);
        // own DMAs of tile t + 1 landed: what may stay in flight is tile t + 2 (and the table piece
        // issued behind the previous barrier, which sits between tiles t + 2 and t + 3 in the queue)
        if (t + 2 < ntile) wait_vm(DMA_TILE + (tab_pending ? 1 : 0));
        else wait_vm(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        tab_pending = false;
        const bool front_now = t + 3 < ntile;
        products(f1, [&](int g) {
            if (g == 0) {
                __builtin_amdgcn_sched_barrier(0);
                if (r_kt != nkt - 1) load_frags(r_next, 0, f0);
                touch_frags(f1);
                // every wave is past the previous chunk's epilogue here: the table set may be
                // replaced by this chunk's
                if (r_kt == 0 && t > 0) { issue_tables(r_chunk); tab_pending = tab_wave; }
                if (front_now) issue_ops(0, S4_SPLIT_ISSUE ? DMA_TILE / 2 : DMA_TILE);
            }
            __builtin_amdgcn_sched_barrier(0);
        });

        if (r_kt == nkt - 1) {
            // the table pieces of this chunk: own piece landed (at most the 9 DMAs issued behind it
            // are in flight), then everybody's
            // (the DMAs issued behind it: a tile and a half with split issue, two tiles without)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S4_SPLIT_ISSUE ? DMA_TILE + DMA_TILE / 2 : 2 * DMA_TILE) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const int jc = r_chunk * BJ;
            const bool has_prev = MODE == 0 && (jc <= jhi) && (jc + BJ - 1 >= jlo);
            int jl0 = wj * 32 * JT + 4 * lh;
            asm volatile("" : "+v"(jl0));
            if (has_prev) {
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    int a0s = 0, jls = -1;
                    const int want = prev_s[wi * 64 + it * 32 + lc] - jc;
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int jl = jl0 + jt * 32 + (r & 3) + 8 * (r >> 2);
                            const bool sel = jl == want;
                            a0s = sel ? acc[jt][it][r] : a0s;
                            jls = sel ? jl : jls;
                            if constexpr (NW == 8) __builtin_amdgcn_sched_barrier(0);
                        }
                    if (jls >= 0)
                        thr_s[wi * 64 + it * 32 + lc] =
                            (s_i[it] * (craw[jc + jls] * sweep_T<PLANES>(a0s, 0, 0)) - yraw[jc + jls]) -
                            eps_s[wi * 64 + it * 32 + lc];
                }
                __syncthreads();
            }
            // float32 chunk epilogue.  The marking test r~_ij <= thr_i in the form
            //     A_i (1 / c'_j) + |w_j|^2 / c'_j <= s_i T'_ij,    A_i = |x_i|^2 - thr_i,
            // evaluated in float32 (an fma, a product, a conversion and a compare per pair -- the
            // float64 form took twice the issue cycles, and at d <= 256 the epilogues, not the matrix
            // products, are most of this kernel).  Marking MORE than the exact test would is always
            // allowed, so the test is made one-sided: every float32 operand is within 2^-24 of its
            // float64 value (table entries and A_i, s_i that are not normal float32 numbers are
            // replaced by "always marked"), the fma, the product and the conversion of T' round
            // once each, so the two sides are off by at most 3.1 2^-24 (|A_i / c'| + |w|^2 / c' +
            // |s_i T'|) <= 3.1 2^-24 (|A_i| + |x_i|^2 + 2.1 max|w|^2) / c' -- thr_i carries that much
            // extra slack (epilogue32_slack, and the 2.5e-7 |A_i| below), and a NaN marks.
            float a32[2] = {0.f, 0.f}, s32[2];
#pragma unroll
            for (int it = 0; it < 2; ++it) s32[it] = (float)s_i[it];
            if constexpr (MODE == 0) {
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const double A = thr_s[wi * 64 + it * 32 + lc];
                    const double Am = A - 2.5e-7 * fabs(A);
                    const bool inf = (A == INFINITY) || (A == -INFINITY);  // no sample / no bound yet
                    const bool ok = f32_representable(Am) && f32_representable(s_i[it]) && s_i[it] != 0.0;
                    a32[it] = inf ? (float)A : (ok ? (float)Am : -INFINITY);
                }
            }
            // Coarse integer test in front of it: a lower bound Lmin_i <= fma(a_i, 1/c'_j, |w_j|^2/c'_j)
            // over the WHOLE chunk (float32 fma and rounding are monotone: smallest table entries,
            // the largest 1/c' for a negative a_i), turned into an integer T'_low with s_i T' < Lmin_i
            // for every T' < T'_low (relative slack 4e-7 and one unit; NaN or -inf -> no bound).  A
            // prototype row whose 64 accumulators all stay below their sample's T'_low cannot be
            // marked by the test above and skips it: on clustered data that is nearly every row.
            int tlow[2] = {(int)0x80000000, (int)0x80000000};
            if constexpr (MODE == 0) {
                const float4 ck = *reinterpret_cast<const float4 *>(chk_g + 4 * r_chunk);
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const float lmin = fmaf(a32[it], a32[it] >= 0.f ? ck.y : ck.z, ck.x);
                    float q = lmin / s32[it];
                    q = q - fabsf(q) * 4e-7f - 2.f;
                    // (NaN, -inf, a non-positive or non-finite s: no bound -- INT_MIN)
                    const bool bounded = (q == q) && q > -2.0e9f && s32[it] > 0.f;
                    tlow[it] = bounded ? (q >= 2.0e9f ? 0x7fffffff : (int)floorf(q)) : (int)0x80000000;
                }
            }
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                uint32_t word = 0;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 y4 = *reinterpret_cast<const float4 *>(ytab + jl0 + jt * 32 + 8 * g);
                    const float4 c4 = *reinterpret_cast<const float4 *>(ctb + jl0 + jt * 32 + 8 * g);
                    const float yv[4] = {y4.x, y4.y, y4.z, y4.w}, cv[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = 4 * g + i;
                        uint64_t pass = 0;
                        if constexpr (MODE == 0) {
                            const uint64_t maybe = __builtin_amdgcn_ballot_w64(acc[jt][0][r] >= tlow[0]) |
                                                   __builtin_amdgcn_ballot_w64(acc[jt][1][r] >= tlow[1]);
                            if (maybe == 0) continue;   // wave-uniform: nobody can pass for this pair of rows
                        }
#pragma unroll
                        for (int it = 0; it < 2; ++it) {
                            const float Tf = (float)acc[jt][it][r];
                            if constexpr (MODE == 0) {
                                pass |= __builtin_amdgcn_ballot_w64(!(fmaf(a32[it], cv[i], yv[i]) > s32[it] * Tf));
                            } else {  // tables in plain form here: r~ - |x_i|^2 = |w|^2 - s (c 2^16) T'
                                const int j = jc + jl0 + jt * 32 + 8 * g + i;
                                const float rv = yv[i] - s32[it] * (cv[i] * Tf);
                                if (j < M && rv < bestv[it]) { bestv[it] = rv; bestj[it] = j * jstride; }
                            }
                        }
                        if constexpr (MODE == 0) {
                            word |= (uint32_t)((uint32_t)pass != 0u) << (8 * g + i);
                            word |= (uint32_t)((uint32_t)(pass >> 32) != 0u) << (4 + 8 * g + i);
                            // (8 wavefronts: one compare pair at a time -- the scalar masks of many
                            // hoisted compares spill, and spilled SGPRs cost VGPRs beyond the 128)
                            if constexpr (NW == 8) __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
                if constexpr (MODE == 0) {
                    const int wbase = jc + wj * 32 * JT + jt * 32;
                    if (wbase < M) {
                        if (M - wbase < 32) word &= (1u << (M - wbase)) - 1u;
                        if (word != 0u && lane == 0) atomicOr(&mask[wbase >> 5], word);
                    }
                }
            }
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                for (int it = 0; it < 2; ++it)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[jt][it][r] = 0;
            r_chunk = (r_chunk + 1 == nchunk) ? 0 : r_chunk + 1;
            load_frags(r_next, 0, f0);
        }
        // (a select, not an assignment per branch: the merge of the two paths otherwise gets an
        // undefined scalar that the compiler fills from an accumulator register -- a wait for the
        // last MFMA of every tile)
        r_kt = (r_kt == nkt - 1) ? 0 : r_kt + 1;
        r_stage = r_next;
    }

    if constexpr (MODE == 1) {
        // seed = arg-min of r~ over the 2 lane halves and the 2 prototype wavefronts
        __syncthreads();
        float *sv = reinterpret_cast<float *>(smem);             // [WJ][128]
        int *sj = reinterpret_cast<int *>(smem + WJ * 128 * 8);  // [WJ][128]
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const float ov = __shfl_xor(bestv[it], 32, 64);
            const int oj = __shfl_xor(bestj[it], 32, 64);
            if (ov < bestv[it] || (ov == bestv[it] && oj < bestj[it])) { bestv[it] = ov; bestj[it] = oj; }
            if (lh == 0) {
                sv[wj * 128 + wi * 64 + it * 32 + lc] = bestv[it];
                sj[wj * 128 + wi * 64 + it * 32 + lc] = bestj[it];
            }
        }
        __syncthreads();
        if (tid < 128 && p0 + tid < N) {
            float bv = sv[tid];
            int bj = sj[tid];
#pragma unroll
            for (int w = 1; w < WJ; ++w) {
                const float ov = sv[w * 128 + tid];
                const int oj = sj[w * 128 + tid];
                if (ov < bv || (ov == bv && oj < bj)) { bv = ov; bj = oj; }
            }
            seed[sample_at(p0 + tid)] = (int64_t)bj;
        }
        return;
    }
    __syncthreads();
    if (wave == 0) {  // compact the marked prototypes, ascending
        uint32_t base = 0;
        uint16_t *out = ulist + (size_t)blockIdx.x * ulist_stride;
        for (int w0 = 0; w0 < nwords; w0 += 64) {
            const int w = w0 + lane;
            uint32_t bits = (w < nwords) ? mask[w] : 0u;
            const uint32_t cnt = __popc(bits);
            uint32_t pre = cnt;
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t v = __shfl_up(pre, off, 64);
                if (lane >= off) pre += v;
            }
            uint32_t pos = base + pre - cnt;
            while (bits) {
                const int b = __ffs(bits) - 1;
                bits &= bits - 1;
                out[pos++] = (uint16_t)(w * 32 + b);
            }
            base += __shfl(pre, 63, 64);
        }
        if (lane == 0) {
            ucount[blockIdx.x] = base;
            atomicAdd(&sched_ctr[sched_bin(base)], 1u);  // bin counts of the exact stage's schedule (2b)
            // sum of the list lengths (what the engine's policy looks at: 8 bytes D2H instead of nb x 4)
            atomicAdd(reinterpret_cast<unsigned long long *>(sched_ctr + SCHED_SUM), (unsigned long long)base);
        }
    }
}

// ---- 2b. launch order of the exact stage ---------------------------------------------------------
// The workgroups of the exact stage differ in length (one step per 16 JTL list entries) and sit in
// the grid in sample order: a two-step list that starts in the last round of resident workgroups
// keeps a mostly idle chip waiting for a whole extra step (C4: ~0.25 ms of a 1.3 ms stage).  The
// workgroup ids are therefore bucketed into SCHED_BINS bins -- class 3 by steps, longest first
// (bins 0 .. 13), then class 2 (14), then class 1 (15) -- and every class kernel walks its range of
// that schedule.  Within a bin the order is whatever the atomics give (blocks of 256 consecutive
// workgroups stay together): the schedule only decides WHEN a workgroup runs, never what it writes.
__global__ __launch_bounds__(256) void sched_fill_kernel(const uint32_t *__restrict__ ucount, int nb,
                                                         uint32_t *__restrict__ ctr,
                                                         int32_t *__restrict__ sched,
                                                         const uint8_t *__restrict__ skip) {
    __shared__ uint32_t h[SCHED_BINS], base[SCHED_BINS];
    if (threadIdx.x < SCHED_BINS) h[threadIdx.x] = 0u;
    __syncthreads();
    const int b = blockIdx.x * 256 + threadIdx.x;
    int bin = -1;
    uint32_t r = 0;
    if (b < nb && !(skip && skip[b])) {  // (skip: workgroups the refinement took, section 2d)
        bin = sched_bin(ucount[b]);
        r = atomicAdd(&h[bin], 1u);
    }
    __syncthreads();
    if (threadIdx.x < SCHED_BINS) {
        uint32_t off = 0;
        for (int u = 0; u < (int)threadIdx.x; ++u) off += ctr[u];
        base[threadIdx.x] = off + (h[threadIdx.x] ? atomicAdd(&ctr[SCHED_BINS + threadIdx.x], h[threadIdx.x]) : 0u);
        if (blockIdx.x == 0) {  // the class ranges, for the class kernels
            uint32_t *range = ctr + 2 * SCHED_BINS;
            if (threadIdx.x == 0) range[0] = 0u;
            if (threadIdx.x == 14) { range[1] = off; range[2] = off; range[3] = ctr[14]; }
            if (threadIdx.x == 15) { range[4] = off; range[5] = ctr[15]; }
        }
    }
    __syncthreads();
    if (bin >= 0) sched[base[bin] + r] = b;
}

// ---- 2c. candidates without a sweep: the triangle inequality --------------------------------------
// Clustered data (what a trained map sits on) lets most of the map be ruled out before any product
// with the sample is formed: with p = the sample's seed,
//     |x_i - w_j| >= |w_p - w_j| - |x_i - w_p|   (real arithmetic, Euclidean norms),
// so a prototype j with |w_p - w_j| >= 2 |x_i - w_p| + m is at least m further from x_i than the seed.
// Two certified ingredients, both from the TOP digit plane alone.  With a^ = s D0 / 127 the row that
// plane stands for, |a_k - a^_k| <= (s / F)(2^15 + 2^7 + 1/2 + 3 u F) <= s / 253, so |a - a^| <= e_a :=
// sqrt(d) s / 253 (Euclidean), and distances between such rows are exact integer sums:
// |x^ - w^|^2 = (s^2 A - 2 s t P + t^2 B) / 127^2, A = sum D0x^2, B = sum D0w^2, P = sum D0x D0w.
//   gap[p][j]  <= |w_p - w_j|^2     float32: (|w^_p - w^_j| - e_p - e_j)^2, rounded down (proto_gap_kernel);
//   bound_i    >= (2 |x_i - w_p| + m_i)^2  with |x_i - w_p| <= |x^_i - w^_p| + e_i + e_p
//                 (prune_mark_kernel: one pass over the top plane of X), and
//                 m_i = sqrt(2 rho_i), rho_i = 4 (d + 16) 2^-53 (|x_i|^2 + max |w|^2) -- at least twice
//                 what the exact kernel's chain can be off the real squared distance by.
// (The bound filter_eps gives for r~ is of no use here: its worst case over the dropped digit
//  products, ~ 0.008 d s t for one plane, exceeds the squared distances between the clusters.)
// gap[p][j] >= bound_i  =>  |x_i - w_j|^2 >= |x_i - w_p|^2 + m_i^2  =>  r_chain(i, j) > r_chain(i, p):
// j can neither win nor tie.  The candidates of a 128-sample workgroup are the prototypes that
// survive for any of its samples; samples arrive in bucket order of their seeds, so a workgroup has
// a few distinct seeds (runs) and tests a few rows of the gap matrix against the runs' largest bounds.
// Same outputs as the sweep (ulist / ucount / schedule counters): the exact stage does not know
// which of the two produced its lists.  What it costs: one pass over the X plane and an M x M matrix
// per epoch instead of N x M digit products; what it yields depends on the data alone (blobs: the
// sample's own cluster; isotropic data: the whole map -- the engine's policy measures it with a
// counting-only launch before it lets the exact stage loose on such lists).
constexpr int PRUNE_MAX_M = 8192;  // the gap matrix: 4 M^2 bytes (256 MB here)

// relative size of what the lower digits and the rounding of the quantisation add to a feature:
// (2^15 + 2^7 + 1/2 + 3 u F) / F
constexpr double PLANE0_ERR = 32897.0 / (127.0 * 65536.0) * (1.0 + 1e-6);

// gap[j ldg + p] for a (32 NB) x (32 NB) tile of (p, j); one wavefront per tile, operands straight from the
// k-tile-major top plane (L2-resident), P = D0 . D0.  NB = 2 (64 x 64 tiles) where that fills the chip; maps below
// ~2000 prototypes have fewer such tiles than four per CU -- 256 at M = 1024, one lone wavefront per CU walking its
// k loop and a float64 epilogue of 64 values per lane with nothing to overlap them with (141 us at C4, on the
// critical path of every epoch that starts from hints) -- and take NB = 1: four times the wavefronts, a quarter
// of the epilogue each
template <int NB>
__global__ __launch_bounds__(64) void proto_gap_kernel(const int8_t *__restrict__ wt, int w_rows, int dpad, int M, int d,
                                                       const double *__restrict__ tw, const double *__restrict__ wn0,
                                                       const double *__restrict__ ww, float *__restrict__ gap, int ldg,
                                                       uint32_t *__restrict__ nnub) {
    // nnub[p] (k = 2 searches; float32 bits, +inf before the launch): an UPPER bound of the distance from
    // prototype p to its nearest other prototype, from the same products: |w^_p - w^_j| + e_p + e_j
    const int lane = threadIdx.x, lc = lane & 31, lh = lane >> 5;
    constexpr int TS = 32 * NB;
    const int pb = blockIdx.x * TS, jb = blockIdx.y * TS;
    const int nks = dpad / 32;  // dpad is a multiple of 64
    const int8_t *base[2][NB];   // [side: 0 = p, 1 = j][32-row block]: this lane's row
    int sw[2][NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int rp = pb + b * 32 + lc, rj = jb + b * 32 + lc;
        base[0][b] = wt + (size_t)rp * FKT; sw[0][b] = (rp >> 2) & 3;
        base[1][b] = wt + (size_t)rj * FKT; sw[1][b] = (rj >> 2) & 3;
    }
    struct Fr { v4i_t v[2][NB]; };  // [side][block]
    auto load = [&](int ks, Fr &f) {
        const size_t tile = (size_t)(ks >> 1) * w_rows * FKT;
        const int c = (ks & 1) * 2 + lh;
#pragma unroll
        for (int sd = 0; sd < 2; ++sd)
#pragma unroll
            for (int b = 0; b < NB; ++b)
                f.v[sd][b] = *reinterpret_cast<const v4i_t *>(base[sd][b] + tile + ((c ^ sw[sd][b]) << 4));
    };
    v16i_t P[NB][NB];  // [jt][it]
#pragma unroll
    for (int jt = 0; jt < NB; ++jt)
#pragma unroll
        for (int it = 0; it < NB; ++it)
#pragma unroll
            for (int r = 0; r < 16; ++r) P[jt][it][r] = 0;
    Fr f0, f1, f2, f3;
    load(0, f0);
    load(1, f1);
    for (int ks = 0; ks < nks; ks += 4) {  // two k-steps in flight behind the two being multiplied
        if (ks + 2 < nks) { load(ks + 2, f2); load(ks + 3, f3); }
#pragma unroll
        for (int jt = 0; jt < NB; ++jt)
#pragma unroll
            for (int it = 0; it < NB; ++it) {
                P[jt][it] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f0.v[1][jt], f0.v[0][it], P[jt][it], 0, 0, 0);
                P[jt][it] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f1.v[1][jt], f1.v[0][it], P[jt][it], 0, 0, 0);
            }
        if (ks + 2 >= nks) break;
        if (ks + 4 < nks) { load(ks + 4, f0); load(ks + 5, f1); }
#pragma unroll
        for (int jt = 0; jt < NB; ++jt)
#pragma unroll
            for (int it = 0; it < NB; ++it) {
                P[jt][it] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f2.v[1][jt], f2.v[0][it], P[jt][it], 0, 0, 0);
                P[jt][it] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f3.v[1][jt], f3.v[0][it], P[jt][it], 0, 0, 0);
            }
    }
    const double root_d = sqrt((double)d) * (1.0 + 1e-12);
    // the tile's 64 column prototypes: scale and digit norm once, through LDS (every lane needs 32 of them)
    __shared__ double tj_s[64], bj_s[64];
    {
        const int j = jb + lane;   // (NB = 1: the upper half is filled and never read)
        tj_s[lane] = j < M ? tw[j] : 0.0;
        // (a row with a NaN or an infinity has digits that mean nothing: 0 |w|^2 turns into a NaN and
        //  the pair gets "no gap known")
        bj_s[lane] = j < M ? wn0[j] + 0.0 * ww[j] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NB; ++it) {
        const int p = pb + it * 32 + lc;
        const bool pok = p < M;
        const double tp = pok ? tw[p] : 0.0, Bp = pok ? wn0[p] + 0.0 * ww[p] : 0.0;
        const double ep = root_d * tp * PLANE0_ERR;
        float near_ub = INFINITY;
#pragma unroll
        for (int jt = 0; jt < NB; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int jl = jt * 32 + 4 * lh + (r & 3) + 8 * (r >> 2), j = jb + jl;
                if (!pok || j >= M) continue;
                const double tj = tj_s[jl], Bj = bj_s[jl], Pv = (double)P[jt][it][r];
                const double sq = tp * tp * Bp + tj * tj * Bj, cr = 2.0 * tp * tj * Pv;
                if (nnub && j != p) {
                    const double uh2 = ((sq - cr) + 1e-12 * (sq + fabs(cr))) * (1.0 / 16129.0) * (1.0 + 1e-12);
                    // (a NaN -- a row with a NaN or an infinity, whose digits mean nothing -- stays a NaN and never
                    //  becomes the minimum: no bound from that pair)
                    const double dh = uh2 > 0.0 ? sqrt(uh2) * (1.0 + 1e-12) : (uh2 == uh2 ? 0.0 : uh2);
                    const double ub = (dh + (ep + root_d * tj * PLANE0_ERR)) * (1.0 + 1e-6);
                    if (ub < 3.0e38) near_ub = fminf(near_ub, __double2float_ru(ub));
                }
                // (the division by 127^2 as a product by its rounded reciprocal: 2^-53 relative, far
                //  inside the 1e-12 margins taken just before and after)
                const double dh2 = ((sq - cr) - 1e-12 * (sq + fabs(cr))) * (1.0 / 16129.0);
                const double lo = (dh2 > 0.0 ? sqrt(dh2) * (1.0 - 1e-12) : 0.0) - (ep + root_d * tj * PLANE0_ERR);
                const double v = lo > 0.0 ? lo * lo * (1.0 - 1e-6) : 0.0;
                // (anything that is not a positive finite number: no gap known)
                gap[(size_t)j * ldg + p] = (v > 0.0 && v < 3.0e38) ? __double2float_rz(v) : 0.f;
            }
        if (nnub && pok && near_ub < INFINITY) atomicMin(&nnub[p], __float_as_uint(near_ub));  // (non-negative floats order like their bits)
    }
}

// the candidate lists of the 128-sample workgroups by the rule above.  count_only: only the sum of
// the list lengths (into sum_out) -- what the lists WOULD be, for the engine's policy.
__global__ __launch_bounds__(256, 6) void prune_mark_kernel(
    const int8_t *__restrict__ xplanes, const double *__restrict__ sx, const double *__restrict__ xx, int64_t N, int d,
    int dpad, const int8_t *__restrict__ wplanes, int w_rows, const double *__restrict__ tw,
    const double *__restrict__ ww, const double *__restrict__ summary, int M,
    const int64_t *__restrict__ prev, const int32_t *__restrict__ order, const float *__restrict__ gap, int ldg,
    uint16_t *__restrict__ ulist, int ulist_stride, uint32_t *__restrict__ ucount, uint32_t *__restrict__ sched_ctr,
    unsigned long long *__restrict__ sum_out, int count_only, const double *__restrict__ dist_prev,
    const double *__restrict__ shift, int32_t *__restrict__ retry_groups, unsigned long long *__restrict__ retry_len,
    int retry_mode, uint32_t retry_above, const uint32_t *__restrict__ nnub) {
    // nnub != nullptr: the lists of a k = 2 search.  With p2 the seed's nearest other prototype, |x - w_p2| <=
    // |x - w_p| + nnub[p]: TWO prototypes are at most that far, so everything at least m further cannot be
    // among the two nearest -- the bound of k = 1 with nnub[p] added to twice the seed distance.
    // retry_mode 1: a workgroup whose list comes out longer than retry_above is not final -- its id goes
    // to retry_groups (its samples' seeds were poor: a cluster none of whose prototypes is in the cheap
    // pre-pass's subset sends its samples to seeds in OTHER clusters, and twice that distance rules
    // nothing out); the launcher re-seeds those workgroups against every prototype and calls again with
    // retry_mode 2: the listed workgroups only, final whatever comes out.
    __shared__ int prev_s[128], run_p[128];
    __shared__ int64_t samp_s[128];
    __shared__ unsigned long long bound_s[128], run_t[128];  // non-negative doubles by their bit patterns
    __shared__ uint32_t mask[PRUNE_MAX_M / 32];
    __shared__ int misc[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int64_t group = blockIdx.x;
    if (retry_mode == 2) {
        if ((unsigned long long)blockIdx.x >= *retry_len) return;
        group = retry_groups[blockIdx.x];
    }
    const int64_t p0 = group * 128;
    PM_STAMP(0);
    const int nwords = (M + 31) / 32;
    const unsigned long long INF_BITS = 0x7ff0000000000000ull;
    for (int w = tid; w < nwords; w += 256) mask[w] = 0u;
    if (tid < 128) {
        const int64_t p = p0 + tid;
        int pj = -2;  // no sample
        int64_t i = 0;
        if (p < N) {
            i = (int64_t)order[p];
            const int64_t q = prev[i];
            pj = (q >= 0 && q < M) ? (int)q : -1;  // -1: a sample without a usable seed
        }
        prev_s[tid] = pj;
        samp_s[tid] = i;
        run_t[tid] = 0ull;
    }
    if (tid == 0) misc[0] = 0;  // 1: some sample has no bound -- every prototype is a candidate
    __syncthreads();
    PM_STAMP(1);   // (sample ids and seeds are here)
    // |x_i - w_seed|^2 from one digit product: 8 threads per sample, 16 bytes of the row each per step
    const double yy_max = summary[2], root_d = sqrt((double)d) * (1.0 + 1e-12);
    // (the four samples of a thread side by side, two 16-byte steps each: 16 loads in flight -- one
    //  sample and one step at a time the kernel waited for a load 28 times in a row)
    const int q = tid & 7, nch = dpad / 16;
    int a0r[4] = {0, 0, 0, 0}, axr[4] = {0, 0, 0, 0}, awr[4] = {0, 0, 0, 0};  // P, A, B of the header
    const int8_t *xrow[4], *wrow[4];
    int wswz[4], xodd = 0;   // (bit r: sample r's row starts in the middle of a cache line)
    bool live[4];
    double hint_up[4];  // >= 0: the seed distance is known without the sample's row (see below)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int il = r * 32 + (tid >> 3);
        const int pj = prev_s[il];
        // Seeds that are the previous epoch's winners come with that epoch's exact distance to the
        // OLD prototype; the new one has moved by shift[p] = |w'_p - w_p| at most (another triangle):
        // |x_i - w'_p| <= dist_i (1 + 1e-7) + sqrt(rho_i) + shift_p  (the distance is the square root
        // of a chain value within rho_i / 2 of the real one, possibly rounded to float32).  Taken when
        // the prototype moved by less than a quarter of that distance -- a frozen or nearly settled
        // map never reads X here; otherwise the row is read as for any other seed.
        hint_up[r] = -1.0;
        if (dist_prev && pj >= 0) {
            const double dp = dist_prev[samp_s[il]], sh = shift[pj];
            if (dp >= 0.0 && dp < INFINITY && sh >= 0.0 && sh <= 0.25 * dp) hint_up[r] = dp * (1.0 + 1e-7) + sh;
        }
        live[r] = pj >= 0 && !(hint_up[r] >= 0.0);
        xrow[r] = xplanes + (size_t)samp_s[il] * dpad;
        // (a row whose pitch is an odd number of 64-byte halves starts in the middle of a cache line every other
        //  sample: its eight lanes then walk the row in 128-byte windows that ARE lines -- the first window half
        //  empty -- instead of straddling two lines with every load; read once and non-temporal, a straddled line
        //  came from memory twice: 1.43 GB for 0.83 GB of plane at d = 784)
        xodd |= (int)((((size_t)samp_s[il] * (size_t)dpad) >> 6) & 1) << r;
        wrow[r] = wplanes + (size_t)(live[r] ? pj : 0) * FKT;
        wswz[r] = live[r] ? (pj >> 2) & 3 : 0;
    }
    // (two of the thread's four samples at a time: 8 loads of 16 bytes in flight per thread, 64 registers for the
    //  kernel -- eight workgroups per CU instead of four share the latency of its ~8 dependent steps)
#pragma unroll
    for (int rh = 0; rh < 4; rh += 2)
        for (int ch0 = q; ch0 < nch + 4; ch0 += 16) {
            v4i_t xv[2][2], wv[2][2];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int ch = ch0 + 8 * u - 4 * ((xodd >> (rh + r)) & 1);
                    const bool ok = live[rh + r] && ch >= 0 && ch < nch;
                    const v4i_t zero = {0, 0, 0, 0};
                    // (non-temporal: a sample's plane row is read once, eight lanes to a cache line)
                    xv[r][u] = ok ? __builtin_nontemporal_load(reinterpret_cast<const v4i_t *>(xrow[rh + r] + ch * 16)) : zero;
                    wv[r][u] = ok ? *reinterpret_cast<const v4i_t *>(wrow[rh + r] + (size_t)(ch >> 2) * w_rows * FKT +
                                                                     (((ch & 3) ^ wswz[rh + r]) << 4)) : zero;
                }
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        a0r[rh + r] = __builtin_amdgcn_sdot4(xv[r][u][e], wv[r][u][e], a0r[rh + r], false);
                        axr[rh + r] = __builtin_amdgcn_sdot4(xv[r][u][e], xv[r][u][e], axr[rh + r], false);
                        awr[rh + r] = __builtin_amdgcn_sdot4(wv[r][u][e], wv[r][u][e], awr[rh + r], false);
                    }
        }
    PM_STAMP(2);   // (the rows have been read)
    // The thread's four samples' sums over the sample's eight lanes (every lane ends up with the totals), then ONE pass
    // of the float64 bound with lane q of the eight taking sample q -- four passes with one lane in eight at work were
    // half of the instructions this kernel issues at d = 128.
#pragma unroll
    for (int round = 0; round < 4; ++round) {
#pragma unroll
        for (int m = 1; m < 8; m <<= 1) {
            a0r[round] += __shfl_xor(a0r[round], m, 64);
            axr[round] += __shfl_xor(axr[round], m, 64);
            awr[round] += __shfl_xor(awr[round], m, 64);
        }
    }
    {
        const int round = q & 3;
        const int il = round * 32 + (tid >> 3);
        const int pj = prev_s[il];
        const int64_t i = samp_s[il];
        const int a0 = round == 0 ? a0r[0] : (round == 1 ? a0r[1] : (round == 2 ? a0r[2] : a0r[3]));
        const int ax = round == 0 ? axr[0] : (round == 1 ? axr[1] : (round == 2 ? axr[2] : axr[3]));
        const int aw = round == 0 ? awr[0] : (round == 1 ? awr[1] : (round == 2 ? awr[2] : awr[3]));
        const double hint = round == 0 ? hint_up[0] : (round == 1 ? hint_up[1] : (round == 2 ? hint_up[2] : hint_up[3]));
        if (q < 4 && pj != -2) {
            unsigned long long bits = INF_BITS;
            if (pj >= 0) {
                const double sv = sx[i], tv = tw[pj];
                const double sq = sv * sv * (double)ax + tv * tv * (double)aw, cr = 2.0 * sv * tv * (double)a0;
                const double dh2 = ((sq - cr) + 1e-12 * (sq + fabs(cr))) / 16129.0;
                // (|x_i|^2 or |w_seed|^2 not finite -- a NaN or an infinity in the row: rho is not, and no bound)
                const double rho = 4.0 * (double)(d + 16) * 1.1102230246251565e-16 * (xx[i] + yy_max) + 0.0 * ww[pj];
                const double up = hint >= 0.0
                                      ? hint + sqrt(rho) * (1.0 + 1e-12)
                                      : (dh2 > 0.0 ? sqrt(dh2) * (1.0 + 1e-12) : 0.0) + root_d * (sv + tv) * PLANE0_ERR;
                if (fabs(up) < INFINITY) {  // (a NaN fails this too)
                    const double nn = nnub ? (double)__uint_as_float(nnub[pj]) : 0.0;  // (+inf: no bound, the whole map)
                    const double b = 2.0 * up * (1.0 + 1e-12) + nn * (1.0 + 1e-12) + (sqrt(2.0 * rho) * 1.0001 + 1e-300);
                    const double b2 = b * b * (1.0 + 1e-12);
                    if (b2 < INFINITY) bits = (unsigned long long)__double_as_longlong(b2);
                }
            }
            bound_s[il] = bits;
            if (bits == INF_BITS) misc[0] = 1;
        }
    }
    __syncthreads();
    PM_STAMP(3);   // (bounds)
    const bool all = misc[0] != 0;
    if (!all) {
        // runs of equal seeds (the samples come sorted by seed; any other order only makes more runs)
        if (wave == 0) {
            const int pa = prev_s[lane], pb_ = prev_s[lane + 64];
            const bool ha = pa != -2 && (lane == 0 || prev_s[lane - 1] != pa);
            const bool hb = pb_ != -2 && prev_s[lane + 63] != pb_;
            const uint64_t ba = __builtin_amdgcn_ballot_w64(ha), bb = __builtin_amdgcn_ballot_w64(hb);
            const uint64_t below = (lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull);  // positions <= lane
            const int ra = __popcll(ba & below) - 1, rb = __popcll(ba) + __popcll(bb & below) - 1;
            if (ha) run_p[ra] = pa;
            if (hb) run_p[rb] = pb_;
            if (pa != -2) atomicMax(&run_t[ra], bound_s[lane]);
            if (pb_ != -2) atomicMax(&run_t[rb], bound_s[lane + 64]);
            if (lane == 0) misc[1] = __popcll(ba) + __popcll(bb);
        }
        __syncthreads();
        const int nrun = misc[1];
        // a wave owns its mask words: four steps of 64 prototypes against four runs at a time -- 16 rows
        // of the gap matrix in flight per lane (one at a time the workgroup waited for L2 once per run and step)
        constexpr int JU = 4;
        for (int jb = wave * 64; jb < M; jb += 256 * JU) {
            bool keep[JU] = {false, false, false, false};
            for (int r = 0; r < nrun; r += 4) {
                int pr[4];
                double T[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int rr = r + v < nrun ? r + v : r;
                    pr[v] = run_p[rr];
                    T[v] = __longlong_as_double((long long)run_t[rr]);
                }
                float g[JU][4];
#pragma unroll
                for (int u = 0; u < JU; ++u) {
                    const int j = jb + 256 * u + lane;
                    const int jc = j < M ? j : M - 1;
#pragma unroll
                    for (int v = 0; v < 4; ++v) g[u][v] = gap[(size_t)pr[v] * ldg + jc];
                }
#pragma unroll
                for (int u = 0; u < JU; ++u) {
                    const int j = jb + 256 * u + lane;
#pragma unroll
                    for (int v = 0; v < 4; ++v) keep[u] |= (j == pr[v]) || !((double)g[u][v] >= T[v]);
                }
            }
#pragma unroll
            for (int u = 0; u < JU; ++u) {
                const int j0 = jb + 256 * u;
                if (j0 >= M) break;
                const uint64_t b = __builtin_amdgcn_ballot_w64(keep[u] && j0 + lane < M);
                if (lane == 0) {
                    mask[j0 >> 5] = (uint32_t)b;
                    if ((j0 >> 5) + 1 < nwords) mask[(j0 >> 5) + 1] = (uint32_t)(b >> 32);
                }
            }
        }
    } else {
        for (int w = tid; w < nwords; w += 256) {
            const int left = M - w * 32;
            mask[w] = left >= 32 ? 0xffffffffu : ((1u << left) - 1u);
        }
    }
    __syncthreads();
    PM_STAMP(4);   // (marks)
    if (wave == 0) {  // compact the marked prototypes, ascending (as the sweeps do)
        uint32_t base = 0;
        uint16_t *out = ulist + (size_t)group * ulist_stride;
        for (int w0 = 0; w0 < nwords; w0 += 64) {
            const int w = w0 + lane;
            uint32_t bits = (w < nwords) ? mask[w] : 0u;
            const uint32_t cnt = __popc(bits);
            uint32_t pre = cnt;
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t v = __shfl_up(pre, off, 64);
                if (lane >= off) pre += v;
            }
            uint32_t pos = base + pre - cnt;
            if (!count_only)
                while (bits) {
                    const int b = __ffs(bits) - 1;
                    bits &= bits - 1;
                    out[pos++] = (uint16_t)(w * 32 + b);
                }
            base += __shfl(pre, 63, 64);
        }
        if (lane == 0) {
            if (retry_mode == 1 && base > retry_above) {
                retry_groups[atomicAdd(retry_len, 1ull)] = (int32_t)group;
            } else {
                if (retry_mode == 0 && base > retry_above) atomicAdd(retry_len, 1ull);  // (only counted: for the policy)
                if (!count_only) {
                    ucount[group] = base;
                    atomicAdd(&sched_ctr[sched_bin(base)], 1u);
                }
                atomicAdd(sum_out, (unsigned long long)base);
            }
        }
    }
    PM_STAMP(5);   // (list written)
}

// ---- 3. exact arg-min over the marked prototypes (float64 MFMA on gathered rows) -----------------
// 128 gathered samples x SJ = 16 JTL listed prototypes per step; 4 wavefronts x 32 samples or 8 x 16
// (NWV), 3-stage LDS-DMA ring.  Three instantiations (JTL = 1, 2, 3), each for the workgroups whose list
// length falls in its class (<= 16, 17..32, > 32) so that the gathered X tile is streamed once for all
// but the longest lists (a fourth class of 64 was measured: no gain at C3 / C4, slower at C2): as one
// launch (subset_exact_all_kernel / subset_exact_split_kernel, behind the workgroup's body) or, k = 2 and
// beside the refinement, as three launches on three streams (subset_exact_kernel).
// NS = stages of the ring (NS - 1 tiles in flight).  3 where the chip is full of workgroups; the 64-sample
// workgroups of a small sample set (one or two rounds of workgroups) take as many stages as fit four
// workgroups per CU: 6 / 5 / 4 for JTL = 1 / 2 / 3 (C2 stage 99 -> 91 us, a 125 k-row share of C4 235 -> 212).
#ifndef DBGSOM_QUAD_MAX
#define DBGSOM_QUAD_MAX 3
#endif
constexpr int QUAD_MAX = DBGSOM_QUAD_MAX;   // a last tile with up to 4 QUAD_MAX entries goes as groups of four
// (stages of the 64-sample workgroups' ring: what fits four workgroups per CU, 40 KB each)
constexpr int split_ring_stages(int jtl, int xs) {
    const int n = (40 * 1024) / (64 * 16 * xs + jtl * 16 * 16 * 8);
    return n > 6 ? 6 : (n < 3 ? 3 : n);
}
template <typename XT, int JTL, int SPLIT, int NS>
constexpr int subset_exact_lds_bytes() { return NS * ((128 / SPLIT) * KT * (int)sizeof(XT) + 16 * JTL * KT * 8); }

// The stage's workgroup: `block` of its class's launch (or of its class's slice of a launch that holds all three
// classes), `smem` the launch's one LDS object.
template <typename XT, int JTL, int NWV, int SPLIT = 1, int K = 1, int NS = 3>
__device__ __forceinline__ void subset_exact_workgroup(
    char *__restrict__ smem, const unsigned block,
    const XT *__restrict__ X, int64_t N, int d, int64_t ldx, const double *__restrict__ xx,
    const double *__restrict__ W, int M, const double *__restrict__ ww,
    const int32_t *__restrict__ order, const uint16_t *__restrict__ ulist, int ulist_stride,
    const uint32_t *__restrict__ ucount, const int32_t *__restrict__ sched,
    const uint32_t *__restrict__ sched_range, int round_f32, int64_t *__restrict__ idx_out,
    double *__restrict__ dist_out) {
    constexpr int SJ = 16 * JTL;
    // NWV wavefronts x 16 IT samples: 4 x 32 (two sample tiles per wavefront) or 8 x 16 (one: half
    // the accumulators, twice the wavefronts per SIMD)
    // SPLIT = 2 (small sample counts: fewer than ~4 workgroups per CU): a 128-sample bucket is done by
    // two workgroups of 4 x 16 samples that share its candidate list -- half the serial chain per
    // workgroup, twice the workgroups to balance over the CUs
    static_assert((NWV == 4 || NWV == 8) && (SPLIT == 1 || (SPLIT == 2 && NWV == 4)), "4 or 8 wavefronts");
    constexpr int IT = 8 / (NWV * SPLIT), WS = 16 * IT, RW = 128 / SPLIT;
    // X tile: 128 rows x KT values, float32 (64-byte rows) or float64 (128-byte rows, laid out like W)
    constexpr int XROW = KT * (int)sizeof(XT), XCH = XROW / 16, XD = RW * XROW / 1024 / NWV;
    constexpr int S_XT = RW * XROW, S_WT = SJ * KT * 8, S_STAGE = S_XT + S_WT;  // 8 / 16 KB (SPLIT: half) + 2 JTL KB
    static_assert(NS >= 3 && NS <= 8, "3 .. 8 stages");
    static_assert(NS * S_STAGE == subset_exact_lds_bytes<XT, JTL, SPLIT, NS>(), "the launch's LDS object");
    // this class's slice of the schedule (2b): entry `block` of it, nothing beyond its end
    const uint32_t *range = sched_range + 2 * (3 - JTL);
    const unsigned entry = block / SPLIT, part = block % SPLIT;
    if (entry >= range[1]) return;
    const int wg = sched[range[0] + entry];
    const int cnt = (int)ucount[wg];

    XT_DECL;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // 4 waves x 32 samples
    const int lr = lane & 15, lq = lane >> 4;
    const int64_t p0 = (int64_t)wg * 128 + (int64_t)part * RW;
    if (p0 >= N) return;  // (the second half of a last, partial bucket)
    const uint16_t *list = ulist + (size_t)wg * ulist_stride;

    // Set-up: a workgroup lives ~35 us at C3 and its first tile used to land after 6 -- four dependent round
    // trips (schedule -> sample ids -> |x|^2 -> list entries -> DMA).  Everything that depends on the bucket alone
    // (the sample ids of the lane's results and of its DMA rows, the first step's list entries) is loaded in ONE
    // round trip, the first tiles' DMAs follow, and |x|^2 -- needed behind the first step only -- comes last.
    double xi[IT];
    int64_t isamp[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int64_t p = p0 + wave * WS + it * 16 + lr;
        isamp[it] = order[p < N ? p : N - 1];
    }
    Best<K> best[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) best[it].init();

    // DMA sources. X tile: 8 (f32) / 16 (f64) instructions, wave w issues q = XD w .. XD w + XD - 1
    // = the rows of its own 32 samples
    const XT *xsrc[XD];
    int xrow[XD];
#pragma unroll
    for (int u = 0; u < XD; ++u) {
        int64_t p = p0 + (64 * (XD * wave + u) + lane) / XCH;
        xrow[u] = order[p < N ? p : N - 1];
    }
    // W tile: SJ rows x 128 B = 2 JTL instructions (8 rows each), issued by waves 0 .. 2 JTL - 1
    // (JTL = 3: waves 0, 1 take two)
    constexpr int W_INSTR = 2 * JTL;
    // 4 wavefronts, 6 instructions: 2, 2, 1, 1; 8 wavefronts: one each for waves 0 .. W_INSTR - 1
    const int n_wdma = NWV == 8 ? (wave < W_INSTR ? 1 : 0)
                                : ((wave < W_INSTR - 4) ? 2 : (wave < W_INSTR ? 1 : 0));
    const int wq0 = NWV == 8 ? wave
                             : ((wave < W_INSTR - 4) ? 2 * wave : (W_INSTR > 4 ? wave + (W_INSTR - 4) : wave));
    const int wlr = lane >> 3, wcp = lane & 7;
    int wfirst[2] = {0, 0};   // (the first step's list entries of the lane's W rows)
#pragma unroll
    for (int u = 0; u < 2; ++u)
        if (u < n_wdma) {
            const int pos = 8 * (wq0 + u) + wlr;
            wfirst[u] = (int)list[pos < cnt ? pos : cnt - 1];
        }
#pragma unroll
    for (int u = 0; u < XD; ++u) {
        const int L = 64 * (XD * wave + u) + lane;
        const int r = L / XCH, cp = L % XCH;
        const int c = cp ^ ((r >> 1) & (XCH - 1));
        xsrc[u] = X + (int64_t)xrow[u] * ldx + c * (16 / (int)sizeof(XT));
    }

    const int nkt = d / KT;
    const int nstep = (cnt + SJ - 1) / SJ;
    const int ntile = nkt * nstep;

    // issue side of the ring.  The W rows of a step (their list entries are global loads) are
    // looked up ONCE per step, not per tile: a load in front of every DMA would drain the whole
    // ring (vmcnt counts in order) each tile.
    int i_kt = 0, i_step = 0, i_stage = 0;
    const double *wrow[2] = {W, W};
#pragma unroll
    for (int u = 0; u < 2; ++u)
        if (u < n_wdma) {
            const int wr = 8 * (wq0 + u) + wlr;
            wrow[u] = W + (int64_t)wfirst[u] * d + (wcp ^ ((wr >> 1) & 7)) * 2;
        }
    auto issue = [&]() {
        if (i_kt == 0 && i_step > 0) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (u < n_wdma) {
                    const int wr = 8 * (wq0 + u) + wlr;
                    const int wc = (wcp ^ ((wr >> 1) & 7)) * 2;
                    int pos = i_step * SJ + wr;
                    pos = pos < cnt ? pos : cnt - 1;
                    wrow[u] = W + (int64_t)list[pos] * d + wc;
                }
        }
        const int k0 = i_kt * KT;
        char *stage = smem + i_stage;
#pragma unroll
        for (int u = 0; u < XD; ++u) fdma16(xsrc[u] + k0, stage + 1024 * (XD * wave + u));
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (u < n_wdma) fdma16(wrow[u] + k0, stage + S_XT + 1024 * (wq0 + u));
        i_stage = (i_stage == (NS - 1) * S_STAGE) ? 0 : i_stage + S_STAGE;
        if (++i_kt == nkt) { i_kt = 0; ++i_step; }
    };

    // Fragment addresses inside a stage.  A row of a tile is 128 bytes (W, float64 X) or 64 (float32 X) in 16-byte
    // chunks, chunk c of row r stored at c ^ swizzle(r) with swizzle = (r >> 1) & 7 (& 3).  Tiles start at
    // multiples of 16 rows and the rows of a group of four at multiples of 4, so the swizzle of a lane's row is the
    // same in every tile, and the chunk of k-step ks is the chunk of k-step 0 with 2 ks (float32 X: ks) xor-ed in:
    // ONE lane-dependent offset per operand, the tile as an immediate, the k-step as an xor of the final address
    // (a stage is a multiple of 128 bytes) -- not a register per (tile, k-step).
    const int hq = lq >> 1;
    const int a_base = S_XT + lr * 128 + (lq & 1) * 8 + 16 * (hq ^ ((lr >> 1) & 7));
    const int aq_row = lane & 3;   // (groups of four, below: rows 4 u + (lane & 3) behind the full tiles)
    const int aq_base = S_XT + aq_row * 128 + (lq & 1) * 8 + 16 * (hq ^ (aq_row >> 1));
    constexpr int B_TILE = 16 * XROW, B_KS = sizeof(XT) == 4 ? 16 : 32;
    const int b_base = sizeof(XT) == 4 ? (wave * WS + lr) * XROW + lq * 4 + 16 * ((lr >> 1) & 3)
                                       : (wave * WS + lr) * XROW + (lq & 1) * 8 + 16 * (hq ^ ((lr >> 1) & 7));
    static_assert(S_STAGE % 128 == 0 && S_XT % 128 == 0, "the k-step is an xor of the address");
    using xfrag_t = std::conditional_t<sizeof(XT) == 4, float, double>;

#pragma unroll
    for (int u = 0; u < NS - 1; ++u)
        if (ntile > u) issue();
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        xi[it] = xx[isamp[it]];
        if (p0 + wave * WS + it * 16 + lr >= N) isamp[it] = -1;   // (a last, partial bucket: nothing stored)
    }
    int t = 0, r_stage = 0;
    // One step of the list: its nkt k-tiles on the first JE 16-prototype tiles, then the candidates' distances.
    // JE is a compile-time figure -- the last step of a list is short (65 entries in 48-entry steps: 48 + 17, two
    // tiles, not three), and a uniform branch per tile INSIDE the k loop kept the compiler from issuing the
    // fragment reads of a tile ahead of its products (one exposed LDS round trip per product group).
    //
    // NQ groups of FOUR prototypes behind the JE full tiles, on v_mfma_f64_4x4x4_4b: four 4x4x4 blocks per
    // instruction, the same four prototypes in every block's A (row lane & 3 of the group, k = lane >> 4), the
    // wavefront's 16 samples over the blocks' B columns -- the B fragment of the 16x16x4 form as it is -- and the
    // result of sample (lane & 15) x prototype (lane >> 4) in one register pair: 16 cycles of the matrix pipe for
    // four prototypes instead of 64 for a quarter-filled tile, the same sequential chain bit for bit
    // (tools/probe_mfma_f64_4x4.hip).  Lists are a cluster's prototypes (20 .. 45 at C4), so a list's last tile
    // is a quarter to three quarters empty more often than not.
    auto run_step = [&](auto je_c, auto nq_c, const int st) {
        constexpr int JE = decltype(je_c)::value, NQ = decltype(nq_c)::value;
        static_assert(JE + (NQ > 0) <= JTL && JE + NQ >= 1 && NQ <= 3, "tiles of the step");
        d4_t acc[JE > 0 ? JE : 1][IT];
        double accq[NQ > 0 ? NQ : 1][IT];
#pragma unroll
        for (int u = 0; u < (NQ > 0 ? NQ : 1); ++u)
#pragma unroll
            for (int it = 0; it < IT; ++it) accq[u][it] = 0.0;
#pragma unroll
        for (int jt = 0; jt < JE; ++jt)
#pragma unroll
            for (int it = 0; it < IT; ++it) acc[jt][it] = d4_t{0.0, 0.0, 0.0, 0.0};
        for (int kt = 0; kt < nkt; ++kt, ++t) {
            // each wavefront waits for ITS OWN DMAs of tile t, the barrier then covers everybody's: what may stay
            // in flight are the tiles issued behind it -- NS - 2 of them, fewer at the end of the walk
            {
                const int behind = ntile - 1 - t;
#define DBGSOM_WAIT_BEHIND(Q)                                                                        \
                if (n_wdma == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((Q) * (XD + 2)) : "memory");      \
                else if (n_wdma == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((Q) * (XD + 1)) : "memory"); \
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((Q) * XD) : "memory")
                static_assert((NS - 2) * (XD + 2) <= 63, "vmcnt is a 6-bit counter");
                if (behind >= NS - 2) { DBGSOM_WAIT_BEHIND(NS - 2); }
                else if (NS > 3 && behind == 1) { DBGSOM_WAIT_BEHIND(1); }
                else if (NS > 4 && behind == 2) { DBGSOM_WAIT_BEHIND(NS > 4 ? 2 : 0); }
                else if (NS > 5 && behind == 3) { DBGSOM_WAIT_BEHIND(NS > 5 ? 3 : 0); }
                else if (NS > 6 && behind == 4) { DBGSOM_WAIT_BEHIND(NS > 6 ? 4 : 0); }
                else if (NS > 7 && behind == 5) { DBGSOM_WAIT_BEHIND(NS > 7 ? 5 : 0); }
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef DBGSOM_WAIT_BEHIND
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            XT_MARK_ONCE(4);   // (the first tile has landed)
            if (t + (NS - 1) < ntile) issue();
            const int a_st = r_stage + a_base, aq_st = r_stage + aq_base, b_st = r_stage + b_base;
            r_stage = (r_stage == (NS - 1) * S_STAGE) ? 0 : r_stage + S_STAGE;
            // every fragment of the tile first, then the products (8 wavefronts of 80 registers: two tiles + two or
            // three groups in two halves, or the fragments spill inside the loop)
            constexpr int KB = (NWV == 8 && sizeof(XT) == 4 && JE + NQ > 3) ? KT / 8 : KT / 4;
#pragma unroll
            for (int k0 = 0; k0 < KT / 4; k0 += KB) {
                double a[KB][JE > 0 ? JE : 1], aq[KB][NQ > 0 ? NQ : 1];
                xfrag_t braw[KB][IT];
#pragma unroll
                for (int kk = 0; kk < KB; ++kk) {
                    const int ks = k0 + kk;
#pragma unroll
                    for (int u = 0; u < JE; ++u)
                        a[kk][u] = *reinterpret_cast<const double *>(smem + (a_st ^ (32 * ks)) + u * 2048);
#pragma unroll
                    for (int u = 0; u < NQ; ++u)   // chunk (2 ks + hq) ^ (2 u + (row >> 1)) = 2 (ks ^ u) ^ (hq ^ (row >> 1))
                        aq[kk][u] = *reinterpret_cast<const double *>(smem + aq_st + (16 * JE + 4 * u) * 128 + 32 * (ks ^ u));
#pragma unroll
                    for (int u = 0; u < IT; ++u)
                        braw[kk][u] = *reinterpret_cast<const xfrag_t *>(smem + (b_st ^ (B_KS * ks)) + u * B_TILE);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < KB; ++kk) {
#pragma unroll
                    for (int jt = 0; jt < JE; ++jt)
#pragma unroll
                        for (int it = 0; it < IT; ++it)
                            acc[jt][it] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk][jt], (double)braw[kk][it], acc[jt][it],
                                                                               0, 0, 0);
#pragma unroll
                    for (int u = 0; u < NQ; ++u)
#pragma unroll
                        for (int it = 0; it < IT; ++it)
                            accq[u][it] = __builtin_amdgcn_mfma_f64_4x4x4f64(aq[kk][u], (double)braw[kk][it], accq[u][it], 0, 0, 0);
                }
                if constexpr (KB < KT / 4) __builtin_amdgcn_sched_barrier(0);
            }
        }
        // list entries and |w|^2 of the lane's candidates, 4 at a time: loads first (clamped positions, no
        // branches), so that their latencies overlap
#pragma unroll
        for (int jt = 0; jt < JE; ++jt) {
            int jv[4];
            double yv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pos = st * SJ + jt * 16 + 4 * r + lq;
                jv[r] = (int)list[pos < cnt ? pos : cnt - 1];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) yv[r] = ww[jv[r]];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // (no branch around the use of the loaded values: a load that is consumed on some paths only is
                //  still "in flight" for the compiler when the k loop starts again, and it then answers the first
                //  reuse of its register -- a fragment read, a DMA address -- with s_waitcnt vmcnt(0): the ring drained
                //  on every tile.  A position behind the end of the list pushes +inf, which never wins.)
                const bool listed = st * SJ + jt * 16 + 4 * r + lq < cnt;
#pragma unroll
                for (int it = 0; it < IT; ++it) {
                    double rv = (xi[it] + (-2.0 * acc[jt][it][r])) + yv[r];
                    if (!(rv > 0.0)) rv = (rv != rv) ? rv : 0.0;
                    best[it].push(listed ? rv : (double)INFINITY, jv[r]);  // list ascends -> j ascends per lane
                }
            }
        }
        if constexpr (NQ > 0) {
            // the groups: the lane's prototype is entry 4 u + (lane >> 4) behind the full tiles -- j still ascends
            int jq[NQ];
            double yq[NQ];
#pragma unroll
            for (int u = 0; u < NQ; ++u) {
                const int pos = st * SJ + JE * 16 + 4 * u + lq;
                jq[u] = (int)list[pos < cnt ? pos : cnt - 1];
            }
#pragma unroll
            for (int u = 0; u < NQ; ++u) yq[u] = ww[jq[u]];
#pragma unroll
            for (int u = 0; u < NQ; ++u) {
                const bool listed = st * SJ + JE * 16 + 4 * u + lq < cnt;
#pragma unroll
                for (int it = 0; it < IT; ++it) {
                    double rv = (xi[it] + (-2.0 * accq[u][it])) + yq[u];
                    if (!(rv > 0.0)) rv = (rv != rv) ? rv : 0.0;
                    best[it].push(listed ? rv : (double)INFINITY, jq[u]);
                }
            }
        }
    };
    for (int st = 0; st < nstep; ++st) {
        if constexpr (K != 1) {
            // (k = 2 walks whole steps: tiles behind the end of the list hold no entries and are not pushed)
            run_step(std::integral_constant<int, JTL>{}, std::integral_constant<int, 0>{}, st);
        } else {
            // the step's entries as full tiles + groups of four: a last tile with up to 12 entries goes as groups
            int e = cnt - st * SJ;
            e = e > SJ ? SJ : e;
            const int tiles = (e + 15) / 16, rem = e - 16 * (tiles - 1);
            const int full = rem <= 4 * QUAD_MAX ? tiles - 1 : tiles, nq = rem <= 4 * QUAD_MAX ? (rem + 3) / 4 : 0;
#define DBGSOM_STEP(F, Q) run_step(std::integral_constant<int, F>{}, std::integral_constant<int, Q>{}, st)
#define DBGSOM_STEP_Q(F)                                                                                  \
            do {                                                                                          \
                if (nq == 0) { if constexpr (F > 0) DBGSOM_STEP(F, 0); }                                  \
                else if constexpr (F < JTL) {                                                             \
                    if (nq == 1) DBGSOM_STEP(F, 1); else if (nq == 2) DBGSOM_STEP(F, 2); else DBGSOM_STEP(F, 3); \
                }                                                                                         \
            } while (0)
            if (full == 0) DBGSOM_STEP_Q(0);
            else if (full == 1) DBGSOM_STEP_Q(1);
            else if constexpr (JTL >= 2) {
                if (full == 2) DBGSOM_STEP_Q(2);
                else if constexpr (JTL >= 3) DBGSOM_STEP_Q(3);
            }
#undef DBGSOM_STEP_Q
#undef DBGSOM_STEP
        }
    }
    XT_MARK(5);   // (the last step's distances are pushed)
#pragma unroll
    for (int it = 0; it < IT; ++it) {
#pragma unroll
        for (int m = 16; m <= 32; m <<= 1) {
            double ov[K];
            int oj[K];
#pragma unroll
            for (int u = 0; u < K; ++u) {
                ov[u] = __shfl_xor(best[it].v[u], m, 64);
                oj[u] = __shfl_xor(best[it].j[u], m, 64);
            }
            best[it].merge(ov, oj);
        }
        if (lq == 0 && isamp[it] >= 0) {
#pragma unroll
            for (int u = 0; u < K; ++u) {
                double dv = sqrt(best[it].v[u]);
                if (round_f32) dv = (double)(float)dv;
                idx_out[isamp[it] * K + u] = (best[it].j[u] == 0x7fffffff) ? (int64_t)-1 : (int64_t)best[it].j[u];
                dist_out[isamp[it] * K + u] = dv;
            }
        }
    }
    XT_MARK(1);
    XT_FLUSH(JTL, cnt, dist_out, isamp[0], K);
}

template <typename XT, int JTL, int NWV, int SPLIT = 1, int K = 1, int NS = 3>
__global__ __launch_bounds__(NWV * 64, (NWV == 8 && sizeof(XT) == 4) ? 6 : 4) void subset_exact_kernel(
    const XT *__restrict__ X, int64_t N, int d, int64_t ldx, const double *__restrict__ xx,
    const double *__restrict__ W, int M, const double *__restrict__ ww,
    const int32_t *__restrict__ order, const uint16_t *__restrict__ ulist, int ulist_stride,
    const uint32_t *__restrict__ ucount, const int32_t *__restrict__ sched,
    const uint32_t *__restrict__ sched_range, int round_f32, int64_t *__restrict__ idx_out,
    double *__restrict__ dist_out) {
    __shared__ __attribute__((aligned(16))) char smem[subset_exact_lds_bytes<XT, JTL, SPLIT, NS>()];
    subset_exact_workgroup<XT, JTL, NWV, SPLIT, K, NS>(smem, blockIdx.x, X, N, d, ldx, xx, W, M, ww, order, ulist, ulist_stride,
                                                       ucount, sched, sched_range, round_f32, idx_out, dist_out);
}

// Few sample buckets (one or two rounds of workgroups: C2, a rank's share): the three list-length classes of the
// 64-sample workgroups in ONE launch, class by class along blockIdx.y (long lists are dispatched first) -- the same
// workgroups as three launches on three streams, without the fork and the join of the streams (13 + 16 us of a
// 0.4 ms epoch).  One block shape (4 wavefronts), one LDS size (40 KB: every class's ring is as deep as fits).
template <typename XT>
__global__ __launch_bounds__(256, 4) void subset_exact_split_kernel(
    const XT *__restrict__ X, int64_t N, int d, int64_t ldx, const double *__restrict__ xx,
    const double *__restrict__ W, int M, const double *__restrict__ ww,
    const int32_t *__restrict__ order, const uint16_t *__restrict__ ulist, int ulist_stride,
    const uint32_t *__restrict__ ucount, const int32_t *__restrict__ sched,
    const uint32_t *__restrict__ sched_range, int round_f32, int64_t *__restrict__ idx_out,
    double *__restrict__ dist_out) {
    constexpr int XS = (int)sizeof(XT);
    constexpr int NS3 = split_ring_stages(3, XS), NS2 = split_ring_stages(2, XS), NS1 = split_ring_stages(1, XS);
    constexpr int B3 = subset_exact_lds_bytes<XT, 3, 2, NS3>(), B2 = subset_exact_lds_bytes<XT, 2, 2, NS2>(),
                  B1 = subset_exact_lds_bytes<XT, 1, 2, NS1>();
    __shared__ __attribute__((aligned(16))) char smem[B3 > B2 ? (B3 > B1 ? B3 : B1) : (B2 > B1 ? B2 : B1)];
    if (blockIdx.y == 0)
        subset_exact_workgroup<XT, 3, 4, 2, 1, NS3>(smem, blockIdx.x, X, N, d, ldx, xx, W, M, ww, order, ulist, ulist_stride,
                                                    ucount, sched, sched_range, round_f32, idx_out, dist_out);
    else if (blockIdx.y == 1)
        subset_exact_workgroup<XT, 2, 4, 2, 1, NS2>(smem, blockIdx.x, X, N, d, ldx, xx, W, M, ww, order, ulist, ulist_stride,
                                                    ucount, sched, sched_range, round_f32, idx_out, dist_out);
    else
        subset_exact_workgroup<XT, 1, 4, 2, 1, NS1>(smem, blockIdx.x, X, N, d, ldx, xx, W, M, ww, order, ulist, ulist_stride,
                                                    ucount, sched, sched_range, round_f32, idx_out, dist_out);
}

// The same for many sample buckets: the 128-sample workgroups of all three classes in one launch, every class as
// 8 wavefronts x 16 samples with the long-list class's LDS size (three workgroups per CU whatever the class: a slot
// that a workgroup leaves fits any other, and the dispatch order IS the schedule's -- long lists first, the short
// ones at the end of the stage).  Against three launches side by side (4 x 32 samples for the two short classes,
// measured the better shape for them): C4 stage 1.036 -> 1.027 ms, epoch 2.220 -> 2.190; C3 0.401 -> 0.381 / 0.998 -> 0.977.
template <typename XT>
__global__ __launch_bounds__(512, sizeof(XT) == 4 ? 6 : 4) void subset_exact_all_kernel(
    const XT *__restrict__ X, int64_t N, int d, int64_t ldx, const double *__restrict__ xx,
    const double *__restrict__ W, int M, const double *__restrict__ ww,
    const int32_t *__restrict__ order, const uint16_t *__restrict__ ulist, int ulist_stride,
    const uint32_t *__restrict__ ucount, const int32_t *__restrict__ sched,
    const uint32_t *__restrict__ sched_range, int round_f32, int64_t *__restrict__ idx_out,
    double *__restrict__ dist_out) {
    __shared__ __attribute__((aligned(16))) char smem[subset_exact_lds_bytes<XT, 3, 1, 3>()];
    // Launch order = dispatch order = the schedule's: long lists first, the short ones at the end of the stage.
    // (The long-list and the middle class ALTERNATING while both last -- one alone asks for all of the matrix pipe,
    //  the other for three quarters of it -- was measured: C4 stage 1.09 -> 1.21 ms.)
    const unsigned n3 = sched_range[1], n2 = sched_range[3], n1 = sched_range[5];
    const unsigned b = blockIdx.x;
    if (b >= n3 + n2 + n1) return;
    const unsigned cls = b < n3 ? 3u : (b < n3 + n2 ? 2u : 1u), e = b < n3 ? b : (b < n3 + n2 ? b - n3 : b - n3 - n2);
    if (cls == 3u)
        subset_exact_workgroup<XT, 3, 8, 1, 1, 3>(smem, e, X, N, d, ldx, xx, W, M, ww, order, ulist, ulist_stride,
                                                  ucount, sched, sched_range, round_f32, idx_out, dist_out);
    else if (cls == 2u)
        subset_exact_workgroup<XT, 2, 8, 1, 1, 3>(smem, e, X, N, d, ldx, xx, W, M, ww, order, ulist, ulist_stride,
                                                  ucount, sched, sched_range, round_f32, idx_out, dist_out);
    else
        subset_exact_workgroup<XT, 1, 8, 1, 1, 3>(smem, e, X, N, d, ldx, xx, W, M, ww, order, ulist, ulist_stride,
                                                  ucount, sched, sched_range, round_f32, idx_out, dist_out);
}

#include "refine.h"

// ---- launchers ----------------------------------------------------------------------------------
struct PlaneBuf {
    int8_t *planes;
    double *scale, *l1, *res16;
};
static size_t carve_planes(PlaneBuf *b, char *base, int64_t rows, int64_t d) {
    const int64_t dpad = filter_dpad(d);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return o; };
    const size_t o0 = take((size_t)3 * rows * dpad), o1 = take((size_t)rows * 8), o2 = take((size_t)rows * 8);
    const size_t o3 = take((size_t)rows * 8);
    if (b) {
        b->planes = (int8_t *)(base + o0); b->scale = (double *)(base + o1); b->l1 = (double *)(base + o2);
        b->res16 = (double *)(base + o3);
    }
    return off;
}

struct FilterWs {
    uint32_t *tickets;     // [0] tile score/select, [1] slice W/tables: "last workgroup" tickets (offset 0 of
                           // the workspace whatever its shape; 0 between launches, see last_workgroup_done)
    int8_t *wt, *wt_sub;   // k-tile-major digit planes of the prototypes / of the pre-pass subset
    double *wscale, *wl1;  // M each
    double *wn0;           // M: sum of the squared top digits of a row (section 2c)
    double *wres16;        // M: |w - w16|, the residual of the top two digit planes (section 2d)
    double *ctab, *yypad, *ctab_sub, *yy_sub, *ictab, *yctab, *yy_part, *summary;
    float *tab32;        // 6 x Mpad float32: [yctab | ictab | yy_sub | ctab_sub 2^16 | yy | ctab 2^16] for the 2-per-CU sweep's epilogue
    float *chk32;        // Mpad / 256 x 4 float32: [min yctab, min ictab, max ictab, -] per 256-prototype chunk
    double *tile_score;  // dpad / 64
    double *tile_part;   // TS_RB x 2 x dpad
    int32_t *kt_sel;     // SW_MAX_KT
    uint16_t *ulist;
    uint32_t *ucount;
    int32_t *sched;       // nb   workgroup ids in launch order of the exact stage
    uint32_t *sched_ctr;  // SCHED_CTR counters of that schedule
    int64_t *seed;     // N   arg-min of the coarse pre-pass (when the caller has no previous winners)
    int32_t *order;    // N   bucket order of the samples by seed
    float *gap;        // Mg x Mg lower bounds of the squared distances between prototypes (2c); M <= PRUNE_MAX_M
    int32_t *retry;    // nb: workgroups of the pruning form to be re-seeded
    uint32_t *nnub;    // Mg float32 bit patterns: upper bounds of the prototypes' nearest-neighbour distances (k = 2)
    unsigned long long *cand;  // N: per-sample candidates (four prototype ids) after the refinement (2d)
    int64_t *rbest;    // N: the refinement's best prototype per sample (bucket key of the pair kernel)
    int32_t *order2;   // N: the samples bucketed by it
    int32_t *ovf;      // 2 N: (sample, workgroup) of the samples whose candidates overflowed
    uint16_t *ovf_cand;  // OV_CAP records of OV_REC: [count | candidates] of the first of them (refine.h)
    uint8_t *gflag;    // nb: 1 = refined (pair kernel); 0 = subset_exact_kernel's
    unsigned long long *rf_ctr;  // RF_CTR counters of the refinement (behind sched_ctr, zeroed with it)
    uint32_t *rf_qlen;           // [4]: lengths of the two class queues, of the overflow list, - (zeroed with it)
    int32_t *rf_queue;           // 2 x nb: the class queues
    int64_t Mg;        // its leading dimension: M rounded up to 64
    void *sort_ws;
    int64_t nb, Mpad;
};
static size_t carve_filter(FilterWs *f, char *base, int64_t N, int64_t d, int64_t M) {
    const int64_t Mpad = (M + 511) / 512 * 512, nb = (N + 127) / 128, dpad = filter_dpad(d);  // whole 512-prototype chunks
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return o; };
    const size_t otk = take(256);
    const size_t ow = take((size_t)3 * Mpad * dpad), ows = take((size_t)3 * Mpad * dpad);
    const size_t osc = take((size_t)M * 8), ol1 = take((size_t)M * 8), on0 = take((size_t)M * 8);
    const size_t ores = take((size_t)M * 8);
    const size_t o0 = take((size_t)Mpad * 8), o1 = take((size_t)Mpad * 8), o2 = take(64);
    const size_t o8 = take((size_t)Mpad * 8), o9 = take((size_t)Mpad * 8);
    const size_t o10 = take((size_t)Mpad * 8), o11 = take((size_t)Mpad * 8), o12 = take((size_t)Mpad * 8);
    const size_t o13 = take((size_t)(dpad / FKT) * 8), o14 = take((size_t)SW_MAX_KT * 4);
    const size_t o18 = take((size_t)6 * Mpad * 4), o19 = take((size_t)(Mpad / 256) * 16);
    const size_t o15 = take((size_t)TS_RB * 2 * dpad * 8);
    const size_t o3 = take((size_t)nb * Mpad * 2), o4 = take((size_t)nb * 4);
    const size_t o5 = take((size_t)N * 8), o6 = take((size_t)N * 4);
    const size_t o7 = take(bucket_sort_workspace_bytes(N, M + 1));   // (+ 1: the "decided" bucket of the deferred form, 2d)
    const size_t o16 = take((size_t)nb * 4), o17 = take((size_t)(SCHED_CTR + 2 * RF_CTR + 4) * 4);
    const size_t o24 = take((size_t)2 * nb * 4);
    const size_t o25 = take((size_t)N * 8), o26 = take((size_t)N * 4), o27 = take((size_t)N * 8);
    const size_t o22 = take((size_t)N * 8), o23 = take((size_t)nb);
    const size_t o29 = take((size_t)(N < OV_CAP ? N : OV_CAP) * OV_REC * 2);
    const int64_t Mg = (M + 63) / 64 * 64;
    const size_t o20 = take(M <= PRUNE_MAX_M ? (size_t)Mg * Mg * 4 : 0);
    const size_t o21 = take((size_t)nb * 4);
    const size_t o28 = take((size_t)Mg * 4);
    if (f) {
        f->retry = (int32_t *)(base + o21);
        f->nnub = (uint32_t *)(base + o28);
        f->ovf_cand = (uint16_t *)(base + o29);
        f->cand = (unsigned long long *)(base + o22); f->gflag = (uint8_t *)(base + o23);
        f->rbest = (int64_t *)(base + o25); f->order2 = (int32_t *)(base + o26); f->ovf = (int32_t *)(base + o27);
        f->rf_ctr = (unsigned long long *)(base + o17 + (size_t)SCHED_CTR * 4);
        f->rf_qlen = (uint32_t *)(base + o17 + (size_t)(SCHED_CTR + 2 * RF_CTR) * 4);
        f->rf_queue = (int32_t *)(base + o24);
        f->gap = M <= PRUNE_MAX_M ? (float *)(base + o20) : nullptr; f->Mg = Mg;
        f->tickets = (uint32_t *)(base + otk);
        f->sched = (int32_t *)(base + o16); f->sched_ctr = (uint32_t *)(base + o17);
        f->wt = (int8_t *)(base + ow); f->wt_sub = (int8_t *)(base + ows);
        f->wscale = (double *)(base + osc); f->wl1 = (double *)(base + ol1); f->wn0 = (double *)(base + on0);
        f->wres16 = (double *)(base + ores);
        f->ctab = (double *)(base + o0); f->yypad = (double *)(base + o1);
        f->ctab_sub = (double *)(base + o8); f->yy_sub = (double *)(base + o9);
        f->ictab = (double *)(base + o10); f->yctab = (double *)(base + o11);
        f->yy_part = (double *)(base + o12);
        f->tile_score = (double *)(base + o13); f->kt_sel = (int32_t *)(base + o14);
        f->tab32 = (float *)(base + o18); f->chk32 = (float *)(base + o19);
        f->tile_part = (double *)(base + o15);
        f->summary = (double *)(base + o2); f->ulist = (uint16_t *)(base + o3);
        f->ucount = (uint32_t *)(base + o4); f->seed = (int64_t *)(base + o5);
        f->order = (int32_t *)(base + o6); f->sort_ws = base + o7; f->nb = nb; f->Mpad = Mpad;
    }
    return off;
}

static int launch_slice(const void *A, int dtype, int64_t rows, int64_t d, int64_t ld,
                        const PlaneBuf &b, hipStream_t s) {
    const int dpad = (int)filter_dpad(d);
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (dtype == DBGSOM_F32)
        hipLaunchKernelGGL(slice_rows_kernel<float>, grid, block, 0, s, (const float *)A, rows, (int)d, ld, dpad, b.planes, b.scale, b.l1, b.res16);
    else if (dtype == DBGSOM_F64)
        hipLaunchKernelGGL(slice_rows_kernel<double>, grid, block, 0, s, (const double *)A, rows, (int)d, ld, dpad, b.planes, b.scale, b.l1, b.res16);
    else
        hipLaunchKernelGGL(slice_rows_kernel<bf16_t>, grid, block, 0, s, (const bf16_t *)A, rows, (int)d, ld, dpad, b.planes, b.scale, b.l1, b.res16);
    return launch_status("slice_rows_kernel");
}

}  // namespace dbgsom

using namespace dbgsom;

// The per-caller state of the filtered search -- stage timer, side streams -- is a FilterAux (common.h): a context
// owns one (FilteredCall::aux), so two contexts driven by one thread keep their own timings and each keeps the
// side-stream overlap on its own device.  Callers of the raw device-level ABI (dbgsom_bmu_filtered,
// dbgsom_filter_timing), which has no handle to hang it on, share this thread's.
namespace {
thread_local dbgsom::FilterAux g_aux;
}  // namespace

extern "C" {

/* diagnostics: when enabled, dbgsom_bmu_filtered records HIP events between its stages;
 * dbgsom_bmu_filtered_stage_ms returns their durations for the LAST call, in milliseconds:
 * [0] slice W + tables, [1] coarse pre-pass (0 when a hint was given), [2] bucket sort,
 * [3] int8 sweep, [4] exact search on the candidates. */
int dbgsom_filter_timing(int enable) { g_aux.timer.enabled = enable != 0; g_aux.timer.valid = false; return DBGSOM_OK; }

int dbgsom_bmu_filtered_stage_ms(double *ms5) { return dbgsom::filter_stage_ms(g_aux, ms5); }

size_t dbgsom_filter_planes_bytes(int64_t rows, int64_t d) {
    if (rows < 1 || d < 1) return 0;
    return carve_planes(nullptr, nullptr, rows, d);
}

int dbgsom_filter_prepare(const void *X_dev, int x_dtype, int64_t N, int64_t d, int64_t ldx,
                          void *planes_dev, size_t planes_bytes, void *stream) {
    DBGSOM_REQUIRE(valid_dtype(x_dtype) && N >= 1 && d >= 1 && ldx >= d && N < 0x7fffffff,
                   "bad samples");
    DBGSOM_REQUIRE(X_dev && planes_dev && is_aligned(planes_dev, 256), "bad pointer");
    if (planes_bytes < dbgsom_filter_planes_bytes(N, d)) {
        set_error("dbgsom_filter_prepare: planes buffer too small");
        return DBGSOM_ENOMEM;
    }
    PlaneBuf b;
    carve_planes(&b, (char *)planes_dev, N, d);
    return launch_slice(X_dev, x_dtype, N, d, ldx, b, (hipStream_t)stream);
}

size_t dbgsom_bmu_filtered_workspace_bytes(int64_t N, int64_t d, int64_t M) {
    if (N < 1 || d < 1 || M < 1) return 0;
    return carve_filter(nullptr, nullptr, N, d, M);
}

/* wavefronts per workgroup of the one-product candidate sweep for this map: 4 (sweep4_i8_kernel,
 * 128 x 256 tile, two workgroups per CU) or 8 (sweep_i8_kernel<0,1,JT>, one per CU).  Measured on
 * the four BASELINE shapes (ms per launch, 8 / 4 wavefronts): C4 1.37 / 1.11, C3 1.12 / 0.88,
 * C5 shard 4.99 / 4.83, C2 0.059 / 0.058 -- the small shape everywhere, although it reads the X
 * plane once per 256 prototypes instead of once per 512 (C5: 16.4 GB per launch, 3.4 TB/s). */
int dbgsom_sweep_shape(int64_t M, int64_t d) {
    static const int forced = [] {  // DBGSOM_SWEEP_SHAPE=4 / 8 forces one
        const char *e = getenv("DBGSOM_SWEEP_SHAPE");
        return e ? atoi(e) : 0;
    }();
    (void)M; (void)d;
    return forced == 8 ? 8 : 4;
}

int dbgsom_bmu_filtered(const void *X_dev, int x_dtype, int64_t N, int64_t d, int64_t ldx,
                        const double *xx_dev, const void *xplanes_dev, const double *W_dev,
                        int64_t M, const double *ww_dev, const int64_t *prev_idx_dev,
                        const int32_t *order_dev, int seed_stride, int sweep_planes,
                        int round_f32, int64_t *idx_dev, double *dist_dev, void *workspace_dev,
                        size_t workspace_bytes, void *stream) {
    FilteredCall a;
    a.X = X_dev; a.x_dtype = x_dtype; a.N = N; a.d = d; a.ldx = ldx; a.xx = xx_dev; a.xplanes = xplanes_dev;
    a.W = W_dev; a.M = M; a.ww = ww_dev; a.prev_idx = prev_idx_dev; a.order = order_dev;
    a.seed_stride = seed_stride & ~DBGSOM_REFINE; a.sweep_planes = sweep_planes; a.round_f32 = round_f32;
    a.idx = idx_dev; a.dist = dist_dev; a.ws = workspace_dev; a.ws_bytes = workspace_bytes;
    a.stream = (hipStream_t)stream;
    a.refine_rows = (seed_stride & DBGSOM_REFINE) ? 192 : 0;
    return launch_bmu_filtered(a);
}

}  // extern "C"

int dbgsom::filter_stage_ms(FilterAux &aux, double *ms5) {
    DBGSOM_REQUIRE(ms5, "null pointer");
    StageTimer &t = aux.timer;
    if (!t.enabled || !t.valid) { set_error("no timed dbgsom_bmu_filtered call"); return DBGSOM_ESTATE; }
    DBGSOM_HIP_CHECK(hipEventSynchronize(t.ev[5]));
    for (int k = 0; k < 5; ++k) {
        float ms = 0.f;
        DBGSOM_HIP_CHECK(hipEventElapsedTime(&ms, t.ev[k], t.ev[k + 1]));
        ms5[k] = ms;
    }
    return DBGSOM_OK;
}

int dbgsom::launch_bmu_filtered(const FilteredCall &call) {
    const void *X_dev = call.X; const int x_dtype = call.x_dtype; const int64_t N = call.N, d = call.d, ldx = call.ldx;
    const double *xx_dev = call.xx; const void *xplanes_dev = call.xplanes; const double *W_dev = call.W;
    const int64_t M = call.M; const double *ww_dev = call.ww; const int64_t *prev_idx_dev = call.prev_idx;
    const int32_t *order_dev = call.order; int seed_stride = call.seed_stride, sweep_planes = call.sweep_planes;
    const int round_f32 = call.round_f32; int64_t *idx_dev = call.idx; double *dist_dev = call.dist;
    void *workspace_dev = call.ws; const size_t workspace_bytes = call.ws_bytes; void *stream = (void *)call.stream;
    FilterAux &aux = call.aux ? *call.aux : g_aux;
    StageTimer &g_timer = aux.timer;
    SideStream &g_side = aux.side;
    const double *g_hint_dist = call.hint_dist, *g_hint_shift = call.hint_shift;
    // 8 wavefronts of 64 x 64 tiles (<= 128 VGPRs: four wavefronts per SIMD, two workgroups per CU)
    // or 4 of 64 x 128 (DBGSOM_SWEEP_WAVES=4; two wavefronts per SIMD).  Measured, ms per launch,
    // 4 / 8: C4 1.13 / 1.04, C3 0.88 / 0.71, C5 shard 4.96 / 4.55
    static const int s4_waves = [] {
        const char *e = getenv("DBGSOM_SWEEP_WAVES");
        return e ? atoi(e) : 8;
    }();
#define S4_LAUNCH(MODE_, NB, ...)                                                                   \
    do {                                                                                           \
        if (s4_waves == 8)                                                                         \
            hipLaunchKernelGGL((sweep4_i8_kernel<MODE_, 8>), dim3((unsigned)(NB)), dim3(512), 0, s, __VA_ARGS__); \
        else                                                                                       \
            hipLaunchKernelGGL((sweep4_i8_kernel<MODE_, 4>), dim3((unsigned)(NB)), dim3(256), 0, s, __VA_ARGS__); \
    } while (0)
    // DBGSOM_SEED_FULL: the seed pre-pass looks at EVERY prototype and every feature (as expensive
    // as the sweep it seeds; what weakly clustered data needs -- the engine's policy decides)
    const bool seed_full = (seed_stride & DBGSOM_SEED_FULL) != 0;
    // DBGSOM_PRUNE: candidates from the triangle inequality instead of the sweep (section 2c);
    // DBGSOM_PRUNE_PROBE: the sweep as usual, and beside it what DBGSOM_PRUNE's lists would add up to
    const bool prune = (seed_stride & DBGSOM_PRUNE) != 0 && M <= PRUNE_MAX_M;
    const bool prune_probe = !prune && (seed_stride & DBGSOM_PRUNE_PROBE) != 0 && M <= PRUNE_MAX_M;
    // DBGSOM_PRUNE_RETRY (stateless searches with cheap seeds): workgroups whose pruned lists come out
    // long are re-seeded against every prototype and pruned again (two more short launches)
    const bool prune_retry = (seed_stride & DBGSOM_PRUNE_RETRY) != 0 && !seed_full && !prev_idx_dev;
    // k = 2 (the two nearest prototypes: topographic error, BaseSom.py:945): the pruning form only
    const bool k2 = call.k == 2;
    DBGSOM_REQUIRE(call.k == 1 || call.k == 2, "k must be 1 or 2");
    DBGSOM_REQUIRE(!call.defer_dist || (call.refine_rows > 0 && M < 0xffff), "deferred distances need the refinement");
    DBGSOM_REQUIRE(!k2 || (prune && call.refine_rows == 0 && M >= 2),
                   "k = 2 needs the pruning form (DBGSOM_PRUNE, M <= 8192) without the refinement");
    seed_stride &= ~(DBGSOM_PRUNE | DBGSOM_PRUNE_PROBE | DBGSOM_PRUNE_RETRY);
    seed_stride = seed_full ? 1 : seed_stride;
    DBGSOM_REQUIRE(seed_stride >= 0 && seed_stride <= 64, "seed_stride outside [0, 64]");
    DBGSOM_REQUIRE(sweep_planes >= 0 && sweep_planes <= 3, "sweep_planes must be 0 .. 3");
    if (sweep_planes == 0) sweep_planes = 2;
    DBGSOM_REQUIRE(x_dtype == DBGSOM_F32 || x_dtype == DBGSOM_F64, "the filtered search takes float32 or float64 samples");
    DBGSOM_REQUIRE(N >= 1 && N < 0x7fffffff && d >= 1 && d % KT == 0 && ldx >= d, "bad sample shape (d must be a multiple of 16)");
    DBGSOM_REQUIRE(M >= 1 && M <= SW_MAX_M, "M outside [1, 16000]");
    DBGSOM_REQUIRE(X_dev && xx_dev && xplanes_dev && W_dev && ww_dev && idx_dev && dist_dev &&
                       workspace_dev, "null pointer");
    DBGSOM_REQUIRE((prev_idx_dev == nullptr) == (order_dev == nullptr),
                   "prev_idx and order come as a pair (both NULL = stateless two-pass search)");
    DBGSOM_REQUIRE(is_aligned(X_dev, 16) && (ldx * (int64_t)dtype_size(x_dtype)) % 16 == 0 && is_aligned(W_dev, 16) &&
                       is_aligned(workspace_dev, 256) && is_aligned(xplanes_dev, 256), "alignment");
    if (workspace_bytes < dbgsom_bmu_filtered_workspace_bytes(N, d, M)) {
        set_error("dbgsom_bmu_filtered: workspace too small");
        return DBGSOM_ENOMEM;
    }
    hipStream_t s = (hipStream_t)stream;
    PlaneBuf xb;
    carve_planes(&xb, (char *)const_cast<void *>(xplanes_dev), N, d);
    FilterWs f;
    carve_filter(&f, (char *)workspace_dev, N, d, M);
    const int dpad = (int)filter_dpad(d);
    g_timer.mark(0, s);
    // the seed pre-pass looks at every `seed_stride`-th prototype (any seed keeps the result exact;
    // a coarser pre-pass is cheaper, its seeds are a little further from the minimum)
    // default (0): the stride that makes the subset ONE 256-prototype chunk of the pre-pass, at
    // least 4 -- list lengths barely depend on it (C3: 58 -> 62 from stride 4 to 8, C4 / C5: none)
    if (seed_stride == 0) {
        seed_stride = (int)((M + 255) / 256);
        seed_stride = seed_stride < 4 ? 4 : (seed_stride > 64 ? 64 : seed_stride);
    }
    while (seed_stride > 1 && (M + seed_stride - 1) / seed_stride < 128) seed_stride >>= 1;
    const int Msub = (int)((M + seed_stride - 1) / seed_stride), Msubpad = (Msub + 255) / 256 * 256;
    // ... and at PREPASS_KTILES k-tiles (64 features each) spread evenly over the row, with the
    // matching partial |w|^2: on every workload measured the candidate lists are as short as with
    // all features, the pre-pass costs 0.35 instead of 0.75 ms at C4 (DBGSOM_PREPASS_KTILES=0: all)
    static const int prepass_env = [] {
        const char *e = getenv("DBGSOM_PREPASS_KTILES");
        return e ? atoi(e) : PREPASS_KTILES;
    }();
    const int nkt_full = dpad / FKT;
    const int nkt_used = (!seed_full && prepass_env >= 2 && prepass_env < nkt_full && nkt_full <= SW_MAX_KT) ? prepass_env : nkt_full;
    if (nkt_used < nkt_full) {
        hipLaunchKernelGGL(tile_partial_kernel, dim3((unsigned)nkt_full, TS_RB), dim3(256), 0, s, W_dev,
                           (int)M, (int)d, dpad, f.tile_part);
        hipLaunchKernelGGL(tile_score_select_kernel, dim3((unsigned)nkt_full), dim3(64), 0, s, f.tile_part, (int)M,
                           dpad, f.tile_score, nkt_full, nkt_used, f.kt_sel, f.tickets + 0);
    }
    WTables tables;
    tables.ww = ww_dev; tables.ctab = f.ctab; tables.yypad = f.yypad; tables.ctab_sub = f.ctab_sub;
    tables.yy_sub = f.yy_sub; tables.ictab = f.ictab; tables.yctab = f.yctab; tables.summary = f.summary;
    tables.sched_ctr = f.sched_ctr; tables.tscale = sweep_tscale(sweep_planes); tables.tab32 = f.tab32; tables.chk32 = f.chk32;
    hipLaunchKernelGGL(slice_w_tiled_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, W_dev,
                       (int)M, (int)d, dpad, (int)f.Mpad, seed_stride, Msubpad, nkt_used,
                       nkt_used < nkt_full ? f.kt_sel : (const int32_t *)nullptr, f.wt,
                       f.wt_sub, f.wscale, f.wl1, f.yy_part, f.wn0, f.wres16, f.tickets + 1, tables);
    g_timer.mark(1, s);
    // The gaps between the prototypes (pruning form) need the digit planes of W and nothing of the samples:
    // in a stateless search they are worked out on the second stream BESIDE the seed pre-pass and the bucket
    // sort (one wavefront per 64 x 64 tile: a few hundred small workgroups next to a launch that fills the
    // chip or, on a rank's share of the samples, does not), DBGSOM_GAP_FORK=0 keeps them in line
    auto launch_gap = [&](hipStream_t gs) -> int {
        const unsigned gt = (unsigned)(f.Mg / 64);
        if (k2) DBGSOM_HIP_CHECK(hipMemsetAsync(f.nnub, 0x7f, (size_t)f.Mg * 4, gs));   // (0x7f7f7f7f: 3.4e38, "no bound")
        if (gt * gt >= 1024u)   // (four 64 x 64 tiles per CU and more: M >= 2048)
            hipLaunchKernelGGL(proto_gap_kernel<2>, dim3(gt, gt), dim3(64), 0, gs, f.wt, (int)f.Mpad, dpad, (int)M, (int)d,
                               f.wscale, f.wn0, ww_dev, f.gap, (int)f.Mg, k2 ? f.nnub : (uint32_t *)nullptr);
        else
            hipLaunchKernelGGL(proto_gap_kernel<1>, dim3(2 * gt, 2 * gt), dim3(64), 0, gs, f.wt, (int)f.Mpad, dpad, (int)M,
                               (int)d, f.wscale, f.wn0, ww_dev, f.gap, (int)f.Mg, k2 ? f.nnub : (uint32_t *)nullptr);
        return DBGSOM_OK;
    };
    static const int gap_fork_env = [] {
        const char *e = getenv("DBGSOM_GAP_FORK");
        return e ? atoi(e) : 1;
    }();
    bool gap_aside = false;
    if ((prune || prune_probe) && !prev_idx_dev && gap_fork_env != 0 && g_side.ready()) {
        DBGSOM_HIP_CHECK(hipEventRecord(g_side.gap_fork, s));
        DBGSOM_HIP_CHECK(hipStreamWaitEvent(g_side.stream, g_side.gap_fork, 0));
        const int rc = launch_gap(g_side.stream);
        if (rc != DBGSOM_OK) return rc;
        DBGSOM_HIP_CHECK(hipEventRecord(g_side.gap_done, g_side.stream));
        gap_aside = true;
    }
    if (!prev_idx_dev) {
        // no previous winners: seed = arg-min of a coarser (3-product) sweep, then bucket the samples
        // the pre-pass is as coarse as the sweep it seeds: one product for the one-product sweep
        // (seeds need not be good, only cheap), three otherwise (data on which the coarse bound
        // fails also gets useless seeds from a one-product pre-pass)
        static const int prepass_shape = [] {  // DBGSOM_PREPASS_SHAPE=8: the 8-wavefront pre-pass
            const char *e = getenv("DBGSOM_PREPASS_SHAPE");
            return e ? atoi(e) : 4;
        }();
        if (sweep_planes == 1 && prepass_shape == 4)
            S4_LAUNCH(1, f.nb, xb.planes, xb.scale,
                               xb.l1, xx_dev, N, (int)d, dpad, f.wt_sub, f.tab32 + 2 * (size_t)f.Mpad,
                               f.tab32 + 3 * (size_t)f.Mpad, f.yy_sub,
                               f.ctab_sub, f.summary, Msub, (const int64_t *)nullptr, (const int32_t *)nullptr,
                               f.ulist, (int)f.Mpad, f.ucount, Msubpad, f.seed, seed_stride, nkt_used, f.kt_sel, f.sched_ctr, f.chk32,
                               (const int32_t *)nullptr, (const unsigned long long *)nullptr);
        else if (sweep_planes == 1)
            hipLaunchKernelGGL((sweep_i8_kernel<1, 1, 2>), dim3((unsigned)f.nb), dim3(FNT), 0, s, xb.planes,
                           xb.scale, xb.l1, xx_dev, N, (int)d, dpad, f.wt_sub, f.yy_sub, f.ctab_sub,
                           f.yy_sub, f.ctab_sub, f.summary, Msub, (const int64_t *)nullptr, (const int32_t *)nullptr,
                           f.ulist, (int)f.Mpad, f.ucount, f.seed, seed_stride, Msubpad, nkt_used, f.kt_sel, f.sched_ctr);
        else
            hipLaunchKernelGGL((sweep_i8_kernel<1, 2, 2>), dim3((unsigned)f.nb), dim3(FNT), 0, s, xb.planes,
                           xb.scale, xb.l1, xx_dev, N, (int)d, dpad, f.wt_sub, f.yy_sub, f.ctab_sub,
                           f.yy_sub, f.ctab_sub, f.summary, Msub, (const int64_t *)nullptr, (const int32_t *)nullptr,
                           f.ulist, (int)f.Mpad, f.ucount, f.seed, seed_stride, Msubpad, nkt_used, f.kt_sel, f.sched_ctr);
        g_timer.mark(2, s);
        const int rc = launch_bucket_sort(f.seed, N, M, f.order, f.sort_ws, s);
        if (rc != DBGSOM_OK) return rc;
        prev_idx_dev = f.seed;
        order_dev = f.order;
    } else {
        g_timer.mark(2, s);
    }
    g_timer.mark(3, s);
#define DBGSOM_SWEEP(P, J)                                                                         \
    hipLaunchKernelGGL((sweep_i8_kernel<0, P, J>), dim3((unsigned)f.nb), dim3(FNT), 0, s, xb.planes,   \
                       xb.scale, xb.l1, xx_dev, N, (int)d, dpad, f.wt, f.yctab, f.ictab, f.yypad,    \
                       f.ctab, f.summary, (int)M, prev_idx_dev, order_dev, f.ulist, (int)f.Mpad,     \
                       f.ucount, (int64_t *)nullptr, 1, (int)f.Mpad, 0, (const int32_t *)nullptr, f.sched_ctr)
    if (prune || prune_probe) {
        if (gap_aside) {
            DBGSOM_HIP_CHECK(hipStreamWaitEvent(s, g_side.gap_done, 0));
        } else {
            const int rc = launch_gap(s);
            if (rc != DBGSOM_OK) return rc;
        }
        unsigned long long *sum = reinterpret_cast<unsigned long long *>(f.sched_ctr + SCHED_SUM) + (prune ? 0 : 1);
        unsigned long long *rlen = reinterpret_cast<unsigned long long *>(f.sched_ctr + SCHED_RETRY);
        const uint32_t retry_above = (uint32_t)(M / 8 > 96 ? M / 8 : 96);
        // (the count is kept either way: the engine turns the re-seeding on when a call reports any)
        hipLaunchKernelGGL(prune_mark_kernel, dim3((unsigned)f.nb), dim3(256), 0, s, xb.planes, xb.scale, xx_dev,
                           N, (int)d, dpad, f.wt, (int)f.Mpad, f.wscale, ww_dev, f.summary, (int)M, prev_idx_dev,
                           order_dev, f.gap, (int)f.Mg, f.ulist, (int)f.Mpad, f.ucount, f.sched_ctr, sum,
                           prune ? 0 : 1, g_hint_shift ? g_hint_dist : (const double *)nullptr, g_hint_shift,
                           f.retry, rlen, prune_retry ? 1 : 0, retry_above, k2 ? f.nnub : (const uint32_t *)nullptr);
        if (prune_retry) {
            // every prototype, every feature, one digit product, for the listed workgroups only
            S4_LAUNCH(1, f.nb, xb.planes, xb.scale, xb.l1, xx_dev, N, (int)d, dpad, f.wt,
                      f.tab32 + 4 * (size_t)f.Mpad, f.tab32 + 5 * (size_t)f.Mpad, f.yypad, f.ctab, f.summary, (int)M,
                      (const int64_t *)nullptr, order_dev, f.ulist, (int)f.Mpad, f.ucount, (int)f.Mpad, f.seed, 1, 0,
                      (const int32_t *)nullptr, f.sched_ctr, f.chk32, (const int32_t *)f.retry, (const unsigned long long *)rlen);
            hipLaunchKernelGGL(prune_mark_kernel, dim3((unsigned)f.nb), dim3(256), 0, s, xb.planes, xb.scale, xx_dev,
                               N, (int)d, dpad, f.wt, (int)f.Mpad, f.wscale, ww_dev, f.summary, (int)M, prev_idx_dev,
                               order_dev, f.gap, (int)f.Mg, f.ulist, (int)f.Mpad, f.ucount, f.sched_ctr, sum,
                               prune ? 0 : 1, (const double *)nullptr, (const double *)nullptr, f.retry, rlen, 2, retry_above,
                               k2 ? f.nnub : (const uint32_t *)nullptr);
        }
    }
    // one digit product: 4-wavefront workgroups unless DBGSOM_SWEEP_SHAPE=8 (see dbgsom_sweep_shape)
    const int sweep_shape = dbgsom_sweep_shape(M, d);
    if (prune) {
        // (no sweep)
    } else if (sweep_planes == 1 && sweep_shape == 4 && order_dev && M <= Sweep4Lds::MAX_M) {
        // one digit product, 4-wavefront workgroups (128 x 256 tile), two of them per CU
        S4_LAUNCH(0, f.nb, xb.planes, xb.scale,
                           xb.l1, xx_dev, N, (int)d, dpad, f.wt, f.tab32, f.tab32 + (size_t)f.Mpad, f.yypad, f.ctab,
                           f.summary, (int)M, prev_idx_dev, order_dev, f.ulist, (int)f.Mpad, f.ucount,
                           (int)f.Mpad, (int64_t *)nullptr, 1, 0, (const int32_t *)nullptr, f.sched_ctr, f.chk32,
                           (const int32_t *)nullptr, (const unsigned long long *)nullptr);
    } else if (sweep_planes == 1) {  // one digit product: 128 x 512 tile (128 x 256 for small maps)
        if (M > 256) DBGSOM_SWEEP(1, 4); else DBGSOM_SWEEP(1, 2);
    } else if (sweep_planes == 3)
        hipLaunchKernelGGL((sweep_i8_kernel<0, 3, 1>), dim3((unsigned)f.nb), dim3(FNT), 0, s, xb.planes,
                           xb.scale, xb.l1, xx_dev, N, (int)d, dpad, f.wt, f.yctab, f.ictab,
                           f.yypad, f.ctab, f.summary, (int)M, prev_idx_dev, order_dev, f.ulist, (int)f.Mpad, f.ucount,
                           (int64_t *)nullptr, 1, (int)f.Mpad, 0, (const int32_t *)nullptr, f.sched_ctr);
    else
        hipLaunchKernelGGL((sweep_i8_kernel<0, 2, 2>), dim3((unsigned)f.nb), dim3(FNT), 0, s, xb.planes,
                           xb.scale, xb.l1, xx_dev, N, (int)d, dpad, f.wt, f.yctab, f.ictab,
                           f.yypad, f.ctab, f.summary, (int)M, prev_idx_dev, order_dev, f.ulist, (int)f.Mpad, f.ucount,
                           (int64_t *)nullptr, 1, (int)f.Mpad, 0, (const int32_t *)nullptr, f.sched_ctr);
    g_timer.mark(4, s);
    if (call.guard_mean > 0.0) {   // (an arm on trial: see FilteredCall::guard_mean)
        unsigned long long sum_h = 0;
        DBGSOM_HIP_CHECK(hipMemcpyAsync(&sum_h, f.sched_ctr + SCHED_SUM, 8, hipMemcpyDeviceToHost, s));
        DBGSOM_HIP_CHECK(hipStreamSynchronize(s));
        if ((double)sum_h > call.guard_mean * (double)f.nb) {
            g_timer.mark(5, s);
            g_timer.valid = g_timer.enabled;
            return DBGSOM_LISTS_LONG;
        }
    }
    // per-sample refinement of the lists (section 2d): workgroups it takes leave the MFMA stage's schedule
    const int rf_rows = call.refine_rows;
    const bool refine = rf_rows > 0;
    // the three list-length classes of the matrix-core stage write disjoint samples: they run side by side
    // (classes 1 and 2 on a second stream forked from the caller's), so that the tail of one launch -- a few
    // long lists on a mostly idle chip -- overlaps the others
    SideStream &side = g_side;
    // (measured also for few buckets, where the fork and join cost ~20 us of bubbles: C2 0.168 -> 0.123
    // ms for the stage, a 125 k-row shard of C4 0.303 -> 0.265; DBGSOM_EXACT_FORK=0 runs them in a row)
    static const int fork_env = [] {
        const char *e = getenv("DBGSOM_EXACT_FORK");
        return e ? atoi(e) : 1;
    }();
    // few sample buckets (C2, a rank's share in strong scaling): two 64-sample workgroups per bucket
    // (DBGSOM_EXACT_SPLIT=0|1 forces).  Measured at C2 (469 buckets): stage 0.164 -> see DESIGN.md
    static const int split_env = [] {
        const char *e = getenv("DBGSOM_EXACT_SPLIT");
        return e ? atoi(e) : -1;
    }();
    // (with the refinement: what is left to this stage is a few workgroups)
    const bool exact_split = split_env >= 0 ? split_env != 0 : (refine || f.nb <= 1024);
    // (k = 1 without the refinement: all three list-length classes go as ONE launch on the caller's stream -- nothing
    //  to fork; DBGSOM_EXACT_MERGED=0: three launches on three streams)
    static const bool merged_env = [] {
        const char *e = getenv("DBGSOM_EXACT_MERGED");
        return e ? atoi(e) != 0 : true;
    }();
    const bool merged = merged_env && !k2 && !refine;
    const bool fork = fork_env != 0 && !merged && side.ready();
    hipStream_t s2 = fork ? side.stream : s, s3 = fork ? side.stream2 : s;
    // With the refinement the matrix-core stage only has the workgroups the refinement does not take (lists
    // beyond its tiles: a few long chains on a mostly idle chip): all of it on the second stream, beside the
    // refinement and the pair kernel on the caller's; the samples whose candidates overflowed on the third.
    hipStream_t s_mfma = refine ? s2 : s;
    const int rows0 = rf_rows <= 32 ? 32 : (rf_rows <= 64 ? 64 : (rf_rows <= 128 ? 128 : 0));
    const int defer_M = (refine && call.defer_dist) ? (int)M : 0;
    if (refine)
        // list-length classes of the refinement: a small tile for the bulk (what the caller expects the
        // lists to be), the largest for the rest; workgroups in neither stay the matrix-core stage's
        hipLaunchKernelGGL(class_fill_kernel, dim3((unsigned)((f.nb + 255) / 256)), dim3(256), 0, s, f.ucount, (int)f.nb,
                           rows0, RF_SEGS * (int)RefineCfg<2, 4>::MAX_CNT, f.rf_queue, f.rf_qlen, f.gflag, f.sched_ctr, order_dev,
                           prev_idx_dev, N, (int)M, f.cand, f.rbest, defer_M);
    else
        // (the bin counts were added up by the sweep's workgroups as they wrote their list lengths)
        hipLaunchKernelGGL(sched_fill_kernel, dim3((unsigned)((f.nb + 255) / 256)), dim3(256), 0, s, f.ucount,
                           (int)f.nb, f.sched_ctr, f.sched, (const uint8_t *)nullptr);
    if (fork) {
        DBGSOM_HIP_CHECK(hipEventRecord(side.forked, s));
        DBGSOM_HIP_CHECK(hipStreamWaitEvent(s2, side.forked, 0));
        DBGSOM_HIP_CHECK(hipStreamWaitEvent(s3, side.forked, 0));
    }
    if (refine) {
        hipLaunchKernelGGL(sched_fill_kernel, dim3((unsigned)((f.nb + 255) / 256)), dim3(256), 0, s_mfma, f.ucount,
                           (int)f.nb, f.sched_ctr, f.sched, (const uint8_t *)f.gflag);
        if (fork) {  // (classes 2 and 1 run on the third stream: behind the schedule)
            DBGSOM_HIP_CHECK(hipEventRecord(side.mid, s_mfma));
            DBGSOM_HIP_CHECK(hipStreamWaitEvent(s3, side.mid, 0));
        }
        // every launch is a few workgroups per CU walking its class's queue
#define DBGSOM_REFINE_LAUNCH(NJ_, JT_, CLS, WGS)                                                                  \
    hipLaunchKernelGGL((refine_i8_kernel<NJ_, JT_>), dim3((unsigned)((f.nb + 7) / 8 * 8 < (WGS) ? (f.nb + 7) / 8 * 8 : (WGS))), dim3(NJ_ * 256), 0, s, \
                       xb.planes, xb.scale, xb.res16, xx_dev, N, (int)d, dpad, f.wt, (int)f.Mpad, f.wscale, ww_dev,    \
                       f.summary, order_dev, f.ulist, (int)f.Mpad, f.ucount, f.rf_queue + (size_t)(CLS) * f.nb,      \
                       f.rf_qlen + (CLS), f.cand, f.rbest, f.rf_ctr, f.ovf, f.rf_qlen + 2, f.ovf_cand, defer_M, idx_dev, dist_dev)
        static const int wgs_env = [] {   // DBGSOM_REFINE_WGS: workgroups per launch (diagnostics; a multiple of 8)
            const char *e = getenv("DBGSOM_REFINE_WGS");
            return e ? atoi(e) / 8 * 8 : 0;
        }();
        const int wgs1 = wgs_env >= 8 ? wgs_env : 1024, wgs2 = wgs_env >= 8 ? wgs_env : 512;
        if (rows0 == 32) DBGSOM_REFINE_LAUNCH(1, 1, 0, wgs1);
        else if (rows0 == 64) DBGSOM_REFINE_LAUNCH(1, 2, 0, wgs1);
        else if (rows0 == 128) DBGSOM_REFINE_LAUNCH(2, 2, 0, wgs2);
        DBGSOM_REFINE_LAUNCH(2, 4, 1, wgs2);
#undef DBGSOM_REFINE_LAUNCH
        // the samples by their refined best prototype: a workgroup of the pair kernel then shares its candidates
        const int64_t Mk = defer_M ? M + 1 : M;   // (deferred: the decided samples behind every real bucket)
        const int rc = launch_bucket_sort(f.rbest, N, Mk, f.order2, f.sort_ws, s);
        if (rc != DBGSOM_OK) return rc;
        const uint32_t *n_active = defer_M ? bucket_sort_seg_start(f.sort_ws, N, Mk) + M : (const uint32_t *)nullptr;
        // (bfloat16-resident samples: the pair kernel reads the stored rows -- half the bytes of the widened
        //  copy the matrix kernels use, the same values)
        const unsigned pgrid = (unsigned)(((N + PS - 1) / PS + 7) / 8 * 8);   // (whole rounds of the 8 XCDs: xcd_group)
        if (call.X_store && call.store_dtype == DBGSOM_BF16 && x_dtype == DBGSOM_F32)
            hipLaunchKernelGGL((pair_exact_kernel<bf16_t, 32, 128>), dim3(pgrid), dim3(256), 0, s, (const bf16_t *)call.X_store, N,
                               (int)d, call.ld_store, xx_dev, W_dev, ww_dev, f.order2, f.cand, round_f32, idx_dev, dist_dev, f.rf_ctr, n_active);
        else if (x_dtype == DBGSOM_F32)
            hipLaunchKernelGGL((pair_exact_kernel<float, 16, 256>), dim3(pgrid), dim3(256), 0, s, (const float *)X_dev, N,
                               (int)d, ldx, xx_dev, W_dev, ww_dev, f.order2, f.cand, round_f32, idx_dev, dist_dev, f.rf_ctr, n_active);
        else
            hipLaunchKernelGGL((pair_exact_kernel<double, 16, 256>), dim3(pgrid), dim3(256), 0, s, (const double *)X_dev, N,
                               (int)d, ldx, xx_dev, W_dev, ww_dev, f.order2, f.cand, round_f32, idx_dev, dist_dev, f.rf_ctr, n_active);
        // (the samples whose candidates overflowed: behind the pair kernel, not beside it -- its uncoalesced
        //  row walks slowed the sort and the pair kernel by more than it takes)
        if (x_dtype == DBGSOM_F32)
            hipLaunchKernelGGL((overflow_exact_kernel<float>), dim3(512), dim3(256), 0, s, (const float *)X_dev, (int)d, ldx,
                               xx_dev, W_dev, ww_dev, f.ulist, (int)f.Mpad, f.ucount, f.ovf, f.rf_qlen + 2, f.ovf_cand, round_f32,
                               idx_dev, dist_dev);
        else
            hipLaunchKernelGGL((overflow_exact_kernel<double>), dim3(512), dim3(256), 0, s, (const double *)X_dev, (int)d, ldx,
                               xx_dev, W_dev, ww_dev, f.ulist, (int)f.Mpad, f.ucount, f.ovf, f.rf_qlen + 2, f.ovf_cand, round_f32,
                               idx_dev, dist_dev);
    }
    // (stages of the 64-sample workgroups' ring: what fits four workgroups per CU, 40 KB each)
#define SPLIT_NS(JTL, XS) split_ring_stages(JTL, XS)
#define DBGSOM_SUBSET_W(JTL, NWV_, STREAM)                                                        \
    do {                                                                                          \
        if (exact_split && x_dtype == DBGSOM_F32)                                                 \
            hipLaunchKernelGGL((subset_exact_kernel<float, JTL, 4, 2, 1, SPLIT_NS(JTL, 4)>), dim3((unsigned)(2 * f.nb)), dim3(256), 0, STREAM, \
                               (const float *)X_dev, N, (int)d, ldx, xx_dev, W_dev, (int)M, ww_dev, \
                               order_dev, f.ulist, (int)f.Mpad, f.ucount, f.sched, f.sched_ctr + 2 * SCHED_BINS, round_f32, idx_dev, dist_dev); \
        else if (exact_split)                                                                     \
            hipLaunchKernelGGL((subset_exact_kernel<double, JTL, 4, 2, 1, SPLIT_NS(JTL, 8)>), dim3((unsigned)(2 * f.nb)), dim3(256), 0, STREAM, \
                               (const double *)X_dev, N, (int)d, ldx, xx_dev, W_dev, (int)M, ww_dev, \
                               order_dev, f.ulist, (int)f.Mpad, f.ucount, f.sched, f.sched_ctr + 2 * SCHED_BINS, round_f32, idx_dev, dist_dev); \
        else if (x_dtype == DBGSOM_F32)                                                           \
            hipLaunchKernelGGL((subset_exact_kernel<float, JTL, NWV_>), dim3((unsigned)f.nb), dim3(NWV_ * 64), 0, STREAM, \
                               (const float *)X_dev, N, (int)d, ldx, xx_dev, W_dev, (int)M, ww_dev, \
                               order_dev, f.ulist, (int)f.Mpad, f.ucount, f.sched, f.sched_ctr + 2 * SCHED_BINS, round_f32, idx_dev, dist_dev); \
        else                                                                                      \
            hipLaunchKernelGGL((subset_exact_kernel<double, JTL, NWV_>), dim3((unsigned)f.nb), dim3(NWV_ * 64), 0, STREAM, \
                               (const double *)X_dev, N, (int)d, ldx, xx_dev, W_dev, (int)M, ww_dev, \
                               order_dev, f.ulist, (int)f.Mpad, f.ucount, f.sched, f.sched_ctr + 2 * SCHED_BINS, round_f32, idx_dev, dist_dev); \
    } while (0)
    // Wavefronts per workgroup of the exact stage: 4 x 32 samples or 8 x 16 (half the accumulators:
    // class 3 goes from 3 to 6 wavefronts per SIMD).  Measured (ms per stage, 4 / 8 / class 3 only):
    // C4 1.23 / 1.23 / 1.16, C3 0.52 / 0.48, C5 shard 5.83 / 5.33 -- 8 for the long lists, 4 for the
    // rest.  DBGSOM_EXACT_WAVES = 4 | 8 | two digits (class 3, classes 2 and 1).
    static const int exact_waves = [] {
        const char *e = getenv("DBGSOM_EXACT_WAVES");
        return e ? atoi(e) : 84;
    }();
    // (two digits: class 3, then classes 2 and 1)
#define DBGSOM_SUBSET(JTL, STREAM)                                                                \
    do {                                                                                          \
        const int w_ = exact_waves >= 10 ? (JTL == 3 ? exact_waves / 10 : exact_waves % 10) : exact_waves; \
        if (w_ == 8) DBGSOM_SUBSET_W(JTL, 8, STREAM);                                             \
        else DBGSOM_SUBSET_W(JTL, 4, STREAM);                                                     \
    } while (0)
    if (k2) {
        // (8 wavefronts x 16 samples: with two (value, index) pairs per sample the 4 x 32 shape needed 256
        //  registers for the long-list class and spilled in the others)
        constexpr int K2_WAVES = 8;
#define DBGSOM_SUBSET_K2(JTL, STREAM)                                                             \
    do {                                                                                          \
        if (x_dtype == DBGSOM_F32)                                                                \
            hipLaunchKernelGGL((subset_exact_kernel<float, JTL, K2_WAVES, 1, 2>), dim3((unsigned)f.nb), dim3(K2_WAVES * 64), 0, STREAM, \
                               (const float *)X_dev, N, (int)d, ldx, xx_dev, W_dev, (int)M, ww_dev, \
                               order_dev, f.ulist, (int)f.Mpad, f.ucount, f.sched, f.sched_ctr + 2 * SCHED_BINS, round_f32, idx_dev, dist_dev); \
        else                                                                                      \
            hipLaunchKernelGGL((subset_exact_kernel<double, JTL, K2_WAVES, 1, 2>), dim3((unsigned)f.nb), dim3(K2_WAVES * 64), 0, STREAM, \
                               (const double *)X_dev, N, (int)d, ldx, xx_dev, W_dev, (int)M, ww_dev, \
                               order_dev, f.ulist, (int)f.Mpad, f.ucount, f.sched, f.sched_ctr + 2 * SCHED_BINS, round_f32, idx_dev, dist_dev); \
    } while (0)
        DBGSOM_SUBSET_K2(3, s_mfma);
        DBGSOM_SUBSET_K2(2, s2);
        DBGSOM_SUBSET_K2(1, s3);
#undef DBGSOM_SUBSET_K2
    } else if (merged && !exact_split) {
        if (x_dtype == DBGSOM_F32)
            hipLaunchKernelGGL((subset_exact_all_kernel<float>), dim3((unsigned)f.nb), dim3(512), 0, s,
                               (const float *)X_dev, N, (int)d, ldx, xx_dev, W_dev, (int)M, ww_dev, order_dev, f.ulist,
                               (int)f.Mpad, f.ucount, f.sched, f.sched_ctr + 2 * SCHED_BINS, round_f32, idx_dev, dist_dev);
        else
            hipLaunchKernelGGL((subset_exact_all_kernel<double>), dim3((unsigned)f.nb), dim3(512), 0, s,
                               (const double *)X_dev, N, (int)d, ldx, xx_dev, W_dev, (int)M, ww_dev, order_dev, f.ulist,
                               (int)f.Mpad, f.ucount, f.sched, f.sched_ctr + 2 * SCHED_BINS, round_f32, idx_dev, dist_dev);
    } else if (merged) {
        if (x_dtype == DBGSOM_F32)
            hipLaunchKernelGGL((subset_exact_split_kernel<float>), dim3((unsigned)(2 * f.nb), 3), dim3(256), 0, s,
                               (const float *)X_dev, N, (int)d, ldx, xx_dev, W_dev, (int)M, ww_dev, order_dev, f.ulist,
                               (int)f.Mpad, f.ucount, f.sched, f.sched_ctr + 2 * SCHED_BINS, round_f32, idx_dev, dist_dev);
        else
            hipLaunchKernelGGL((subset_exact_split_kernel<double>), dim3((unsigned)(2 * f.nb), 3), dim3(256), 0, s,
                               (const double *)X_dev, N, (int)d, ldx, xx_dev, W_dev, (int)M, ww_dev, order_dev, f.ulist,
                               (int)f.Mpad, f.ucount, f.sched, f.sched_ctr + 2 * SCHED_BINS, round_f32, idx_dev, dist_dev);
    } else {
        DBGSOM_SUBSET(3, s_mfma);
        DBGSOM_SUBSET(2, refine ? s3 : s2);
        DBGSOM_SUBSET(1, s3);
    }
#undef DBGSOM_SUBSET
#undef DBGSOM_SUBSET_W
#undef SPLIT_NS
#undef DBGSOM_SWEEP
#undef S4_LAUNCH
    if (fork) {
        DBGSOM_HIP_CHECK(hipEventRecord(side.joined, s2));
        DBGSOM_HIP_CHECK(hipEventRecord(side.joined2, s3));
        DBGSOM_HIP_CHECK(hipStreamWaitEvent(s, side.joined, 0));
        DBGSOM_HIP_CHECK(hipStreamWaitEvent(s, side.joined2, 0));
    }
    g_timer.mark(5, s);
    g_timer.valid = g_timer.enabled;
    return launch_status("filtered bmu kernels");
}

extern "C" {


/* (internal, engine.hip) device address of the sum of the candidate-list lengths of the last call */
const unsigned long long *dbgsom_filter_count_sum_ptr(const void *workspace_dev, int64_t N, int64_t d, int64_t M) {
    FilterWs f;
    carve_filter(&f, (char *)const_cast<void *>(workspace_dev), N, d, M);
    return reinterpret_cast<const unsigned long long *>(f.sched_ctr + SCHED_SUM);
}

/* diagnostics: sizes of the per-workgroup candidate lists of the last dbgsom_bmu_filtered call */
int dbgsom_bmu_filtered_counts(const void *workspace_dev, int64_t N, int64_t d, int64_t M,
                               uint32_t *counts_host, int64_t n_counts, void *stream) {
    DBGSOM_REQUIRE(workspace_dev && counts_host, "null pointer");
    FilterWs f;
    carve_filter(&f, (char *)const_cast<void *>(workspace_dev), N, d, M);
    DBGSOM_REQUIRE(n_counts == f.nb, "n_counts must be ceil(N / 128)");
    DBGSOM_HIP_CHECK(hipMemcpyAsync(counts_host, f.ucount, (size_t)f.nb * 4, hipMemcpyDeviceToHost,
                                    (hipStream_t)stream));
    DBGSOM_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return DBGSOM_OK;
}

int dbgsom_bmu_filtered_refine_counts(const void *workspace_dev, int64_t N, int64_t d, int64_t M,
                                      uint64_t *out4, void *stream) {
    DBGSOM_REQUIRE(workspace_dev && out4, "null pointer");
    FilterWs f;
    carve_filter(&f, (char *)const_cast<void *>(workspace_dev), N, d, M);
    DBGSOM_HIP_CHECK(hipMemcpyAsync(out4, f.rf_ctr, (size_t)RF_CTR * 8, hipMemcpyDeviceToHost, (hipStream_t)stream));
    DBGSOM_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return DBGSOM_OK;
}

/* the same copy, only queued on the stream: counts_host must be page-locked, the caller
 * synchronises (the estimator fetches it in the round trip that brings back the epoch's results) */
int dbgsom_bmu_filtered_counts_async(const void *workspace_dev, int64_t N, int64_t d, int64_t M,
                                     uint32_t *counts_host, int64_t n_counts, void *stream) {
    DBGSOM_REQUIRE(workspace_dev && counts_host, "null pointer");
    FilterWs f;
    carve_filter(&f, (char *)const_cast<void *>(workspace_dev), N, d, M);
    DBGSOM_REQUIRE(n_counts == f.nb, "n_counts must be ceil(N / 128)");
    DBGSOM_HIP_CHECK(hipMemcpyAsync(counts_host, f.ucount, (size_t)f.nb * 4, hipMemcpyDeviceToHost,
                                    (hipStream_t)stream));
    return DBGSOM_OK;
}

}  // extern "C"

#if defined(DBGSOM_EXPERIMENTS) && PRUNE_STAMPS
extern "C" int dbgsom_experiment_prune_stamps(unsigned long long *out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prune_stamps), (size_t)n * 8, 0, hipMemcpyDeviceToHost);
}
#endif
