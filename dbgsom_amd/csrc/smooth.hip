// Neighbourhood-weighted batch update on gfx950 (float64 throughout).
//
// Replaces steps 3-5 of BaseSom._update_weights (reference dbgsom/BaseSom.py:506-522) and
// _calculate_gaussian_neighborhood (:525-531):
//     c_j  = S_j / K_j  (rows placed per `layout`; COMPACT reproduces quirk Q1, :1045,1053)
//     g_ij = exp(-(hop_ij^2 / (2 sigma^2))) * a_j
//     W'_i = sum_j g_ij c_j / sum_j g_ij ;   change_total = sum_i |W_i - W'_i|_2
// The reference materialises an (M, M, d) float64 temporary for the same sum; here it is one
// M x M x d float64 GEMM (2 M^2 d flops, < 0.2 % of the BMU work at M ~ 1000), LDS-tiled.
#include <math.h>
#include <stdlib.h>

#include "common.h"

namespace dbgsom {

// G and C are stored for the GEMM's LDS-DMA tiles: G with a leading dimension of Mp = M rounded up to 16
// (zeros behind column M), C with Mp rows (zero rows behind row M) and one tile of slack behind it -- a
// k-tile is then always whole, zeros add nothing to an fma chain, and no tile read leaves the buffers.
static inline int64_t smooth_mp(int64_t M) { return (M + 15) / 16 * 16; }

struct SmoothWs {
    double *C;      // Mp x d  Voronoi centres in the requested layout
    double *G;      // M x Mp  h * a^T
    double *den;    // M
    double *rowchg; // M
    uint32_t *ticket;  // "last workgroup" ticket of the row-change kernel
    double *part;   // splits x M x d   partial products of the split-K GEMM (splits > 1)
};

// The GEMM's grid is ceil(d / 64) x ceil(M / 64) workgroups, each walking all M / 16 k-tiles one
// after the other: on a map of ~1000 neurons that is fewer workgroups than CUs and a 64-tile
// dependent chain.  The k range is therefore cut into `splits` consecutive pieces (a function of
// the shape alone) computed side by side and added in piece order: still bitwise reproducible.
static int gemm_splits(int64_t M, int64_t d) {
    const int64_t tiles = ((d + 63) / 64) * ((M + 63) / 64), nkt = (M + 15) / 16;
    int64_t ks = (768 + tiles - 1) / tiles;
    if (ks > 8) ks = 8;
    if (ks > nkt / 4) ks = nkt / 4;
    return (int)(ks < 1 ? 1 : ks);
}

static size_t carve_smooth(SmoothWs *w, char *base, int64_t M, int64_t d) {
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return o; };
    const int64_t Mp = smooth_mp(M);
    const size_t oC = take(((size_t)Mp * d + 64) * 8), oG = take(((size_t)M * Mp + 16) * 8);
    const size_t oD = take((size_t)M * 8), oR = take((size_t)M * 8), oK = take((size_t)M * 4);
    const int ks = gemm_splits(M, d);
    const size_t oP = take(ks > 1 ? (size_t)ks * M * d * 8 : 0);
    if (w) {
        w->part = (double *)(base + oP);
        w->C = (double *)(base + oC); w->G = (double *)(base + oG);
        w->den = (double *)(base + oD); w->rowchg = (double *)(base + oR);
        w->ticket = (uint32_t *)(base + oK);
    }
    return off;
}

size_t smooth_workspace_bytes(int64_t M, int64_t d) {
    if (M < 1 || d < 1) return 0;
    return carve_smooth(nullptr, nullptr, M, d);
}

// One launch prepares both GEMM operands (it used to be a memset and three kernels):
//   workgroup j < M:  row j of the centres -- C[dst(j), :] = S[j, :] / K[j] for a non-empty neuron,
//                     dst = rank among the non-empty neurons (COMPACT, quirk Q1) or j (ALIGNED);
//                     the rows nobody writes (COMPACT: the last M - #non-empty; ALIGNED: the empty
//                     neurons') are zero-filled by the workgroup of the same number;
//                     row j of G = exp(-(hop^2 / (2 sigma^2))) * a and den[j] = its sum (fixed tree).
// The rank of a neuron is a count over a[0 .. j): every workgroup counts for itself (M reads).
__global__ __launch_bounds__(256) void smooth_prep_kernel(const double *__restrict__ S,
                                                          const double *__restrict__ K,
                                                          const double *__restrict__ a,
                                                          const float *__restrict__ hop, int M, int d,
                                                          int layout, double two_sigma_sq,
                                                          double *__restrict__ C, double *__restrict__ G,
                                                          double *__restrict__ den,
                                                          uint32_t *__restrict__ ticket, int Mp) {
    __shared__ double red[256];
    __shared__ int cnt[256], cnt_all[256];
    const int i = blockIdx.x, t = threadIdx.x;
    if (i == 0 && t == 0) *ticket = 0u;  // of the row-change kernel behind the GEMM
    // non-empty neurons before i, and in all
    int before = 0, all = 0;
    for (int j = t; j < M; j += 256) {
        const int ne = a[j] > 0.0;
        all += ne;
        before += (j < i) ? ne : 0;
    }
    cnt[t] = before;
    cnt_all[t] = all;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (t < w) { cnt[t] += cnt[t + w]; cnt_all[t] += cnt_all[t + w]; }
        __syncthreads();
    }
    const int rank = cnt[0], nne = cnt_all[0];
    const bool alive = a[i] > 0.0;
    if (alive) {
        const int dst = (layout == DBGSOM_CENTRES_COMPACT) ? rank : i;
        const double k = K[i];
        for (int c = t; c < d; c += 256) C[(size_t)dst * d + c] = S[(size_t)i * d + c] / k;
    }
    const bool zero_row = (layout == DBGSOM_CENTRES_COMPACT) ? (i >= nne) : !alive;
    if (zero_row)
        for (int c = t; c < d; c += 256) C[(size_t)i * d + c] = 0.0;
    if (i == 0)  // (the zero rows of C behind row M)
        for (int e = t; e < (Mp - M) * d; e += 256) C[(size_t)M * d + e] = 0.0;
    double s = 0.0;
    for (int j = t; j < Mp; j += 256) {
        double g = 0.0;
        if (j < M) {
            const double h = (double)hop[(size_t)i * M + j];
            g = exp(-((h * h) / two_sigma_sq)) * a[j];
            s += g;
        }
        G[(size_t)i * Mp + j] = g;
    }
    red[t] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (t < w) red[t] += red[t + w];
        __syncthreads();
    }
    if (t == 0) den[i] = red[0];
}

// W'[i, c] = (sum_j G[i, j] C[j, c]) / den[i] on the f64 matrix cores: 64 x 64 tile per workgroup, 4
// wavefronts of 2 x 2 v_mfma_f64_16x16x4_f64 tiles, j ascending inside one accumulator chain (sequential
// like the reference's np.sum over axis 1).  Operand tiles (64 rows of G x 16 j, 16 rows of C x 64 columns:
// 8 KB each) come through a 3-stage LDS-DMA ring -- global_load_lds_dwordx4, counted s_waitcnt vmcnt, ONE
// raw barrier per k-tile -- the ring of bmu_dma_kernel (0.85 of the f64 matrix peak on the same kind of
// product; the register-staged tiles with two barriers per k-tile this replaces reached 0.57).  The
// accumulation order is unchanged: results are the former kernel's bit for bit.
typedef double sd4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *sm_lds_ptr_t;
typedef const __attribute__((address_space(1))) void *sm_gbl_ptr_t;
__device__ __forceinline__ void sm_dma16(const void *src, void *lds_dst) {
    __builtin_amdgcn_global_load_lds((sm_gbl_ptr_t)src, (sm_lds_ptr_t)lds_dst, 16, 0, 0);
}
constexpr int GT = 64, GK = 16;
constexpr int GR = 128;   // rows of G per workgroup of the DMA kernel (columns: GT)
constexpr int SG_A = GR * GK * 8, SG_B = GK * GT * 8, SG_STAGE = SG_A + SG_B;   // 16 + 8 KB
__global__ __launch_bounds__(256, 2) void smooth_gemm_kernel(const double *__restrict__ G,
                                                             const double *__restrict__ C,
                                                             const double *__restrict__ den, int M, int Mp,
                                                             int d, int splits,
                                                             double *__restrict__ part,
                                                             double *__restrict__ Wn) {
    __shared__ __attribute__((aligned(16))) char smem[3 * SG_STAGE];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 1, wc = wave & 1;  // 2 x 2 wavefronts, 64 rows x 32 columns each (4 x 2 matrix tiles)
    const int lr = lane & 15, lq = lane >> 4;
    const int i0 = blockIdx.y * GR, c0 = blockIdx.x * GT;
    // this workgroup's piece of the k range (whole k-tiles)
    const int per = ((M + GK - 1) / GK + splits - 1) / splits * GK;
    const int jlo = blockIdx.z * per, jhi = min(Mp, jlo + per);
    const int ntile = jhi > jlo ? (jhi - jlo) / GK : 0;
    // DMA sources.  A tile: 128 rows x 128 bytes, 8 rows per instruction, wavefront w issues rows 32 w .. 32 w +
    // 31 (four instructions); LDS chunk cp of row r holds the row's 16-byte chunk cp ^ ((r >> 1) & 7).
    // B tile: 16 rows x 512 bytes, 2 rows per instruction, wavefront w issues rows 4 w .. 4 w + 3; LDS chunk
    // cp of row j holds chunk cp ^ ((j & 1) << 3): the two j of a read group hit different bank halves.
    const double *asrc[4], *bsrc[2];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int r = 32 * wave + 8 * u + (lane >> 3), cp = lane & 7;
        int gi = i0 + r;
        gi = gi < M ? gi : M - 1;   // (rows behind M: any row will do, the results are not stored)
        asrc[u] = G + (size_t)gi * Mp + ((cp ^ ((r >> 1) & 7)) << 1);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int j = 4 * wave + 2 * u + (lane >> 5), cq = lane & 31;
        bsrc[u] = C + (size_t)j * d + c0 + ((cq ^ ((j & 1) << 3)) << 1);   // (columns behind d: the slack, not stored)
    }
    int i_t = 0, i_stage = 0;
    auto issue = [&]() {
        char *stage = smem + i_stage;
        const int j0 = jlo + i_t * GK;
#pragma unroll
        for (int u = 0; u < 4; ++u) sm_dma16(asrc[u] + j0, stage + 1024 * (4 * wave + u));
#pragma unroll
        for (int u = 0; u < 2; ++u) sm_dma16(bsrc[u] + (size_t)j0 * d, stage + SG_A + 1024 * (2 * wave + u));
        i_stage = (i_stage == 2 * SG_STAGE) ? 0 : i_stage + SG_STAGE;
        ++i_t;
    };
    int a_off[4], a_swz[4], b_off[2];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int ra = wr * 64 + u * 16 + lr;
        a_off[u] = ra * 128 + (lq & 1) * 8;
        a_swz[u] = (ra >> 1) & 7;
    }
#pragma unroll
    for (int v = 0; v < 2; ++v) b_off[v] = (wc * 32 + v * 16 + lr) * 8;   // byte offset of the column in a 512-byte row
    sd4_t acc[4][2];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) acc[u][v] = sd4_t{0.0, 0.0, 0.0, 0.0};
    if (ntile > 0) issue();
    if (ntile > 1) issue();
    int r_stage = 0;
    for (int tl = 0; tl < ntile; ++tl) {
        if (tl + 1 < ntile) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const char *stage = smem + r_stage;
        r_stage = (r_stage == 2 * SG_STAGE) ? 0 : r_stage + SG_STAGE;
#pragma unroll
        for (int ks = 0; ks < GK / 4; ++ks) {
            // (the DMAs of tile tl + 2 go out behind the first eight matrix instructions: an LDS-DMA holds the
            //  wavefront's issue for ~100 cycles, which the matrix pipe then covers)
            if (ks == 1 && tl + 2 < ntile) {
                __builtin_amdgcn_sched_barrier(0);
                issue();
                __builtin_amdgcn_sched_barrier(0);
            }
            double a[4], b[2];
            const int jr = 4 * ks + lq;   // row of the B tile
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ca = (2 * ks + (lq >> 1)) ^ a_swz[u];
                a[u] = *reinterpret_cast<const double *>(stage + a_off[u] + ca * 16);
            }
#pragma unroll
            for (int v = 0; v < 2; ++v)
                b[v] = *reinterpret_cast<const double *>(stage + SG_A + jr * 512 + (b_off[v] ^ ((jr & 1) << 7)));
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 2; ++v)
                    acc[u][v] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[v], acc[u][v], 0, 0, 0);
        }
    }
    // D layout: reg r of lane l = D[row = (l >> 4) + 4 r][col = l & 15]
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + wr * 64 + u * 16 + lq + 4 * r;
            if (i >= M) continue;
            const double dn = den[i];
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const int c = c0 + wc * 32 + v * 16 + lr;
                if (c >= d) continue;
                if (splits == 1) Wn[(size_t)i * d + c] = acc[u][v][r] / dn;
                else part[((size_t)blockIdx.z * M + i) * d + c] = acc[u][v][r];
            }
        }
}

// The same product with register-staged tiles (two barriers per k-tile): rows of C that are not 16-byte
// multiples (odd d through the device-level ABI) cannot be the source of an LDS-DMA.  Same accumulation
// order, same results.
constexpr int GS_A = GK + 2, GS_B = GT + 16;
__global__ __launch_bounds__(256) void smooth_gemm_generic_kernel(const double *__restrict__ G,
                                                          const double *__restrict__ C,
                                                          const double *__restrict__ den, int M, int Mp,
                                                          int d, int splits,
                                                          double *__restrict__ part,
                                                          double *__restrict__ Wn) {
    __shared__ double gs[GT * GS_A];   // [i][j]   rows padded to 18: conflict-free A fragments
    __shared__ double cs[GK * GS_B];   // [j][c]   rows padded to 80: conflict-free B fragments
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1;  // 2 x 2 wavefronts, 32 x 32 outputs each
    const int lr = lane & 15, lq = lane >> 4;
    const int i0 = blockIdx.y * GT, c0 = blockIdx.x * GT;
    // this workgroup's piece of the k range (whole k-tiles)
    const int per = ((M + GK - 1) / GK + splits - 1) / splits * GK;
    const int jlo = blockIdx.z * per, jhi = min(M, jlo + per);
    sd4_t acc[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) acc[u][v] = sd4_t{0.0, 0.0, 0.0, 0.0};
    // global -> register -> LDS staging, one k-tile ahead: the loads of tile j0 + GK are in flight
    // while tile j0 feeds the matrix cores (64 k-tiles of 2 x 1024 values per workgroup; without
    // the prefetch every tile exposed a full L2/HBM round trip)
    constexpr int PER = GT * GK / 256;  // 4 elements of each operand tile per thread
    double gr[PER], cr[PER];
    auto fetch = [&](int j0) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int e = t + 256 * u;
            const int r = e / GK, k = e % GK;  // G tile: 64 rows x 16 j
            const int gi = i0 + r, gj = j0 + k;
            gr[u] = (gi < M && gj < jhi) ? G[(size_t)gi * Mp + gj] : 0.0;
            const int kr = e / GT, cc = e % GT;  // C tile: 16 j x 64 cols
            const int cj = j0 + kr, col = c0 + cc;
            cr[u] = (cj < jhi && col < d) ? C[(size_t)cj * d + col] : 0.0;
        }
    };
    fetch(jlo);
    for (int j0 = jlo; j0 < jhi; j0 += GK) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int e = t + 256 * u;
            gs[(e / GK) * GS_A + e % GK] = gr[u];
            cs[(e / GT) * GS_B + e % GT] = cr[u];
        }
        __syncthreads();
        if (j0 + GK < jhi) fetch(j0 + GK);
#pragma unroll
        for (int ks = 0; ks < GK / 4; ++ks) {
            double a[2], b[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                a[u] = gs[(wr * 32 + u * 16 + lr) * GS_A + ks * 4 + lq];
                b[u] = cs[(ks * 4 + lq) * GS_B + wc * 32 + u * 16 + lr];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int v = 0; v < 2; ++v)
                    acc[u][v] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[v], acc[u][v], 0, 0, 0);
        }
    }
    // D layout: reg r of lane l = D[row = (l >> 4) + 4 r][col = l & 15]
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + wr * 32 + u * 16 + lq + 4 * r;
            if (i >= M) continue;
            const double dn = den[i];
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const int c = c0 + wc * 32 + v * 16 + lr;
                if (c >= d) continue;
                if (splits == 1) Wn[(size_t)i * d + c] = acc[u][v][r] / dn;
                else part[((size_t)blockIdx.z * M + i) * d + c] = acc[u][v][r];
            }
        }
}

// rowchg[i] = |W_i - W'_i|_2; with a split-K GEMM, W'_i is first put together from the pieces (in
// piece order) and divided by den[i].  One wavefront per row, four rows per workgroup (a quarter of the
// tickets, no LDS tree per row: the launch took 28 us for 1024 rows of 784 with a workgroup per row); the
// squares are added per lane in column order and across the lanes in a fixed butterfly.
constexpr int RC_ROWS = 4;
__global__ __launch_bounds__(256) void rowchange_kernel(const double *__restrict__ Wo,
                                                        double *__restrict__ Wn, int M, int d,
                                                        int splits, const double *__restrict__ part,
                                                        const double *__restrict__ den,
                                                        double *__restrict__ rowchg,
                                                        uint32_t *__restrict__ ticket,
                                                        double *__restrict__ change_total) {
    const int t = threadIdx.x, lane = t & 63;
    const int i = blockIdx.x * RC_ROWS + (t >> 6);
    if (i < M) {
        double s = 0.0;
        const double dn = den[i];
        for (int c = lane; c < d; c += 64) {
            double wn;
            if (splits == 1) {
                wn = Wn[(size_t)i * d + c];
            } else {
                double p = part[(size_t)i * d + c];
                for (int z = 1; z < splits; ++z) p += part[((size_t)z * M + i) * d + c];
                wn = p / dn;
                Wn[(size_t)i * d + c] = wn;
            }
            const double df = Wo[(size_t)i * d + c] - wn;
            s += df * df;
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
        if (lane == 0) rowchg[i] = sqrt(s);
    }
    // change_total = sum_i rowchg[i]: the workgroup that finishes last adds them up, in the fixed
    // order of a 1024-leaf strided binary tree -- bitwise reproducible
    if (!last_workgroup_done(ticket, gridDim.x)) return;
    __shared__ double tot[256];
    double p[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        double v = 0.0;
        for (int j = t + 256 * u; j < M; j += 1024) v += rowchg[j];
        p[u] = v;
    }
    tot[t] = (p[0] + p[2]) + (p[1] + p[3]);   // tree levels 512 and 256 of the 1024-leaf tree
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (t < w) tot[t] += tot[t + w];
        __syncthreads();
    }
    if (t == 0) change_total[0] = tot[0];
}

// ---- smoothing sharded over the ranks of a sample-sharded job (columns of W') -------------------
// W'[:, c] depends on column c of the centres and on nothing else of S, so rank r of G smooths the
// columns [r cb, (r + 1) cb) only: the epoch's collective becomes a reduce-scatter of column blocks
// S[:, block] (every rank then holds the reduced block it smooths; the small vectors [K | a | E | status] are
// all-reduced beside it, bit-identical on every rank), the GEMM shrinks to M x M x cb per rank, and an
// all-gather of the W' blocks gives every rank the same W' bit for bit.  The k range is cut as for the whole matrix (gemm_splits of the FULL
// shape), so a column of W' is the same chain of the same pieces in either form.

// [nblk][ S block (M x cb) ] from the row-major sums [S (M x d) | tail]; columns behind d: 0.  The small vectors
// [K | a | E | status] are NOT part of a block: a reduce-scatter gives every block its own summation chain, so
// copies of them riding in the blocks could differ in the last bit from rank to rank; they go through one
// ordinary all-reduce instead (engine.hip), whose result is the same on every rank.
__global__ __launch_bounds__(256) void pack_blocks_kernel(const double *__restrict__ sums, int M, int d, int cb,
                                                          int nblk, int64_t blk, double *__restrict__ out) {
    const int i = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
    double *dst = out + (size_t)b * blk;
    for (int cc = t; cc < cb; cc += 256) {
        const int c = b * cb + cc;
        dst[(size_t)i * cb + cc] = c < d ? sums[(size_t)i * d + c] : 0.0;
    }
}

// W'_block[i, c] = (sum of the split-K pieces in piece order) / den[i]
__global__ __launch_bounds__(256) void combine_block_kernel(const double *__restrict__ part, const double *__restrict__ den,
                                                            int M, int cb, int splits, double *__restrict__ Wb) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)M * cb) return;
    double p = part[e];
    for (int z = 1; z < splits; ++z) p += part[(size_t)z * M * cb + e];
    Wb[e] = p / den[e / cb];
}

// the gathered blocks [nblk][M][cb] -> W' row-major (M x d), and the row changes as rowchange_kernel forms them
__global__ __launch_bounds__(256) void rowchange_blocks_kernel(const double *__restrict__ Wo, double *__restrict__ Wn,
                                                               int M, int d, int cb,
                                                               const double *__restrict__ blocks,
                                                               double *__restrict__ rowchg,
                                                               uint32_t *__restrict__ ticket,
                                                               double *__restrict__ change_total) {
    const int t = threadIdx.x, lane = t & 63;
    const int i = blockIdx.x * RC_ROWS + (t >> 6);
    if (i < M) {
        double s = 0.0;
        for (int c = lane; c < d; c += 64) {
            const int b = c / cb;
            const double wn = blocks[((size_t)b * M + i) * cb + (c - b * cb)];
            Wn[(size_t)i * d + c] = wn;
            const double df = Wo[(size_t)i * d + c] - wn;
            s += df * df;
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
        if (lane == 0) rowchg[i] = sqrt(s);
    }
    if (!last_workgroup_done(ticket, gridDim.x)) return;
    __shared__ double tot[256];
    double p[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        double v = 0.0;
        for (int j = t + 256 * u; j < M; j += 1024) v += rowchg[j];
        p[u] = v;
    }
    tot[t] = (p[0] + p[2]) + (p[1] + p[3]);
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (t < w) tot[t] += tot[t + w];
        __syncthreads();
    }
    if (t == 0) change_total[0] = tot[0];
}

int64_t smooth_block_cols(int64_t d, int nranks) {
    const int64_t cb = (d + nranks - 1) / nranks;
    return (cb + 1) / 2 * 2;   // (16-byte rows for the GEMM's LDS-DMA)
}
int64_t smooth_block_elems(int64_t M, int64_t d, int nranks) {
    return M * smooth_block_cols(d, nranks);
}

int launch_pack_blocks(const double *sums, int64_t M, int64_t d, int nranks, double *out, hipStream_t s) {
    const int64_t cb = smooth_block_cols(d, nranks), blk = smooth_block_elems(M, d, nranks);
    hipLaunchKernelGGL(pack_blocks_kernel, dim3((unsigned)M, (unsigned)nranks), dim3(256), 0, s, sums, (int)M, (int)d, (int)cb,
                       nranks, blk, out);
    return launch_status("pack_blocks_kernel");
}

// this rank's block of W' (M x cb, into Wb) from its reduced block of the sums
int launch_smooth_block(const double *S_b, const double *K, const double *a, int64_t M, int64_t cb, int64_t d_full,
                        const float *hop, double sigma, int layout, double *Wb, void *ws, size_t ws_bytes,
                        hipStream_t s) {
    DBGSOM_REQUIRE(M >= 1 && M <= 0x7fff && cb >= 2 && cb % 2 == 0 && cb <= d_full + 1 && d_full <= 0x7ffffff0, "bad shape");
    DBGSOM_REQUIRE(S_b && K && a && hop && Wb && ws && sigma > 0.0, "bad arguments");
    if (ws_bytes < smooth_workspace_bytes(M, d_full)) { set_error("dbgsom_smooth: workspace too small"); return DBGSOM_ENOMEM; }
    SmoothWs w;
    carve_smooth(&w, (char *)ws, M, d_full);
    const int Mi = (int)M, ci = (int)cb, Mp = (int)smooth_mp(M);
    hipLaunchKernelGGL(smooth_prep_kernel, dim3((unsigned)M), dim3(256), 0, s, S_b, K, a, hop, Mi, ci, layout,
                       2.0 * (sigma * sigma), w.C, w.G, w.den, w.ticket, Mp);
    const int splits = gemm_splits(M, d_full);   // (the pieces of the whole matrix: the same bits in either form)
    hipLaunchKernelGGL(smooth_gemm_kernel, dim3((unsigned)((cb + GT - 1) / GT), (unsigned)((M + GR - 1) / GR), (unsigned)splits),
                       dim3(256), 0, s, w.G, w.C, w.den, Mi, Mp, ci, splits, w.part, Wb);
    if (splits > 1)
        hipLaunchKernelGGL(combine_block_kernel, dim3((unsigned)((M * cb + 255) / 256)), dim3(256), 0, s, w.part, w.den, Mi, ci,
                           splits, Wb);
    return launch_status("smooth block kernels");
}

// W' (row-major) and the convergence norm from the gathered blocks
int launch_rowchange_blocks(const double *blocks, int64_t M, int64_t d, int64_t cb, const double *W_old, double *W_new,
                            double *change_total, void *ws, hipStream_t s) {
    SmoothWs w;
    carve_smooth(&w, (char *)ws, M, d);
    hipLaunchKernelGGL(rowchange_blocks_kernel, dim3((unsigned)((M + RC_ROWS - 1) / RC_ROWS)), dim3(256), 0, s, W_old, W_new,
                       (int)M, (int)d, (int)cb, blocks, w.rowchg, w.ticket, change_total);
    return launch_status("rowchange_blocks_kernel");
}

int launch_smooth(const double *sums, int64_t M, int64_t d, const float *hop, double sigma,
                  int layout, const double *W_old, double *W_new, double *change_total, void *ws,
                  size_t ws_bytes, hipStream_t s) {
    DBGSOM_REQUIRE(M >= 1 && M <= 0x7fff && d >= 1 && d <= 0x7ffffff0, "bad shape");
    DBGSOM_REQUIRE(layout == DBGSOM_CENTRES_COMPACT || layout == DBGSOM_CENTRES_ALIGNED,
                   "layout must be DBGSOM_CENTRES_COMPACT/ALIGNED");
    DBGSOM_REQUIRE(sums && hop && W_old && W_new && change_total && ws, "null pointer");
    DBGSOM_REQUIRE(W_old != W_new, "W_new must not alias W_old");
    DBGSOM_REQUIRE(is_aligned(ws, 256), "workspace must be 256-byte aligned");
    DBGSOM_REQUIRE(sigma > 0.0, "sigma must be positive");
    if (ws_bytes < smooth_workspace_bytes(M, d)) {
        set_error("dbgsom_smooth: workspace too small (%zu < %zu)", ws_bytes,
                  smooth_workspace_bytes(M, d));
        return DBGSOM_ENOMEM;
    }
    SmoothWs w;
    carve_smooth(&w, (char *)ws, M, d);
    const int Mi = (int)M, di = (int)d;
    const double *S = sums, *K = sums + (size_t)M * d, *a = K + M;
    const int Mp = (int)smooth_mp(M);
    hipLaunchKernelGGL(smooth_prep_kernel, dim3((unsigned)M), dim3(256), 0, s, S, K, a, hop, Mi, di, layout,
                       2.0 * (sigma * sigma), w.C, w.G, w.den, w.ticket, Mp);
    const int splits = gemm_splits(M, d);
    dim3 grid((unsigned)((d + GT - 1) / GT), (unsigned)((M + GT - 1) / GT), (unsigned)splits);
    static const bool force_generic = [] {   // DBGSOM_SMOOTH_GENERIC=1: the register-staged kernel everywhere (tests)
        const char *e = getenv("DBGSOM_SMOOTH_GENERIC");
        return e && atoi(e) != 0;
    }();
    if (d % 2 == 0 && !force_generic)
        hipLaunchKernelGGL(smooth_gemm_kernel, dim3((unsigned)((d + GT - 1) / GT), (unsigned)((M + GR - 1) / GR), (unsigned)splits), dim3(256), 0, s, w.G, w.C, w.den, Mi, Mp, di, splits,
                           w.part, W_new);
    else
        hipLaunchKernelGGL(smooth_gemm_generic_kernel, grid, dim3(256), 0, s, w.G, w.C, w.den, Mi, Mp, di, splits,
                           w.part, W_new);
    hipLaunchKernelGGL(rowchange_kernel, dim3((unsigned)((M + RC_ROWS - 1) / RC_ROWS)), dim3(256), 0, s, W_old, W_new, Mi, di,
                       splits, w.part, w.den, w.rowchg, w.ticket, change_total);
    return launch_status("smooth kernels");
}

}  // namespace dbgsom
