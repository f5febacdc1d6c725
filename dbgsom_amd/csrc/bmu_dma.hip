// BMU search, LDS-DMA form (the fast path for 16-byte aligned rows with d % 16 == 0).
//
// Same arithmetic and the same (sample, prototype) -> accumulator-chain mapping as bmu.hip --
// v_mfma_f64_16x16x4_f64 over the whole feature dimension in one sequential fma chain per pair, so
// the two kernels (and oracle/bmu_chain.c) agree bit for bit.  What differs is how operands reach
// LDS: `global_load_lds_dwordx4` writes the X and W tiles straight into a 3-stage LDS ring (no
// staging VGPRs, no ds_write, no conversion pass), tile t+2 is issued while tile t feeds the
// MFMAs and is waited for with a counted vmcnt, one raw s_barrier per tile.
//
// LDS-DMA writes are lane-linear (wave-uniform base + 16 B x lane), so the bank-conflict swizzle
// is applied to the per-lane SOURCE address and undone on the fragment read:
//   W tile  : 128 rows x 128 B, 16-B chunk c of row r is stored at chunk c ^ ((r >> 1) & 7)
//             -> the ds_read_b64 fragment reads are conflict free
//   X tile  : f32: 128 rows x 64 B, chunk c at c ^ ((r >> 1) & 3) (2-way on ds_read_b32, the
//             best a 64-B row allows); f64: laid out like W
// float32 samples stay float32 in LDS and are widened on the fragment read (exact).
#include <stdlib.h>

#include "bmu_common.h"

namespace dbgsom {

constexpr int NSTAGE = 3;

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *gbl_ptr_t;

__device__ __forceinline__ void dma16(const void *src, void *lds_dst) {
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)lds_dst, 16, 0, 0);
}

template <typename XT>
struct XTile {
    static constexpr int ROW_BYTES = KT * (int)sizeof(XT);  // 64 (f32) or 128 (f64)
    static constexpr int CHUNKS = ROW_BYTES / 16;            // 4 or 8
    static constexpr int BYTES = BI * ROW_BYTES;             // 8 KB or 16 KB
    static constexpr int DMA_PER_WAVE = BYTES / 1024 / 4;    // wave-instructions per tile per wave
};

constexpr int W_ROW_BYTES = KT * 8, W_CHUNKS = 8;

// JTW = 16-prototype tiles per wavefront: the workgroup sweeps the prototypes in chunks of
// BJW = 32 JTW (128 for the general case; 64 / 32 for small maps, where a 128-wide chunk would
// be mostly padding -- a growing map spends most of its epochs below 100 neurons).
template <typename XT, int K, int JTW>
__global__ __launch_bounds__(NT, 2) void bmu_dma_kernel(
    const XT *__restrict__ X, int64_t N, int d, int64_t ldx, const double *__restrict__ xx,
    const double *__restrict__ W, int M, const double *__restrict__ ww, int round_f32,
    int64_t *__restrict__ idx_out, double *__restrict__ dist_out) {
    using XL = XTile<XT>;
    constexpr int BJW = 32 * JTW, W_BYTES = BJW * W_ROW_BYTES, W_DMA_PER_WAVE = JTW;
    constexpr int STAGE_BYTES = XL::BYTES + W_BYTES;
    // ONE shared array (a second __shared__ object next to an LDS-DMA target makes hipcc drain
    // vmcnt before every ds_read)
    __shared__ __attribute__((aligned(16))) char smem[NSTAGE * STAGE_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 1, wj = wave & 1;
    const int lr = lane & 15, lq = lane >> 4;
    const int64_t i0 = (int64_t)blockIdx.x * BI;

    double xi[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int64_t i = i0 + wi * 64 + it * 16 + lr;
        xi[it] = (i < N) ? xx[i] : 0.0;
    }
    Best<K> best[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) best[it].init();

    // ---- per-lane DMA sources ----------------------------------------------------------------
    // X: wave w issues instructions q = XD*w .. XD*w + XD-1; LDS linear chunk L = 64 q + lane
    const XT *xsrc[XL::DMA_PER_WAVE];
#pragma unroll
    for (int u = 0; u < XL::DMA_PER_WAVE; ++u) {
        const int L = 64 * (XL::DMA_PER_WAVE * wave + u) + lane;
        const int r = L / XL::CHUNKS, cp = L % XL::CHUNKS;
        const int c = cp ^ ((r >> 1) & (XL::CHUNKS - 1));
        int64_t row = i0 + r;
        row = row < N ? row : N - 1;  // clamped rows are computed but never stored
        xsrc[u] = X + row * ldx + c * (16 / (int)sizeof(XT));
    }
    // W: instructions q = 4 w .. 4 w + 3; row within the chunk and logical 16-B chunk per lane
    int wrow[W_DMA_PER_WAVE], wcol[W_DMA_PER_WAVE];
#pragma unroll
    for (int u = 0; u < W_DMA_PER_WAVE; ++u) {
        const int L = 64 * (W_DMA_PER_WAVE * wave + u) + lane;
        const int r = L / W_CHUNKS, cp = L % W_CHUNKS;
        wrow[u] = r;
        wcol[u] = (cp ^ ((r >> 1) & 7)) * 2;
    }

    const int nkt = d / KT;
    const int nchunk = (M + BJW - 1) / BJW;
    const int ntile = nkt * nchunk;

    auto issue = [&](int t) {  // enqueue the DMA of tile t into ring slot t % NSTAGE
        const int c_t = t / nkt, k0 = (t - c_t * nkt) * KT, jc_t = c_t * BJW;
        char *stage = smem + (t % NSTAGE) * STAGE_BYTES;
#pragma unroll
        for (int u = 0; u < XL::DMA_PER_WAVE; ++u)
            dma16(xsrc[u] + k0, stage + 1024 * (XL::DMA_PER_WAVE * wave + u));
#pragma unroll
        for (int u = 0; u < W_DMA_PER_WAVE; ++u) {
            int j = jc_t + wrow[u];
            j = j < M ? j : M - 1;
            dma16(W + (int64_t)j * d + k0 + wcol[u],
                  stage + XL::BYTES + 1024 * (W_DMA_PER_WAVE * wave + u));
        }
    };
    constexpr int DMA_PER_TILE = XL::DMA_PER_WAVE + W_DMA_PER_WAVE;  // per wave

    // ---- fragment read offsets (bytes inside a stage) ----------------------------------------
    int a_off[JTW], a_swz[JTW], b_off[4], b_swz[4];
#pragma unroll
    for (int u = 0; u < JTW; ++u) {
        const int ra = wj * 16 * JTW + u * 16 + lr;
        a_off[u] = XL::BYTES + ra * W_ROW_BYTES + (lq & 1) * 8;
        a_swz[u] = (ra >> 1) & 7;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int rb = wi * 64 + u * 16 + lr;
        if constexpr (sizeof(XT) == 4) {
            b_off[u] = rb * XL::ROW_BYTES + lq * 4;
            b_swz[u] = (rb >> 1) & 3;
        } else {
            b_off[u] = rb * XL::ROW_BYTES + (lq & 1) * 8;
            b_swz[u] = (rb >> 1) & 7;
        }
    }

    d4_t acc[JTW][4];
#pragma unroll
    for (int jt = 0; jt < JTW; ++jt)
#pragma unroll
        for (int it = 0; it < 4; ++it) acc[jt][it] = d4_t{0.0, 0.0, 0.0, 0.0};

    issue(0);
    if (ntile > 1) issue(1);

    int kt = 0, jc = 0;
    for (int t = 0; t < ntile; ++t) {
        // tile t has landed once all but the younger tile's DMAs of THIS wave are done ...
        if (t + 1 < ntile) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_TILE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ... and every wave has said so; the same barrier retires all reads of slot (t-1) % 3
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + 2 < ntile) issue(t + 2);

        const char *stage = smem + (t % NSTAGE) * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < KT / 4; ++ks) {
            double a[JTW], b[4];
#pragma unroll
            for (int u = 0; u < JTW; ++u) {
                const int ca = (2 * ks + (lq >> 1)) ^ a_swz[u];
                a[u] = *reinterpret_cast<const double *>(stage + a_off[u] + ca * 16);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if constexpr (sizeof(XT) == 4) {
                    const int cb = ks ^ b_swz[u];
                    b[u] = (double)*reinterpret_cast<const float *>(stage + b_off[u] + cb * 16);
                } else {
                    const int cb = (2 * ks + (lq >> 1)) ^ b_swz[u];
                    b[u] = *reinterpret_cast<const double *>(stage + b_off[u] + cb * 16);
                }
            }
#pragma unroll
            for (int jt = 0; jt < JTW; ++jt)
#pragma unroll
                for (int it = 0; it < 4; ++it)
                    acc[jt][it] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[jt], b[it], acc[jt][it],
                                                                       0, 0, 0);
        }
        if (kt == nkt - 1) {
            // chunk epilogue (plain loads of |w|^2: once per chunk, L2 resident)
#pragma unroll
            for (int jt = 0; jt < JTW; ++jt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = jc + wj * 16 * JTW + jt * 16 + 4 * r + lq;
                    if (j < M) {
                        const double y = ww[j];
#pragma unroll
                        for (int it = 0; it < 4; ++it) {
                            double rv = (xi[it] + (-2.0 * acc[jt][it][r])) + y;
                            if (!(rv > 0.0)) rv = (rv != rv) ? rv : 0.0;
                            best[it].push(rv, j);
                        }
                    }
                }
            }
#pragma unroll
            for (int jt = 0; jt < JTW; ++jt)
#pragma unroll
                for (int it = 0; it < 4; ++it) acc[jt][it] = d4_t{0.0, 0.0, 0.0, 0.0};
            kt = 0;
            jc += BJW;
        } else {
            ++kt;
        }
    }

#pragma unroll
    for (int it = 0; it < 4; ++it) {
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
    }

    __syncthreads();  // all DMA drained (vmcnt(0) above), all fragment reads done
    double *mv = reinterpret_cast<double *>(smem);                    // [2][BI][K]
    int *mj = reinterpret_cast<int *>(smem + 2 * BI * K * 8);         // [2][BI][K]
    if (lq == 0) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int sidx = wi * 64 + it * 16 + lr;
#pragma unroll
            for (int u = 0; u < K; ++u) {
                mv[(wj * BI + sidx) * K + u] = best[it].v[u];
                mj[(wj * BI + sidx) * K + u] = best[it].j[u];
            }
        }
    }
    __syncthreads();
    if (tid < BI) {
        const int64_t i = i0 + tid;
        if (i < N) {
            Best<K> bb;
            double ov[K];
            int oj[K];
#pragma unroll
            for (int u = 0; u < K; ++u) {
                bb.v[u] = mv[(0 * BI + tid) * K + u];
                bb.j[u] = mj[(0 * BI + tid) * K + u];
                ov[u] = mv[(1 * BI + tid) * K + u];
                oj[u] = mj[(1 * BI + tid) * K + u];
            }
            bb.merge(ov, oj);
#pragma unroll
            for (int u = 0; u < K; ++u) {
                double dv = sqrt(bb.v[u]);
                if (round_f32) dv = (double)(float)dv;
                idx_out[i * K + u] = (bb.j[u] == 0x7fffffff) ? (int64_t)-1 : (int64_t)bb.j[u];
                dist_out[i * K + u] = dv;
            }
        }
    }
}

// prototype chunk width (16-prototype tiles per wavefront) with the least padded work.  Measured
// cost of one chunk pass over the samples relative to the 32-wide one: float32 samples
// 1 : 1.74 : 3.33 (N = 1e6, d = 784: 0.93 / 1.62 / 3.10 ms); float64 samples 1 : 1.66 and no
// 128-wide form here (X tile 16 KB + W tile 16 KB = one workgroup per CU: the register-staged
// kernel is faster, 3.6 on this scale per 128 prototypes).  0 = take the register-staged kernel.
static int dma_chunk_tiles(int x_dtype, int64_t M) {
    const double n1 = (double)((M + 31) / 32), n2 = (double)((M + 63) / 64), n4 = (double)((M + 127) / 128);
    if (x_dtype == DBGSOM_F32) {
        const double c1 = 1.00 * n1, c2 = 1.74 * n2, c4 = 3.33 * n4;
        return (c4 <= c2 && c4 <= c1) ? 4 : (c2 <= c1 ? 2 : 1);
    }
    const double c1 = 1.00 * n1, c2 = 1.66 * n2, cg = 3.6 * n4;
    if (cg <= c1 && cg <= c2) return 0;
    return c2 <= c1 ? 2 : 1;
}

bool bmu_dma_usable(const void *X, int x_dtype, int64_t d, int64_t ldx, const void *W, int64_t M) {
    if (x_dtype != DBGSOM_F32 && x_dtype != DBGSOM_F64) return false;
    const size_t xe = dtype_size(x_dtype);
    return d % KT == 0 && is_aligned(X, 16) && (ldx * xe) % 16 == 0 && is_aligned(W, 16) &&
           dma_chunk_tiles(x_dtype, M) != 0;
}

int launch_bmu_dma(const void *X, int x_dtype, int64_t N, int64_t d, int64_t ldx, const double *xx,
                   const double *W, int64_t M, const double *ww, int k, int round_f32,
                   int64_t *idx, double *dist, hipStream_t s) {
    const int64_t nb = (N + BI - 1) / BI;
    dim3 grid((unsigned)nb), block(NT);
#define DBGSOM_DMA_LAUNCH(XT, KK, JTW)                                                            \
    hipLaunchKernelGGL((bmu_dma_kernel<XT, KK, JTW>), grid, block, 0, s, (const XT *)X, N, (int)d, \
                       ldx, xx, W, (int)M, ww, round_f32, idx, dist)
    const int jtw = dma_chunk_tiles(x_dtype, M);
#define DBGSOM_DMA_WIDTH(XT, KK)                                                                  \
    do {                                                                                          \
        if (jtw == 1) DBGSOM_DMA_LAUNCH(XT, KK, 1);                                               \
        else if (jtw == 2) DBGSOM_DMA_LAUNCH(XT, KK, 2);                                          \
        else DBGSOM_DMA_LAUNCH(XT, KK, 4);                                                        \
    } while (0)
    if (x_dtype == DBGSOM_F32) {
        if (k == 1) DBGSOM_DMA_WIDTH(float, 1); else DBGSOM_DMA_WIDTH(float, 2);
    } else {
        if (k == 1) DBGSOM_DMA_WIDTH(double, 1); else DBGSOM_DMA_WIDTH(double, 2);
    }
#undef DBGSOM_DMA_WIDTH
#undef DBGSOM_DMA_LAUNCH
    return launch_status("bmu_dma_kernel");
}

}  // namespace dbgsom
