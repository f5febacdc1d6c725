// Section 2d of filter.hip (included there, inside namespace dbgsom): per-SAMPLE refinement of the
// per-workgroup candidate lists, and the exact search on what is left of them.
//
// Replaces the same reference step as the rest of the file: BaseSom._get_winning_neurons
// (BaseSom.py:446-464).
//
// The lists the sweep / the triangle inequality leave are per 128-sample workgroup: on clustered data
// "the sample's cluster" (C4: 33 prototypes, C5 shard: 136), and the float64 matrix cores then form
// every (sample, list entry) product although a sample has one winner and a handful of rivals.
//   refine_i8_kernel   the top TWO digit planes of a row are its 16-bit rounding a16 = s (256 D0 + D1) /
//                      F16, F16 = 127 * 2^8.  For the gathered samples x the workgroup's list the four
//                      digit products give T = sum Q16x Q16w EXACTLY (int32 per level), so
//                          v_ij = |w_j|^2 - 2 s_i t_j T / F16^2 = r(x16_i, w16_j) - |x_i|^2 (+ norms as stored)
//                      and |x.w - x16.w16| <= |x - x16| |w| + (|x| + |x - x16|) |w - w16| (Cauchy-Schwarz) with
//                      the residual norms measured when the planes were cut (plane16_residual):
//                          eps_i = 2 [rx_i (max|w| + max rw) + |x_i| max rw] (1 + 1e-9) + rounding_i
//                      bounds |v_ij + |x_i|^2 - r_chain(i, j)| for every j.  A prototype that wins or ties
//                      has v_ij <= min_j' v_ij' + 2 eps_i: those (at most RF_C per sample, as positions in the
//                      list) are the sample's candidates; more than RF_C -> "the whole list".
//   pair_exact_kernel  the exact float64 chain (the arithmetic of subset_exact_kernel / bmu.hip: acc =
//                      fma(x_k, w_k, acc), k ascending -- what v_mfma_f64_16x16x4_f64 computes) for the
//                      (sample, candidate) PAIRS only, one pair per lane on the vector ALU, the gathered X
//                      tile streamed ONCE; arg-min by (value, index) over a sample's pairs.
// The arg-min over a superset of the possible winners is the arg-min over everything: winners and
// distances stay bit-identical to the all-pairs search.  A workgroup whose list is longer than the
// refinement's tile, or whose pairs / distinct candidates exceed the pair kernel's LDS tables, keeps
// gflag = 0 and goes through subset_exact_kernel as before (the schedule's bin counts are corrected
// here for the others).
#pragma once

typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));
constexpr int RF_C = 4;               // candidate slots per sample
constexpr int RF_MAX_PAIRS = 512;     // pairs per 128-sample workgroup of the pair kernel (two per lane)
constexpr int RF_MAX_UNION = 64;      // distinct prototypes among them (rows of its W tile); gflag 1: <= 32, 2: <= 64
constexpr uint32_t RF_NONE = 0xffffffffu, RF_ALL = 0xfefefefeu;

template <int NJ, int JT>
struct RefineCfg {
    static constexpr int NW = 4 * NJ;          // wavefronts: 4 (32 samples each) x NJ (parts of the list)
    static constexpr int ROWS = 32 * JT * NJ;  // list entries one workgroup can take
    static constexpr int X_PLANE = 128 * FKT, W_PLANE = ROWS * FKT;
    static constexpr int STAGE = 2 * (X_PLANE + W_PLANE);
    static constexpr int RING = FSTAGES * STAGE;
    static constexpr int VM = ROWS * 128 * 4;  // v_ij as float32, [list entry][sample]: aliases the ring
    static constexpr int MAIN = RING > VM ? RING : VM;
    static constexpr int OFF_TAB = MAIN;       // |w_j|^2 and 2 t_j / F16^2 of the list entries
    static constexpr int OFF_MISC = OFF_TAB + ROWS * 16;
    static constexpr int BYTES = OFF_MISC + 64;
    static constexpr int X_OPS = 16 / NW;            // LDS-DMA instructions per wavefront and k-tile
    static constexpr int W_OPS = (ROWS / 8) / NW;
    static constexpr int OPS = X_OPS + W_OPS;
    static_assert(16 % NW == 0 && (ROWS / 8) % NW == 0, "whole DMA instructions per wavefront");
    static constexpr int MAX_CNT = ROWS < 0xfd ? ROWS : 0xfd;  // list positions are bytes (0xfe, 0xff: markers)
};

template <int NJ, int JT>
__global__ __launch_bounds__(NJ * 256, 2) void refine_i8_kernel(
    const int8_t *__restrict__ xplanes, const double *__restrict__ sx, const double *__restrict__ xres,
    const double *__restrict__ xx, int64_t N, int d, int dpad, const int8_t *__restrict__ wplanes, int w_rows,
    const double *__restrict__ tw, const double *__restrict__ ww, const double *__restrict__ summary,
    const int32_t *__restrict__ order, const uint16_t *__restrict__ ulist, int ulist_stride,
    const uint32_t *__restrict__ ucount, const int32_t *__restrict__ queue, const uint32_t *__restrict__ queue_len,
    uint32_t *__restrict__ cand, uint8_t *__restrict__ gflag, uint32_t *__restrict__ sched_ctr,
    unsigned long long *__restrict__ rf_ctr, int32_t *__restrict__ pair_queue, uint32_t *__restrict__ pair_len, int nb) {
    // (one launch per list-length class, a few workgroups per CU walking the class's queue of 128-sample
    //  workgroups -- class_fill_kernel; pair_queue / pair_len: [2] queues of the pair kernel, by union size)
    using C = RefineCfg<NJ, JT>;
    __shared__ __attribute__((aligned(16))) char smem[C::BYTES];
    double *tab_y = reinterpret_cast<double *>(smem + C::OFF_TAB), *tab_c = tab_y + C::ROWS;
    uint32_t *misc = reinterpret_cast<uint32_t *>(smem + C::OFF_MISC);  // [0..7] union bits, [8] pairs, [9] all
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave & 3, wj = wave >> 2;
    const int lc = lane & 31, lh = lane >> 5;
    const int qn = (int)*queue_len;
    for (int entry = blockIdx.x; entry < qn; entry += gridDim.x) {
    if (entry != (int)blockIdx.x) __syncthreads();  // the previous workgroup's tables are done with
    const int group = queue[entry];
    const int64_t p0 = (int64_t)group * 128;
    const int cnt = __builtin_amdgcn_readfirstlane((int)ucount[group]);  // 1 <= cnt <= MAX_CNT (class_fill_kernel)
    const uint16_t *list = ulist + (size_t)group * ulist_stride;
    if (tid < 10) misc[tid] = 0u;
    for (int l = tid; l < C::ROWS; l += C::NW * 64) {
        const int j = (int)list[l < cnt ? l : cnt - 1];
        tab_y[l] = ww[j];
        tab_c[l] = 2.0 * tw[j] / (F16 * F16);
    }
    // ---- DMA sources ------------------------------------------------------------------------------
    // X: op o = u NW + wave, row block o % 8 (16 rows), plane o / 8; lane -> row 16 block + lane / 4,
    // LDS chunk lane % 4 holds the row's chunk (lane % 4) ^ swz(row) (the image the fragments read)
    const size_t xps = (size_t)N * dpad, wps = (size_t)w_rows * dpad;
    const int8_t *xsrc[C::X_OPS];
    int xdst[C::X_OPS];
#pragma unroll
    for (int u = 0; u < C::X_OPS; ++u) {
        const int o = u * C::NW + wave, blk = o & 7, pl = o >> 3;
        const int r = 16 * blk + (lane >> 2);
        int64_t p = p0 + r;
        p = p < N ? p : N - 1;
        xsrc[u] = xplanes + pl * xps + (size_t)order[p] * dpad + (((lane & 3) ^ ((r >> 2) & 3)) << 4);
        xdst[u] = pl * C::X_PLANE + blk * 1024;
    }
    // W (k-tile-major, the chunks of a row already swizzled by ITS index): op ow = u NW + wave, block
    // ow % (ROWS / 16), plane ow / (ROWS / 16); the row lands at its list position l
    const int8_t *wsrc[C::W_OPS];
    int wdst[C::W_OPS];
#pragma unroll
    for (int u = 0; u < C::W_OPS; ++u) {
        constexpr int NBLK = C::ROWS / 16;
        const int ow = u * C::NW + wave, blk = ow % NBLK, pl = ow / NBLK;
        const int l = 16 * blk + (lane >> 2);
        const int j = (int)list[l < cnt ? l : cnt - 1];
        wsrc[u] = wplanes + pl * wps + (size_t)j * FKT + (((lane & 3) ^ ((l >> 2) & 3) ^ ((j >> 2) & 3)) << 4);
        wdst[u] = 2 * C::X_PLANE + pl * C::W_PLANE + blk * 1024;
    }
    const int nkt = dpad / FKT;
    int i_kt = 0, i_stage = 0;
    auto issue = [&]() {
        char *stage = smem + i_stage;
#pragma unroll
        for (int u = 0; u < C::X_OPS; ++u) fdma16(xsrc[u] + (size_t)i_kt * FKT, stage + xdst[u]);
#pragma unroll
        for (int u = 0; u < C::W_OPS; ++u) fdma16(wsrc[u] + (size_t)i_kt * w_rows * FKT, stage + wdst[u]);
        i_stage = (i_stage == (FSTAGES - 1) * C::STAGE) ? 0 : i_stage + C::STAGE;
        ++i_kt;
    };
    // fragment offsets (bytes inside a stage; chunk (2 ks + lh) ^ swz = (2 ks) ^ (lh ^ swz))
    int xoff, woff[JT];
    {
        const int r = wi * 32 + lc;
        xoff = r * FKT + ((lh ^ ((r >> 2) & 3)) * 16);
    }
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
        const int r = wj * 32 * JT + jt * 32 + lc;
        woff[jt] = 2 * C::X_PLANE + r * FKT + ((lh ^ ((r >> 2) & 3)) * 16);
    }
    const int njt = min(JT, (cnt - wj * 32 * JT + 31) / 32);  // list tiles of this wavefront that hold entries
    v16i_t acc[JT][3];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int lv = 0; lv < 3; ++lv)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[jt][lv][r] = 0;

    issue();
    if (nkt > 1) issue();
    int r_stage = 0;
    for (int t = 0; t < nkt; ++t) {
        if (t + 1 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::OPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + 2 < nkt) issue();
        const char *stage = smem + r_stage;
        r_stage = (r_stage == (FSTAGES - 1) * C::STAGE) ? 0 : r_stage + C::STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            v4i_t x0 = *reinterpret_cast<const v4i_t *>(stage + (xoff ^ (ks * 32)));
            v4i_t x1 = *reinterpret_cast<const v4i_t *>(stage + C::X_PLANE + (xoff ^ (ks * 32)));
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                if (jt < njt) {  // (wave-uniform)
                    v4i_t w0 = *reinterpret_cast<const v4i_t *>(stage + (woff[jt] ^ (ks * 32)));
                    v4i_t w1 = *reinterpret_cast<const v4i_t *>(stage + C::W_PLANE + (woff[jt] ^ (ks * 32)));
                    acc[jt][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w0, x0, acc[jt][0], 0, 0, 0);
                    acc[jt][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w1, x0, acc[jt][1], 0, 0, 0);
                    acc[jt][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w0, x1, acc[jt][1], 0, 0, 0);
                    acc[jt][2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w1, x1, acc[jt][2], 0, 0, 0);
                }
            }
        }
    }
    __syncthreads();  // every wavefront is done with the ring: v_ij takes its place
    float *vm = reinterpret_cast<float *>(smem);
    {
        const int col = wi * 32 + lc;
        int64_t p = p0 + col;
        p = p < N ? p : N - 1;
        const double s_i = sx[order[p]];
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            if (jt < njt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int l = wj * 32 * JT + jt * 32 + 8 * (r >> 2) + (r & 3) + 4 * lh;
                    const double T = ((double)acc[jt][0][r] * 256.0 + (double)acc[jt][1][r]) * 256.0 + (double)acc[jt][2][r];
                    vm[l * 128 + col] = (float)(tab_y[l] - s_i * (tab_c[l] * T));
                }
            }
        }
    }
    __syncthreads();
    if (tid < 128) {
        const int64_t p = p0 + tid;
        if (p < N) {
            const int64_t i = order[p];
            const double yy_max = summary[2], rw = summary[3], xv = xx[i], rx = xres[i];
            const double xn = sqrt(xv) * (1.0 + 1e-9), wn = sqrt(yy_max) * (1.0 + 1e-9);
            const double rounding = 4.0 * (double)(d + 16) * 1.1102230246251565e-16 * (xv + yy_max);
            const double eps2 = 2.0 * (2.0 * (rx * (wn + rw) + xn * rw) * (1.0 + 1e-9) + rounding);
            float m = INFINITY;
            for (int l = 0; l < cnt; ++l) m = fminf(m, vm[l * 128 + tid]);  // (fminf passes over a NaN)
            // v <= min + 2 eps, with both sides' rounding to float32 on the safe side; a NaN anywhere keeps
            const double thr = (double)m + eps2 + 2.4e-7 * (fabs((double)m) + eps2);
            int n = 0;
            uint32_t slots = RF_NONE;
            for (int l = 0; l < cnt; ++l) {
                if (!((double)vm[l * 128 + tid] > thr)) {
                    if (n < RF_C) slots = (slots & ~(0xffu << (8 * n))) | ((uint32_t)l << (8 * n));
                    ++n;
                }
            }
            if (n > RF_C) {
                slots = RF_ALL;
                n = cnt;
                misc[9] = 1u;
            } else {
#pragma unroll
                for (int e = 0; e < RF_C; ++e) {
                    const uint32_t l = (slots >> (8 * e)) & 0xffu;
                    if (e < n) atomicOr(&misc[l >> 5], 1u << (l & 31));
                }
            }
            cand[p] = slots;
            atomicAdd(&misc[8], (uint32_t)n);
        }
    }
    __syncthreads();
    if (tid == 0) {
        int uni = 0;
        if (misc[9]) uni = cnt;
        else
            for (int w = 0; w < 8; ++w) uni += __popc(misc[w]);
        const bool ok = misc[8] <= (uint32_t)RF_MAX_PAIRS && uni <= RF_MAX_UNION;
        gflag[group] = ok ? 1 : 0;
        if (ok) {
            atomicSub(&sched_ctr[sched_bin((uint32_t)cnt)], 1u);  // not a workgroup of the MFMA stage after all
            atomicAdd(rf_ctr + 0, (unsigned long long)misc[8]);
            atomicAdd(rf_ctr + 1, 1ull);
            const int u = uni <= 32 ? 0 : 1;
            pair_queue[(size_t)u * nb + atomicAdd(&pair_len[u], 1u)] = group;
        } else {
            atomicAdd(rf_ctr + 2, 1ull);
        }
    }
    }  // (queue)
}

// which refinement class (tile) takes a 128-sample workgroup: queue 0 = lists of 1 .. rows0 entries, queue 1 =
// longer ones up to the largest tile's; longer still (or empty): nobody's, gflag stays 0
__global__ __launch_bounds__(256) void class_fill_kernel(const uint32_t *__restrict__ ucount, int nb, int rows0,
                                                         int rows1, int32_t *__restrict__ queue,
                                                         uint32_t *__restrict__ queue_len, uint8_t *__restrict__ gflag) {
    __shared__ uint32_t h[2], base[2];
    if (threadIdx.x < 2) h[threadIdx.x] = 0u;
    __syncthreads();
    const int b = blockIdx.x * 256 + threadIdx.x;
    int cls = -1;
    uint32_t r = 0;
    if (b < nb) {
        const int cnt = (int)ucount[b];
        cls = cnt < 1 ? -1 : (cnt <= rows0 ? 0 : (cnt <= rows1 ? 1 : -1));
        gflag[b] = 0;
        if (cls >= 0) r = atomicAdd(&h[cls], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 2) base[threadIdx.x] = h[threadIdx.x] ? atomicAdd(&queue_len[threadIdx.x], h[threadIdx.x]) : 0u;
    __syncthreads();
    if (cls >= 0) queue[(size_t)cls * nb + base[cls] + r] = b;
}

// ---- exact chain on the (sample, candidate) pairs ------------------------------------------------------
// One workgroup per refined 128-sample workgroup; lane = pair (two rounds of 256), X tile (128 gathered
// rows x KT) and W tile (the <= UN distinct candidates x KT) in a 4-stage LDS-DMA ring, three tiles in
// flight (the kernel is bound by the gathered rows it streams, once: bytes in flight are what counts).
// UN = 32 / 64: one queue each (refine_i8_kernel sorts the workgroups by the number of distinct candidates),
// a few workgroups per CU walk it.  K = 1.
template <typename XT, int UN>
__global__ __launch_bounds__(256, 2) void pair_exact_kernel(
    const XT *__restrict__ X, int64_t N, int d, int64_t ldx, const double *__restrict__ xx,
    const double *__restrict__ W, const double *__restrict__ ww, const int32_t *__restrict__ order,
    const uint16_t *__restrict__ ulist, int ulist_stride, const uint32_t *__restrict__ ucount,
    const uint32_t *__restrict__ cand, const int32_t *__restrict__ queue, const uint32_t *__restrict__ queue_len,
    int round_f32, int64_t *__restrict__ idx_out, double *__restrict__ dist_out) {
    constexpr int PX_STAGES = sizeof(XT) == 4 ? 4 : 3;  // (float64 rows: three stages, two tiles in flight)
    constexpr int XROW = KT * (int)sizeof(XT), XCH = XROW / 16, XD = 128 * XROW / 1024 / 4;
    constexpr int WD = UN / 32;  // W tile: UN rows x 128 B = UN / 8 instructions, UN / 32 per wavefront
    constexpr int S_XT = 128 * XROW, S_WT = UN * KT * 8, S_STAGE = S_XT + S_WT;
    static_assert(PX_STAGES * S_STAGE >= RF_MAX_PAIRS * 8, "the pairs' results take the ring's place");
    // ONE LDS object: with several, the compiler tags every access with the object's alias scope and then
    // answers each LDS read that may alias an LDS-DMA in flight with s_waitcnt vmcnt(0) -- no prefetch left
    constexpr int RING = PX_STAGES * S_STAGE;
    __shared__ __attribute__((aligned(16))) char smem[RING + UN * 4 + 130 * 4 + 8 * 4 + 2 * RF_MAX_PAIRS];
    int *rows = reinterpret_cast<int *>(smem + RING);
    int *off_s = rows + UN, *wtot = off_s + 128;
    uint32_t *bits = reinterpret_cast<uint32_t *>(wtot + 2);
    uint8_t *psamp = reinterpret_cast<uint8_t *>(bits + 8), *pslot = psamp + RF_MAX_PAIRS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qn = (int)*queue_len;
    for (int entry = blockIdx.x; entry < qn; entry += gridDim.x) {
    if (entry != (int)blockIdx.x) __syncthreads();  // the previous workgroup's results have been read
    const int group = queue[entry];
    const int64_t p0 = (int64_t)group * 128;
    const int cnt = (int)ucount[group];
    const uint16_t *list = ulist + (size_t)group * ulist_stride;

    // ---- pairs and the distinct candidates --------------------------------------------------------
    if (tid < 8) bits[tid] = 0u;
    uint32_t slots = RF_NONE;
    int n = 0;
    if (tid < 128 && p0 + tid < N) {
        slots = cand[p0 + tid];
        if (slots == RF_ALL) n = cnt;
        else
#pragma unroll
            for (int e = 0; e < RF_C; ++e) n += ((slots >> (8 * e)) & 0xffu) != 0xffu;
    }
    __syncthreads();
    if (tid < 128) {
        if (slots == RF_ALL) {
            for (int l = 0; l < cnt; ++l) atomicOr(&bits[l >> 5], 1u << (l & 31));
        } else {
#pragma unroll
            for (int e = 0; e < RF_C; ++e) {
                const uint32_t l = (slots >> (8 * e)) & 0xffu;
                if (l != 0xffu) atomicOr(&bits[l >> 5], 1u << (l & 31));
            }
        }
        // exclusive scan of n over the 128 sample threads
        int pre = n;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(pre, o, 64);
            if (lane >= o) pre += v;
        }
        if (lane == 63) wtot[tid >> 6] = pre;
        off_s[tid] = pre - n;
    }
    __syncthreads();
    if (tid >= 64 && tid < 128) off_s[tid] += wtot[0];
    const int npairs = wtot[0] + wtot[1];
    __syncthreads();
    auto slot_of = [&](uint32_t l) {
        int s = 0;
        for (uint32_t w = 0; w < (l >> 5); ++w) s += __popc(bits[w]);
        return s + __popc(bits[l >> 5] & ((1u << (l & 31)) - 1u));
    };
    if (tid < cnt && ((bits[tid >> 5] >> (tid & 31)) & 1u)) rows[slot_of((uint32_t)tid)] = (int)list[tid];
    if (tid < 128 && n > 0) {
        int q = off_s[tid];
        if (slots == RF_ALL) {
            for (int l = 0; l < cnt; ++l, ++q) { psamp[q] = (uint8_t)tid; pslot[q] = (uint8_t)l; }  // (every bit set: slot = l)
        } else {
#pragma unroll
            for (int e = 0; e < RF_C; ++e) {
                const uint32_t l = (slots >> (8 * e)) & 0xffu;
                if (l != 0xffu) { psamp[q] = (uint8_t)tid; pslot[q] = (uint8_t)slot_of(l); ++q; }
            }
        }
    }
    __syncthreads();
    int nun = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) nun += __popc(bits[w]);

    // ---- DMA sources (X as subset_exact_kernel; W: 8 rows per instruction) -----------------------------
    const XT *xsrc[XD];
#pragma unroll
    for (int u = 0; u < XD; ++u) {
        const int L = 64 * (XD * wave + u) + lane;
        const int r = L / XCH, cp = L % XCH;
        const int c = cp ^ ((r >> 1) & (XCH - 1));
        int64_t p = p0 + r;
        p = p < N ? p : N - 1;
        xsrc[u] = X + (int64_t)order[p] * ldx + c * (16 / (int)sizeof(XT));
    }
    const double *wsrc[WD];
#pragma unroll
    for (int u = 0; u < WD; ++u) {
        const int wr = 8 * (WD * wave + u) + (lane >> 3), wcp = lane & 7;
        const int wc = (wcp ^ ((wr >> 1) & 7)) * 2;
        wsrc[u] = W + (int64_t)rows[wr < nun ? wr : (nun > 0 ? nun - 1 : 0)] * d + wc;
    }
    const int nkt = d / KT;
    int i_kt = 0, i_stage = 0;
    auto issue = [&]() {
        char *stage = smem + i_stage;
        const int k0 = i_kt * KT;
#pragma unroll
        for (int u = 0; u < XD; ++u) fdma16(xsrc[u] + k0, stage + 1024 * (XD * wave + u));
#pragma unroll
        for (int u = 0; u < WD; ++u) fdma16(wsrc[u] + k0, stage + S_XT + 1024 * (WD * wave + u));
        i_stage = (i_stage == (PX_STAGES - 1) * S_STAGE) ? 0 : i_stage + S_STAGE;
        ++i_kt;
    };
    // this lane's (at most two) pairs
    int xo[2], xs_[2], wo[2], ws_[2];
    bool live[2];
    double acc[2] = {0.0, 0.0};
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
        const int q = tid + 256 * pp;
        live[pp] = q < npairs;
        const int r = live[pp] ? (int)psamp[q] : 0, sl = live[pp] ? (int)pslot[q] : 0;
        xo[pp] = r * XROW; xs_[pp] = (r >> 1) & (XCH - 1);
        wo[pp] = S_XT + sl * 128; ws_[pp] = (sl >> 1) & 7;
    }
    const bool second = npairs > 256;  // (workgroup-uniform)
    issue();
    if (nkt > 1) issue();
    if (PX_STAGES == 4 && nkt > 2) issue();
    int r_stage = 0;
    for (int t = 0; t < nkt; ++t) {
        // the wavefront's own DMAs of tile t have landed (PX_STAGES - 2 younger tiles may stay in flight),
        // the barrier covers everybody's
        if (PX_STAGES == 4 && t + 2 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (XD + WD)) : "memory");
        else if (t + 1 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XD + WD) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + PX_STAGES - 1 < nkt) issue();  // into the stage tile t - 1 was read from
        const char *stage = smem + r_stage;
        r_stage = (r_stage == (PX_STAGES - 1) * S_STAGE) ? 0 : r_stage + S_STAGE;
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            if (pp == 1 && !second) break;
            double xv[KT], wv[KT];
            if constexpr (sizeof(XT) == 4) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const v4f_t f = *reinterpret_cast<const v4f_t *>(stage + xo[pp] + ((c ^ xs_[pp]) << 4));
                    xv[4 * c] = (double)f.x; xv[4 * c + 1] = (double)f.y; xv[4 * c + 2] = (double)f.z; xv[4 * c + 3] = (double)f.w;
                }
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const d2_t f = *reinterpret_cast<const d2_t *>(stage + xo[pp] + ((c ^ xs_[pp]) << 4));
                    xv[2 * c] = f.x; xv[2 * c + 1] = f.y;
                }
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const d2_t f = *reinterpret_cast<const d2_t *>(stage + wo[pp] + ((c ^ ws_[pp]) << 4));
                wv[2 * c] = f.x; wv[2 * c + 1] = f.y;
            }
            // (the prototype is the first factor, as the A operand of the matrix instruction)
#pragma unroll
            for (int k = 0; k < KT; ++k) acc[pp] = fma(wv[k], xv[k], acc[pp]);
        }
    }
    __syncthreads();  // the ring is done with: the pairs' results take its place
    double *pv = reinterpret_cast<double *>(smem);
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
        const int q = tid + 256 * pp;
        if (q < npairs) {
            const int r = (int)psamp[q];
            const int j = rows[(int)pslot[q]];
            const double xi = xx[order[p0 + r]];
            double rv = (xi + (-2.0 * acc[pp])) + ww[j];
            if (!(rv > 0.0)) rv = (rv != rv) ? rv : 0.0;
            pv[q] = rv;
        }
    }
    __syncthreads();
    if (tid < 128 && p0 + tid < N) {
        double bv = INFINITY;
        int bj = 0x7fffffff;
        const int q0 = off_s[tid];
        for (int e = 0; e < n; ++e) {
            const double v = pv[q0 + e];
            const int j = rows[(int)pslot[q0 + e]];
            if (v < bv) { bv = v; bj = j; }  // (Best<1>::push: the pairs of a sample come with ascending index)
        }
        const int64_t i = order[p0 + tid];
        double dv = sqrt(bv);
        if (round_f32) dv = (double)(float)dv;
        idx_out[i] = (bj == 0x7fffffff) ? (int64_t)-1 : (int64_t)bj;
        dist_out[i] = dv;
    }
    }  // (queue)
}
