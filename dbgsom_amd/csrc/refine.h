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
constexpr int RF_SEGS = 2;            // tile-sized segments a candidate list may have in the refinement
constexpr int RF_MAX_PAIRS = 512;     // pairs per 128-sample workgroup of the pair kernel (two per lane)
constexpr int RF_HASH = 512;          // slots of its table of distinct candidates
constexpr unsigned long long RF_NONE = 0xffffffffffffffffull;  // four empty slots (prototype ids are < 0xffff)
constexpr int OV_REC = 32;            // uint16 per record of an overflowing sample: [count | up to 31 candidates]
constexpr int OV_CAP = 1 << 17;       // records (samples beyond: the whole list, as with count 0)

// Workgroups are dealt to the 8 XCDs round robin (b % 8), each XCD with an L2 of its own.  Neighbouring
// 128- / 64-sample workgroups share their prototypes (samples arrive bucketed): give every XCD a contiguous
// eighth of them, so that a prototype tile is fetched into ONE L2 instead of eight.  Bijective on a grid
// that is a multiple of 8; the order only decides who runs where.
__device__ __forceinline__ int xcd_group(int b, int n_groups8) { return (b & 7) * (n_groups8 >> 3) + (b >> 3); }

// the least float32 that is >= t (a NaN stays a NaN: every comparison with it fails, everything is kept)
__device__ __forceinline__ float f32_at_least(double t) {
    float f = (float)t;
    if ((double)f < t) f = __uint_as_float(f >= 0.f ? __float_as_uint(f) + 1u : __float_as_uint(f) - 1u);
    return f;
}

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
    static constexpr int OFF_META = OFF_MISC + 64;   // per sample: 2 eps_i, s_i (doubles), its index (int32)
    static constexpr int BYTES = OFF_META + 128 * 20;
    static constexpr int X_OPS = 16 / NW;            // LDS-DMA instructions per wavefront and k-tile
    static constexpr int W_OPS = (ROWS / 8) / NW;
    static constexpr int OPS = X_OPS + W_OPS;
    static_assert(16 % NW == 0 && (ROWS / 8) % NW == 0, "whole DMA instructions per wavefront");
    static constexpr int MAX_CNT = ROWS;
};

#ifndef REFINE_DEPHASE
#define REFINE_DEPHASE 1
#endif
template <int NJ, int JT>
__global__ __launch_bounds__(NJ * 256, NJ < 2 ? 2 : NJ) void refine_i8_kernel(
    const int8_t *__restrict__ xplanes, const double *__restrict__ sx, const double *__restrict__ xres,
    const double *__restrict__ xx, int64_t N, int d, int dpad, const int8_t *__restrict__ wplanes, int w_rows,
    const double *__restrict__ tw, const double *__restrict__ ww, const double *__restrict__ summary,
    const int32_t *__restrict__ order, const uint16_t *__restrict__ ulist, int ulist_stride,
    const uint32_t *__restrict__ ucount, const int32_t *__restrict__ queue, const uint32_t *__restrict__ queue_len,
    unsigned long long *__restrict__ cand, int64_t *__restrict__ rbest, unsigned long long *__restrict__ rf_ctr,
    int32_t *__restrict__ ovf, uint32_t *__restrict__ ovf_len, uint16_t *__restrict__ ovf_cand, int defer_M,
    int64_t *__restrict__ idx_out, double *__restrict__ dist_out) {
    // defer_M > 0 (= M; FilteredCall::defer_dist): a sample left with ONE candidate has its winner -- written
    // here with dist = -1, evaluated by the caller on its own pass over the rows -- and leaves the pair
    // kernel's work (bucket key M: behind every real bucket); only the undecided samples are bucketed.
    // (one launch per list-length class, a few workgroups per CU walking the class's queue of 128-sample
    //  workgroups -- class_fill_kernel)
    using C = RefineCfg<NJ, JT>;
    __shared__ __attribute__((aligned(16))) char smem[C::BYTES];
    double *tab_y = reinterpret_cast<double *>(smem + C::OFF_TAB), *tab_c = tab_y + C::ROWS;
    uint32_t *misc = reinterpret_cast<uint32_t *>(smem + C::OFF_MISC);  // [8] pairs of this workgroup
    double *eps_s = reinterpret_cast<double *>(smem + C::OFF_META), *sx_s = eps_s + 128;
    int32_t *isamp_s = reinterpret_cast<int32_t *>(sx_s + 128);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave & 3, wj = wave >> 2;
    const int lc = lane & 31, lh = lane >> 5;
    const int qn = (int)*queue_len;
    // (XCD x = blockIdx % 8 walks the x-th eighth of the queue, gridDim / 8 workgroups side by side)
    const int per_xcd = (qn + 7) / 8, q_lo = (int)(blockIdx.x & 7) * per_xcd, q_hi = min(qn, q_lo + per_xcd);
    const int q_first = q_lo + (int)(blockIdx.x >> 3), q_step = (int)(gridDim.x >> 3);
    if ((int)(blockIdx.x >> 3) >= q_step) return;  // (a grid that is no multiple of 8: the odd workgroups have no part)
    // (the loads a workgroup's set-up hangs on are taken out of its way: the NEXT entry's group and list length
    //  are fetched while this one runs; the set-up itself issues what needs one round trip -- list entries,
    //  sample indices -- then the first two tiles' DMAs, and only then what needs a second: norms, scales)
    int group_next = q_first < q_hi ? queue[q_first] : 0;
    int cnt_next = (int)ucount[group_next];
    for (int entry = q_first; entry < q_hi; entry += q_step) {
    if (entry != q_first) __syncthreads();  // the previous workgroup's tables are done with
    const int group = __builtin_amdgcn_readfirstlane(group_next);
    const int64_t p0 = (int64_t)group * 128;
    const int cnt_all = __builtin_amdgcn_readfirstlane(cnt_next);  // 1 <= cnt_all <= RF_SEGS ROWS (class_fill_kernel)
    if (entry + q_step < q_hi) group_next = queue[entry + q_step];
    const uint16_t *list_all = ulist + (size_t)group * ulist_stride;
    // A list longer than the tile (a workgroup whose samples come from two clusters: 4 % of the C5 shard's) is
    // taken in segments of ROWS entries, each a pass over the planes of its own; a sample's candidates of
    // a segment are chosen against the SEGMENT's minimum (a superset), merged by the sample's thread and
    // filtered against the minimum of the whole list at the end.
    const int nseg = (cnt_all + C::ROWS - 1) / C::ROWS;
    float g_m = INFINITY, g_v[RF_SEGS * RF_C];
    int g_best = (int)list_all[0], g_n = 0, g_id[RF_SEGS * RF_C];
    bool g_ovf = false;
    int g_k = -1, g_rc = 0;  // this sample's record of overflowing candidates (index, entries so far)
#pragma unroll
    for (int e = 0; e < RF_SEGS * RF_C; ++e) { g_v[e] = 0.f; g_id[e] = 0xffff; }
    for (int seg = 0; seg < nseg; ++seg) {
    if (seg) __syncthreads();  // the previous segment's v_ij have been read
    const uint16_t *list = list_all + seg * C::ROWS;
    const int cnt = min(C::ROWS, cnt_all - seg * C::ROWS);
    if (tid < 10 && seg == 0) misc[tid] = 0u;
    static_assert(C::ROWS <= C::NW * 64, "one table entry per thread");
    const int tj = tid < C::ROWS ? (int)list[tid < cnt ? tid : cnt - 1] : 0;          // (first hop)
    int64_t p_me = p0 + (tid & 127);
    p_me = p_me < N ? p_me : N - 1;
    const int i_me = (seg == 0 && tid < 128) ? order[p_me] : 0;                          // (first hop)
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
    // (16-row blocks behind the end of the list are not fetched: their LDS rows keep whatever they held, and
    //  nothing reads the v_ij of a row >= cnt; w_ops = this wavefront's live instructions, the same every k-tile)
    const int8_t *wsrc[C::W_OPS];
    int wdst[C::W_OPS];
    bool wlive[C::W_OPS];
    int w_ops = 0;
#pragma unroll
    for (int u = 0; u < C::W_OPS; ++u) {
        constexpr int NBLK = C::ROWS / 16;
        const int ow = u * C::NW + wave, blk = ow % NBLK, pl = ow / NBLK;
        const int l = 16 * blk + (lane >> 2);
        const int j = (int)list[l < cnt ? l : cnt - 1];
        wsrc[u] = wplanes + pl * wps + (size_t)j * FKT + (((lane & 3) ^ ((l >> 2) & 3) ^ ((j >> 2) & 3)) << 4);
        wdst[u] = 2 * C::X_PLANE + pl * C::W_PLANE + blk * 1024;
        wlive[u] = 16 * blk < cnt;
        w_ops += wlive[u] ? 1 : 0;
    }
    w_ops = __builtin_amdgcn_readfirstlane(w_ops);
    const int nkt = dpad / FKT;
    int i_kt = 0, i_stage = 0;
    auto issue = [&]() {
        char *stage = smem + i_stage;
#pragma unroll
        for (int u = 0; u < C::X_OPS; ++u) fdma16(xsrc[u] + (size_t)i_kt * FKT, stage + xdst[u]);
#pragma unroll
        for (int u = 0; u < C::W_OPS; ++u)
            if (wlive[u]) fdma16(wsrc[u] + (size_t)i_kt * w_rows * FKT, stage + wdst[u]);
        i_stage = (i_stage == (FSTAGES - 1) * C::STAGE) ? 0 : i_stage + C::STAGE;
        ++i_kt;
    };
    auto wait_tile = [&]() {  // all but this wavefront's youngest tile have landed (s_waitcnt needs an immediate)
        static_assert(C::W_OPS <= 4, "cases of wait_tile");
        if (w_ops == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::X_OPS) : "memory");
        else if (w_ops == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::X_OPS + 1) : "memory");
        else if (w_ops == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::X_OPS + 2) : "memory");
        else if (w_ops == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::X_OPS + 3) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::X_OPS + 4) : "memory");
    };
    // fragment offsets (bytes inside a stage; chunk (2 ks + lh) ^ swz = (2 ks) ^ (lh ^ swz))
    int xoff, woff[JT];
    {
        const int r = wi * 32 + lc;
        xoff = r * FKT + ((lh ^ ((r >> 2) & 3)) * 16);
    }
    // (the 32-entry tiles of the list are dealt to the NJ parts round robin -- tile jt NJ + wj -- so that a short
    //  list keeps every wavefront busy: 136 entries in a 256-row tile are 3 + 2 tiles, not 4 + 1)
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
        const int r = (jt * NJ + wj) * 32 + lc;
        woff[jt] = 2 * C::X_PLANE + r * FKT + ((lh ^ ((r >> 2) & 3)) * 16);
    }
    const int njt = min(JT, ((cnt + 31) / 32 - wj + NJ - 1) / NJ);  // list tiles of this wavefront that hold entries
    v16i_t acc[JT][3];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int lv = 0; lv < 3; ++lv)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[jt][lv][r] = 0;

    issue();
    if (nkt > 1) issue();
    {   // second hop, under the DMAs in flight: the tables of the list entries, the samples' constants
        double ty = 0.0, tc = 0.0, xv = 0.0, rx = 0.0, sv = 0.0;
        if (tid < C::ROWS) { ty = ww[tj]; tc = tw[tj]; }
        if (seg == 0 && tid < 128) { xv = xx[i_me]; rx = xres[i_me]; sv = sx[i_me]; }
        if (tid < C::ROWS) {
            tab_y[tid] = ty;
            tab_c[tid] = 2.0 * tc / (F16 * F16);
        }
        if (seg == 0 && tid < 128) {
            const double yy_max = summary[2], rw = summary[3];
            const double xn = sqrt(xv) * (1.0 + 1e-9), wn = sqrt(yy_max) * (1.0 + 1e-9);
            const double rounding = 4.0 * (double)(d + 16) * 1.1102230246251565e-16 * (xv + yy_max);
            eps_s[tid] = 2.0 * (2.0 * (rx * (wn + rw) + xn * rw) * (1.0 + 1e-9) + rounding);
            sx_s[tid] = sv;
            isamp_s[tid] = i_me;
        }
    }
    int r_stage = 0;
    for (int t = 0; t < nkt; ++t) {
        if (t + 1 < nkt) wait_tile();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // (the two wavefronts of a SIMD -- wave w and w + 4 -- take turns: one issues its LDS-DMAs of tile
        //  t + 2 in front of its matrix products, the other behind them, so that the one's issue stalls run
        //  under the other's products instead of both stalling right behind the barrier)
        const bool issue_first = REFINE_DEPHASE == 0 || NJ < 2 || (wj & 1) == 0;
        if (issue_first && t + 2 < nkt) issue();
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
        if (!issue_first && t + 2 < nkt) issue();
    }
    __syncthreads();  // every wavefront is done with the ring: v_ij takes its place
    float *vm = reinterpret_cast<float *>(smem);
    {
        const int col = wi * 32 + lc;
        const double s_i = sx_s[col];
        if (seg == nseg - 1 && entry + q_step < q_hi) cnt_next = (int)ucount[group_next];  // (its first hop is long back)
        // (the lane's first list row, made opaque here: left visible, the 48 JT table / output addresses are
        //  hoisted out of the queue loop, kept across the matrix loop and spilled)
        int lb = wj * 32 + 4 * lh;
        asm volatile("" : "+v"(lb));
        const double *ty = tab_y + lb, *tc = tab_c + lb;
        float *vcol = vm + lb * 128 + col;
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            if (jt < njt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int lo = jt * NJ * 32 + 8 * (r >> 2) + (r & 3);
                    const double T = ((double)acc[jt][0][r] * 256.0 + (double)acc[jt][1][r]) * 256.0 + (double)acc[jt][2][r];
                    vcol[lo * 128] = (float)(ty[lo] - s_i * (tc[lo] * T));
                    // (four values at a time: left to itself the scheduler converts all 16 JT accumulators first)
                    if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    __syncthreads();
    {
        // selection: SP threads per sample, each over a contiguous part of the list (parts in list order, so
        // that candidates come out ascending); scratch behind the v_ij matrix, inside the ring's bytes
        constexpr int SP = C::NW * 64 / 128 > 4 ? 4 : C::NW * 64 / 128;
        static_assert(C::MAIN - C::VM >= 4 * 128 * 24, "selection scratch behind the v_ij matrix");
        float *pm = reinterpret_cast<float *>(smem + C::MAIN - 4 * 128 * 24);
        int *plm = reinterpret_cast<int *>(pm + 4 * 128), *pn = plm + 4 * 128;
        uint32_t *ppos = reinterpret_cast<uint32_t *>(pn + 4 * 128);
        unsigned long long *psl = reinterpret_cast<unsigned long long *>(ppos + 4 * 128);
        const int part = tid >> 7, sidx = tid & 127;
        const bool act = part < SP && p0 + sidx < N;
        const int l0 = part * cnt / SP, l1 = (part + 1) * cnt / SP;
        if (act) {
            float m = INFINITY;
            int lm = l0;
            // (eight LDS reads in flight at a time: one after the other, a read's latency per list entry)
            for (int lb8 = l0; lb8 < l1; lb8 += 8) {  // (a NaN never becomes the minimum)
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = vm[min(lb8 + u, l1 - 1) * 128 + sidx];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (lb8 + u < l1 && v[u] < m) { m = v[u]; lm = lb8 + u; }
            }
            pm[part * 128 + sidx] = m;
            plm[part * 128 + sidx] = lm;
        }
        __syncthreads();
        double eps2 = 0.0;
        if (act) {
            eps2 = eps_s[sidx];
            float m = INFINITY;
#pragma unroll
            for (int q = 0; q < SP; ++q) m = fminf(m, pm[q * 128 + sidx]);  // (fminf passes over a NaN)
            // v <= min + 2 eps, with both sides' rounding to float32 on the safe side; a NaN anywhere keeps
            const double thr = (double)m + eps2 + 2.4e-7 * (fabs((double)m) + eps2);
            const float thr_f = f32_at_least(thr);   // (compared in float32: a bound >= thr keeps at least as many)
            int n = 0;
            unsigned long long slots = RF_NONE;
            uint32_t pos = 0u;
            for (int lb8 = l0; lb8 < l1; lb8 += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = vm[min(lb8 + u, l1 - 1) * 128 + sidx];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int l = lb8 + u;
                    if (l < l1 && !(v[u] > thr_f)) {
                        if (n < RF_C) {
                            slots = (slots & ~(0xffffull << (16 * n))) | ((unsigned long long)list[l] << (16 * n));
                            pos |= (uint32_t)l << (8 * n);
                        }
                        ++n;
                    }
                }
            }
            psl[part * 128 + sidx] = slots;
            ppos[part * 128 + sidx] = pos;
            pn[part * 128 + sidx] = n;
        }
        __syncthreads();
        if (act && part == 0) {  // this segment's candidates (ascending) behind those of the earlier ones
            int n = 0;
#pragma unroll
            for (int q = 0; q < SP; ++q) {
                const float mq = pm[q * 128 + sidx];
                if (mq < g_m) { g_m = mq; g_best = (int)list[plm[q * 128 + sidx]]; }
                const int nq = pn[q * 128 + sidx];
                const unsigned long long sq = psl[q * 128 + sidx];
                const uint32_t pq = ppos[q * 128 + sidx];
#pragma unroll
                for (int e = 0; e < RF_C; ++e) {
                    if (e < nq && n + e < RF_C) {
                        const float v = vm[((pq >> (8 * e)) & 0xffu) * 128 + sidx];
                        const int id = (int)((sq >> (16 * e)) & 0xffffull);
#pragma unroll
                        for (int z = 0; z < RF_SEGS * RF_C; ++z)  // (no dynamic register indexing)
                            if (z == g_n + n + e) { g_v[z] = v; g_id[z] = id; }
                    }
                }
                n += nq;
            }
            if (n > RF_C) {
                // more than the slots hold: ALL of this segment's candidates go to the sample's record for
                // overflow_exact_kernel (the segment's v_ij are in LDS now; same test as above)
                g_ovf = true;
                if (g_k < 0) g_k = (int)atomicAdd(ovf_len, 1u);
                if ((uint32_t)g_k < (uint32_t)OV_CAP) {
                    float m = INFINITY;
#pragma unroll
                    for (int q = 0; q < SP; ++q) m = fminf(m, pm[q * 128 + sidx]);
                    const float thr_f = f32_at_least((double)m + eps2 + 2.4e-7 * (fabs((double)m) + eps2));
                    uint16_t *rec = ovf_cand + (size_t)g_k * OV_REC;
                    for (int lb8 = 0; lb8 < cnt; lb8 += 8) {
                        float v[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) v[u] = vm[min(lb8 + u, cnt - 1) * 128 + sidx];
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            if (lb8 + u < cnt && !(v[u] > thr_f)) {
                                if (g_rc < OV_REC - 1) rec[1 + g_rc] = list[lb8 + u];
                                ++g_rc;
                            }
                    }
                }
            } else {
                g_n += n;
            }
        }
    }
    }  // (segments)
    if (tid < 128 && p0 + tid < N) {
        // the candidates of all segments against the minimum of the whole list
        const int64_t isamp = isamp_s[tid];
        const double eps2 = eps_s[tid];
        const double thr = (double)g_m + eps2 + 2.4e-7 * (fabs((double)g_m) + eps2);
        int n = 0;
        unsigned long long slots = RF_NONE;
#pragma unroll
        for (int z = 0; z < RF_SEGS * RF_C; ++z) {
            if (z < g_n && !((double)g_v[z] > thr)) {
                if (n < RF_C) slots = (slots & ~(0xffffull << (16 * n))) | ((unsigned long long)g_id[z] << (16 * n));
                ++n;
            }
        }
        if (n > RF_C || g_ovf) {  // too many to keep apart: exactly, by overflow_exact_kernel
            slots = RF_NONE;
            atomicAdd(rf_ctr + 2, 1ull);
            if (g_k < 0) g_k = (int)atomicAdd(ovf_len, 1u);
            ovf[2 * (size_t)g_k] = (int32_t)isamp;
            ovf[2 * (size_t)g_k + 1] = group;
            if ((uint32_t)g_k < (uint32_t)OV_CAP) {
                // the record: the candidates of the segments that overflowed (written there), then those of
                // the others (registers, against the minimum of the whole list); more than it holds: count 0
                // = the workgroup's whole list
                uint16_t *rec = ovf_cand + (size_t)g_k * OV_REC;
#pragma unroll
                for (int z = 0; z < RF_SEGS * RF_C; ++z)
                    if (z < g_n && !((double)g_v[z] > thr)) {
                        if (g_rc < OV_REC - 1) rec[1 + g_rc] = (uint16_t)g_id[z];
                        ++g_rc;
                    }
                rec[0] = (uint16_t)(g_rc < OV_REC ? g_rc : 0);
            }
            n = 0;
        }
        int64_t key = (int64_t)g_best;  // bucket key of the pair kernel: any prototype does, the likely winner is best
        if (defer_M > 0) {
            if (n == 1) {
                idx_out[isamp] = (int64_t)(slots & 0xffffull);
                dist_out[isamp] = -1.0;
                slots = RF_NONE;
                n = 0;
            }
            if (n == 0) key = (int64_t)defer_M;   // (decided, or the overflow kernel's)
        }
        cand[isamp] = slots;
        rbest[isamp] = key;
        atomicAdd(&misc[8], (uint32_t)n);
    }
    __syncthreads();
    if (tid == 0) {
        atomicAdd(rf_ctr + 0, (unsigned long long)misc[8]);
        atomicAdd(rf_ctr + 1, 1ull);
    }
    }  // (queue)
}

// which refinement class (tile) takes a 128-sample workgroup: queue 0 = lists of 1 .. rows0 entries, queue 1 =
// longer ones up to the largest tile's (gflag 1, out of the matrix-core stage's bin counts); longer still (or
// empty): gflag 0, the matrix-core stage's, and its samples get a bucket key (their seed) and no candidates
// for the pair kernel
__global__ __launch_bounds__(256) void class_fill_kernel(const uint32_t *__restrict__ ucount, int nb, int rows0,
                                                         int rows1, int32_t *__restrict__ queue,
                                                         uint32_t *__restrict__ queue_len, uint8_t *__restrict__ gflag,
                                                         uint32_t *__restrict__ sched_ctr,
                                                         const int32_t *__restrict__ order,
                                                         const int64_t *__restrict__ prev, int64_t N, int M,
                                                         unsigned long long *__restrict__ cand,
                                                         int64_t *__restrict__ rbest, int defer_M) {
    __shared__ uint32_t h[3], base[2];
    __shared__ int skipped[256];
    if (threadIdx.x < 3) h[threadIdx.x] = 0u;
    __syncthreads();
    const int b = blockIdx.x * 256 + threadIdx.x;
    int cls = -1;
    uint32_t r = 0;
    if (b < nb) {
        const int cnt = (int)ucount[b];
        cls = cnt < 1 ? -1 : (cnt <= rows0 ? 0 : (cnt <= rows1 ? 1 : -1));
        gflag[b] = cls >= 0 ? 1 : 0;
        if (cls >= 0) {
            r = atomicAdd(&h[cls], 1u);
            atomicSub(&sched_ctr[sched_bin((uint32_t)cnt)], 1u);  // not a workgroup of the matrix-core stage after all
        } else {
            skipped[atomicAdd(&h[2], 1u)] = b;
        }
    }
    __syncthreads();
    if (threadIdx.x < 2) base[threadIdx.x] = h[threadIdx.x] ? atomicAdd(&queue_len[threadIdx.x], h[threadIdx.x]) : 0u;
    const int nskip = (int)h[2];
    for (int e = threadIdx.x; e < nskip * 128; e += 256) {
        const int64_t p = (int64_t)skipped[e >> 7] * 128 + (e & 127);
        if (p < N) {
            const int64_t i = order[p], q = prev[i];
            cand[i] = RF_NONE;
            rbest[i] = defer_M > 0 ? (int64_t)defer_M : ((q >= 0 && q < M) ? q : 0);
        }
    }
    __syncthreads();
    if (cls >= 0) queue[(size_t)cls * nb + base[cls] + r] = b;
}

// ---- exact chain on the (sample, candidate) pairs ------------------------------------------------------
// The samples arrive bucketed by their refined best prototype (order2), so the PS = 64 samples of a workgroup
// share a handful of candidates: lane = pair (<= 256), X tile (64 gathered rows x 256 bytes) and W tile (UN
// distinct candidates x the same features) in a 3-stage LDS-DMA ring, two tiles in flight.  The kernel is
// bound by the gathered rows it streams, once -- and by how they are asked for: 64-byte pieces of a row
// (one 16-feature tile at a time, as subset_exact_kernel reads them) reached 4.4 TB/s, every piece a
// different DRAM page; a piece here is 256 contiguous bytes.  More than UN distinct candidates (rare):
// further passes over the rows, UN candidates each.  K = 1.
constexpr int PS = 64;           // samples per workgroup
constexpr int PX_PAIRS = PS * RF_C;
template <typename XT, int UN, int PIECE>
__global__ __launch_bounds__(256, 2) void pair_exact_kernel(
    const XT *__restrict__ X, int64_t N, int d, int64_t ldx, const double *__restrict__ xx,
    const double *__restrict__ W, const double *__restrict__ ww, const int32_t *__restrict__ order2,
    const unsigned long long *__restrict__ cand, int round_f32, int64_t *__restrict__ idx_out,
    double *__restrict__ dist_out, unsigned long long *__restrict__ rf_ctr, const uint32_t *__restrict__ n_active) {
    // n_active (FilteredCall::defer_dist): the first *n_active positions of order2 hold the undecided samples
    constexpr int STAGES = 3;
    constexpr int ES = (int)sizeof(XT);
    constexpr int KP = PIECE / ES;              // features per tile: 64 (float32, bfloat16 in 128-byte pieces) / 32 (float64)
    constexpr int XCH = PIECE / 16;             // 16-byte chunks of an X piece
    constexpr int XSH = PIECE == 256 ? 0 : 1;   // rows per 256 bytes of the X tile, as a shift (swizzle below)
    constexpr int WROW = KP * 8, WCH = WROW / 16;  // W piece: 512 / 256 bytes, 32 / 16 chunks
    constexpr int S_XT = PS * PIECE, S_WT = UN * WROW, S_STAGE = S_XT + S_WT;
    constexpr int XD = S_XT / 1024 / 4, WDI = S_WT / 1024;  // DMA instructions: X per wavefront; W in all
    static_assert(PIECE == 256 || PIECE == 128, "X pieces of one or two cache lines");
    static_assert(WDI % 4 == 0, "whole W instructions per wavefront");
    constexpr int WD = WDI / 4;
    static_assert(STAGES * S_STAGE >= PX_PAIRS * 8, "the pairs' results take the ring's place");
    // ONE LDS object: with several, the compiler tags every access with its object's alias scope and then
    // answers each LDS read that may alias an LDS-DMA in flight with s_waitcnt vmcnt(0) -- no prefetch left
    constexpr int RING = STAGES * S_STAGE;
    constexpr int O_TAB = RING, O_ROWS = O_TAB + RF_HASH * 4, O_DENSE = O_ROWS + PX_PAIRS * 4,
                  O_PSLOT = O_DENSE + RF_HASH * 2, O_OFF = O_PSLOT + PX_PAIRS * 2, O_MISC = O_OFF + PS * 4,
                  O_PSAMP = O_MISC + 16 * 4, BYTES = O_PSAMP + PX_PAIRS;
    __shared__ __attribute__((aligned(16))) char smem[BYTES];
    uint32_t *table = reinterpret_cast<uint32_t *>(smem + O_TAB);
    int *rows = reinterpret_cast<int *>(smem + O_ROWS);
    uint16_t *dense = reinterpret_cast<uint16_t *>(smem + O_DENSE), *pslot = reinterpret_cast<uint16_t *>(smem + O_PSLOT);
    int *off_s = reinterpret_cast<int *>(smem + O_OFF), *misc = reinterpret_cast<int *>(smem + O_MISC);
    uint8_t *psamp = reinterpret_cast<uint8_t *>(smem + O_PSAMP);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t NA = n_active ? (int64_t)*n_active : N;   // (the positions that can hold pairs)
    // (the eight XCDs share the ACTIVE workgroups: an eighth of them each, contiguous)
    const int ng8 = (int)(((NA + PS - 1) / PS + 7) / 8 * 8);
    if ((int)blockIdx.x >= ng8) return;
    const int64_t p0 = (int64_t)xcd_group((int)blockIdx.x, ng8) * PS;
    if (p0 >= NA) return;

    // ---- pairs and the distinct candidates --------------------------------------------------------
    for (int h = tid; h < RF_HASH; h += 256) table[h] = 0xffffffffu;
    unsigned long long slots = RF_NONE;
    int64_t isamp = -1;
    int n = 0;
    if (tid < PS) {  // (wavefront 0)
        if (p0 + tid < NA) {
            isamp = order2[p0 + tid];
            slots = cand[isamp];
#pragma unroll
            for (int e = 0; e < RF_C; ++e) n += ((slots >> (16 * e)) & 0xffffull) != 0xffffull;
        }
        int pre = n;  // exclusive scan of n over the sample threads
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(pre, o, 64);
            if (lane >= o) pre += v;
        }
        if (lane == 63) misc[0] = pre;
        off_s[tid] = pre - n;
    }
    __syncthreads();
    const int npairs = misc[0];
    if (npairs == 0) return;  // (samples of workgroups the refinement did not take: the matrix-core stage's)
    if (tid < PS) {
        int q = off_s[tid];
#pragma unroll
        for (int e = 0; e < RF_C; ++e) {
            const uint32_t id = (uint32_t)((slots >> (16 * e)) & 0xffffull);
            if (id != 0xffffu) {
                uint32_t h = (id * 2654435761u) >> 23;  // 9 bits
                for (;;) {
                    const uint32_t old = atomicCAS(&table[h], 0xffffffffu, id);
                    if (old == 0xffffffffu || old == id) break;
                    h = (h + 1) & (RF_HASH - 1);
                }
                psamp[q] = (uint8_t)tid;
                pslot[q] = (uint16_t)h;  // (the table slot for now)
                ++q;
            }
        }
    }
    __syncthreads();
    {   // dense numbering of the occupied table slots: 2 per thread, scan over the 256 threads
        const bool o0 = table[2 * tid] != 0xffffffffu, o1 = table[2 * tid + 1] != 0xffffffffu;
        int pre = (int)o0 + (int)o1;
        const int mine = pre;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(pre, o, 64);
            if (lane >= o) pre += v;
        }
        if (lane == 63) misc[4 + wave] = pre;
        __syncthreads();
        int basew = 0;
        for (int w = 0; w < wave; ++w) basew += misc[4 + w];
        const int first = basew + pre - mine;
        if (o0) { dense[2 * tid] = (uint16_t)first; rows[first] = (int)table[2 * tid]; }
        if (o1) { dense[2 * tid + 1] = (uint16_t)(first + (int)o0); rows[first + (int)o0] = (int)table[2 * tid + 1]; }
    }
    __syncthreads();
    const int nun = misc[4] + misc[5] + misc[6] + misc[7];
    if (tid < npairs) pslot[tid] = dense[pslot[tid]];
    __syncthreads();

    // ---- DMA sources: an X instruction = 4 rows x 256 bytes, a W instruction = 1024 / WROW rows ----------
    // LDS chunk cp of row r holds the row's chunk cp ^ swz(r), swz = the row's number among the rows that share
    // its banks: a lane reads its pair's row chunk by chunk, the lanes of a 16-lane read group hit 16 different
    // bank quads (or the same address)
    const XT *xsrc[XD];
    int xchunk[XD];
#pragma unroll
    for (int u = 0; u < XD; ++u) {
        const int L = 64 * (XD * wave + u) + lane;
        const int r = L / XCH, cp = L % XCH;
        xchunk[u] = cp ^ ((r >> XSH) & (XCH - 1));
        int64_t p = p0 + r;
        p = p < NA ? p : NA - 1;
        xsrc[u] = X + (int64_t)order2[p] * ldx;
    }
    const int nkt = (d + KP - 1) / KP;
    const int npass = (nun + UN - 1) / UN;
    if (tid == 0) atomicAdd(rf_ctr + 3, (unsigned long long)nun);  // (diagnostics: distinct candidates met)
    double acc = 0.0;
    const bool mine = tid < npairs;
    const int prow = mine ? (int)psamp[tid] : 0, psl = mine ? (int)pslot[tid] : -1;
    const int xo = prow * PIECE, xsw = (prow >> XSH) & (XCH - 1);
    for (int pass = 0; pass < npass; ++pass) {
        if (pass) __syncthreads();  // the previous pass's last tiles have been read
        const double *wsrc[WD];
        int wchunk[WD];
#pragma unroll
        for (int u = 0; u < WD; ++u) {
            const int L = 64 * (WD * wave + u) + lane;
            const int wr = L / WCH, wcp = L % WCH;
            wchunk[u] = wcp ^ (wr & 15);
            const int sl = pass * UN + wr;
            wsrc[u] = W + (int64_t)rows[sl < nun ? sl : nun - 1] * d;
        }
        const int sl = psl - pass * UN;
        const bool live = psl >= 0 && sl >= 0 && sl < UN;
        const int wo = S_XT + (live ? sl : 0) * WROW, wsw = (live ? sl : 0) & 15;
        int i_kt = 0, i_stage = 0;
        auto issue = [&]() {
            char *stage = smem + i_stage;
            const int k0 = i_kt * KP;
            // (a chunk behind the row's end -- the last tile of a row whose length is no multiple of the
            //  tile -- is never read: any address inside the row will do)
#pragma unroll
            for (int u = 0; u < XD; ++u) {
                const int k = k0 + xchunk[u] * (16 / ES);
                fdma16(xsrc[u] + (k < d ? k : 0), stage + 1024 * (XD * wave + u));
            }
#pragma unroll
            for (int u = 0; u < WD; ++u) {
                const int k = k0 + wchunk[u] * 2;
                fdma16(wsrc[u] + (k < d ? k : 0), stage + S_XT + 1024 * (WD * wave + u));
            }
            i_stage = (i_stage == (STAGES - 1) * S_STAGE) ? 0 : i_stage + S_STAGE;
            ++i_kt;
        };
        issue();
        if (nkt > 1) issue();
        int r_stage = 0;
        for (int t = 0; t < nkt; ++t) {
            // the wavefront's own DMAs of tile t have landed (one younger tile may stay in flight), the barrier
            // covers everybody's
            if (t + 1 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XD + WD) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (t + 2 < nkt) issue();  // into the stage tile t - 1 was read from
            const char *stage = smem + r_stage;
            r_stage = (r_stage == (STAGES - 1) * S_STAGE) ? 0 : r_stage + S_STAGE;
            if (live) {
                const int nsub = min(KP, d - t * KP) / KT;  // 16-feature blocks of this tile (d % 16 == 0)
                for (int sb = 0; sb < nsub; ++sb) {
                    double xv[KT], wv[KT];
                    if constexpr (ES == 4) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const v4f_t f = *reinterpret_cast<const v4f_t *>(stage + xo + (((4 * sb + c) ^ xsw) << 4));
                            xv[4 * c] = (double)f.x; xv[4 * c + 1] = (double)f.y; xv[4 * c + 2] = (double)f.z; xv[4 * c + 3] = (double)f.w;
                        }
                    } else if constexpr (ES == 2) {  // bfloat16: the upper half of a float32, exactly
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            const v4i_t f = *reinterpret_cast<const v4i_t *>(stage + xo + (((2 * sb + c) ^ xsw) << 4));
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                xv[8 * c + 2 * e] = (double)__uint_as_float(((uint32_t)f[e]) << 16);
                                xv[8 * c + 2 * e + 1] = (double)__uint_as_float(((uint32_t)f[e]) & 0xffff0000u);
                            }
                        }
                    } else {
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            const d2_t f = *reinterpret_cast<const d2_t *>(stage + xo + (((8 * sb + c) ^ xsw) << 4));
                            xv[2 * c] = f.x; xv[2 * c + 1] = f.y;
                        }
                    }
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const d2_t f = *reinterpret_cast<const d2_t *>(stage + wo + (((8 * sb + c) ^ wsw) << 4));
                        wv[2 * c] = f.x; wv[2 * c + 1] = f.y;
                    }
                    // (the prototype is the first factor, as the A operand of the matrix instruction)
#pragma unroll
                    for (int k = 0; k < KT; ++k) acc = fma(wv[k], xv[k], acc);
                }
            }
        }
    }
    __syncthreads();  // the ring is done with: the pairs' results take its place
    double *pv = reinterpret_cast<double *>(smem);
    if (mine) {
        const int j = rows[psl];
        const double xi = xx[order2[p0 + prow]];
        double rv = (xi + (-2.0 * acc)) + ww[j];
        if (!(rv > 0.0)) rv = (rv != rv) ? rv : 0.0;
        pv[tid] = rv;
    }
    __syncthreads();
    if (tid < PS && n > 0) {
        double bv = INFINITY;
        int bj = 0x7fffffff;
        const int q0 = off_s[tid];
        for (int e = 0; e < n; ++e) {
            const double v = pv[q0 + e];
            const int j = rows[(int)pslot[q0 + e]];
            if (v < bv) { bv = v; bj = j; }  // (Best<1>::push: the pairs of a sample come with ascending index)
        }
        double dv = sqrt(bv);
        if (round_f32) dv = (double)(float)dv;
        idx_out[isamp] = (bj == 0x7fffffff) ? (int64_t)-1 : (int64_t)bj;
        dist_out[isamp] = dv;
    }
}

// ---- samples whose candidates did not fit RF_C slots --------------------------------------------------
// (rare -- duplicated prototypes, a collapsed map, rows the bound says nothing about.)  One 256-thread
// workgroup per sample, a few of them walking the queue.
//   record count m > 0: the sample's m <= 31 candidates (refine_i8_kernel left them): lane r of the first
//     wavefront runs the chain of candidate r; the rows come through LDS in chunks of 128 doubles, fetched by
//     the whole workgroup in 16-byte pieces one chunk ahead (row pitch 1040 bytes: the 16 lanes of a
//     ds_read_b128 group hit 16 different 4-bank sets), the sample's chunk beside them, widened.
//   count 0 (a list in segments, more candidates than a record holds, a record beyond OV_CAP): the workgroup's
//     whole list, thread = list entry, plain loops -- 16 KB of prototype row per entry straight from memory.
constexpr int OV_KC = 128, OV_PITCH = OV_KC * 8 + 16;
template <typename XT>
__global__ __launch_bounds__(256, 2) void overflow_exact_kernel(
    const XT *__restrict__ X, int d, int64_t ldx, const double *__restrict__ xx, const double *__restrict__ W,
    const double *__restrict__ ww, const uint16_t *__restrict__ ulist, int ulist_stride,
    const uint32_t *__restrict__ ucount, const int32_t *__restrict__ ovf, const uint32_t *__restrict__ ovf_len,
    const uint16_t *__restrict__ ovf_cand, int round_f32, int64_t *__restrict__ idx_out,
    double *__restrict__ dist_out) {
    __shared__ __attribute__((aligned(16))) char tile[2][(OV_REC) * OV_PITCH];  // row OV_REC - 1: the sample's chunk
    __shared__ double sv[256];
    __shared__ int sj[256];
    const int tid = threadIdx.x;
    const uint32_t qn = *ovf_len;
    for (uint32_t e = blockIdx.x; e < qn; e += gridDim.x) {
        const int64_t i = ovf[2 * (size_t)e];
        const int group = ovf[2 * (size_t)e + 1];
        const XT *x = X + i * ldx;
        const uint16_t *rec = ovf_cand + (size_t)e * OV_REC;
        const int m = e < (uint32_t)OV_CAP ? (int)rec[0] : 0;
        Best<1> best;
        best.init();
        if (m > 0) {
            // pieces of a chunk: piece q = tid + 256 u -> candidate q / 64, 16 bytes q % 64 of its 1024
            constexpr int NP = (OV_REC - 1) * 64 / 256 + 1;
            const double *src[NP];
            int dst[NP];
            bool live[NP];
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                const int q = tid + 256 * u, r = q >> 6, part = q & 63;
                live[u] = r < m;
                const int j = (int)rec[1 + (live[u] ? r : 0)];
                src[u] = W + (int64_t)j * d + 2 * part;
                dst[u] = r * OV_PITCH + 16 * part;
            }
            d2_t stage[NP];
            double xs = 0.0;
            const int nkc = (d + OV_KC - 1) / OV_KC;
            auto fetch = [&](int kc) {
                const int k0 = kc * OV_KC;
#pragma unroll
                for (int u = 0; u < NP; ++u)
                    if (live[u]) {
                        const int k = k0 + 2 * ((tid + 256 * u) & 63);   // (d % 16 == 0: a piece is inside the row or behind it)
                        stage[u] = k < d ? *reinterpret_cast<const d2_t *>(src[u] + k0) : d2_t{0.0, 0.0};
                    }
                if (tid < OV_KC) xs = k0 + tid < d ? widen(x[k0 + tid]) : 0.0;
            };
            auto put = [&](int buf) {
#pragma unroll
                for (int u = 0; u < NP; ++u)
                    if (live[u]) *reinterpret_cast<d2_t *>(tile[buf] + dst[u]) = stage[u];
                if (tid < OV_KC) *reinterpret_cast<double *>(tile[buf] + (OV_REC - 1) * OV_PITCH + 8 * tid) = xs;
            };
            const bool mine = tid < m;
            double acc = 0.0;
            fetch(0);
            put(0);
            __syncthreads();
            for (int kc = 0; kc < nkc; ++kc) {
                if (kc + 1 < nkc) fetch(kc + 1);
                if (mine) {  // (zeros behind column d: fma(0, 0, acc) == acc)
                    const char *row = tile[kc & 1] + tid * OV_PITCH;
                    const char *xr = tile[kc & 1] + (OV_REC - 1) * OV_PITCH;
#pragma unroll 8
                    for (int u = 0; u < OV_KC / 2; ++u) {
                        const d2_t wv = *reinterpret_cast<const d2_t *>(row + 16 * u);
                        const d2_t xv = *reinterpret_cast<const d2_t *>(xr + 16 * u);
                        acc = fma(wv[0], xv[0], acc);
                        acc = fma(wv[1], xv[1], acc);
                    }
                }
                if (kc + 1 < nkc) put((kc + 1) & 1);
                __syncthreads();
            }
            if (mine) {
                const int j = (int)rec[1 + tid];
                double rv = (xx[i] + (-2.0 * acc)) + ww[j];
                if (!(rv > 0.0)) rv = (rv != rv) ? rv : 0.0;
                best.push(rv, j);
            }
        } else {
            const int cnt = (int)ucount[group];
            const uint16_t *list = ulist + (size_t)group * ulist_stride;
            for (int l = tid; l < cnt; l += 256) {  // (ascending per thread)
                const int j = (int)list[l];
                const double *w = W + (int64_t)j * d;
                double acc = 0.0;
                for (int k = 0; k < d; k += 16) {  // (d % 16 == 0; the loads of a block first)
                    double wv[16], xv[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) { wv[u] = w[k + u]; xv[u] = widen(x[k + u]); }
#pragma unroll
                    for (int u = 0; u < 16; ++u) acc = fma(wv[u], xv[u], acc);
                }
                double rv = (xx[i] + (-2.0 * acc)) + ww[j];
                if (!(rv > 0.0)) rv = (rv != rv) ? rv : 0.0;
                best.push(rv, j);
            }
        }
        sv[tid] = best.v[0];
        sj[tid] = best.j[0];
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if (tid < w && lex_lt(sv[tid + w], sj[tid + w], sv[tid], sj[tid])) { sv[tid] = sv[tid + w]; sj[tid] = sj[tid + w]; }
            __syncthreads();
        }
        if (tid == 0) {
            double dv = sqrt(sv[0]);
            if (round_f32) dv = (double)(float)dv;
            idx_out[i] = (sj[0] == 0x7fffffff) ? (int64_t)-1 : (int64_t)sj[0];
            dist_out[i] = dv;
        }
        __syncthreads();
    }
}
