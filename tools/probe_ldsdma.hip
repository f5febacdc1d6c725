// GPU probe (diagnostic tool, not part of the library): how many bytes per clock does a CU take in through
// the LDS-DMA path (global_load_lds_dwordx4, 1 KiB per wave-instruction) from an L2-resident buffer, alone
// and beside int8 MFMAs issued by the same waves?  And through plain 16-byte loads into registers?
//   hipcc --offload-arch=gfx950 -O2 tools/probe_ldsdma.hip -o /tmp/probe_ldsdma && /tmp/probe_ldsdma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v16i_t __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *gbl_ptr_t;

__device__ __forceinline__ void dma16(const void *src, void *dst) {
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)dst, 16, 0, 0);
}

// MODE 0: LDS-DMA only; 1: LDS-DMA + NM MFMAs per group of PIECES DMAs; 2: MFMAs only; 3: register loads only;
// 4: register loads + MFMAs
template <int MODE, int NW, int PIECES, int NM, int LDS_KB>
__global__ __launch_bounds__(NW * 64) void intake(const char *__restrict__ src, size_t span, int iters, int *out) {
    __shared__ __attribute__((aligned(16))) char smem[LDS_KB * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v16i_t acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0;
    v4i_t a = {lane, 1, 2, 3}, b = {3, 2, 1, lane};
    v4i_t sink = {0, 0, 0, 0};
    // every wave walks its own KiB pieces of the span; consecutive waves / workgroups take consecutive pieces
    size_t base = (((size_t)blockIdx.x * NW + wave) * PIECES * 1024) & (span - 1);   // span: a power of two (+ 8 KiB slack)
    const size_t step = (size_t)gridDim.x * NW * PIECES * 1024;
    char *dst = smem + wave * PIECES * 1024;   // (every group overwrites the wave's own LDS pieces: nothing reads them)
    for (int it = 0; it < iters; ++it) {
        const size_t off = base + 16 * lane;
        if constexpr (MODE == 0 || MODE == 1) {
#pragma unroll
            for (int p = 0; p < PIECES; ++p) dma16(src + off + p * 1024, dst + p * 1024);
        }
        if constexpr (MODE == 3 || MODE == 4) {
#pragma unroll
            for (int p = 0; p < PIECES; ++p) {
                v4i_t v = *reinterpret_cast<const v4i_t *>(src + off + p * 1024);
                asm volatile("" : "+v"(v));
                sink ^= v;
            }
        }
        if constexpr (MODE == 1 || MODE == 2 || MODE == 4) {
#pragma unroll
            for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[m & 3], 0, 0, 0);
        }
        if constexpr (MODE == 0 || MODE == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");  // one group stays in flight
        base = (base + step) & (span - 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int s = sink[0] ^ sink[1] ^ sink[2] ^ sink[3];
#pragma unroll
    for (int t = 0; t < 4; ++t) s += acc[t][0] + acc[t][5];
    if (s == 0x7fffffff) out[0] = s + smem[lane];
}

template <int MODE, int NW, int PIECES, int NM, int LDS_KB>
static void run(const char *name, const char *src, size_t span, int *out, int wgs_per_cu) {
    const int iters = 4000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((intake<MODE, NW, PIECES, NM, LDS_KB>), dim3(grid), dim3(NW * 64), 0, 0, src, span, 200, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((intake<MODE, NW, PIECES, NM, LDS_KB>), dim3(grid), dim3(NW * 64), 0, 0, src, span, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (MODE == 2) ? 0.0 : (double)grid * NW * PIECES * 1024.0 * iters;
    const double mfma = (MODE == 1 || MODE == 2 || MODE == 4) ? (double)grid * NW * NM * iters : 0.0;
    const double cyc = ms * 1e-3 * 2.4e9;  // at 2.4 GHz
    printf("%-58s %8.3f ms  %7.1f B/clk/CU (%6.2f TB/s chip)  MFMA busy %5.1f %% of 4 SIMDs\n", name, ms,
           bytes / 256.0 / cyc, bytes / (ms * 1e-3) / 1e12, 100.0 * mfma * 32.0 / (256.0 * 4.0 * cyc));
}

int main() {
    const size_t span = 1 << 20;  // 1 MiB: stays in every XCD's L2
    char *src; int *out;
    hipMalloc(&src, span + 8192); hipMemset(src, 1, span + 8192); hipMalloc(&out, 64);
    char *big; const size_t bigspan = (size_t)2 << 30;
    hipMalloc(&big, bigspan + 8192); hipMemset(big, 1, bigspan + 8192);
    run<0, 8, 6, 0, 96>("LDS-DMA only, 8 waves, 6 KiB/wave/group, 1 WG/CU", src, span, out, 1);
    run<0, 8, 3, 0, 48>("LDS-DMA only, 8 waves, 3 KiB/wave/group, 2 WG/CU", src, span, out, 2);
    run<0, 4, 6, 0, 48>("LDS-DMA only, 4 waves, 6 KiB/wave/group, 2 WG/CU", src, span, out, 2);
    run<0, 4, 6, 0, 96>("LDS-DMA only, 4 waves, 6 KiB/wave/group, 1 WG/CU", src, span, out, 1);
    run<0, 16, 3, 0, 96>("LDS-DMA only, 16 waves, 3 KiB/wave/group, 1 WG/CU", src, span, out, 1);
    run<0, 8, 6, 0, 96>("LDS-DMA only, 8 waves, from a 2 GiB span (HBM)", big, bigspan, out, 1);
    run<2, 8, 6, 24, 96>("MFMA only, 8 waves x 24 per group", src, span, out, 1);
    run<1, 8, 6, 24, 96>("LDS-DMA 6 KiB + 24 MFMA per wave-group, 8 waves (the sweep)", src, span, out, 1);
    run<1, 8, 3, 24, 96>("LDS-DMA 3 KiB + 24 MFMA per wave-group, 8 waves", src, span, out, 1);
    run<1, 8, 2, 24, 96>("LDS-DMA 2 KiB + 24 MFMA per wave-group, 8 waves", src, span, out, 1);
    run<1, 8, 3, 8, 48>("LDS-DMA 3 KiB + 8 MFMA per wave-group, 8 waves, 2 WG/CU", src, span, out, 2);
    run<3, 8, 6, 0, 1>("register loads only, 8 waves, 6 x 16 B/lane per group", src, span, out, 1);
    run<3, 16, 6, 0, 1>("register loads only, 16 waves", src, span, out, 1);
    run<4, 8, 6, 24, 1>("register loads 6 KiB + 24 MFMA per wave-group, 8 waves", src, span, out, 1);
    run<4, 8, 3, 24, 1>("register loads 3 KiB + 24 MFMA per wave-group, 8 waves", src, span, out, 1);
    return 0;
}
