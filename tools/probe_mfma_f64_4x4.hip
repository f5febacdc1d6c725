// GPU probe (diagnostic tool, not part of the library): lane layout and rounding of v_mfma_f64_4x4x4_4b_f64
// (4 blocks of 4x4x4 per instruction, 16 cycles: a quarter of the 16x16x4 form's pipe time for 16 B-columns),
// and its issue rate.  Is it, like the 16x16x4 form, the sequential fma chain over k?
//   hipcc --offload-arch=gfx950 -O2 tools/probe_mfma_f64_4x4.hip -o /tmp/probe44 && /tmp/probe44
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

__global__ void one(const double *A, const double *B, const double *C, double *D) {
    const int l = threadIdx.x;
    D[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], C[l], 0, 0, 0);
}
__global__ void rate(double *out, int iters) {
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[t], 0, 0, 0);
    }
    double s = 0;
    for (int t = 0; t < 8; ++t) s += acc[t];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void chain_latency(double *out, int iters) {   // ONE dependent chain per wave
    double acc = 0, a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc, 0, 0, 0);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
typedef double d4_t __attribute__((ext_vector_type(4)));
__global__ void chain_latency16(double *out, int iters) {
    d4_t acc = {0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

static const int PERM[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
// lane of (x0, x1, x2) when role r sits in base-4 digit PERM[p][r]
static int lane_of(int p, int x0, int x1, int x2) {
    int dig[3]; dig[PERM[p][0]] = x0; dig[PERM[p][1]] = x1; dig[PERM[p][2]] = x2;
    return dig[0] + 4 * dig[1] + 16 * dig[2];
}

int main() {
    double hA[64], hB[64], hC[64], hD[64];
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dC, 512); hipMalloc(&dD, 512);
    srand(7);
    const int TR = 60;
    static double sA[TR][64], sB[TR][64], sC[TR][64], sD[TR][64];
    for (int t = 0; t < TR; ++t) {
        for (int e = 0; e < 64; ++e) { hA[e] = (rand() / (double)RAND_MAX - 0.5) * 7.3; hB[e] = (rand() / (double)RAND_MAX - 0.5) * 3.1; hC[e] = (rand() / (double)RAND_MAX - 0.5) * 11.0; }
        hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice); hipMemcpy(dC, hC, 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(one, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(hD, dD, 512, hipMemcpyDeviceToHost);
        memcpy(sA[t], hA, 512); memcpy(sB[t], hB, 512); memcpy(sC[t], hC, 512); memcpy(sD[t], hD, 512);
    }
    // roles: A (block, i, k), B (block, k, j), D (block, i, j): which digit permutation each?
    int found = 0;
    for (int pa = 0; pa < 6; ++pa) for (int pb = 0; pb < 6; ++pb) for (int pd = 0; pd < 6; ++pd) {
        long bad[4] = {0, 0, 0, 0};
        for (int t = 0; t < TR; ++t)
            for (int b = 0; b < 4; ++b) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
                const int ld = lane_of(pd, b, i, j);
                const double c = sC[t][ld], got = sD[t][ld];
                double ch = c, rv = c, nf = c;
                double pr[4];
                for (int k = 0; k < 4; ++k) pr[k] = sA[t][lane_of(pa, b, i, k)] * sB[t][lane_of(pb, b, k, j)];
                for (int k = 0; k < 4; ++k) ch = fma(sA[t][lane_of(pa, b, i, k)], sB[t][lane_of(pb, b, k, j)], ch);
                for (int k = 3; k >= 0; --k) rv = fma(sA[t][lane_of(pa, b, i, k)], sB[t][lane_of(pb, b, k, j)], rv);
                for (int k = 0; k < 4; ++k) { volatile double p = pr[k]; nf = nf + p; }
                double pw = c + (fma(sA[t][lane_of(pa, b, i, 1)], sB[t][lane_of(pb, b, 1, j)], pr[0]) +
                                 fma(sA[t][lane_of(pa, b, i, 3)], sB[t][lane_of(pb, b, 3, j)], pr[2]));
                bad[0] += memcmp(&got, &ch, 8) != 0; bad[1] += memcmp(&got, &rv, 8) != 0;
                bad[2] += memcmp(&got, &nf, 8) != 0; bad[3] += memcmp(&got, &pw, 8) != 0;
            }
        // a layout "fits" when some model is nearly right (all four models agree to ~1e-15 relative anyway):
        // accept if any model has zero mismatches
        for (int m = 0; m < 4; ++m)
            if (bad[m] == 0) {
                printf("layout A digits(block,i,k)=(%d,%d,%d) B digits(block,k,j)=(%d,%d,%d) D digits(block,i,j)=(%d,%d,%d): model %s fits all %d outputs\n",
                       PERM[pa][0], PERM[pa][1], PERM[pa][2], PERM[pb][0], PERM[pb][1], PERM[pb][2], PERM[pd][0], PERM[pd][1], PERM[pd][2],
                       m == 0 ? "fma-chain k0..3" : (m == 1 ? "reversed chain" : (m == 2 ? "unfused" : "pairwise")), TR * 64);
                ++found;
            }
    }
    if (!found) printf("no (digit layout, rounding model) reproduces the instruction bit for bit\n");
    const int iters = 400000, blocks = 512;
    double *dout; hipMalloc(&dout, blocks * 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    hipLaunchKernelGGL(rate, dim3(blocks), dim3(256), 0, 0, dout, 100);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(rate, dim3(blocks), dim3(256), 0, 0, dout, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("4x4x4 rate, 8 accumulators x 4 waves x %d blocks: %.3f ms -> %.1f cycles per instruction per SIMD at 2.4 GHz, %.1f TFLOP/s\n", blocks, ms,
           ms * 1e-3 * 2.4e9 / ((double)iters * 8 * blocks / 256.0), (double)blocks * 4 * iters * 8 * 512.0 / (ms * 1e-3) / 1e12);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(chain_latency, dim3(256), dim3(64), 0, 0, dout, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("4x4x4 dependent chain, one wave per CU: %.1f cycles per instruction at 2.4 GHz\n", ms * 1e-3 * 2.4e9 / iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(chain_latency16, dim3(256), dim3(64), 0, 0, dout, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("16x16x4 dependent chain, one wave per CU: %.1f cycles per instruction at 2.4 GHz\n", ms * 1e-3 * 2.4e9 / iters);
    return 0;
}
