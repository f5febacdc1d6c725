// GPU probe (diagnostic tool, not part of the library): what does v_mfma_f64_16x16x4_f64 compute
// bit for bit, and how fast does it issue?  Build+run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 tools/probe_mfma_f64.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef double d4_t __attribute__((ext_vector_type(4)));

__global__ void one_mfma(const double *A, const double *B, const double *C, double *D) {
    // A: 16x4 (row i, k), B: 4x16 (k, col j), C/D: 16x16
    const int l = threadIdx.x;
    const double a = A[(l & 15) * 4 + (l >> 4)];
    const double b = B[(l >> 4) * 16 + (l & 15)];
    d4_t c;
    for (int r = 0; r < 4; ++r) c[r] = C[((l >> 4) + 4 * r) * 16 + (l & 15)];
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}

__global__ void rate(double *out, int iters) {
    d4_t acc[8];
    for (int t = 0; t < 8; ++t) acc[t] = d4_t{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
    }
    double s = 0;
    for (int t = 0; t < 8; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    double hA[64], hB[64], hC[256], hD[256];
    srand(1);
    int bad_chain = 0, bad_rev = 0, bad_pair = 0, bad_nofma = 0;
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC); hipMalloc(&dD, sizeof hD);
    for (int trial = 0; trial < 200; ++trial) {
        for (int e = 0; e < 64; ++e) { hA[e] = (rand() / (double)RAND_MAX - 0.5) * 7.3; hB[e] = (rand() / (double)RAND_MAX - 0.5) * 3.1; }
        for (int e = 0; e < 256; ++e) hC[e] = (rand() / (double)RAND_MAX - 0.5) * 11.0;
        hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
        hipMemcpy(dC, hC, sizeof hC, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(one_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            double c = hC[i * 16 + j];
            double ch = c, rv = c, nf = c;
            for (int k = 0; k < 4; ++k) ch = fma(hA[i * 4 + k], hB[k * 16 + j], ch);
            for (int k = 3; k >= 0; --k) rv = fma(hA[i * 4 + k], hB[k * 16 + j], rv);
            for (int k = 0; k < 4; ++k) { volatile double p = hA[i * 4 + k] * hB[k * 16 + j]; nf = nf + p; }
            double p01 = fma(hA[i * 4 + 1], hB[16 + j], hA[i * 4] * hB[j]);
            double p23 = fma(hA[i * 4 + 3], hB[48 + j], hA[i * 4 + 2] * hB[32 + j]);
            double pr = c + (p01 + p23);
            double got = hD[i * 16 + j];
            bad_chain += memcmp(&got, &ch, 8) != 0; bad_rev += memcmp(&got, &rv, 8) != 0;
            bad_pair += memcmp(&got, &pr, 8) != 0; bad_nofma += memcmp(&got, &nf, 8) != 0;
        }
    }
    printf("mfma_f64_16x16x4 vs host models over 200x256 outputs: mismatches  fma-chain k0..3: %d  reversed: %d  pairwise: %d  unfused: %d\n",
           bad_chain, bad_rev, bad_pair, bad_nofma);
    // issue rate: 8 independent accumulators per wave, 4 waves per block (1 per SIMD), 256*2 blocks
    const int iters = 400000, blocks = 512;
    double *dout; hipMalloc(&dout, blocks * 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate, dim3(blocks), dim3(256), 0, 0, dout, 100);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(rate, dim3(blocks), dim3(256), 0, 0, dout, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * iters * 8 * 2048.0;
    printf("f64 MFMA rate: %.1f TFLOP/s (%.3f ms)\n", flops / (ms * 1e-3) / 1e12, ms);
    return 0;
}
