// GPU probe (diagnostic): operand / result lane maps of v_mfma_i32_32x32x32_i8 on gfx950 and its
// issue rate.  hipcc --offload-arch=gfx950 -O2 tools/probe_mfma_i8.hip -o /tmp/probe_i8 && /tmp/probe_i8
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// assumed maps: A[i = l&31][k = 16*(l>>5) + b], B[k = 16*(l>>5) + b][j = l&31], b = byte 0..15 of the
// 4 dwords; D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]
__global__ void one(const int8_t *A, const int8_t *B, int *D) {
    const int l = threadIdx.x;
    v4i a, b;
    for (int w = 0; w < 4; ++w) {
        unsigned pa = 0, pb = 0;
        for (int e = 0; e < 4; ++e) {
            const int k = 16 * (l >> 5) + 4 * w + e;
            pa |= (unsigned)(uint8_t)A[(l & 31) * 32 + k] << (8 * e);
            pb |= (unsigned)(uint8_t)B[k * 32 + (l & 31)] << (8 * e);
        }
        a[w] = (int)pa; b[w] = (int)pb;
    }
    v16i c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}

__global__ void rate(int *out, int iters) {
    v16i acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0;
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {3, 2, 1, (int)threadIdx.x};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[t], 0, 0, 0);
    int s = 0;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    int8_t hA[1024], hB[1024]; int hD[1024];
    srand(3);
    for (int e = 0; e < 1024; ++e) { hA[e] = (int8_t)(rand() % 129 - 64); hB[e] = (int8_t)(rand() % 129 - 64); }
    int8_t *dA, *dB; int *dD;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 4096);
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(one, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        int s = 0;
        for (int k = 0; k < 32; ++k) s += (int)hA[i * 32 + k] * (int)hB[k * 32 + j];
        bad += (s != hD[i * 32 + j]);
    }
    printf("mfma_i32_32x32x32_i8 with the assumed lane maps: %d of 1024 outputs differ\n", bad);
    const int iters = 200000, blocks = 512;
    int *dout; hipMalloc(&dout, blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate, dim3(blocks), dim3(256), 0, 0, dout, 100);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(rate, dim3(blocks), dim3(256), 0, 0, dout, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("i8 MFMA rate: %.1f TOP/s (%.2f ms)\n", (double)blocks * 4 * iters * 4 * 65536.0 / (ms * 1e-3) / 1e12, ms);
    return 0;
}
