// GPU probe (diagnostic): issue rate of v_mfma_f64_16x16x4_f64 against the number of independent
// accumulators per wave and waves per SIMD.
//   hipcc --offload-arch=gfx950 -O2 tools/probe_mfma_f64_lat.hip -o /tmp/probe_lat && /tmp/probe_lat
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4_t __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void rate(double *out, int iters) {
    d4_t acc[NACC];
    for (int t = 0; t < NACC; ++t) acc[t] = d4_t{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
    }
    double s = 0;
    for (int t = 0; t < NACC; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(double *dout, int waves_per_simd) {
    const int iters = 200000 / NACC, blocks = 256 * waves_per_simd;  // 4 waves per block = 1 per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate<NACC>, dim3(blocks), dim3(256), 0, 0, dout, 10);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(rate<NACC>, dim3(blocks), dim3(256), 0, 0, dout, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * 4 * iters * NACC;
    printf("acc %d  waves/SIMD %d : %.1f TFLOP/s   %.1f ns per MFMA per SIMD\n", NACC, waves_per_simd,
           n * 2048.0 / (ms * 1e-3) / 1e12, ms * 1e6 / (n / 1024.0));
}

int main() {
    double *dout; hipMalloc(&dout, 256 * 8 * 256 * 8);
    for (int w = 1; w <= 4; ++w) {
        run<1>(dout, w); run<2>(dout, w); run<4>(dout, w); run<6>(dout, w); run<8>(dout, w);
    }
    return 0;
}
