// GPU probe (diagnostic tool): what does one step of a DEPENDENT v_mfma_f64_4x4x4 chain cost when the B operand is
// produced between the matrix instructions (as segsum_chain_kernel does: LDS read -> v_cvt_f64_f32 -> mfma)?
//   hipcc --offload-arch=gfx950 -O2 tools/probe_mfma_chain.hip -o /tmp/pc && /tmp/pc
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ void chain(double *out, const float *src, int iters) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = src[i];
    __syncthreads();
    double acc = 0, a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    float f = threadIdx.x * 0.5f;
    const float *p = lds + threadIdx.x;
    for (int it = 0; it < iters; it += 4) {
        if (MODE == 0) {          // invariant operands
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc, 0, 0, 0);
        } else if (MODE == 1) {   // an independent VALU conversion between the matrix instructions
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                double t;
                asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(t) : "v"(f));
                acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc, 0, 0, 0);
                asm volatile("" :: "v"(t));
            }
        } else if (MODE == 2) {   // B = that conversion's result (converted right in front of its instruction)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                double t;
                asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(t) : "v"(f));
                acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, t, acc, 0, 0, 0);
            }
        } else if (MODE == 3) {   // four conversions first, then four matrix instructions
            double t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(t[u]) : "v"(f));
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, t[u], acc, 0, 0, 0);
        } else if (MODE == 4) {   // B from LDS (read four steps ahead), converted in front of its instruction
            float r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) r[u] = p[(64 * (it + u)) & 4095];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, (double)r[u], acc, 0, 0, 0);
        } else if (MODE == 5) {   // A changes too (invariant register pair per step, no producer)
            double a2 = a + 1.0;
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_4x4x4f64((u & 1) ? a : a2, b, acc, 0, 0, 0);
        } else if (MODE == 6) {   // the 16x16x4 form with B converted in front (subset_exact_kernel's pattern)
            typedef double d4_t __attribute__((ext_vector_type(4)));
            static d4_t a16;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MODE>
static void run(const char *what, double *dout, const float *dsrc, int waves) {
    const int iters = 200000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    hipLaunchKernelGGL(chain<MODE>, dim3(256), dim3(64 * waves), 0, 0, dout, dsrc, 400);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(chain<MODE>, dim3(256), dim3(64 * waves), 0, 0, dout, dsrc, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-72s %d wave(s) per CU: %6.1f ns per step = %5.1f cycles at 2.4 GHz\n", what, waves, ms * 1e6 / iters, ms * 1e-3 * 2.4e9 / iters);
}

int main() {
    double *dout; float *dsrc;
    hipMalloc(&dout, 256 * 1024 * 8); hipMalloc(&dsrc, 4096 * 4); hipMemset(dsrc, 0, 4096 * 4);
    for (int waves = 1; waves <= 4; waves += 3) {
        run<0>("dependent 4x4x4 chain, invariant operands", dout, dsrc, waves);
        run<1>("... an independent v_cvt_f64_f32 between the instructions", dout, dsrc, waves);
        run<2>("... B = v_cvt_f64_f32 issued right in front of its instruction", dout, dsrc, waves);
        run<3>("... four conversions, then four instructions", dout, dsrc, waves);
        run<4>("... B read from LDS four steps ahead, converted in front", dout, dsrc, waves);
        run<5>("... A alternates between two registers", dout, dsrc, waves);
    }
    return 0;
}
