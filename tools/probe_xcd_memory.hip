// Are the eight XCDs alike for memory?  Two kernels whose workgroups the hardware deals round robin to the XCDs:
//   stream : a workgroup reads 128 KB of a 3 GB buffer in 16-byte pieces per lane (non-temporal), like the sums kernel
//   chase  : a workgroup makes 12 dependent random 64-byte reads, like the pruning pass
// Every workgroup stamps s_memrealtime at its start and end and its XCC_ID; printed per XCD: workgroups, mean life,
// when its last workgroup ended.
//   hipcc -O3 --offload-arch=gfx950 tools/probe_xcd_memory.hip -o /tmp/probe_xcd && /tmp/probe_xcd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float f4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void stream_kernel(const f4_t *__restrict__ buf, size_t per_wg, float *__restrict__ out,
                                                     unsigned long long *__restrict__ stamps) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const f4_t *p = buf + (size_t)blockIdx.x * per_wg;
    f4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = threadIdx.x; i < per_wg; i += 256 * 4) {
        f4_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (i + 256 * u < per_wg) ? __builtin_nontemporal_load(p + i + 256 * u) : acc;
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) out[blockIdx.x] = acc[0];
    __syncthreads();
    if (threadIdx.x == 0) {
        stamps[3 * blockIdx.x] = t0;
        stamps[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        stamps[3 * blockIdx.x + 2] = __builtin_amdgcn_s_getreg((3 << 11) | 20);
    }
}

__global__ __launch_bounds__(256) void chase_kernel(const unsigned *__restrict__ next, unsigned n, int hops,
                                                    unsigned *__restrict__ out, unsigned long long *__restrict__ stamps) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned j = (blockIdx.x * 256u + threadIdx.x) * 2654435761u % n;
    for (int h = 0; h < hops; ++h) j = next[(size_t)j * 16];   // (one 64-byte line per hop)
    if (j == 0xffffffffu) out[blockIdx.x] = j;
    __syncthreads();
    if (threadIdx.x == 0) {
        stamps[3 * blockIdx.x] = t0;
        stamps[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        stamps[3 * blockIdx.x + 2] = __builtin_amdgcn_s_getreg((3 << 11) | 20);
    }
}

static void report(const char *name, const std::vector<unsigned long long> &st, int nwg) {
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int b = 0; b < nwg; ++b) { t0 = std::min(t0, st[3 * b]); t1 = std::max(t1, st[3 * b + 1]); }
    printf("%s: %d workgroups, span %.1f us\n", name, nwg, (t1 - t0) / 100.0);
    for (int x = 0; x < 8; ++x) {
        double life = 0; int n = 0; unsigned long long last = 0, last_start = 0;
        for (int b = 0; b < nwg; ++b)
            if ((int)(st[3 * b + 2] & 15) == x) {
                life += (st[3 * b + 1] - st[3 * b]) / 100.0; ++n;
                last = std::max(last, st[3 * b + 1]); last_start = std::max(last_start, st[3 * b]);
            }
        if (n) printf("  XCD %d: %5d workgroups, life %7.2f us, last start %7.1f, last end %7.1f\n", x, n, life / n,
                      (last_start - t0) / 100.0, (last - t0) / 100.0);
    }
}

int main() {
    const int nwg = 24576;
    const size_t per_wg = 128 * 1024 / 16;   // 16-byte pieces per workgroup
    f4_t *buf; float *out; unsigned long long *stamps; unsigned *next, *out2;
    CHECK(hipMalloc(&buf, (size_t)nwg * per_wg * 16));
    CHECK(hipMemset(buf, 0, (size_t)nwg * per_wg * 16));
    CHECK(hipMalloc(&out, nwg * 4)); CHECK(hipMalloc(&out2, nwg * 4));
    CHECK(hipMalloc(&stamps, (size_t)nwg * 24));
    const unsigned n = 1u << 24;             // 2^24 lines of 64 bytes: 1 GB
    CHECK(hipMalloc(&next, (size_t)n * 64));
    {
        std::vector<unsigned> h((size_t)n * 16, 0u);
        unsigned long long s = 88172645463325252ull;
        for (unsigned i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[(size_t)i * 16] = (unsigned)(s % n); }
        CHECK(hipMemcpy(next, h.data(), (size_t)n * 64, hipMemcpyHostToDevice));
    }
    std::vector<unsigned long long> st((size_t)nwg * 3);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(stream_kernel, dim3(nwg), dim3(256), 0, 0, buf, per_wg, out, stamps);
        CHECK(hipDeviceSynchronize());
    }
    CHECK(hipMemcpy(st.data(), stamps, (size_t)nwg * 24, hipMemcpyDeviceToHost));
    report("stream (128 KB per workgroup, 3 GB)", st, nwg);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(chase_kernel, dim3(8192), dim3(256), 0, 0, next, n, 12, out2, stamps);
        CHECK(hipDeviceSynchronize());
    }
    CHECK(hipMemcpy(st.data(), stamps, (size_t)8192 * 24, hipMemcpyDeviceToHost));
    report("chase (12 dependent random 64-byte reads in 1 GB)", st, 8192);
    return 0;
}
