// Microbenchmark 2: one lane per node, NL dwordx4 loads from the same random 128 B-aligned record (pointer chasing).
// If the time per step grows with NL the fetch is bound by lane requests (texture-addresser rate); if it is flat it is
// bound by distinct cache lines / latency.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int NL>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ nodes, uint32_t mask, int steps, uint32_t* out) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t cur = (tid * 2654435761u) & mask;
    float acc = 0.f;
    for (int s = 0; s < steps; s++) {
        const float4* np = nodes + 8 * (size_t)cur;
        float4 q[NL];
#pragma unroll
        for (int j = 0; j < NL; j++) q[j] = np[j];
#pragma unroll
        for (int j = 0; j < NL; j++) asm volatile("" : "+v"(q[j].x), "+v"(q[j].y), "+v"(q[j].z), "+v"(q[j].w));
        uint32_t nxt = 0;
#pragma unroll
        for (int j = 0; j < NL; j++) { acc += q[j].y; nxt ^= __float_as_uint(q[j].x); }
        cur = ((nxt ^ (s * 40503u)) + tid * 2246822519u) * 2654435761u >> 7 & mask;  // per-thread stream: walks must not coalesce
    }
    if (acc == 123.456f) out[0] = 1;
    out[1 + (tid & 1023)] = cur;
}

int main(int argc, char** argv) {
    int log2n = argc > 1 ? atoi(argv[1]) : 17;  // 2^17 records * 128 B = 16 MiB
    uint32_t n = 1u << log2n;
    std::vector<float> h((size_t)n * 32);
    uint32_t x = 12345;
    for (size_t i = 0; i < h.size(); i++) { x = x * 1664525u + 1013904223u; uint32_t v = x >> 4; memcpy(&h[i], &v, 4); }
    float4* d; uint32_t* out;
    CK(hipMalloc(&d, (size_t)n * 128)); CK(hipMalloc(&out, 8192));
    CK(hipMemcpy(d, h.data(), (size_t)n * 128, hipMemcpyHostToDevice));
    const int steps = 64, blocks = 256 * 24;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int nl : {1, 2, 4, 5, 6, 8}) {
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            for (int it = 0; it < 5; it++) {
                if (nl == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, n - 1, steps, out);
                if (nl == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, n - 1, steps, out);
                if (nl == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, d, n - 1, steps, out);
                if (nl == 5) hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(256), 0, 0, d, n - 1, steps, out);
                if (nl == 6) hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(256), 0, 0, d, n - 1, steps, out);
                if (nl == 8) hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(256), 0, 0, d, n - 1, steps, out);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            double visits = 5.0 * blocks * 256.0 * steps;
            if (rep) printf("table %4u MiB  loads/visit %d (%3d B): %7.2f Gvisit/s  %6.2f TB/s requested  (%.3f ms)\n", (unsigned)((size_t)n * 128 >> 20), nl, nl * 16, visits / ms / 1e6, visits * nl * 16 / ms / 1e9, ms / 5);
        }
    }
    return 0;
}
