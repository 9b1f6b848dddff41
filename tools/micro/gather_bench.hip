// Microbenchmark: random 64-byte "node" fetches from a table, as 1 / 2 / 4 lanes per node.
//   mode 1: one lane loads the whole 64 B node (4 x dwordx4)          -- k_extend today
//   mode 2: two lanes share a node, each loads 32 B (2 x dwordx4)
//   mode 4: four lanes share a node, each loads 16 B (1 x dwordx4)
// Each "step" the next node index depends on the loaded data (pointer chasing, like traversal).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int LANES>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ nodes, uint32_t mask, int steps, uint32_t* out) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t ray = tid / LANES, sub = tid % LANES;
    uint32_t cur = (ray * 2654435761u) & mask;
    float acc = 0.f;
    for (int s = 0; s < steps; s++) {
        const float4* np = nodes + 4 * (size_t)cur;
        uint32_t nxt;
        if (LANES == 1) {
            float4 a = np[0], b = np[1], c = np[2], d = np[3];
            acc += a.x + b.y + c.z;
            nxt = __float_as_uint(d.x);
        } else if (LANES == 2) {
            float4 a = np[2 * sub], b = np[2 * sub + 1];
            acc += a.x + b.y;
            uint32_t mine = __float_as_uint(b.x);
            nxt = __shfl(mine, (threadIdx.x & 63 & ~1u) + 1);  // the ref lives in the second half
        } else {
            float4 a = np[sub];
            acc += a.x;
            uint32_t mine = __float_as_uint(a.x);
            nxt = __shfl(mine, (threadIdx.x & 63 & ~3u) + 3);
        }
        cur = ((nxt ^ (s * 40503u)) + tid * 2246822519u) * 2654435761u >> 7 & mask;  // per-thread stream: walks must not coalesce
    }
    if (acc == 123.456f) out[0] = 1;
    out[1 + (tid & 1023)] = cur;
}

int main(int argc, char** argv) {
    int log2n = argc > 1 ? atoi(argv[1]) : 18;  // 2^18 nodes * 64 B = 16 MiB
    uint32_t n = 1u << log2n;
    std::vector<float> h((size_t)n * 16);
    uint32_t x = 12345;
    for (size_t i = 0; i < h.size(); i++) { x = x * 1664525u + 1013904223u; uint32_t v = x >> 4; memcpy(&h[i], &v, 4); }
    float4* d; uint32_t* out;
    CK(hipMalloc(&d, (size_t)n * 64)); CK(hipMalloc(&out, 8192));
    CK(hipMemcpy(d, h.data(), (size_t)n * 64, hipMemcpyHostToDevice));
    const int steps = 64, blocks = 256 * 24;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode : {1, 2, 4}) {
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            for (int it = 0; it < 5; it++) {
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, n - 1, steps, out);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, n - 1, steps, out);
                if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, d, n - 1, steps, out);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            double nodes = 5.0 * blocks * 256.0 / mode * steps;
            if (rep) printf("table %4u MiB  lanes/node %d : %7.2f Gnode/s  %7.2f TB/s of node bytes  (%.3f ms)\n", (unsigned)((size_t)n * 64 >> 20), mode, nodes / ms / 1e6, nodes * 64 / ms / 1e9, ms / 5);
        }
    }
    return 0;
}
