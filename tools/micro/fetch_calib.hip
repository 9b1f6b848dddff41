// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access shapes of the path tracer's kernels ("what's weak" 6).
// MI355X_MICROARCH.md establishes FETCH_SIZE = 1/2 of the bytes for WIDE COALESCED 16 B/lane streaming reads and calls every
// other shape uncalibrated.  k_extend / k_shade gather 64-byte records (4 x dwordx4 per lane at a per-lane address) and 16-byte
// texels.  Each kernel below reads a KNOWN number of bytes, every record exactly once (a bijective scramble of the index), from
// a table far beyond L2 (and, for the large size, beyond the 256 MiB Infinity Cache):
//   calib_stream16   lane i reads float4 i                      (the calibrated case: expect FETCH_SIZE = bytes / 2)
//   calib_gather16   lane i reads ONE random float4             (sky texel gathers)
//   calib_gather64   lane i reads ONE random 64 B record        (BVH node / shading record: 4 x dwordx4)
//   calib_gather128  lane i reads ONE random 128 B record       (8 x dwordx4: a whole 128-byte line)
// Run:  rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- ./fetch_calib [log2_bytes]
// and compare the per-kernel FETCH_SIZE with the "bytes" column printed here (tools/fetch_calib_report.py).
// The timing columns say what the fabric really moves: if 64 B and 128 B records run at the same RECORD rate, a 64 B gather costs
// a full 128 B line on the memory side.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e = (x);                                                            \
        if (e != hipSuccess) {                                                         \
            printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__);       \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

// bijection on [0, 2^bits): multiply by an odd constant, then xor-shift (both invertible mod 2^bits)
__device__ __forceinline__ uint32_t scramble(uint32_t i, uint32_t bits) {
    const uint32_t mask = bits >= 32 ? 0xFFFFFFFFu : ((1u << bits) - 1u);
    uint32_t x = (i * 2654435761u) & mask;
    x ^= x >> (bits / 2 + 1);
    x = (x * 2246822519u) & mask;
    return x;
}

__global__ __launch_bounds__(256) void calib_stream16(const float4* __restrict__ t, uint32_t n, float* out) {
    float acc = 0.0f;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 v = t[i];
        acc += v.x + v.w;
    }
    if (acc == 123.456f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_calib_gather16(const float4* __restrict__ t, uint32_t n_rec, uint32_t bits, float* out) {
    float acc = 0.0f;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_rec; i += gridDim.x * blockDim.x) {
        const float4 v = t[scramble(i, bits)];
        acc += v.x + v.w;
    }
    if (acc == 123.456f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_calib_gather64(const float4* __restrict__ t, uint32_t n_rec, uint32_t bits, float* out) {
    float acc = 0.0f;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_rec; i += gridDim.x * blockDim.x) {
        const float4* p = t + 4 * (size_t)scramble(i, bits);
        const float4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc += a.x + b.y + c.z + d.w;
    }
    if (acc == 123.456f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_calib_gather128(const float4* __restrict__ t, uint32_t n_rec, uint32_t bits, float* out) {
    float acc = 0.0f;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_rec; i += gridDim.x * blockDim.x) {
        const float4* p = t + 8 * (size_t)scramble(i, bits);
        const float4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4], f = p[5], g = p[6], h = p[7];
        acc += a.x + b.y + c.z + d.w + e.x + f.y + g.z + h.w;
    }
    if (acc == 123.456f) out[0] = acc;
}

int main(int argc, char** argv) {
    const int log2_bytes = argc > 1 ? atoi(argv[1]) : 31;  // 2 GiB by default: 8 x the Infinity Cache
    if (log2_bytes < 20 || log2_bytes > 33) {
        printf("log2_bytes must be in [20, 33]\n");
        return 1;
    }
    const size_t bytes = (size_t)1 << log2_bytes;
    float4* t = nullptr;
    float* out = nullptr;
    CK(hipMalloc((void**)&t, bytes));
    CK(hipMalloc((void**)&out, 64));
    CK(hipMemset(t, 0x3c, bytes));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int blocks = 256 * 16;
    auto timed = [&](const char* name, size_t rec_bytes, auto launch) {
        float ms = 0.0f;
        CK(hipEventRecord(e0));
        launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double n_rec = (double)bytes / (double)rec_bytes;
        printf("CALIB %-18s table_bytes %zu record_bytes %zu records %.0f bytes %zu ms %.3f GB/s %.1f Grecords/s %.3f\n", name, bytes, rec_bytes, n_rec, bytes, ms,
               (double)bytes / ms / 1e6, n_rec / ms / 1e6);
    };
    const uint32_t n16 = (uint32_t)(bytes / 16 > 0xFFFFFFFFull ? 0xFFFFFFFFull : bytes / 16);
    const uint32_t bits16 = (uint32_t)log2_bytes - 4, bits64 = (uint32_t)log2_bytes - 6, bits128 = (uint32_t)log2_bytes - 7;
    for (int rep = 0; rep < 2; rep++) {  // the second round is the one to read (first: cold TLBs)
        timed("calib_stream16", 16, [&] { hipLaunchKernelGGL(calib_stream16, dim3(blocks), dim3(256), 0, 0, t, n16, out); });
        timed("k_calib_gather16", 16, [&] { hipLaunchKernelGGL(k_calib_gather16, dim3(blocks), dim3(256), 0, 0, t, 1u << bits16, bits16, out); });
        timed("k_calib_gather64", 64, [&] { hipLaunchKernelGGL(k_calib_gather64, dim3(blocks), dim3(256), 0, 0, t, 1u << bits64, bits64, out); });
        timed("k_calib_gather128", 128, [&] { hipLaunchKernelGGL(k_calib_gather128, dim3(blocks), dim3(256), 0, 0, t, 1u << bits128, bits128, out); });
    }
    CK(hipFree(t));
    CK(hipFree(out));
    return 0;
}
