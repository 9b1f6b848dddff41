// rt3_internal.hpp -- declarations shared by the kernel translation units and the C-ABI host layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "rt3_device.hpp"

#define RT3_FLAG_NEE_SKY 1u
#define RT3_FLAG_BLUENOISE 2u
#define RT3_FLAG_SPECULAR 4u
#define RT3_FLAG_FACEFORWARD 8u
#define RT3_FLAG_PROBE_RADIANCE 16u

namespace rt3 {

constexpr int kExtendBlock = 256;          // threads per traversal workgroup (4 waves)
constexpr unsigned kExtendMaxBlocks = 2048;  // 256 CUs x 8: grid-stride beyond that
// traversal node layouts
constexpr int kLayoutBinary64 = 0;   // 64 B: two fp32 child boxes + two references
constexpr int kLayoutWide128 = 1;    // 128 B: four {fp32 min, max, ref, pad} slots
constexpr int kLayoutWide48Q = 3;    // 48 B: as kLayoutWide64Q, references implied (node_base / tri_base + a nibble per child): 3 loads per node
constexpr int kC48Stride = 3;        // float4s per compact node
constexpr int kLayoutWide64Q = 2;    // 64 B: origin + power-of-two steps + four 8-bit boxes + four references
constexpr uint32_t kMaxStack = 64;         // traversal stack entries: LDS short stack (12) + private spill (52)
constexpr uint32_t kTopCacheNodes = 128;   // top-of-tree nodes the traversal kernels hold in LDS (8 KiB)

struct ShadeLaunch {
    GConstDev g;
    SceneDev sc;
    const uint32_t* pixels;
    const uint2* pixbn;
    uint32_t npix, width, s0, bounce;
    const void* gbuffer;
    const float* depth;
    const float* in_rays;
    const float* in_hits;
    const float* in_T;
    const uint32_t* in_count;
    uint32_t n_first;
    float* out_rays;
    float* out_T;
    uint32_t* out_count;
    float* sh_rays;
    float* sh_contrib;
    uint32_t* sh_count;
    float* lacc;
    size_t stride;
    uint32_t max_n;  // upper bound of live paths (grid sizing)
};

void launch_raygen(hipStream_t st, const GConstDev& g, const uint32_t* pixels, uint32_t npix, float* rays, size_t stride);
void launch_extend(hipStream_t st, bool count, int layout, const float4* nodes, const float4* tris, const float4* top, uint32_t n_top, const float* rays, size_t stride,
                   const uint32_t* count_ptr, uint32_t count_imm, uint32_t max_n, float* hits, uint32_t* cn, uint32_t* ct,
                   unsigned long long* totals, uint32_t* work_counter, bool payload = false);
void launch_shadow(hipStream_t st, bool count, int layout, const float4* nodes, const float4* tris, const float4* top, uint32_t n_top, const float* rays, size_t stride,
                   const uint32_t* count_ptr, uint32_t count_imm, uint32_t max_n, const float* contrib, const uint32_t* pid, float* lacc,
                   size_t lstride, uint32_t* occluded_out, uint32_t* cn, uint32_t* ct, unsigned long long* totals, uint32_t* work_counter);
void launch_trace(hipStream_t st, bool count, int layout, const float4* nodes, const float4* tris, const float4* top, uint32_t n_top, const float* ext_rays, const float* sh_rays,
                  size_t stride, const uint32_t* ext_count, const uint32_t* sh_count, uint32_t max_n, float* hits, const float* contrib, float* lacc,
                  unsigned long long* totals, uint32_t* work_ext, uint32_t* work_sh);
void launch_gbuffer(hipStream_t st, const SceneDev& sc, const uint32_t* pixels, uint32_t npix, uint32_t width, const float* hits,
                    size_t stride, void* gbuffer, float* depth);
void launch_shade(hipStream_t st, bool first, const ShadeLaunch& L);
void launch_pixbn(hipStream_t st, const uint32_t* pixels, uint32_t npix, const uint8_t* bluenoise, uint32_t bn_w, uint32_t bn_h, uint2* out);
void launch_accumulate(hipStream_t st, const GConstDev& g, const uint32_t* pixels, uint32_t npix, uint32_t width, const float* depth,
                       const float* lacc, size_t stride, uint32_t sb, int first_batch, int last_batch, float* radsum, void* light,
                       const void* prev);
void launch_postprocess(hipStream_t st, const GConstDev& g, const SceneDev& sc, const uint32_t* pixels, uint32_t npix, uint32_t width,
                        const float* depth, const void* in, void* out);
void launch_pack_tiles(hipStream_t st, const uint32_t* pixels, uint32_t npix, uint32_t width, const void* img, void* dst);
void launch_unpack_tiles(hipStream_t st, const uint32_t* pixels, uint32_t npix, uint32_t width, const void* src, void* img);

// probe-GI passes (rt3_probes.hip); atlas images are (8 * probes_x) x (8 * probes_y)
void launch_sis(hipStream_t st, uint32_t W, uint32_t probes_x, uint32_t probes_y, const void* gbuffer, void* out, float* debug);
void launch_probe_raygen(hipStream_t st, const GConstDev& g, uint32_t W, uint32_t probes_x, uint32_t probes_y, const float* depth, const void* directions,
                         void* atlas, float* rays, size_t stride, void* d2);
void launch_probe_store(hipStream_t st, const SceneDev& sc, uint32_t flags, float blend, uint32_t probes_x, uint32_t probes_y, const float* hits,
                        const void* d2, const void* prev, void* atlas);
void launch_sh_conversion(hipStream_t st, uint32_t probes_x, uint32_t probes_y, const void* atlas, void* out);
void launch_interpolate(hipStream_t st, const GConstDev& g, uint32_t W, uint32_t H, const void* gbuffer, const float* depth, const void* sh, void* light);
void launch_selftest_probes(hipStream_t st, int op, const uint32_t* in, uint32_t n, uint32_t* out);

void launch_prim_geom(hipStream_t st, const uint32_t* first_prim, uint32_t n_geoms, uint32_t n, uint32_t* prim_geom);
void set_refill_lanes(uint32_t v);
void set_pool_chunk(uint32_t v);
void set_trace_blocks(uint32_t v);
bool selftest_widths(int op, uint32_t* in_w, uint32_t* out_w);
void launch_selftest(hipStream_t st, int op, const uint32_t* in, uint32_t n, uint32_t* out);

// LBVH build (rt3_lbvh.hip).  All pointers are device memory owned by the caller except the scratch the builder
// allocates and frees itself.  Returns hipSuccess or the failing HIP error; *max_depth is read back to the host.
// Scratch memory of the builder: ONE device allocation, handed out by a bump pointer and kept by the context from build to build
// (a build made ~50 hipMalloc / hipFree pairs before, a third of its wall time on a 260 k-triangle scene).
struct BuildArena {
    char* base = nullptr;
    size_t cap = 0, used = 0;
    hipError_t reserve(size_t bytes) {  // a fresh build: everything handed out before is void
        used = 0;
        if (bytes <= cap) return hipSuccess;
        if (base) (void)hipFree(base);
        base = nullptr;
        cap = 0;
        hipError_t e = hipMalloc((void**)&base, bytes);
        if (e == hipSuccess) cap = bytes;
        return e;
    }
    template <typename T>
    hipError_t take(T** p, size_t bytes) {
        const size_t at = (used + 255) & ~(size_t)255;
        if (at + bytes > cap) return hipErrorOutOfMemory;  // the builder's bound on its own scratch was wrong: fail, never overrun
        *p = reinterpret_cast<T*>(base + at);
        used = at + bytes;
        return hipSuccess;
    }
    void release() {
        if (base) (void)hipFree(base);
        base = nullptr;
        cap = used = 0;
    }
};
struct LbvhResult {
    float4* nodes = nullptr;   // n_nodes x node_bytes: 64 B {box0, box1, ref0, ref1, pad} or 128 B 4 x {min, max, ref, pad}
    uint32_t node_bytes = 128;
    int layout = kLayoutWide128;
    float4* tris = nullptr;    // n_tris x 3 float4 (48 B), Morton order
    uint4* tri_shade = nullptr;   // n_tris x 16 B, flattened primitive order: three octahedral vertex normals + flattened geometry index
    float2* tri_uv = nullptr;     // n_tris x 3 float2: vertex uvs
    float4* top = nullptr;     // quantised four-wide layout: the first n_top nodes in breadth-first order (64 B each), child references to
    uint32_t n_top = 0;        // cached nodes rewritten as 0x40000000 | slot -- the traversal kernels keep this copy in LDS
    uint32_t n_nodes = 0, n_tris = 0, max_depth = 0;
    uint32_t bulk_copies = 0;  // array-sized host <-> device copies this build made (0 on the default path)
};
hipError_t lbvh_build(hipStream_t st, const float* verts, const uint32_t* indices, const FlatGeomDev* geoms, const uint32_t* prim_geom,
                      const uint32_t* first_prim, uint32_t n_prims, uint32_t leaf_max, uint32_t node_width, uint32_t node_quant, uint32_t collapse_mode,
                      uint32_t sah_top, uint32_t sah_device, BuildArena& arena, LbvhResult* out);

hipError_t lbvh_make_top(hipStream_t st, const float4* nodes, uint32_t n_nodes, float4** top, uint32_t* n_top);

}  // namespace rt3
