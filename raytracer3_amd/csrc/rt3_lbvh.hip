// rt3_lbvh.hip -- GPU LBVH build for gfx950; stands in for create_acceleration_structure
// (src/renderer/vulkan/raytracing.rs:88-148, flags PREFER_FAST_TRACE :103,131) which hands the job to the Vulkan driver.
//
//   k_prim_bounds  triangle boxes + scene / centroid bounds (wave reduce, ordered-uint atomics)
//   k_morton       63-bit Morton code of the box centre (21 bits per axis)
//   hipcub radix sort of (code, primitive) pairs, 64-bit keys, stable -> ties keep primitive order
//   k_leaves       Morton-ordered triangle records {v0,v1,v2,prim} (48 B) + padded leaf boxes
//   k_hierarchy    Karras 2012 radix-tree topology, one thread per internal node
//   k_refit        bottom-up boxes of every binary node: the second thread to arrive at a node (agent-scope atomic +
//                  fences) merges the two child boxes and climbs on
//   k_keep_flags   binary depth of every node (parent walk); multi-triangle leaves: a node covering <= leaf_max
//                  triangles is referenced as a leaf; wide nodes: even-depth nodes survive and absorb their children
//   hipcub exclusive scan of the keep flags -> dense node numbering in index order
//   k_emit_nodes   64 B binary nodes {box0, box1, ref0, ref1} or 128 B four-wide nodes 4 x {min, max, ref, pad}
// min/max are exact, so the tree is a pure function of the input and is compared bit for bit with the CPU oracle.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "rt3_device.hpp"
#include "rt3_internal.hpp"

namespace rt3 {

__device__ __forceinline__ uint32_t float_to_ordered(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ordered_to_float(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }

// The world-space triangle of flattened primitive `prim`: the vertices of its geometry under its instance's matrix (as uploaded when
// that matrix is the identity).  Every user goes through here, so two triangles that share an edge see bit-identical end points.
__device__ __forceinline__ void fetch_triangle(const float* verts, const uint32_t* indices, const FlatGeomDev* geoms, const uint32_t* prim_geom,
                                               const uint32_t* first_prim, uint32_t prim, V3& a, V3& b, V3& c) {
    uint32_t g = prim_geom[prim];
    const FlatGeomDev& fg = geoms[g];
    uint32_t io = fg.g.index_offset + 3u * (prim - first_prim[g]);
    const float* v0 = verts + 8 * (size_t)(fg.g.vertex_offset + indices[io]);
    const float* v1 = verts + 8 * (size_t)(fg.g.vertex_offset + indices[io + 1]);
    const float* v2 = verts + 8 * (size_t)(fg.g.vertex_offset + indices[io + 2]);
    a = v3(v0[0], v0[1], v0[2]);
    b = v3(v1[0], v1[1], v1[2]);
    c = v3(v2[0], v2[1], v2[2]);
    if (!fg.identity) {
        a = transform_point(fg.m, a);
        b = transform_point(fg.m, b);
        c = transform_point(fg.m, c);
    }
}
// flattened primitive -> flattened geometry: the last entry of first_prim (ascending, n_geoms of them) that is <= prim
__global__ void k_prim_geom(const uint32_t* first_prim, uint32_t n_geoms, uint32_t n, uint32_t* prim_geom) {
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
        uint32_t lo = 0, hi = n_geoms;  // invariant: first_prim[lo] <= p, answer in [lo, hi)
        while (hi - lo > 1u) {
            uint32_t mid = (lo + hi) >> 1;
            if (first_prim[mid] <= p) lo = mid;
            else hi = mid;
        }
        prim_geom[p] = lo;
    }
}
void launch_prim_geom(hipStream_t st, const uint32_t* first_prim, uint32_t n_geoms, uint32_t n, uint32_t* prim_geom) {
    if (n == 0 || n_geoms == 0) return;
    hipLaunchKernelGGL(k_prim_geom, dim3((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256), dim3(256), 0, st, first_prim, n_geoms, n, prim_geom);
}

// bounds[0..2] scene min, [3..5] scene max, [6..8] centroid min, [9..11] centroid max (ordered-uint encoded)
__global__ void k_prim_bounds(const float* verts, const uint32_t* indices, const FlatGeomDev* geoms, const uint32_t* prim_geom,
                              const uint32_t* first_prim, uint32_t n, float* bmin, float* bmax, uint32_t* bounds) {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        V3 a, b, c;
        fetch_triangle(verts, indices, geoms, prim_geom, first_prim, i, a, b, c);
        float mn[3] = {fmin_sel(a.x, fmin_sel(b.x, c.x)), fmin_sel(a.y, fmin_sel(b.y, c.y)), fmin_sel(a.z, fmin_sel(b.z, c.z))};
        float mx[3] = {fmax_sel(a.x, fmax_sel(b.x, c.x)), fmax_sel(a.y, fmax_sel(b.y, c.y)), fmax_sel(a.z, fmax_sel(b.z, c.z))};
#pragma unroll
        for (int k = 0; k < 3; k++) {
            bmin[3 * (size_t)i + k] = mn[k];
            bmax[3 * (size_t)i + k] = mx[k];
            float ce = (mn[k] + mx[k]) * 0.5f;
            lo[k] = fmin_sel(lo[k], mn[k]);
            hi[k] = fmax_sel(hi[k], mx[k]);
            clo[k] = fmin_sel(clo[k], ce);
            chi[k] = fmax_sel(chi[k], ce);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo[k] = fmin_sel(lo[k], __shfl_xor(lo[k], off));
            hi[k] = fmax_sel(hi[k], __shfl_xor(hi[k], off));
            clo[k] = fmin_sel(clo[k], __shfl_xor(clo[k], off));
            chi[k] = fmax_sel(chi[k], __shfl_xor(chi[k], off));
        }
    }
    // one set of 12 atomics per workgroup, and few workgroups (the launch caps the grid): the 12 words share a cache line, on which
    // returning or not, atomics retire at ~90 per microsecond -- 48 k of them (one set per wave of a full grid) took 0.55 ms
    __shared__ uint32_t s_b[12];
    if (threadIdx.x < 12) s_b[threadIdx.x] = (threadIdx.x % 6) < 3 ? 0xFFFFFFFFu : 0u;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            atomicMin(&s_b[k], float_to_ordered(lo[k]));
            atomicMax(&s_b[3 + k], float_to_ordered(hi[k]));
            atomicMin(&s_b[6 + k], float_to_ordered(clo[k]));
            atomicMax(&s_b[9 + k], float_to_ordered(chi[k]));
        }
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        if ((threadIdx.x % 6) < 3) atomicMin(&bounds[threadIdx.x], s_b[threadIdx.x]);
        else atomicMax(&bounds[threadIdx.x], s_b[threadIdx.x]);
    }
}

// shading record of hit_logic.slang:10-27: the three vertex normals (octahedral, 2 x 16 bit: rt3_device.hpp) + the flattened geometry
// index, 16 B; the three uv pairs go to their own stream, which only textured geometries read
__global__ void k_tri_shade(const float* verts, const uint32_t* indices, const FlatGeomDev* geoms, const uint32_t* prim_geom,
                            const uint32_t* first_prim, uint32_t n, uint4* rec, float2* uv) {
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
        uint32_t g = prim_geom[p];
        const GeometryInfoDev& gi = geoms[g].g;
        uint32_t io = gi.index_offset + 3u * (p - first_prim[g]);
        const float* v0 = verts + 8 * (size_t)(gi.vertex_offset + indices[io]);
        const float* v1 = verts + 8 * (size_t)(gi.vertex_offset + indices[io + 1]);
        const float* v2 = verts + 8 * (size_t)(gi.vertex_offset + indices[io + 2]);
        rec[p] = make_uint4(octa_encode16(v3(v0[3], v0[4], v0[5])), octa_encode16(v3(v1[3], v1[4], v1[5])), octa_encode16(v3(v2[3], v2[4], v2[5])), g);
        uv[3 * (size_t)p + 0] = make_float2(v0[6], v0[7]);
        uv[3 * (size_t)p + 1] = make_float2(v1[6], v1[7]);
        uv[3 * (size_t)p + 2] = make_float2(v2[6], v2[7]);
    }
}

__device__ __forceinline__ uint64_t expand21(uint32_t v) {
    uint64_t x = v & 0x1FFFFFu;
    x = (x | (x << 32)) & 0x001F00000000FFFFull;
    x = (x | (x << 16)) & 0x001F0000FF0000FFull;
    x = (x | (x << 8)) & 0x100F00F00F00F00Full;
    x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}

__global__ void k_morton(const float* bmin, const float* bmax, const uint32_t* bounds, uint32_t n, uint64_t* keys, uint32_t* vals) {
    float cmin[3], cext[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        cmin[k] = ordered_to_float(bounds[6 + k]);
        cext[k] = ordered_to_float(bounds[9 + k]) - cmin[k];
    }
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint32_t q[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float ce = (bmin[3 * (size_t)i + k] + bmax[3 * (size_t)i + k]) * 0.5f;
            float nrm = cext[k] > 0.0f ? (ce - cmin[k]) / cext[k] : 0.0f;
            q[k] = (uint32_t)fmin_sel(nrm * 2097152.0f, 2097151.0f);
        }
        keys[i] = (expand21(q[0]) << 2) | (expand21(q[1]) << 1) | expand21(q[2]);
        vals[i] = i;
    }
}

__global__ void k_leaves(const float* verts, const uint32_t* indices, const FlatGeomDev* geoms, const uint32_t* prim_geom,
                         const uint32_t* first_prim, const uint32_t* sorted_prim, const float* bmin, const float* bmax,
                         const uint32_t* bounds, uint32_t n, float4* tris, float* lmin, float* lmax) {
    float ex = ordered_to_float(bounds[3]) - ordered_to_float(bounds[0]);
    float ey = ordered_to_float(bounds[4]) - ordered_to_float(bounds[1]);
    float ez = ordered_to_float(bounds[5]) - ordered_to_float(bounds[2]);
    float pad = fmax_sel(ex, fmax_sel(ey, ez)) * 1.0e-5f;  // conservative leaf padding
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        uint32_t p = sorted_prim[k];
        V3 a, b, c;
        fetch_triangle(verts, indices, geoms, prim_geom, first_prim, p, a, b, c);
        // the three vertices exactly as uploaded (not v0 + edges): triangles that share an edge must see bit-identical end points
        // for the watertight edge functions of the triangle test
        tris[3 * (size_t)k + 0] = make_float4(a.x, a.y, a.z, b.x);
        tris[3 * (size_t)k + 1] = make_float4(b.y, b.z, c.x, c.y);
        tris[3 * (size_t)k + 2] = make_float4(c.z, __uint_as_float(p), 0.0f, 0.0f);
#pragma unroll
        for (int j = 0; j < 3; j++) {
            lmin[3 * (size_t)k + j] = bmin[3 * (size_t)p + j] - pad;
            lmax[3 * (size_t)k + j] = bmax[3 * (size_t)p + j] + pad;
        }
    }
}

__device__ __forceinline__ int delta(const uint64_t* codes, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    uint64_t a = codes[i], b = codes[j];
    if (a != b) return __clzll((long long)(a ^ b));
    return 64 + __clz((int)((uint32_t)i ^ (uint32_t)j));
}

// Karras, "Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees", HPG 2012, section 4
__global__ void k_hierarchy(const uint64_t* codes, int n, uint32_t* left, uint32_t* right, uint32_t* parent_internal, uint32_t* parent_leaf,
                            uint32_t* range_lo, uint32_t* range_cnt) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n - 1; i += gridDim.x * blockDim.x) {
        int d = (delta(codes, n, i, i + 1) - delta(codes, n, i, i - 1)) >= 0 ? 1 : -1;
        int dmin = delta(codes, n, i, i - d);
        int lmax = 2;
        while (delta(codes, n, i, i + lmax * d) > dmin) lmax *= 2;
        int l = 0;
        for (int t = lmax / 2; t >= 1; t /= 2)
            if (delta(codes, n, i, i + (l + t) * d) > dmin) l += t;
        int j = i + l * d;
        int dnode = delta(codes, n, i, j);
        int s = 0, t = l;
        do {
            t = (t + 1) >> 1;
            if (delta(codes, n, i, i + (s + t) * d) > dnode) s += t;
        } while (t > 1);
        int gamma = i + s * d + (d < 0 ? d : 0);
        int lo = i < j ? i : j, hi = i < j ? j : i;
        range_lo[i] = (uint32_t)lo;
        range_cnt[i] = (uint32_t)(hi - lo + 1);
        if (lo == gamma) {
            left[i] = 0x80000000u | (uint32_t)gamma;
            parent_leaf[gamma] = (uint32_t)i;
        } else {
            left[i] = (uint32_t)gamma;
            parent_internal[gamma] = (uint32_t)i;
        }
        if (hi == gamma + 1) {
            right[i] = 0x80000000u | (uint32_t)(gamma + 1);
            parent_leaf[gamma + 1] = (uint32_t)i;
        } else {
            right[i] = (uint32_t)(gamma + 1);
            parent_internal[gamma + 1] = (uint32_t)i;
        }
        if (i == 0) parent_internal[0] = 0xFFFFFFFFu;
    }
}

// bottom-up refit.  nbox holds each binary node's own box (6 floats).  Inter-workgroup hand-off of a child's box goes
// through an agent-scope fence + returning atomic on the node's arrival counter, then an agent-scope fence on the
// consumer before it reads (per-XCD L2s are not coherent).
__global__ void k_refit(const uint32_t* left, const uint32_t* right, const uint32_t* parent_internal, const uint32_t* parent_leaf,
                        const float* lmin, const float* lmax, uint32_t n, float* nbox, uint32_t* arrive) {
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        uint32_t cur = parent_leaf[k];
        while (cur != 0xFFFFFFFFu) {
            __threadfence();
            uint32_t prev = atomicAdd(&arrive[cur], 1u);
            if (prev == 0u) break;  // first arrival: the sibling subtree is not finished yet
            __threadfence();
            uint32_t ch[2] = {left[cur], right[cur]};
            float mn[2][3], mx[2][3];
#pragma unroll
            for (int c = 0; c < 2; c++) {
                if (ch[c] & 0x80000000u) {
                    uint32_t q = ch[c] & 0x7FFFFFFFu;
#pragma unroll
                    for (int j = 0; j < 3; j++) {
                        mn[c][j] = lmin[3 * (size_t)q + j];
                        mx[c][j] = lmax[3 * (size_t)q + j];
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 3; j++) {
                        mn[c][j] = __hip_atomic_load(&nbox[6 * (size_t)ch[c] + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        mx[c][j] = __hip_atomic_load(&nbox[6 * (size_t)ch[c] + 3 + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 3; j++) {
                __hip_atomic_store(&nbox[6 * (size_t)cur + j], fmin_sel(mn[0][j], mn[1][j]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&nbox[6 * (size_t)cur + 3 + j], fmax_sel(mx[0][j], mx[1][j]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            cur = parent_internal[cur];
        }
    }
}

// The boxes of the nodes at and below the cluster roots of the SAH top (subtrees of at most T triangles), straight from each node's
// leaf range: what the SAH top reads before it re-links everything above (whose boxes it then writes itself) -- no climb, no fences.
// min / max are exact and associative: the same boxes as the bottom-up refit.
__global__ void k_refit_clusters(const uint32_t* range_lo, const uint32_t* range_cnt, const float* lmin, const float* lmax, uint32_t nn, uint32_t T, float* nbox) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nn; i += gridDim.x * blockDim.x) {
        const uint32_t cnt = range_cnt[i];
        if (cnt > T) continue;
        const uint32_t lo = range_lo[i];
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t q = lo; q < lo + cnt; q++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
                mn[j] = fmin_sel(mn[j], lmin[3 * (size_t)q + j]);
                mx[j] = fmax_sel(mx[j], lmax[3 * (size_t)q + j]);
            }
#pragma unroll
        for (int j = 0; j < 3; j++) {
            nbox[6 * (size_t)i + j] = mn[j];
            nbox[6 * (size_t)i + 3 + j] = mx[j];
        }
    }
}

// binary depth (root = 0) by walking the parents; keep[i] = 1 if node i survives into the traversal array:
// it covers more than leaf_max triangles (or is the root) and, for four-wide nodes, sits at an even depth.
__global__ void k_keep_flags(const uint32_t* parent_internal, const uint32_t* live_f, uint32_t nn, int wide,
                             uint32_t* keep, uint32_t* max_levels) {
    uint32_t best = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nn; i += gridDim.x * blockDim.x) {
        uint32_t d = 0, p = parent_internal[i];
        while (p != 0xFFFFFFFFu) {
            d++;
            p = parent_internal[p];
        }
        bool live = live_f[i] != 0u;
        bool k = live && (!wide || (d & 1u) == 0u);
        keep[i] = k ? 1u : 0u;
        if (k) {
            uint32_t lvl = (wide ? d / 2u : d) + 2u;  // levels from the root down to this node's leaf slots
            best = lvl > best ? lvl : best;
        }
    }
    if (best) atomicMax(max_levels, best);
}

__device__ __forceinline__ void slot_of(uint32_t ch, const float* lmin, const float* lmax, const float* nbox, const uint32_t* range_lo,
                                        const uint32_t* range_cnt, const uint32_t* newidx, const uint32_t* live, float mn[3], float mx[3], uint32_t& ref) {
    if (ch & 0x80000000u) {
        uint32_t q = ch & 0x7FFFFFFFu;
        ref = 0x80000000u | q;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            mn[j] = lmin[3 * (size_t)q + j];
            mx[j] = lmax[3 * (size_t)q + j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            mn[j] = nbox[6 * (size_t)ch + j];
            mx[j] = nbox[6 * (size_t)ch + 3 + j];
        }
        // live[ch]: the node stays a node (it covers more than leaf_max triangles, or the cost-driven collapse decided so); otherwise
        // it is referenced as a multi-triangle leaf over its range
        ref = live[ch] ? newidx[ch] : (0x80000000u | ((range_cnt[ch] - 1u) << 28) | range_lo[ch]);
    }
}

// 64-byte four-wide node with 8-bit child boxes on a per-axis power-of-two grid anchored at the node's min corner:
//   float4 0: origin.xyz, exponents ex | ey << 8 | ez << 16 (quantize_words; quantize_node turns them into three float steps: word 3, words 14 / 15)
//   float4 1 + first half of 2: 4 x {qlo.xyz, qhi.xyz} bytes
//   float4 2 second half + float4 3 first half: the four references
// Conservative with respect to the decode expression origin + float(q) * scale used by the traversal kernels.
__device__ void quantize_words(const float (*mn)[3], const float (*mx)[3], const uint32_t* ref, uint32_t ns, uint32_t* w) {
#pragma unroll
    for (int k = 0; k < 16; k++) w[k] = 0u;
    uint32_t qb[4][6];
    for (int a = 0; a < 3; a++) {
        float lo = mn[0][a], hi = mx[0][a];
        for (uint32_t k = 1; k < ns; k++) {
            lo = fmin_sel(lo, mn[k][a]);
            hi = fmax_sel(hi, mx[k][a]);
        }
        w[a] = __float_as_uint(lo);
        float sdiv = (hi - lo) / 255.0f;
        uint32_t bits = __float_as_uint(sdiv), e = (bits >> 23) & 0xFFu;
        if (bits & 0x7FFFFFu) e += 1;
        if (e < 1) e = 1;
        for (;;) {  // grow the step until every child fits in 8 bits
            float scale = __uint_as_float(e << 23);
            uint32_t worst = 0;
            for (uint32_t k = 0; k < ns; k++) {
                float fl = floorf((mn[k][a] - lo) / scale);
                uint32_t ql = fl > 255.0f ? 255u : (uint32_t)fl;
                while (ql > 0 && lo + (float)ql * scale > mn[k][a]) ql--;
                float fh = ceilf((mx[k][a] - lo) / scale);
                uint32_t qh = fh > 1024.0f ? 1024u : (uint32_t)fh;
                while (qh < 1024u && lo + (float)qh * scale < mx[k][a]) qh++;
                worst = qh > worst ? qh : worst;
                qb[k][a] = ql;
                qb[k][3 + a] = qh > 255u ? 255u : qh;
            }
            if (worst <= 255u) break;
            e++;
        }
        w[3] |= e << (8 * a);
    }
    for (uint32_t k = 0; k < 4; k++) {
        for (int j = 0; j < 6; j++) {
            uint32_t v = k < ns ? qb[k][j] : (j < 3 ? 255u : 0u);
            uint32_t byte = 6 * k + j;
            w[4 + (byte >> 2)] |= v << (8 * (byte & 3u));
        }
        w[10 + k] = k < ns ? ref[k] : 0xFFFFFFFFu;
    }
}
__device__ void quantize_node(const float (*mn)[3], const float (*mx)[3], const uint32_t* ref, uint32_t ns, float4* out) {
    uint32_t w[16];
    quantize_words(mn, mx, ref, ns, w);
    // the 64-byte node carries its three steps as FLOATS (word 3 and the two spare words 14, 15): the walk multiplies them into the
    // ray's inverse direction at every node and used to rebuild each from its exponent byte first (a shift and a mask per axis)
    const uint32_t ex = w[3];
    w[3] = (ex & 0xFFu) << 23;
    w[14] = ((ex >> 8) & 0xFFu) << 23;
    w[15] = ((ex >> 16) & 0xFFu) << 23;
#pragma unroll
    for (int k = 0; k < 4; k++)
        out[k] = make_float4(__uint_as_float(w[4 * k]), __uint_as_float(w[4 * k + 1]), __uint_as_float(w[4 * k + 2]), __uint_as_float(w[4 * k + 3]));
}
// compact 48-byte node: words 0..9 as above; the references are implied -- internal children are numbered consecutively
// from node_base, the triangles of leaf children are stored consecutively from tri_base (slot order), and one nibble
// per child says what it is: 0 internal, 8 | (count - 1) leaf, 7 empty.  Nibbles 0,1 -> bits 24..31 of word 3,
// nibble 2 / 3 -> top of word 10 (node_base) / word 11 (tri_base).
__device__ void compact_node(const float (*mn)[3], const float (*mx)[3], const uint32_t* meta, uint32_t ns, uint32_t node_base, uint32_t tri_base,
                             float4* out) {
    uint32_t w[16];
    const uint32_t zero[4] = {0, 0, 0, 0};
    quantize_words(mn, mx, zero, ns, w);
    uint32_t m[4];
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) m[k] = k < ns ? meta[k] : 7u;
    w[3] |= (m[0] << 24) | (m[1] << 28);
    w[10] = (node_base & 0x0FFFFFFFu) | (m[2] << 28);
    w[11] = (tri_base & 0x0FFFFFFFu) | (m[3] << 28);
#pragma unroll
    for (int k = 0; k < 3; k++)
        out[k] = make_float4(__uint_as_float(w[4 * k]), __uint_as_float(w[4 * k + 1]), __uint_as_float(w[4 * k + 2]), __uint_as_float(w[4 * k + 3]));
}

// live[i] = node i stays a node of the traversal array (collapse 0 / 1: it is the root or covers more than leaf_max triangles)
__global__ void k_live_flags(const uint32_t* range_cnt, uint32_t nn, uint32_t leaf_max, uint32_t* live) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nn; i += gridDim.x * blockDim.x) live[i] = (i == 0u || range_cnt[i] > leaf_max) ? 1u : 0u;
}

// ---- cost-driven collapse (RT3_OPT_WIDE_COLLAPSE = 2; the oracle's orc_accel_build has the recurrences, after Ylitie, Karras, Laine 2017):
// bottom-up over the final binary tree -- a leaf starts at its parent, the second arrival at a node computes it from its two children --
//   C(n,1) = min( A_n cnt_n [a leaf of <= leaf_max triangles],  A_n + D(n,4) [a four-wide node] ),  C(n,m) = min( D(n,m), C(n,m-1) ),
//   D(n,j) = min over 0 < k < j of C(left,k) + C(right,j-k),  C(triangle,.) = A,
// fp32, the oracle's order, strict '<'.  Also writes the TRUE triangle count of every node (range_cnt) and live[] (not a leaf).
// dk bits: 0-1 k of D(n,4); 2 k of D(n,3) minus 1; 3 C(n,1) is a leaf; 4 C(n,2) = D(n,2); 5 C(n,3) = D(n,3).
__device__ __forceinline__ float half_area_box(const float* mn, const float* mx) {
    const float ex = mx[0] - mn[0], ey = mx[1] - mn[1], ez = mx[2] - mn[2];
    return (ex * ey + ey * ez) + ez * ex;
}
__global__ void k_dp_up(const uint32_t* left, const uint32_t* right, const uint32_t* parent_internal, const uint32_t* parent_leaf, const float* lmin,
                        const float* lmax, const float* nbox, uint32_t n, uint32_t leaf_max, uint32_t* range_cnt, float* dc, uint32_t* dk, uint32_t* live,
                        uint32_t* arrive) {
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < n; q += gridDim.x * blockDim.x) {
        uint32_t cur = parent_leaf[q];
        while (cur != 0xFFFFFFFFu) {
            __threadfence();
            const uint32_t prev = atomicAdd(&arrive[cur], 1u);
            if (prev == 0u) break;  // first arrival: the sibling subtree is not finished yet
            __threadfence();
            const uint32_t ch[2] = {left[cur], right[cur]};
            float C[2][3];
            uint32_t cnt = 0;
#pragma unroll
            for (int c = 0; c < 2; c++) {
                if (ch[c] & 0x80000000u) {
                    const uint32_t t = ch[c] & 0x7FFFFFFFu;
                    const float a = half_area_box(lmin + 3 * (size_t)t, lmax + 3 * (size_t)t);
                    C[c][0] = C[c][1] = C[c][2] = a;
                    cnt += 1u;
                } else {
#pragma unroll
                    for (int j = 0; j < 3; j++) C[c][j] = __hip_atomic_load(&dc[3 * (size_t)ch[c] + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    cnt += __hip_atomic_load(&range_cnt[ch[c]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            const float d2 = C[0][0] + C[1][0];
            float d3 = C[0][0] + C[1][1];
            uint32_t k3 = 1;
            {
                const float b = C[0][1] + C[1][0];
                if (b < d3) { d3 = b; k3 = 2; }
            }
            float d4 = C[0][0] + C[1][2];
            uint32_t k4 = 1;
            {
                float b = C[0][1] + C[1][1];
                if (b < d4) { d4 = b; k4 = 2; }
                b = C[0][2] + C[1][0];
                if (b < d4) { d4 = b; k4 = 3; }
            }
            const float A = half_area_box(nbox + 6 * (size_t)cur, nbox + 6 * (size_t)cur + 3);
            const float cint = A + d4, cleaf = (float)cnt * A;
            const uint32_t leaf1 = (cur != 0u && cnt <= leaf_max && cleaf <= cint) ? 1u : 0u;
            const float c1 = leaf1 ? cleaf : cint;
            const uint32_t s2 = d2 < c1 ? 1u : 0u;
            const float c2 = s2 ? d2 : c1;
            const uint32_t s3 = d3 < c2 ? 1u : 0u;
            const float c3 = s3 ? d3 : c2;
            __hip_atomic_store(&dc[3 * (size_t)cur], c1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&dc[3 * (size_t)cur + 1], c2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&dc[3 * (size_t)cur + 2], c3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&range_cnt[cur], cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            dk[cur] = k4 | ((k3 - 1u) << 2) | (leaf1 << 3) | (s2 << 4) | (s3 << 5);
            live[cur] = leaf1 ? 0u : 1u;
            cur = parent_internal[cur];
        }
    }
}
// TREE ORDER: the position of a triangle (the first triangle of a node) in the depth-first order of the final binary tree = the sum, over
// the ancestors it hangs under on the RIGHT, of the triangle count of their left child.  One walk to the root per leaf / node, no
// synchronisation; every subtree becomes a contiguous range (so any node may be referenced as a multi-triangle leaf).
__device__ __forceinline__ uint32_t tree_position(uint32_t me_ref, uint32_t parent, const uint32_t* left, const uint32_t* right, const uint32_t* parent_internal,
                                                  const uint32_t* range_cnt) {
    uint32_t pos = 0;
    while (parent != 0xFFFFFFFFu) {
        if (right[parent] == me_ref) {
            const uint32_t l = left[parent];
            pos += (l & 0x80000000u) ? 1u : range_cnt[l];
        }
        me_ref = parent;
        parent = parent_internal[parent];
    }
    return pos;
}
__global__ void k_tree_order(const uint32_t* left, const uint32_t* right, const uint32_t* parent_internal, const uint32_t* parent_leaf, const uint32_t* range_cnt,
                             uint32_t n, uint32_t nn, uint32_t* newpos, uint32_t* range_lo) {
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n + nn; t += gridDim.x * blockDim.x) {
        if (t < n) newpos[t] = tree_position(0x80000000u | t, parent_leaf[t], left, right, parent_internal, range_cnt);
        else range_lo[t - n] = tree_position(t - n, parent_internal[t - n], left, right, parent_internal, range_cnt);
    }
}
// move the triangle records and leaf boxes to their tree-order places; rewrite the leaf references of the links
__global__ void k_tree_reorder(const uint32_t* newpos, uint32_t n, uint32_t nn, const float4* tris_in, float4* tris_out, const float* lmin_in, const float* lmax_in,
                               float* lmin_out, float* lmax_out, uint32_t* left, uint32_t* right) {
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n + nn; t += gridDim.x * blockDim.x) {
        if (t < n) {
            const uint32_t p = newpos[t];
#pragma unroll
            for (int k = 0; k < 3; k++) tris_out[3 * (size_t)p + k] = tris_in[3 * (size_t)t + k];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                lmin_out[3 * (size_t)p + j] = lmin_in[3 * (size_t)t + j];
                lmax_out[3 * (size_t)p + j] = lmax_in[3 * (size_t)t + j];
            }
        } else {
            const uint32_t i = t - n, l = left[i], r = right[i];
            if (l & 0x80000000u) left[i] = 0x80000000u | newpos[l & 0x7FFFFFFFu];
            if (r & 0x80000000u) right[i] = 0x80000000u | newpos[r & 0x7FFFFFFFu];
        }
    }
}

// child slots of surviving node i, in tree order.  Four-wide nodes, collapse 0 (even binary depth): a child that is itself
// a live internal node is absorbed (its two children take its place).  Collapse 1 (surface area, default): the two child
// slots are grown to (up to) four by repeatedly replacing the live internal slot of largest surface area by its two
// children (ties: first slot); half area = (ex*ey + ey*ez) + ez*ex in fp32, like the oracle.
// Collapse 2 (cost-driven, default since round 3): the choices k_dp_up recorded in dk are unfolded top-down, left subtree's slots first
// (the oracle's dp_slots).
__device__ __forceinline__ uint32_t gather_slots(uint32_t i, const uint32_t* left, const uint32_t* right, const uint32_t* live,
                                                 const float* nbox, const uint32_t* dk, int wide, int collapse, uint32_t sl[4]) {
    uint32_t ns = 0;
    if (wide && collapse == 2) {
        uint32_t sn[8], sm[8];
        int sp = 0;
        const uint32_t k4 = dk[i] & 3u;
        sl[0] = sl[1] = sl[2] = sl[3] = 0xFFFFFFFFu;
        sn[sp] = right[i]; sm[sp++] = 4u - k4;
        sn[sp] = left[i]; sm[sp++] = k4;
        while (sp > 0) {
            const uint32_t nd = sn[--sp];
            uint32_t m = sm[sp];
            if (nd & 0x80000000u) { sl[ns++] = nd; continue; }
            const uint32_t f = dk[nd];
            if (m == 3u && !(f & 32u)) m = 2u;
            if (m == 2u && !(f & 16u)) m = 1u;
            if (m == 1u) { sl[ns++] = nd; continue; }
            const uint32_t kl = m == 2u ? 1u : ((f >> 2) & 1u) + 1u;
            sn[sp] = right[nd]; sm[sp++] = m - kl;
            sn[sp] = left[nd]; sm[sp++] = kl;
        }
        return ns;
    }
    if (wide && collapse) {
        ns = 2;
        sl[0] = left[i];
        sl[1] = right[i];
        sl[2] = sl[3] = 0xFFFFFFFFu;
        for (int it = 0; it < 2; it++) {
            int best = -1;
            float ba = -1.0f;
            for (uint32_t k = 0; k < ns; k++) {
                const uint32_t ch = sl[k];
                if ((ch & 0x80000000u) || !live[ch]) continue;
                const float ex = nbox[6 * (size_t)ch + 3] - nbox[6 * (size_t)ch], ey = nbox[6 * (size_t)ch + 4] - nbox[6 * (size_t)ch + 1],
                            ez = nbox[6 * (size_t)ch + 5] - nbox[6 * (size_t)ch + 2];
                const float a = (ex * ey + ey * ez) + ez * ex;
                if (a > ba) {
                    ba = a;
                    best = (int)k;
                }
            }
            if (best < 0) break;
            const uint32_t ch = sl[best];
            for (int k = (int)ns; k > best + 1; k--) sl[k] = sl[k - 1];
            sl[best] = left[ch];
            sl[best + 1] = right[ch];
            ns++;
        }
        return ns;
    }
    const uint32_t c2[2] = {left[i], right[i]};
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const uint32_t ch = c2[c];
        if (wide && !(ch & 0x80000000u) && live[ch]) {
            sl[ns++] = left[ch];
            sl[ns++] = right[ch];
        } else {
            sl[ns++] = ch;
        }
    }
    for (uint32_t k = ns; k < 4; k++) sl[k] = 0xFFFFFFFFu;
    return ns;
}

// surface-area collapse, one four-wide level per launch: every node of the frontier survives; its live internal slots
// form the next frontier
__global__ void k_wide_level(const uint32_t* left, const uint32_t* right, const uint32_t* live, const float* nbox, const uint32_t* dk, int collapse,
                             const uint32_t* frontier, const uint32_t* n_frontier_ptr, uint32_t* keep, uint32_t* next, uint32_t* n_next) {
    const uint32_t n_frontier = *n_frontier_ptr;  // written by the level before: levels are launched back to back, no host round trip
    for (uint32_t f = blockIdx.x * blockDim.x + threadIdx.x; f < n_frontier; f += gridDim.x * blockDim.x) {
        const uint32_t i = frontier[f];
        keep[i] = 1u;
        uint32_t sl[4];
        const uint32_t ns = gather_slots(i, left, right, live, nbox, dk, 1, collapse, sl);
        for (uint32_t k = 0; k < ns; k++)
            if (!(sl[k] & 0x80000000u) && live[sl[k]]) next[atomicAdd(n_next, 1u)] = sl[k];
    }
}

// compact layout, pass 1: per surviving node the number of internal child slots and of triangles in leaf slots
__global__ void k_child_counts(const uint32_t* left, const uint32_t* right, const uint32_t* range_cnt, const float* nbox, const uint32_t* keep, uint32_t nn,
                               const uint32_t* live, const uint32_t* dk, int collapse, uint32_t* n_internal, uint32_t* n_leaf_tris) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nn; i += gridDim.x * blockDim.x) {
        uint32_t ci = 0, ti = 0;
        if (keep[i]) {
            uint32_t sl[4];
            const uint32_t ns = gather_slots(i, left, right, live, nbox, dk, 1, collapse, sl);
            for (uint32_t k = 0; k < ns; k++) {
                if (sl[k] & 0x80000000u) ti += 1u;
                else if (live[sl[k]]) ci += 1u;
                else ti += range_cnt[sl[k]];
            }
        }
        n_internal[i] = ci;
        n_leaf_tris[i] = ti;
    }
}
// compact layout, pass 2 (after the exclusive sums): a child's index is 1 + node_base(parent) + rank among the internal slots
__global__ void k_assign_index(const uint32_t* left, const uint32_t* right, const uint32_t* live, const float* nbox, const uint32_t* keep, uint32_t nn,
                               const uint32_t* dk, int collapse, const uint32_t* cbase, uint32_t* newidx) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nn; i += gridDim.x * blockDim.x) {
        if (i == 0) newidx[0] = 0u;
        if (!keep[i]) continue;
        uint32_t sl[4], rank = 0;
        const uint32_t ns = gather_slots(i, left, right, live, nbox, dk, 1, collapse, sl);
        for (uint32_t k = 0; k < ns; k++)
            if (!(sl[k] & 0x80000000u) && live[sl[k]]) newidx[sl[k]] = 1u + cbase[i] + rank++;
    }
}

// one thread per surviving node: gather its 2 (binary) or 2..4 (wide: internal children are absorbed) child slots
__global__ void k_emit_nodes(const uint32_t* left, const uint32_t* right, const uint32_t* range_lo, const uint32_t* range_cnt,
                             const uint32_t* keep, const uint32_t* newidx, const float* lmin, const float* lmax, const float* nbox,
                             uint32_t nn, const uint32_t* live, const uint32_t* dk, int wide, int quant, int collapse, float4* nodes, const uint32_t* cbase, const uint32_t* tbase,
                             const float4* tris_morton, float4* tris_out) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nn; i += gridDim.x * blockDim.x) {
        if (!keep[i]) continue;
        uint32_t sl4[4];
        const uint32_t ns4 = gather_slots(i, left, right, live, nbox, dk, wide, collapse, sl4);
        const uint32_t s0 = sl4[0], s1 = sl4[1], s2 = sl4[2], s3 = sl4[3];
        const bool v2 = ns4 > 2, v3 = ns4 > 3;
        const uint32_t o = newidx[i];
        float mn[3], mx[3];
        uint32_t ref;
        if (wide && quant) {
            const uint32_t sl[4] = {s0, s1, s2, s3};
            const uint32_t ns = 2u + (v2 ? 1u : 0u) + (v3 ? 1u : 0u);
            float qmn[4][3], qmx[4][3];
            uint32_t qref[4] = {0, 0, 0, 0};
            for (uint32_t k = 0; k < ns; k++) slot_of(sl[k], lmin, lmax, nbox, range_lo, range_cnt, newidx, live, qmn[k], qmx[k], qref[k]);
            if (quant == 2) {
                uint32_t meta[4] = {7u, 7u, 7u, 7u}, tcur = tbase[i];
                for (uint32_t k = 0; k < ns; k++) {
                    if (qref[k] & 0x80000000u) {  // move the leaf's triangles to their place behind tri_base
                        const uint32_t first = qref[k] & 0x0FFFFFFFu, cnt = ((qref[k] >> 28) & 7u) + 1u;
                        meta[k] = 8u | (cnt - 1u);
                        for (uint32_t t = 0; t < 3u * cnt; t++) tris_out[3 * (size_t)tcur + t] = tris_morton[3 * (size_t)first + t];
                        tcur += cnt;
                    } else {
                        meta[k] = 0u;
                    }
                }
                compact_node(qmn, qmx, meta, ns, 1u + cbase[i], tbase[i], nodes + kC48Stride * (size_t)o);
            } else {
                quantize_node(qmn, qmx, qref, ns, nodes + 4 * (size_t)o);
            }
        } else if (wide) {
            const float inf = INFINITY;
            const uint32_t sl[4] = {s0, s1, s2, s3};
            const bool vl[4] = {true, true, v2, v3};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (vl[k]) {
                    slot_of(sl[k], lmin, lmax, nbox, range_lo, range_cnt, newidx, live, mn, mx, ref);
                } else {
                    mn[0] = mn[1] = mn[2] = inf;
                    mx[0] = mx[1] = mx[2] = -inf;
                    ref = 0xFFFFFFFFu;
                }
                nodes[8 * (size_t)o + 2 * k] = make_float4(mn[0], mn[1], mn[2], mx[0]);
                nodes[8 * (size_t)o + 2 * k + 1] = make_float4(mx[1], mx[2], __uint_as_float(ref), 0.0f);
            }
        } else {
            float mn1[3], mx1[3];
            uint32_t ref1;
            slot_of(s0, lmin, lmax, nbox, range_lo, range_cnt, newidx, live, mn, mx, ref);
            slot_of(s1, lmin, lmax, nbox, range_lo, range_cnt, newidx, live, mn1, mx1, ref1);
            nodes[4 * (size_t)o + 0] = make_float4(mn[0], mn[1], mn[2], mx[0]);
            nodes[4 * (size_t)o + 1] = make_float4(mx[1], mx[2], mn1[0], mn1[1]);
            nodes[4 * (size_t)o + 2] = make_float4(mn1[2], mx1[0], mx1[1], mx1[2]);
            nodes[4 * (size_t)o + 3] = make_float4(__uint_as_float(ref), __uint_as_float(ref1), 0.0f, 0.0f);
        }
    }
}

// Top-of-tree cache for the traversal kernels (quantised 64-byte four-wide layout): the first up-to-kTopCacheNodes nodes in breadth-first
// order, copied verbatim except that a reference to a child that is itself in the cache becomes 0x40000000 | slot.  Slot 0 = the root.
// The canonical node array is left alone (it is what the parity tests compare with the oracle); the cache is a pure acceleration of
// the GPU walk and changes neither hits nor per-ray visit counts.  One thread: 128 nodes.
__global__ void k_top_cache(const float4* __restrict__ nodes, uint32_t n_nodes, uint32_t* __restrict__ top_words /* 16 per slot */, uint32_t* __restrict__ n_top_out) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    uint32_t queue[kTopCacheNodes];
    uint32_t head = 0, tail = 0;
    if (n_nodes) queue[tail++] = 0u;
    while (head < tail) {
        const uint32_t node = queue[head];
        const uint32_t* src = reinterpret_cast<const uint32_t*>(nodes + 4 * (size_t)node);
        uint32_t* dst = top_words + 16 * (size_t)head;
        for (int k = 0; k < 16; k++) dst[k] = src[k];
        for (int k = 0; k < 4; k++) {
            const uint32_t ref = src[10 + k];
            if (ref == 0xFFFFFFFFu || (ref & 0x80000000u)) continue;  // empty slot / leaf
            if (tail < kTopCacheNodes) {
                dst[10 + k] = 0x40000000u | tail;
                queue[tail++] = ref;
            }
        }
        head++;
    }
    *n_top_out = tail;
}

// (re)make the LDS top-of-tree copy for a node array in the quantised 64-byte layout (rt3_accel_import)
hipError_t lbvh_make_top(hipStream_t st, const float4* nodes, uint32_t n_nodes, float4** top, uint32_t* n_top) {
    *n_top = 0;
    if (!*top) {
        hipError_t e = hipMalloc(top, (size_t)kTopCacheNodes * 64);
        if (e != hipSuccess) return e;
    }
    uint32_t* d_ntop = nullptr;
    hipError_t e = hipMalloc(&d_ntop, 4);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_top_cache, dim3(1), dim3(1), 0, st, nodes, n_nodes, (uint32_t*)*top, d_ntop);
    e = hipMemcpyAsync(n_top, d_ntop, 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d_ntop);
    return e;
}

// single-triangle scene: root with the leaf in slot 0 and empty other slots
__global__ void k_single(const float* lmin, const float* lmax, int wide, int quant, float4* nodes) {
    const float inf = INFINITY;
    if (wide && quant) {
        float qmn[4][3], qmx[4][3];
        uint32_t qref[4] = {0x80000000u, 0, 0, 0};
        for (int j = 0; j < 3; j++) {
            qmn[0][j] = lmin[j];
            qmx[0][j] = lmax[j];
        }
        const uint32_t qmeta[4] = {8u, 7u, 7u, 7u};
        if (quant == 2) compact_node(qmn, qmx, qmeta, 1, 1u, 0u, nodes);
        else quantize_node(qmn, qmx, qref, 1, nodes);
    } else if (wide) {
        nodes[0] = make_float4(lmin[0], lmin[1], lmin[2], lmax[0]);
        nodes[1] = make_float4(lmax[1], lmax[2], __uint_as_float(0x80000000u), 0.0f);
        for (int k = 1; k < 4; k++) {
            nodes[2 * k] = make_float4(inf, inf, inf, -inf);
            nodes[2 * k + 1] = make_float4(-inf, -inf, __uint_as_float(0xFFFFFFFFu), 0.0f);
        }
    } else {
        nodes[0] = make_float4(lmin[0], lmin[1], lmin[2], lmax[0]);
        nodes[1] = make_float4(lmax[1], lmax[2], inf, inf);
        nodes[2] = make_float4(inf, -inf, -inf, -inf);
        nodes[3] = make_float4(__uint_as_float(0x80000000u), __uint_as_float(0xFFFFFFFFu), 0.0f, 0.0f);
    }
}

// ---- SAH top (hierarchical LBVH, after Pantaleoni & Luebke 2010 / Garanzha et al. 2011), on the host.  The Karras tree is kept
// below "cluster roots" (maximal subtrees of at most T triangles: contiguous Morton ranges); the C - 1 nodes above the C cluster
// roots are re-linked into a tree built top-down by binned SAH (16 bins on the cluster centroids, cost = half area x triangle
// count) over the cluster boxes.  Node indices are reused (the top of a binary tree with C leaves has C - 1 nodes), the root
// stays node 0.  fp32 in a fixed order, the oracle runs the same algorithm.  A re-linked node is never a multi-triangle leaf
// (its triangles are not contiguous), so its count is kept above T.
namespace {
struct SahCluster {
    uint32_t ref, cnt;
    float mn[3], mx[3];
};
inline float half_area3(const float* mn, const float* mx) {
    const float ex = mx[0] - mn[0], ey = mx[1] - mn[1], ez = mx[2] - mn[2];
    return (ex * ey + ey * ez) + ez * ex;
}
inline float fminh(float a, float b) { return a < b ? a : b; }
inline float fmaxh(float a, float b) { return a > b ? a : b; }
// left / right / rcnt: the Karras tree (modified in place); parents are rewritten for every re-linked edge
bool sah_top_relink(uint32_t nn, uint32_t* left, uint32_t* right, uint32_t* rcnt, uint32_t* pint, uint32_t* pleaf, const float* lmin, const float* lmax,
                    const float* nbox, uint32_t T) {
    std::vector<SahCluster> cl;
    std::vector<uint32_t> pool;
    for (uint32_t i = 0; i < nn; i++) {
        if (!(i == 0 || rcnt[i] > T)) continue;
        pool.push_back(i);
        const uint32_t c2[2] = {left[i], right[i]};
        for (int c = 0; c < 2; c++) {
            const uint32_t ch = c2[c];
            if (!(ch & 0x80000000u) && rcnt[ch] > T) continue;  // another top node
            SahCluster k;
            k.ref = ch;
            if (ch & 0x80000000u) {
                const uint32_t q = ch & 0x7FFFFFFFu;
                k.cnt = 1;
                for (int a = 0; a < 3; a++) { k.mn[a] = lmin[3 * (size_t)q + a]; k.mx[a] = lmax[3 * (size_t)q + a]; }
            } else {
                k.cnt = rcnt[ch];
                for (int a = 0; a < 3; a++) { k.mn[a] = nbox[6 * (size_t)ch + a]; k.mx[a] = nbox[6 * (size_t)ch + 3 + a]; }
            }
            cl.push_back(k);
        }
    }
    const uint32_t nc = (uint32_t)cl.size();
    if (nc < 3 || pool.size() != nc - 1) return false;
    std::vector<uint32_t> idx(nc), tmp(nc);
    for (uint32_t i = 0; i < nc; i++) idx[i] = i;
    // A subtree over n clusters takes exactly n - 1 pool nodes, numbered in pre-order: the node of a job is pool[job.pool], its
    // left subtree (nl clusters) owns pool[job.pool + 1 ..], its right subtree pool[job.pool + nl ..].  Jobs therefore touch disjoint
    // ranges of idx / tmp / pool and disjoint tree nodes whatever order they run in, which lets the big right subtrees near the
    // root run on their own threads with a result that does not depend on scheduling.
    struct Job { uint32_t a, n, pool, patch; };
    const float inf = INFINITY;
    auto patch_parent = [&](uint32_t patch, uint32_t ref) {
        if (patch == 0xFFFFFFFFu) return;
        const uint32_t parent = patch >> 1;
        if (patch & 1u) right[parent] = ref;
        else left[parent] = ref;
        if (ref & 0x80000000u) pleaf[ref & 0x7FFFFFFFu] = parent;
        else pint[ref] = parent;
    };
    // splits one job; returns the number of clusters that go left
    auto split = [&](const Job& j, uint32_t node) -> uint32_t {
        float cmn[3] = {inf, inf, inf}, cmx[3] = {-inf, -inf, -inf};
        uint32_t total = 0;
        for (uint32_t k = 0; k < j.n; k++) {
            const SahCluster& c = cl[idx[j.a + k]];
            total += c.cnt;
            for (int a = 0; a < 3; a++) {
                const float ce = (c.mn[a] + c.mx[a]) * 0.5f;
                cmn[a] = fminh(cmn[a], ce);
                cmx[a] = fmaxh(cmx[a], ce);
            }
        }
        float best_cost = inf;
        int best_axis = -1, best_split = 0;
        for (int a = 0; a < 3; a++) {
            const float ext = cmx[a] - cmn[a];
            if (!(ext > 0.0f)) continue;
            float bmn[16][3], bmx[16][3];
            uint32_t bc[16];
            for (int b = 0; b < 16; b++) {
                bc[b] = 0;
                for (int q = 0; q < 3; q++) { bmn[b][q] = inf; bmx[b][q] = -inf; }
            }
            for (uint32_t k = 0; k < j.n; k++) {
                const SahCluster& c = cl[idx[j.a + k]];
                const float ce = (c.mn[a] + c.mx[a]) * 0.5f;
                int b = (int)(((ce - cmn[a]) / ext) * 16.0f);
                if (b > 15) b = 15;
                bc[b] += c.cnt;
                for (int q = 0; q < 3; q++) { bmn[b][q] = fminh(bmn[b][q], c.mn[q]); bmx[b][q] = fmaxh(bmx[b][q], c.mx[q]); }
            }
            float rmn[16][3], rmx[16][3];
            uint32_t rc[16];
            for (int b = 15; b >= 0; b--) {
                for (int q = 0; q < 3; q++) {
                    rmn[b][q] = b == 15 ? bmn[b][q] : fminh(bmn[b][q], rmn[b + 1][q]);
                    rmx[b][q] = b == 15 ? bmx[b][q] : fmaxh(bmx[b][q], rmx[b + 1][q]);
                }
                rc[b] = bc[b] + (b == 15 ? 0u : rc[b + 1]);
            }
            float lmn[3] = {inf, inf, inf}, lmx[3] = {-inf, -inf, -inf};
            uint32_t lc = 0;
            for (int sp = 1; sp < 16; sp++) {
                for (int q = 0; q < 3; q++) { lmn[q] = fminh(lmn[q], bmn[sp - 1][q]); lmx[q] = fmaxh(lmx[q], bmx[sp - 1][q]); }
                lc += bc[sp - 1];
                if (lc == 0 || rc[sp] == 0) continue;
                const float cost = half_area3(lmn, lmx) * (float)lc + half_area3(rmn[sp], rmx[sp]) * (float)rc[sp];
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_split = sp; }
            }
        }
        uint32_t nl = 0;
        if (best_axis < 0) {
            nl = j.n / 2;  // coincident centroids: halve in index order
        } else {           // stable partition by bin
            const float ext = cmx[best_axis] - cmn[best_axis];
            uint32_t w = 0, r = 0;
            for (uint32_t k = 0; k < j.n; k++) {
                const SahCluster& c = cl[idx[j.a + k]];
                const float ce = (c.mn[best_axis] + c.mx[best_axis]) * 0.5f;
                int b = (int)(((ce - cmn[best_axis]) / ext) * 16.0f);
                if (b > 15) b = 15;
                if (b < best_split) idx[j.a + w++] = idx[j.a + k];
                else tmp[j.a + r++] = idx[j.a + k];
            }
            for (uint32_t k = 0; k < r; k++) idx[j.a + w + k] = tmp[j.a + k];
            nl = w;
        }
        rcnt[node] = total > T ? total : T + 1u;
        return nl;
    };
    // one thread's share: an explicit stack; subtrees of at least `fork_min` clusters are handed to `spawn` instead (if given)
    std::function<void(Job, const std::function<bool(const Job&)>&)> run = [&](Job root, const std::function<bool(const Job&)>& spawn) {
        std::vector<Job> st;
        st.push_back(root);
        while (!st.empty()) {
            const Job j = st.back();
            st.pop_back();
            if (j.n == 1) {
                patch_parent(j.patch, cl[idx[j.a]].ref);
                continue;
            }
            const uint32_t node = pool[j.pool];
            const uint32_t nl = split(j, node);
            patch_parent(j.patch, node);
            const Job rj{j.a + nl, j.n - nl, j.pool + nl, (node << 1) | 1u}, lj{j.a, nl, j.pool + 1, node << 1};
            if (!(spawn && spawn(rj))) st.push_back(rj);
            st.push_back(lj);
        }
    };
    const unsigned hw = std::thread::hardware_concurrency();
    const unsigned max_threads = hw > 16 ? 16 : (hw ? hw : 1);
    const uint32_t fork_min = nc / (4 * max_threads) > 2048 ? nc / (4 * max_threads) : 2048;
    std::mutex mu;
    std::vector<std::thread> threads;
    std::function<bool(const Job&)> spawn = [&](const Job& j) -> bool {
        if (j.n < fork_min) return false;
        std::lock_guard<std::mutex> g(mu);
        if (threads.size() + 1 >= max_threads * 4) return false;  // bounded: at most a few dozen short-lived threads per build
        threads.emplace_back([&run, &spawn, j] { run(j, spawn); });
        return true;
    };
    run(Job{0, nc, 0, 0xFFFFFFFFu}, max_threads > 1 ? spawn : std::function<bool(const Job&)>());
    for (size_t t = 0;; t++) {  // threads may still be spawning threads: join until the list stops growing
        std::thread th;
        {
            std::lock_guard<std::mutex> g(mu);
            if (t >= threads.size()) break;
            th = std::move(threads[t]);
        }
        th.join();
    }
    pint[0] = 0xFFFFFFFFu;
    return true;
}
}  // namespace

// ---- SAH top on the GPU (RT3_OPT_SAH_TOP_DEVICE, default): the same algorithm as sah_top_relink above, same fp32 expressions in the
// same order, so the tree is bit-identical to the host's and the oracle's -- every reduction in it is a min, a max or an integer
// sum (exact, order independent), and the partitions are stable.  No bulk D2H / H2D copies: the Karras arrays are re-linked in place.
//   k_sah_mark / scans / k_sah_gather   top nodes -> pool[] (ascending), their cluster children -> cl_*[] (node order, left first)
//   k_sahh_*     segments of more than kSahHuge clusters, tiled over several workgroups: one launch per phase and level
//   k_sah_block  one workgroup per segment of kSahSmall .. kSahHuge clusters, one launch per level: centroid bounds and 3 x 16 bins by
//                LDS atomics on order-preserving uints, the split sweep by thread 0, a stable in-place partition in 256-wide chunks
//   k_sah_small  one thread per remaining segment (<= kSahSmall clusters) running the sequential algorithm with its own stack
// A subtree over n clusters owns n - 1 pool nodes in pre-order (as on the host), so segments touch disjoint ranges.
namespace {
constexpr uint32_t kSahSmall = 16;    // segments of at most this many clusters are finished by one thread
constexpr uint32_t kSahHuge = 4096;   // segments of more clusters are tiled over several workgroups (k_sahh_*), the others get one of 256 threads
struct SahSeg { uint32_t a, n, pool, patch; };
struct SahArrays {
    uint32_t *left, *right, *rcnt, *pint, *pleaf;   // the Karras tree, re-linked in place
    const uint32_t *cl_ref, *cl_cnt;                // clusters
    const float *cl_mn, *cl_mx;                     // 3 floats each
    const uint32_t* pool;
    uint32_t *idx, *tmp;
    uint32_t T;
    float* nbox;  // every re-linked node's box (the union of its clusters' boxes) is written by the kernel that splits it: no second refit
};
__device__ __forceinline__ void sah_store_box(const SahArrays& A, uint32_t node, const float* mn, const float* mx) {
    for (int q = 0; q < 3; q++) {
        A.nbox[6 * (size_t)node + q] = mn[q];
        A.nbox[6 * (size_t)node + 3 + q] = mx[q];
    }
}
__device__ __forceinline__ float sah_half_area(const float* mn, const float* mx) {
    const float ex = mx[0] - mn[0], ey = mx[1] - mn[1], ez = mx[2] - mn[2];
    return (ex * ey + ey * ez) + ez * ex;
}
__device__ __forceinline__ void sah_patch_parent(const SahArrays& A, uint32_t patch, uint32_t ref) {
    if (patch == 0xFFFFFFFFu) return;
    const uint32_t parent = patch >> 1;
    if (patch & 1u) A.right[parent] = ref;
    else A.left[parent] = ref;
    if (ref & 0x80000000u) A.pleaf[ref & 0x7FFFFFFFu] = parent;
    else A.pint[ref] = parent;
}
// the sweep over the 15 split planes of one axis (sah_top_relink::split, inner part); bins hold decoded floats
__device__ __forceinline__ void sah_sweep_axis(int axis, const float (*bmn)[3], const float (*bmx)[3], const uint32_t* bc, float& best_cost, int& best_axis,
                                               int& best_split) {
    const float inf = INFINITY;
    float rmn[16][3], rmx[16][3];
    uint32_t rc[16];
    for (int b = 15; b >= 0; b--) {
        for (int q = 0; q < 3; q++) {
            rmn[b][q] = b == 15 ? bmn[b][q] : fmin_sel(bmn[b][q], rmn[b + 1][q]);
            rmx[b][q] = b == 15 ? bmx[b][q] : fmax_sel(bmx[b][q], rmx[b + 1][q]);
        }
        rc[b] = bc[b] + (b == 15 ? 0u : rc[b + 1]);
    }
    float lmn[3] = {inf, inf, inf}, lmx[3] = {-inf, -inf, -inf};
    uint32_t lc = 0;
    for (int sp = 1; sp < 16; sp++) {
        for (int q = 0; q < 3; q++) {
            lmn[q] = fmin_sel(lmn[q], bmn[sp - 1][q]);
            lmx[q] = fmax_sel(lmx[q], bmx[sp - 1][q]);
        }
        lc += bc[sp - 1];
        if (lc == 0 || rc[sp] == 0) continue;
        const float cost = sah_half_area(lmn, lmx) * (float)lc + sah_half_area(rmn[sp], rmx[sp]) * (float)rc[sp];
        if (cost < best_cost) {
            best_cost = cost;
            best_axis = axis;
            best_split = sp;
        }
    }
}
__device__ __forceinline__ int sah_bin(float ce, float cmn, float ext) {
    int b = (int)(((ce - cmn) / ext) * 16.0f);
    return b > 15 ? 15 : b;
}

__global__ void k_sah_mark(const uint32_t* left, const uint32_t* right, const uint32_t* rcnt, uint32_t nn, uint32_t T, uint32_t* top, uint32_t* ncl) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nn; i += gridDim.x * blockDim.x) {
        const bool t = i == 0 || rcnt[i] > T;
        uint32_t c = 0;
        if (t) {
            const uint32_t c2[2] = {left[i], right[i]};
            for (int k = 0; k < 2; k++)
                if ((c2[k] & 0x80000000u) || rcnt[c2[k]] <= T) c++;
        }
        top[i] = t ? 1u : 0u;
        ncl[i] = c;
    }
}
__global__ void k_sah_gather(const uint32_t* left, const uint32_t* right, const uint32_t* rcnt, uint32_t nn, uint32_t T, const uint32_t* top, const uint32_t* pool_pos,
                             const uint32_t* cl_pos, const float* lmin, const float* lmax, const float* nbox, uint32_t* pool, uint32_t* cl_ref, uint32_t* cl_cnt,
                             float* cl_mn, float* cl_mx, uint32_t* idx) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nn; i += gridDim.x * blockDim.x) {
        if (!top[i]) continue;
        pool[pool_pos[i]] = i;
        uint32_t k = cl_pos[i];
        const uint32_t c2[2] = {left[i], right[i]};
        for (int c = 0; c < 2; c++) {
            const uint32_t ch = c2[c];
            if (!(ch & 0x80000000u) && rcnt[ch] > T) continue;  // another top node
            cl_ref[k] = ch;
            if (ch & 0x80000000u) {
                const uint32_t q = ch & 0x7FFFFFFFu;
                cl_cnt[k] = 1;
                for (int a = 0; a < 3; a++) {
                    cl_mn[3 * (size_t)k + a] = lmin[3 * (size_t)q + a];
                    cl_mx[3 * (size_t)k + a] = lmax[3 * (size_t)q + a];
                }
            } else {
                cl_cnt[k] = rcnt[ch];
                for (int a = 0; a < 3; a++) {
                    cl_mn[3 * (size_t)k + a] = nbox[6 * (size_t)ch + a];
                    cl_mx[3 * (size_t)k + a] = nbox[6 * (size_t)ch + 3 + a];
                }
            }
            idx[k] = k;
            k++;
        }
    }
}

// children of a split segment: single clusters are linked at once, the others queued by size (huge / big -> next level, small -> k_sah_small)
struct SahQueues {
    SahSeg *huge, *big, *small;
    uint32_t* counts;   // [0] huge, [1] big segments of the next level
    uint32_t* n_small;
};
__device__ __forceinline__ void sah_emit_child(const SahArrays& A, SahSeg c, const SahQueues& Q) {
    if (c.n == 1) sah_patch_parent(A, c.patch, A.cl_ref[A.idx[c.a]]);
    else if (c.n > kSahHuge) Q.huge[atomicAdd(&Q.counts[0], 1u)] = c;
    else if (c.n > kSahSmall) Q.big[atomicAdd(&Q.counts[1], 1u)] = c;
    else Q.small[atomicAdd(Q.n_small, 1u)] = c;
}
// Reductions over a FULL wave (all 64 lanes active -- every caller iterates whole waves): four DPP steps inside each row of 16 lanes
// (quad xor 1, quad xor 2, half-row mirror, row mirror), then the four row results through v_readlane.  ~15 instructions; the
// __shfl_xor butterfly these replace is six dependent ds_bpermute round trips through the LDS crossbar (~800 cycles).
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppMirror = 0x140;
__device__ __forceinline__ float lane_f(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ __forceinline__ float wave_min_f(float v) {
    v = fmin_sel(v, dpp_f<kDppXor1>(v));
    v = fmin_sel(v, dpp_f<kDppXor2>(v));
    v = fmin_sel(v, dpp_f<kDppHalfMirror>(v));
    v = fmin_sel(v, dpp_f<kDppMirror>(v));
    return fmin_sel(fmin_sel(lane_f(v, 0), lane_f(v, 16)), fmin_sel(lane_f(v, 32), lane_f(v, 48)));
}
__device__ __forceinline__ float wave_max_f(float v) {
    v = fmax_sel(v, dpp_f<kDppXor1>(v));
    v = fmax_sel(v, dpp_f<kDppXor2>(v));
    v = fmax_sel(v, dpp_f<kDppHalfMirror>(v));
    v = fmax_sel(v, dpp_f<kDppMirror>(v));
    return fmax_sel(fmax_sel(lane_f(v, 0), lane_f(v, 16)), fmax_sel(lane_f(v, 32), lane_f(v, 48)));
}
__device__ __forceinline__ uint32_t wave_sum_u(uint32_t v) {
    v += dpp_u<kDppXor1>(v);
    v += dpp_u<kDppXor2>(v);
    v += dpp_u<kDppHalfMirror>(v);
    v += dpp_u<kDppMirror>(v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) + (uint32_t)__builtin_amdgcn_readlane((int)v, 16) + (uint32_t)__builtin_amdgcn_readlane((int)v, 32) +
           (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
}

// One workgroup per segment.  Contention-free by construction: centroid bounds are reduced in registers and then across the wave;
// bins are reduced across the lanes of a wave that share a bin (clusters are in Morton sub-order, a wave's 64 consecutive ones
// fall into one to three bins) and only the wave's leader lane touches the LDS counters.  min / max / integer sums: exact in any order.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_sah_block(SahArrays A, const SahSeg* segs, const uint32_t* n_segs, SahQueues Q) {
    __shared__ uint32_t s_cmn[3], s_cmx[3], s_total, s_amn[3], s_amx[3];
    __shared__ uint32_t s_bmn[3][16][3], s_bmx[3][16][3], s_bc[3][16];
    __shared__ int s_axis, s_split;
    __shared__ float s_cost[3][16];
    // the segment's cluster ids and their three bin numbers, staged once: the partition then needs no global read at all and places
    // every element directly (ranks from one scan over per-wave counts) -- it was 3 barriers and two dependent loads per 256 elements
    constexpr int NW = BLOCK / 64, NE = (int)kSahHuge / BLOCK;
    static_assert(NW * NE == 64, "one wave scans the per-(chunk, wave) counts");
    __shared__ uint32_t s_c[kSahHuge];
    __shared__ uint16_t s_bins[kSahHuge];
    __shared__ uint32_t s_offl[64], s_offr[64], s_nl;
    if (blockIdx.x >= *n_segs) return;  // the grid is sized for the most segments a level can have: no host round trip per level
    const SahSeg j = segs[blockIdx.x];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t node = A.pool[j.pool];
    const float inf = INFINITY;
    if (tid < 3) {
        s_cmn[tid] = s_amn[tid] = float_to_ordered(inf);
        s_cmx[tid] = s_amx[tid] = float_to_ordered(-inf);
    }
    if (tid == 0) s_total = 0;
    for (uint32_t k = tid; k < 3 * 16 * 3; k += BLOCK) {
        (&s_bmn[0][0][0])[k] = float_to_ordered(inf);
        (&s_bmx[0][0][0])[k] = float_to_ordered(-inf);
    }
    for (uint32_t k = tid; k < 3 * 16; k += BLOCK) (&s_bc[0][0])[k] = 0;
    __syncthreads();
    {  // centroid bounds, triangle total
        float tmn[3] = {inf, inf, inf}, tmx[3] = {-inf, -inf, -inf}, amn[3] = {inf, inf, inf}, amx[3] = {-inf, -inf, -inf};
        uint32_t tc = 0;
        for (uint32_t k0 = tid; k0 < j.n; k0 += 4 * BLOCK) {  // four independent elements per trip: this loop lives off loads in flight
            uint32_t c[4], cc[4];
            float mn[4][3], mx[4][3];
#pragma unroll
            for (int e = 0; e < 4; e++) c[e] = k0 + e * BLOCK < j.n ? A.idx[j.a + k0 + e * BLOCK] : 0xFFFFFFFFu;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const bool ok = c[e] != 0xFFFFFFFFu;
                if (ok) s_c[k0 + e * BLOCK] = c[e];
                cc[e] = ok ? A.cl_cnt[c[e]] : 0u;
                for (int a = 0; a < 3; a++) {
                    mn[e][a] = ok ? A.cl_mn[3 * (size_t)c[e] + a] : 0.0f;
                    mx[e][a] = ok ? A.cl_mx[3 * (size_t)c[e] + a] : 0.0f;
                }
            }
#pragma unroll
            for (int e = 0; e < 4; e++) {
                if (c[e] == 0xFFFFFFFFu) continue;
                tc += cc[e];
                for (int a = 0; a < 3; a++) {
                    const float ce = (mn[e][a] + mx[e][a]) * 0.5f;
                    tmn[a] = fmin_sel(tmn[a], ce);
                    tmx[a] = fmax_sel(tmx[a], ce);
                    amn[a] = fmin_sel(amn[a], mn[e][a]);
                    amx[a] = fmax_sel(amx[a], mx[e][a]);
                }
            }
        }
        tc = wave_sum_u(tc);
        for (int a = 0; a < 3; a++) {
            tmn[a] = wave_min_f(tmn[a]);
            tmx[a] = wave_max_f(tmx[a]);
            amn[a] = wave_min_f(amn[a]);
            amx[a] = wave_max_f(amx[a]);
        }
        if (lane == 0) {
            atomicAdd(&s_total, tc);
            for (int a = 0; a < 3; a++) {
                atomicMin(&s_cmn[a], float_to_ordered(tmn[a]));
                atomicMax(&s_cmx[a], float_to_ordered(tmx[a]));
                atomicMin(&s_amn[a], float_to_ordered(amn[a]));
                atomicMax(&s_amx[a], float_to_ordered(amx[a]));
            }
        }
    }
    __syncthreads();
    float cmn[3], ext[3];
    for (int a = 0; a < 3; a++) {
        cmn[a] = ordered_to_float(s_cmn[a]);
        ext[a] = ordered_to_float(s_cmx[a]) - cmn[a];
    }
    const uint32_t n_round = (j.n + 63u) & ~63u;  // whole waves iterate: cross-lane reductions inside
    for (uint32_t k0 = tid; k0 < n_round; k0 += 4 * BLOCK) {  // 3 x 16 bins; the loads of four elements are issued together
        uint32_t ce4[4], cnt4[4];
        float mn4[4][3], mx4[4][3];
#pragma unroll
        for (int e = 0; e < 4; e++) ce4[e] = k0 + e * BLOCK < j.n ? s_c[k0 + e * BLOCK] : 0xFFFFFFFFu;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const bool ok = ce4[e] != 0xFFFFFFFFu;
            cnt4[e] = ok ? A.cl_cnt[ce4[e]] : 0u;
            for (int q = 0; q < 3; q++) {
                mn4[e][q] = ok ? A.cl_mn[3 * (size_t)ce4[e] + q] : inf;
                mx4[e][q] = ok ? A.cl_mx[3 * (size_t)ce4[e] + q] : -inf;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; e++) {
            if (k0 - tid + e * BLOCK >= n_round) break;  // (uniform over the wave: n_round and the wave's base are multiples of 64)
            const bool valid = ce4[e] != 0xFFFFFFFFu;
            const uint32_t cnt = cnt4[e];
            const float* mn = mn4[e];
            const float* mx = mx4[e];
            uint32_t packed = 0;
            for (int a = 0; a < 3; a++) {
                if (!(ext[a] > 0.0f)) continue;  // (uniform over the workgroup)
                const int b = valid ? sah_bin((mn[a] + mx[a]) * 0.5f, cmn[a], ext[a]) : -1;
                packed |= valid ? (uint32_t)b << (4 * a) : 0u;
                // a wave's 64 consecutive clusters (Morton sub-order) share one bin in the big segments near the root: one reduction
                // across the wave and seven LDS atomics by its first lane.  Otherwise every lane adds its own cluster: lanes of one
                // bin serialise inside the atomic, the others proceed in parallel (a reduction round per distinct bin cost more)
                const unsigned long long vmask = __ballot(valid);
                if (vmask == 0ull) continue;
                const int leader = __ffsll((long long)vmask) - 1;
                const int bb = __builtin_amdgcn_readlane(b, leader);
                if (__ballot(valid && b == bb) == vmask) {
                    const uint32_t sc = wave_sum_u(cnt);  // (idle lanes carry 0 / +inf / -inf)
                    float rmn[3], rmx[3];
                    for (int q = 0; q < 3; q++) {
                        rmn[q] = wave_min_f(mn[q]);
                        rmx[q] = wave_max_f(mx[q]);
                    }
                    if ((int)lane == leader) {
                        atomicAdd(&s_bc[a][bb], sc);
                        for (int q = 0; q < 3; q++) {
                            atomicMin(&s_bmn[a][bb][q], float_to_ordered(rmn[q]));
                            atomicMax(&s_bmx[a][bb][q], float_to_ordered(rmx[q]));
                        }
                    }
                } else if (valid) {
                    atomicAdd(&s_bc[a][b], cnt);
                    for (int q = 0; q < 3; q++) {
                        atomicMin(&s_bmn[a][b][q], float_to_ordered(mn[q]));
                        atomicMax(&s_bmx[a][b][q], float_to_ordered(mx[q]));
                    }
                }
            }
            if (valid) s_bins[k0 + e * BLOCK] = (uint16_t)packed;
        }
    }
    __syncthreads();
    if (tid < 48) {  // the 3 x 15 split planes in parallel: the same left / right boxes, counts and cost expression as the sequential sweep
        const int a = (int)tid >> 4, sp = (int)tid & 15;
        float cost = inf;
        if (sp >= 1 && ext[a] > 0.0f) {
            float lmn[3] = {inf, inf, inf}, lmx[3] = {-inf, -inf, -inf}, rmn[3] = {inf, inf, inf}, rmx[3] = {-inf, -inf, -inf};
            uint32_t lc = 0, rc = 0;
            for (int b = 0; b < sp; b++) {
                lc += s_bc[a][b];
                for (int q = 0; q < 3; q++) {
                    lmn[q] = fmin_sel(lmn[q], ordered_to_float(s_bmn[a][b][q]));
                    lmx[q] = fmax_sel(lmx[q], ordered_to_float(s_bmx[a][b][q]));
                }
            }
            for (int b = 15; b >= sp; b--) {
                rc += s_bc[a][b];
                for (int q = 0; q < 3; q++) {
                    rmn[q] = fmin_sel(rmn[q], ordered_to_float(s_bmn[a][b][q]));
                    rmx[q] = fmax_sel(rmx[q], ordered_to_float(s_bmx[a][b][q]));
                }
            }
            if (lc != 0 && rc != 0) cost = sah_half_area(lmn, lmx) * (float)lc + sah_half_area(rmn, rmx) * (float)rc;
        }
        s_cost[a][sp] = cost;
    }
    __syncthreads();
    if (tid == 0) {  // first minimum in (axis, plane) order, like the sequential sweep's strict `<`
        float best_cost = inf;
        int best_axis = -1, best_split = 0;
        for (int a = 0; a < 3; a++)
            for (int sp = 1; sp < 16; sp++)
                if (s_cost[a][sp] < best_cost) {
                    best_cost = s_cost[a][sp];
                    best_axis = a;
                    best_split = sp;
                }
        s_axis = best_axis;
        s_split = best_split;
    }
    __syncthreads();
    const int axis = s_axis, split = s_split;
    uint32_t nl;
    if (axis < 0) {
        nl = j.n / 2;  // coincident centroids: halve in index order
    } else {
        // stable partition: lefts keep their order in idx[a ..], rights theirs behind them.  The position of element k (chunk e = k / BLOCK,
        // wave w, lane) is the number of lefts (rights) in the (chunk, wave) pairs before (e, w) -- one 64-entry scan -- plus those below
        // its lane in its own ballot.
        const unsigned long long below = (1ull << lane) - 1ull;
        const uint32_t n_chunks = (j.n + BLOCK - 1) / BLOCK;
        for (uint32_t e = 0; e < (uint32_t)NE; e++) {
            uint32_t cl = 0, cr = 0;
            if (e < n_chunks) {
                const uint32_t k = e * BLOCK + tid;
                const bool valid = k < j.n;
                const bool goes_left = valid && (int)((s_bins[valid ? k : 0] >> (4 * axis)) & 15u) < split;
                cl = (uint32_t)__popcll(__ballot(goes_left));
                cr = (uint32_t)__popcll(__ballot(valid && !goes_left));
            }
            if (lane == 0) {
                s_offl[e * NW + wave] = cl;
                s_offr[e * NW + wave] = cr;
            }
        }
        __syncthreads();
        if (wave == 0) {  // exclusive scan of the 64 counts
            const uint32_t vl = s_offl[lane], vr = s_offr[lane];
            uint32_t il = vl, ir = vr;
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) {
                const uint32_t tl = __shfl_up(il, m), tr = __shfl_up(ir, m);
                if ((int)lane >= m) {
                    il += tl;
                    ir += tr;
                }
            }
            s_offl[lane] = il - vl;
            s_offr[lane] = ir - vr;
            if (lane == 63) s_nl = il;
        }
        __syncthreads();
        nl = s_nl;
        for (uint32_t e = 0; e < n_chunks; e++) {
            const uint32_t k = e * BLOCK + tid;
            const bool valid = k < j.n;
            const bool goes_left = valid && (int)((s_bins[valid ? k : 0] >> (4 * axis)) & 15u) < split;
            const unsigned long long ml = __ballot(goes_left), mr = __ballot(valid && !goes_left);
            if (valid) {
                const uint32_t pos = goes_left ? s_offl[e * NW + wave] + (uint32_t)__popcll(ml & below) : nl + s_offr[e * NW + wave] + (uint32_t)__popcll(mr & below);
                A.idx[j.a + pos] = s_c[k];
            }
        }
    }
    if (tid == 0) {
        const uint32_t total = s_total;
        A.rcnt[node] = total > A.T ? total : A.T + 1u;
        const float bmn[3] = {ordered_to_float(s_amn[0]), ordered_to_float(s_amn[1]), ordered_to_float(s_amn[2])};
        const float bmx[3] = {ordered_to_float(s_amx[0]), ordered_to_float(s_amx[1]), ordered_to_float(s_amx[2])};
        sah_store_box(A, node, bmn, bmx);
        sah_patch_parent(A, j.patch, node);
        sah_emit_child(A, SahSeg{j.a, nl, j.pool + 1, node << 1}, Q);
        sah_emit_child(A, SahSeg{j.a + nl, j.n - nl, j.pool + nl, (node << 1) | 1u}, Q);
    }
}

// ---- segments of more than kSahHuge clusters: the same split, tiled over several workgroups (one launch per phase and level).
// A tile = kSahTile consecutive clusters of one segment; per-segment state lives in global memory and is reduced with atomics on
// order-preserving uints (min / max) and integers (sums): exact, order independent.
constexpr uint32_t kSahTile = 2048;
struct SahHuge {
    uint32_t cmn[3], cmx[3], total, amn[3], amx[3];  // centroid bounds, triangles, box of the whole segment
    uint32_t bmn[3][16][3], bmx[3][16][3], bc[3][16];
    int axis, split;
    uint32_t nl, tile_base, ntiles;
};
struct SahTile { uint32_t seg, t; };

__global__ void k_sahh_tiles(const SahSeg* segs, const uint32_t* n_segs, SahHuge* hs, SahTile* tiles, uint32_t* n_tiles) {
    const uint32_t s = blockIdx.x;
    if (s >= *n_segs) return;
    __shared__ uint32_t s_base;
    const SahSeg j = segs[s];
    const uint32_t nt = (j.n + kSahTile - 1) / kSahTile;
    SahHuge& h = hs[s];
    for (uint32_t k = threadIdx.x; k < 3 * 16 * 3; k += blockDim.x) {
        (&h.bmn[0][0][0])[k] = float_to_ordered(INFINITY);
        (&h.bmx[0][0][0])[k] = float_to_ordered(-INFINITY);
    }
    for (uint32_t k = threadIdx.x; k < 3 * 16; k += blockDim.x) (&h.bc[0][0])[k] = 0;
    if (threadIdx.x < 3) {
        h.cmn[threadIdx.x] = h.amn[threadIdx.x] = float_to_ordered(INFINITY);
        h.cmx[threadIdx.x] = h.amx[threadIdx.x] = float_to_ordered(-INFINITY);
    }
    if (threadIdx.x == 0) {
        h.total = 0;
        h.axis = -1;
        h.split = 0;
        h.nl = 0;
        h.ntiles = nt;
        s_base = atomicAdd(n_tiles, nt);
        h.tile_base = s_base;
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < nt; t += blockDim.x) tiles[s_base + t] = SahTile{s, t};
}
__global__ __launch_bounds__(256) void k_sahh_bounds(SahArrays A, const SahSeg* segs, SahHuge* hs, const SahTile* tiles, const uint32_t* n_tiles) {
    if (blockIdx.x >= *n_tiles) return;
    const SahTile tl = tiles[blockIdx.x];
    const SahSeg j = segs[tl.seg];
    const uint32_t lo = tl.t * kSahTile, hi = lo + kSahTile < j.n ? lo + kSahTile : j.n;
    const float inf = INFINITY;
    float tmn[3] = {inf, inf, inf}, tmx[3] = {-inf, -inf, -inf}, amn[3] = {inf, inf, inf}, amx[3] = {-inf, -inf, -inf};
    uint32_t tc = 0;
    for (uint32_t k = lo + threadIdx.x; k < hi; k += 256) {
        const uint32_t c = A.idx[j.a + k];
        tc += A.cl_cnt[c];
        for (int a = 0; a < 3; a++) {
            const float mn = A.cl_mn[3 * (size_t)c + a], mx = A.cl_mx[3 * (size_t)c + a];
            const float ce = (mn + mx) * 0.5f;
            tmn[a] = fmin_sel(tmn[a], ce);
            tmx[a] = fmax_sel(tmx[a], ce);
            amn[a] = fmin_sel(amn[a], mn);
            amx[a] = fmax_sel(amx[a], mx);
        }
    }
    tc = wave_sum_u(tc);
    for (int a = 0; a < 3; a++) {
        tmn[a] = wave_min_f(tmn[a]);
        tmx[a] = wave_max_f(tmx[a]);
        amn[a] = wave_min_f(amn[a]);
        amx[a] = wave_max_f(amx[a]);
    }
    if ((threadIdx.x & 63u) == 0) {
        SahHuge& h = hs[tl.seg];
        atomicAdd(&h.total, tc);
        for (int a = 0; a < 3; a++) {
            atomicMin(&h.cmn[a], float_to_ordered(tmn[a]));
            atomicMax(&h.cmx[a], float_to_ordered(tmx[a]));
            atomicMin(&h.amn[a], float_to_ordered(amn[a]));
            atomicMax(&h.amx[a], float_to_ordered(amx[a]));
        }
    }
}
__global__ __launch_bounds__(256) void k_sahh_bins(SahArrays A, const SahSeg* segs, SahHuge* hs, const SahTile* tiles, const uint32_t* n_tiles) {
    if (blockIdx.x >= *n_tiles) return;
    __shared__ uint32_t s_bmn[3][16][3], s_bmx[3][16][3], s_bc[3][16];
    const SahTile tl = tiles[blockIdx.x];
    const SahSeg j = segs[tl.seg];
    SahHuge& h = hs[tl.seg];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const float inf = INFINITY;
    for (uint32_t k = tid; k < 3 * 16 * 3; k += 256) {
        (&s_bmn[0][0][0])[k] = float_to_ordered(inf);
        (&s_bmx[0][0][0])[k] = float_to_ordered(-inf);
    }
    for (uint32_t k = tid; k < 3 * 16; k += 256) (&s_bc[0][0])[k] = 0;
    __syncthreads();
    float cmn[3], ext[3];
    for (int a = 0; a < 3; a++) {
        cmn[a] = ordered_to_float(h.cmn[a]);
        ext[a] = ordered_to_float(h.cmx[a]) - cmn[a];
    }
    const uint32_t lo = tl.t * kSahTile, hi = lo + kSahTile < j.n ? lo + kSahTile : j.n;
    const uint32_t hi_round = lo + ((hi - lo + 63u) & ~63u);  // whole waves iterate: cross-lane reductions inside
    for (uint32_t k = lo + tid; k < hi_round; k += 256) {
        const bool valid = k < hi;
        uint32_t cnt = 0;
        float mn[3] = {inf, inf, inf}, mx[3] = {-inf, -inf, -inf};
        if (valid) {
            const uint32_t c = A.idx[j.a + k];
            cnt = A.cl_cnt[c];
            for (int q = 0; q < 3; q++) {
                mn[q] = A.cl_mn[3 * (size_t)c + q];
                mx[q] = A.cl_mx[3 * (size_t)c + q];
            }
        }
        for (int a = 0; a < 3; a++) {
            if (!(ext[a] > 0.0f)) continue;
            const int b = valid ? sah_bin((mn[a] + mx[a]) * 0.5f, cmn[a], ext[a]) : -1;
            const unsigned long long vmask = __ballot(valid);  // (k_sah_block has the argument)
            if (vmask == 0ull) continue;
            const int leader = __ffsll((long long)vmask) - 1;
            const int bb = __builtin_amdgcn_readlane(b, leader);
            if (__ballot(valid && b == bb) == vmask) {
                const uint32_t sc = wave_sum_u(cnt);
                float rmn[3], rmx[3];
                for (int q = 0; q < 3; q++) {
                    rmn[q] = wave_min_f(mn[q]);
                    rmx[q] = wave_max_f(mx[q]);
                }
                if ((int)lane == leader) {
                    atomicAdd(&s_bc[a][bb], sc);
                    for (int q = 0; q < 3; q++) {
                        atomicMin(&s_bmn[a][bb][q], float_to_ordered(rmn[q]));
                        atomicMax(&s_bmx[a][bb][q], float_to_ordered(rmx[q]));
                    }
                }
            } else if (valid) {
                atomicAdd(&s_bc[a][b], cnt);
                for (int q = 0; q < 3; q++) {
                    atomicMin(&s_bmn[a][b][q], float_to_ordered(mn[q]));
                    atomicMax(&s_bmx[a][b][q], float_to_ordered(mx[q]));
                }
            }
        }
    }
    __syncthreads();
    for (uint32_t k = tid; k < 3 * 16 * 3; k += 256) {  // this tile's bins into the segment's
        const uint32_t vmn = (&s_bmn[0][0][0])[k], vmx = (&s_bmx[0][0][0])[k];
        if (vmn != float_to_ordered(inf)) atomicMin(&(&h.bmn[0][0][0])[k], vmn);
        if (vmx != float_to_ordered(-inf)) atomicMax(&(&h.bmx[0][0][0])[k], vmx);
    }
    for (uint32_t k = tid; k < 3 * 16; k += 256)
        if ((&s_bc[0][0])[k]) atomicAdd(&(&h.bc[0][0])[k], (&s_bc[0][0])[k]);
}
__global__ __launch_bounds__(64) void k_sahh_pick(const SahSeg* segs, const uint32_t* n_segs, SahHuge* hs) {
    const uint32_t s = blockIdx.x;
    if (s >= *n_segs) return;
    __shared__ float s_cost[3][16];
    SahHuge& h = hs[s];
    const uint32_t tid = threadIdx.x;
    const float inf = INFINITY;
    float ext[3];
    for (int a = 0; a < 3; a++) ext[a] = ordered_to_float(h.cmx[a]) - ordered_to_float(h.cmn[a]);
    if (tid < 48) {
        const int a = (int)tid >> 4, sp = (int)tid & 15;
        float cost = inf;
        if (sp >= 1 && ext[a] > 0.0f) {
            float lmn[3] = {inf, inf, inf}, lmx[3] = {-inf, -inf, -inf}, rmn[3] = {inf, inf, inf}, rmx[3] = {-inf, -inf, -inf};
            uint32_t lc = 0, rc = 0;
            for (int b = 0; b < sp; b++) {
                lc += h.bc[a][b];
                for (int q = 0; q < 3; q++) {
                    lmn[q] = fmin_sel(lmn[q], ordered_to_float(h.bmn[a][b][q]));
                    lmx[q] = fmax_sel(lmx[q], ordered_to_float(h.bmx[a][b][q]));
                }
            }
            for (int b = 15; b >= sp; b--) {
                rc += h.bc[a][b];
                for (int q = 0; q < 3; q++) {
                    rmn[q] = fmin_sel(rmn[q], ordered_to_float(h.bmn[a][b][q]));
                    rmx[q] = fmax_sel(rmx[q], ordered_to_float(h.bmx[a][b][q]));
                }
            }
            if (lc != 0 && rc != 0) cost = sah_half_area(lmn, lmx) * (float)lc + sah_half_area(rmn, rmx) * (float)rc;
        }
        s_cost[a][sp] = cost;
    }
    __syncthreads();
    if (tid == 0) {
        float best_cost = inf;
        int best_axis = -1, best_split = 0;
        for (int a = 0; a < 3; a++)
            for (int sp = 1; sp < 16; sp++)
                if (s_cost[a][sp] < best_cost) {
                    best_cost = s_cost[a][sp];
                    best_axis = a;
                    best_split = sp;
                }
        h.axis = best_axis;
        h.split = best_split;
        if (best_axis < 0) h.nl = segs[s].n / 2;
    }
}
// lefts of every tile (axis >= 0 only)
__global__ __launch_bounds__(256) void k_sahh_count(SahArrays A, const SahSeg* segs, const SahHuge* hs, const SahTile* tiles, const uint32_t* n_tiles, uint32_t* tile_left) {
    if (blockIdx.x >= *n_tiles) return;
    __shared__ uint32_t s_n;
    const SahTile tl = tiles[blockIdx.x];
    const SahSeg j = segs[tl.seg];
    const SahHuge& h = hs[tl.seg];
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    if (h.axis >= 0) {
        const float cmn = ordered_to_float(h.cmn[h.axis]), ext = ordered_to_float(h.cmx[h.axis]) - cmn;
        const uint32_t lo = tl.t * kSahTile, hi = lo + kSahTile < j.n ? lo + kSahTile : j.n;
        uint32_t n = 0;
        for (uint32_t k = lo + threadIdx.x; k < hi; k += 256) {
            const uint32_t c = A.idx[j.a + k];
            const float ce = (A.cl_mn[3 * (size_t)c + h.axis] + A.cl_mx[3 * (size_t)c + h.axis]) * 0.5f;
            n += sah_bin(ce, cmn, ext) < h.split ? 1u : 0u;
        }
        n = wave_sum_u(n);
        if ((threadIdx.x & 63u) == 0) atomicAdd(&s_n, n);
    }
    __syncthreads();
    if (threadIdx.x == 0) tile_left[blockIdx.x] = s_n;
}
// per segment: exclusive offsets of its tiles' lefts / rights, and the number of lefts
__global__ void k_sahh_scan(const SahSeg* segs, const uint32_t* n_segs, SahHuge* hs, const uint32_t* tile_left, uint32_t* tile_woff, uint32_t* tile_roff) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= *n_segs) return;
    SahHuge& h = hs[s];
    if (h.axis < 0) return;
    const SahSeg j = segs[s];
    uint32_t w = 0, r = 0;
    for (uint32_t t = 0; t < h.ntiles; t++) {
        const uint32_t lo = t * kSahTile, hi = lo + kSahTile < j.n ? lo + kSahTile : j.n;
        const uint32_t l = tile_left[h.tile_base + t];
        tile_woff[h.tile_base + t] = w;
        tile_roff[h.tile_base + t] = r;
        w += l;
        r += (hi - lo) - l;
    }
    h.nl = w;
}
// stable scatter of a tile into tmp at its final positions: lefts at a + woff.., rights at a + nl + roff..
__global__ __launch_bounds__(256) void k_sahh_scatter(SahArrays A, const SahSeg* segs, const SahHuge* hs, const SahTile* tiles, const uint32_t* n_tiles, const uint32_t* tile_woff,
                                                      const uint32_t* tile_roff) {
    if (blockIdx.x >= *n_tiles) return;
    __shared__ uint32_t s_w, s_r, s_wave[4][2];
    const SahTile tl = tiles[blockIdx.x];
    const SahSeg j = segs[tl.seg];
    const SahHuge& h = hs[tl.seg];
    if (h.axis < 0) return;  // halved in index order: nothing moves
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const float cmn = ordered_to_float(h.cmn[h.axis]), ext = ordered_to_float(h.cmx[h.axis]) - cmn;
    const uint32_t lo = tl.t * kSahTile, hi = lo + kSahTile < j.n ? lo + kSahTile : j.n;
    if (tid == 0) {
        s_w = tile_woff[blockIdx.x];
        s_r = tile_roff[blockIdx.x];
    }
    __syncthreads();
    for (uint32_t base = lo; base < hi; base += 256) {
        const uint32_t k = base + tid;
        const bool valid = k < hi;
        uint32_t c = 0;
        bool goes_left = false;
        if (valid) {
            c = A.idx[j.a + k];
            const float ce = (A.cl_mn[3 * (size_t)c + h.axis] + A.cl_mx[3 * (size_t)c + h.axis]) * 0.5f;
            goes_left = sah_bin(ce, cmn, ext) < h.split;
        }
        const unsigned long long ml = __ballot(valid && goes_left), mr = __ballot(valid && !goes_left);
        if (lane == 0) {
            s_wave[wave][0] = (uint32_t)__popcll(ml);
            s_wave[wave][1] = (uint32_t)__popcll(mr);
        }
        __syncthreads();
        uint32_t wl = s_w, wr = s_r;
        for (uint32_t q = 0; q < wave; q++) {
            wl += s_wave[q][0];
            wr += s_wave[q][1];
        }
        const unsigned long long below = (1ull << lane) - 1ull;
        if (valid) {
            if (goes_left) A.tmp[j.a + wl + (uint32_t)__popcll(ml & below)] = c;
            else A.tmp[j.a + h.nl + wr + (uint32_t)__popcll(mr & below)] = c;
        }
        __syncthreads();
        if (tid == 0) {
            s_w += s_wave[0][0] + s_wave[1][0] + s_wave[2][0] + s_wave[3][0];
            s_r += s_wave[0][1] + s_wave[1][1] + s_wave[2][1] + s_wave[3][1];
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k_sahh_copy(SahArrays A, const SahSeg* segs, const SahHuge* hs, const SahTile* tiles, const uint32_t* n_tiles) {
    if (blockIdx.x >= *n_tiles) return;
    const SahTile tl = tiles[blockIdx.x];
    const SahSeg j = segs[tl.seg];
    if (hs[tl.seg].axis < 0) return;
    const uint32_t lo = tl.t * kSahTile, hi = lo + kSahTile < j.n ? lo + kSahTile : j.n;
    for (uint32_t k = lo + threadIdx.x; k < hi; k += 256) A.idx[j.a + k] = A.tmp[j.a + k];
}
__global__ void k_sahh_emit(SahArrays A, const SahSeg* segs, const uint32_t* n_segs, const SahHuge* hs, SahQueues Q) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= *n_segs) return;
    const SahSeg j = segs[s];
    const SahHuge& h = hs[s];
    const uint32_t node = A.pool[j.pool], nl = h.nl, total = h.total;
    A.rcnt[node] = total > A.T ? total : A.T + 1u;
    const float bmn[3] = {ordered_to_float(h.amn[0]), ordered_to_float(h.amn[1]), ordered_to_float(h.amn[2])};
    const float bmx[3] = {ordered_to_float(h.amx[0]), ordered_to_float(h.amx[1]), ordered_to_float(h.amx[2])};
    sah_store_box(A, node, bmn, bmx);
    sah_patch_parent(A, j.patch, node);
    sah_emit_child(A, SahSeg{j.a, nl, j.pool + 1, node << 1}, Q);
    sah_emit_child(A, SahSeg{j.a + nl, j.n - nl, j.pool + nl, (node << 1) | 1u}, Q);
}

// Segments of at most kSahSmall clusters: sah_top_relink's loop, one 16-lane group per segment (four segments per wave).  A lane holds one
// cluster of the sub-segment being split; the 15 split planes of each axis are costed by lanes 1..15 of the group, every lane sweeping
// the (at most 16) clusters broadcast from their lanes -- unions of the same boxes and sums of the same counts as the bins of the
// sequential sweep, min / max / integer adds being exact in any order -- and the group takes the first minimum in (axis, plane) order
// like its strict `<`.  The segment's cluster order lives in LDS (the global index array is not needed past this point: single
// clusters are linked as soon as they fall out), the sub-segment stack too.
constexpr uint32_t kSahGroup = 16;
static_assert(kSahGroup == kSahSmall, "a group's lanes hold a whole small segment");
__device__ __forceinline__ float group_min_f(float v) {
#pragma unroll
    for (int m = kSahGroup / 2; m >= 1; m >>= 1) v = fmin_sel(v, __shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ float group_max_f(float v) {
#pragma unroll
    for (int m = kSahGroup / 2; m >= 1; m >>= 1) v = fmax_sel(v, __shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ uint32_t group_sum_u(uint32_t v) {
#pragma unroll
    for (int m = kSahGroup / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__global__ __launch_bounds__(256) void k_sah_small(SahArrays A, const SahSeg* segs, uint32_t n_segs) {
    constexpr uint32_t G = kSahGroup, NG = 256 / G;
    __shared__ uint32_t s_cl[NG][G];
    __shared__ SahSeg s_stack[NG][G + 2];
    const uint32_t tid = threadIdx.x, g = tid / G, l = tid % G;
    const uint32_t s = blockIdx.x * NG + g;
    if (s >= n_segs) return;  // (a whole group at a time)
    const uint32_t gshift = (tid & 63u) & ~(G - 1u);  // the group's first lane within the wave
    const unsigned long long gmask = 0xFFFFull << gshift;
    const uint32_t below = (1u << l) - 1u;
    const float inf = INFINITY;
    {
        const SahSeg seg0 = segs[s];
        if (l < seg0.n) s_cl[g][l] = A.idx[seg0.a + l];
        if (l == 0) s_stack[g][0] = SahSeg{0u, seg0.n, seg0.pool, seg0.patch};  // .a: relative to the segment's start from here on
    }
    int sp = 1;
    while (sp > 0) {
        __builtin_amdgcn_wave_barrier();  // (LDS is in order within a wave; this keeps the compiler from moving the reads up)
        const SahSeg j = s_stack[g][--sp];
        const uint32_t node = A.pool[j.pool];
        const bool valid = l < j.n;
        const uint32_t c = valid ? s_cl[g][j.a + l] : 0u;
        float mn[3], mx[3], ce[3];
        const uint32_t cnt = valid ? A.cl_cnt[c] : 0u;
        for (int q = 0; q < 3; q++) {
            mn[q] = valid ? A.cl_mn[3 * (size_t)c + q] : inf;
            mx[q] = valid ? A.cl_mx[3 * (size_t)c + q] : -inf;
            ce[q] = (mn[q] + mx[q]) * 0.5f;
        }
        const uint32_t total = group_sum_u(cnt);
        float cmn[3], ext[3], amn[3], amx[3];
        int b[3];
        for (int q = 0; q < 3; q++) {
            cmn[q] = group_min_f(valid ? ce[q] : inf);
            ext[q] = group_max_f(valid ? ce[q] : -inf) - cmn[q];
            amn[q] = group_min_f(mn[q]);
            amx[q] = group_max_f(mx[q]);
            b[q] = (valid && ext[q] > 0.0f) ? sah_bin(ce[q], cmn[q], ext[q]) : 16;  // (an idle lane adds nothing wherever it lands)
        }
        // lane l: the plane between bins l-1 and l of every axis
        float lmn[3][3], lmx[3][3], rmn[3][3], rmx[3][3];
        uint32_t lc[3] = {0, 0, 0}, rc[3] = {0, 0, 0};
        for (int a = 0; a < 3; a++)
            for (int q = 0; q < 3; q++) {
                lmn[a][q] = rmn[a][q] = inf;
                lmx[a][q] = rmx[a][q] = -inf;
            }
        for (uint32_t k = 0; k < j.n; k++) {
            const int src = (int)(gshift + k);
            float kmn[3], kmx[3];
            int kb[3];
            const uint32_t kc = __shfl(cnt, src);
            for (int q = 0; q < 3; q++) {
                kmn[q] = __shfl(mn[q], src);
                kmx[q] = __shfl(mx[q], src);
                kb[q] = __shfl(b[q], src);
            }
            for (int a = 0; a < 3; a++) {
                const bool left = kb[a] < (int)l;
                lc[a] += left ? kc : 0u;
                rc[a] += left ? 0u : kc;
                for (int q = 0; q < 3; q++) {
                    lmn[a][q] = fmin_sel(lmn[a][q], left ? kmn[q] : inf);
                    lmx[a][q] = fmax_sel(lmx[a][q], left ? kmx[q] : -inf);
                    rmn[a][q] = fmin_sel(rmn[a][q], left ? inf : kmn[q]);
                    rmx[a][q] = fmax_sel(rmx[a][q], left ? -inf : kmx[q]);
                }
            }
        }
        float best_cost = inf;
        uint32_t best_key = 0xFFFFFFFFu;  // axis * 16 + plane
        for (int a = 0; a < 3; a++) {
            if (l == 0 || !(ext[a] > 0.0f) || lc[a] == 0 || rc[a] == 0) continue;
            const float cost = sah_half_area(lmn[a], lmx[a]) * (float)lc[a] + sah_half_area(rmn[a], rmx[a]) * (float)rc[a];
            if (cost < best_cost) {
                best_cost = cost;
                best_key = (uint32_t)a * 16u + l;
            }
        }
#pragma unroll
        for (int m = G / 2; m >= 1; m >>= 1) {
            const float oc = __shfl_xor(best_cost, m);
            const uint32_t ok = __shfl_xor(best_key, m);
            if (oc < best_cost || (oc == best_cost && ok < best_key)) {
                best_cost = oc;
                best_key = ok;
            }
        }
        uint32_t nl, newpos = l;
        if (best_key == 0xFFFFFFFFu) {
            nl = j.n / 2;  // coincident centroids: halve in index order
        } else {
            const int axis = (int)(best_key >> 4), split = (int)(best_key & 15u);
            const int bx = axis == 0 ? b[0] : (axis == 1 ? b[1] : b[2]);
            const bool gl = valid && bx < split, gr = valid && !gl;
            const uint32_t ml = (uint32_t)((__ballot(gl) & gmask) >> gshift), mr = (uint32_t)((__ballot(gr) & gmask) >> gshift);
            nl = (uint32_t)__popc(ml);
            newpos = gl ? (uint32_t)__popc(ml & below) : nl + (uint32_t)__popc(mr & below);  // stable on both sides
        }
        __builtin_amdgcn_wave_barrier();
        if (valid) s_cl[g][j.a + newpos] = c;
        __builtin_amdgcn_wave_barrier();
        const uint32_t nr = j.n - nl;
        if (l == 0) {
            A.rcnt[node] = total > A.T ? total : A.T + 1u;
            sah_store_box(A, node, amn, amx);
            sah_patch_parent(A, j.patch, node);
            const SahSeg cl{j.a, nl, j.pool + 1, node << 1}, cr{j.a + nl, nr, j.pool + nl, (node << 1) | 1u};
            int w = sp;
            if (nr == 1) sah_patch_parent(A, cr.patch, A.cl_ref[s_cl[g][cr.a]]);
            else s_stack[g][w++] = cr;
            if (nl == 1) sah_patch_parent(A, cl.patch, A.cl_ref[s_cl[g][cl.a]]);
            else s_stack[g][w] = cl;
        }
        sp += (nr > 1 ? 1 : 0) + (nl > 1 ? 1 : 0);
    }
}
}  // namespace

// returns hipSuccess and *relinked = false when the tree has fewer than three clusters (nothing to do, like the host path)
static hipError_t sah_top_relink_gpu(hipStream_t st, uint32_t n, uint32_t nn, uint32_t* left, uint32_t* right, uint32_t* rcnt, uint32_t* pint, uint32_t* pleaf,
                                     const float* lmin, const float* lmax, float* nbox, uint32_t T, BuildArena& arena, bool* relinked) {
    *relinked = false;
    hipError_t err = hipSuccess;
    uint32_t *top = nullptr, *ncl = nullptr, *pool_pos = nullptr, *cl_pos = nullptr, *pool = nullptr, *cl_ref = nullptr, *cl_cnt = nullptr, *idx = nullptr, *tmp = nullptr,
             *counters = nullptr;
    float *cl_mn = nullptr, *cl_mx = nullptr;
    SahSeg *seg_a = nullptr, *seg_b = nullptr, *seg_small = nullptr;
    SahHuge* hs = nullptr;
    SahTile* tiles = nullptr;
    uint32_t *tile_left = nullptr, *tile_woff = nullptr, *tile_roff = nullptr;
    void* scan_tmp = nullptr;
    size_t scan_bytes = 0;
    uint32_t tails[4] = {0, 0, 0, 0}, npool = 0, nc = 0;
    const unsigned grid = (unsigned)(((uint64_t)nn + 255) / 256 > 4096 ? 4096 : ((uint64_t)nn + 255) / 256);
#define SAH_CHECK(x)            \
    do {                        \
        err = (x);              \
        if (err != hipSuccess) goto sah_done; \
    } while (0)
    SAH_CHECK(arena.take(&top, (size_t)nn * 4));
    SAH_CHECK(arena.take(&ncl, (size_t)nn * 4));
    SAH_CHECK(arena.take(&pool_pos, (size_t)nn * 4));
    SAH_CHECK(arena.take(&cl_pos, (size_t)nn * 4));
    hipLaunchKernelGGL(k_sah_mark, dim3(grid), dim3(256), 0, st, left, right, rcnt, nn, T, top, ncl);
    SAH_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, top, pool_pos, (int)nn, st));
    SAH_CHECK(arena.take(&scan_tmp, scan_bytes ? scan_bytes : 16));
    SAH_CHECK(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, top, pool_pos, (int)nn, st));
    SAH_CHECK(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, ncl, cl_pos, (int)nn, st));
    SAH_CHECK(hipMemcpyAsync(&tails[0], pool_pos + (nn - 1), 4, hipMemcpyDeviceToHost, st));
    SAH_CHECK(hipMemcpyAsync(&tails[1], top + (nn - 1), 4, hipMemcpyDeviceToHost, st));
    SAH_CHECK(hipMemcpyAsync(&tails[2], cl_pos + (nn - 1), 4, hipMemcpyDeviceToHost, st));
    SAH_CHECK(hipMemcpyAsync(&tails[3], ncl + (nn - 1), 4, hipMemcpyDeviceToHost, st));
    SAH_CHECK(hipStreamSynchronize(st));
    npool = tails[0] + tails[1];
    nc = tails[2] + tails[3];
    if (nc < 3 || npool != nc - 1) goto sah_done;  // (the host path's `return false`)
    (void)n;
    SAH_CHECK(arena.take(&pool, (size_t)npool * 4));
    SAH_CHECK(arena.take(&cl_ref, (size_t)nc * 4));
    SAH_CHECK(arena.take(&cl_cnt, (size_t)nc * 4));
    SAH_CHECK(arena.take(&cl_mn, (size_t)nc * 12));
    SAH_CHECK(arena.take(&cl_mx, (size_t)nc * 12));
    SAH_CHECK(arena.take(&idx, (size_t)nc * 4));
    SAH_CHECK(arena.take(&tmp, (size_t)nc * 4));
    SAH_CHECK(arena.take(&seg_a, 2 * ((size_t)nc / kSahSmall + 2) * sizeof(SahSeg)));  // [huge | big] of the current level
    SAH_CHECK(arena.take(&seg_b, 2 * ((size_t)nc / kSahSmall + 2) * sizeof(SahSeg)));  // ... of the next one
    SAH_CHECK(arena.take(&seg_small, ((size_t)nc / 2 + 2) * sizeof(SahSeg)));
    SAH_CHECK(arena.take(&counters, 64));
    SAH_CHECK(hipMemsetAsync(counters, 0, 64, st));
    {
        const size_t mh = (size_t)nc / kSahHuge + 1, mt = (size_t)nc / kSahTile + mh + 1;
        SAH_CHECK(arena.take(&hs, mh * sizeof(SahHuge)));
        SAH_CHECK(arena.take(&tiles, mt * sizeof(SahTile)));
        SAH_CHECK(arena.take(&tile_left, mt * 4));
        SAH_CHECK(arena.take(&tile_woff, mt * 4));
        SAH_CHECK(arena.take(&tile_roff, mt * 4));
    }
    hipLaunchKernelGGL(k_sah_gather, dim3(grid), dim3(256), 0, st, left, right, rcnt, nn, T, top, pool_pos, cl_pos, lmin, lmax, nbox, pool, cl_ref, cl_cnt, cl_mn, cl_mx, idx);
    {
        SahArrays A{left, right, rcnt, pint, pleaf, cl_ref, cl_cnt, cl_mn, cl_mx, pool, idx, tmp, T, nbox};
        const size_t half = (size_t)nc / kSahSmall + 2;
        const SahSeg root{0, nc, 0, 0xFFFFFFFFu};
        // counters: [0..2] huge / big segment counts of the level being processed + spare, [4..6] of the next level, [8] small segments
        uint32_t cnt[3] = {0, 0, 0};
        cnt[nc > kSahHuge ? 0 : (nc > kSahSmall ? 1 : 2)] = 1;
        SAH_CHECK(hipMemcpyAsync(nc > kSahHuge ? seg_a : (nc > kSahSmall ? seg_a + half : seg_small), &root, sizeof(root), hipMemcpyHostToDevice, st));
        SAH_CHECK(hipMemcpyAsync(counters, cnt, 8, hipMemcpyHostToDevice, st));
        SAH_CHECK(hipMemcpyAsync(counters + 8, &cnt[2], 4, hipMemcpyHostToDevice, st));
        const bool trace = getenv("RT3_TRACE_BUILD") != nullptr;
        auto tnow = [] { return std::chrono::steady_clock::now(); };
        auto tl = tnow();
        // Levels are launched back to back with grids sized for the most segments a level can hold (workgroups beyond the level's
        // count return at once); the host looks at the counters only every 24 levels.  A level's segments have more than kSahSmall
        // (kSahHuge) clusters each, so there are at most nc / kSahSmall (nc / kSahHuge) of them.
        const uint32_t max_huge = nc / kSahHuge + 1, max_big = nc / kSahSmall + 1, max_tiles = nc / kSahTile + max_huge + 1;
        bool huge_possible = nc > kSahHuge;  // (checked again with the counters after every burst of levels)
        int level = 0, cur = 0;
        for (;;) {
            for (int burst = 0; burst < (huge_possible ? 8 : 24); burst++, level++, cur ^= 1) {
                uint32_t* c_cur = counters + 4 * cur;
                uint32_t* c_next = counters + 4 * (cur ^ 1);
                SAH_CHECK(hipMemsetAsync(c_next, 0, 8, st));
                const SahQueues Q{seg_b, seg_b + half, seg_small, c_next, counters + 8};
                if (huge_possible) {  // segments of more than kSahHuge clusters, tiled over several workgroups: one launch per phase
                    uint32_t* n_tiles = counters + 12;
                    SAH_CHECK(hipMemsetAsync(n_tiles, 0, 4, st));
                    hipLaunchKernelGGL(k_sahh_tiles, dim3(max_huge), dim3(64), 0, st, seg_a, c_cur, hs, tiles, n_tiles);
                    hipLaunchKernelGGL(k_sahh_bounds, dim3(max_tiles), dim3(256), 0, st, A, seg_a, hs, tiles, n_tiles);
                    hipLaunchKernelGGL(k_sahh_bins, dim3(max_tiles), dim3(256), 0, st, A, seg_a, hs, tiles, n_tiles);
                    hipLaunchKernelGGL(k_sahh_pick, dim3(max_huge), dim3(64), 0, st, seg_a, c_cur, hs);
                    hipLaunchKernelGGL(k_sahh_count, dim3(max_tiles), dim3(256), 0, st, A, seg_a, hs, tiles, n_tiles, tile_left);
                    hipLaunchKernelGGL(k_sahh_scan, dim3((max_huge + 63) / 64), dim3(64), 0, st, seg_a, c_cur, hs, tile_left, tile_woff, tile_roff);
                    hipLaunchKernelGGL(k_sahh_scatter, dim3(max_tiles), dim3(256), 0, st, A, seg_a, hs, tiles, n_tiles, tile_woff, tile_roff);
                    hipLaunchKernelGGL(k_sahh_copy, dim3(max_tiles), dim3(256), 0, st, A, seg_a, hs, tiles, n_tiles);
                    hipLaunchKernelGGL(k_sahh_emit, dim3((max_huge + 63) / 64), dim3(64), 0, st, A, seg_a, c_cur, hs, Q);
                }
                hipLaunchKernelGGL(k_sah_block<256>, dim3(max_big), dim3(256), 0, st, A, seg_a + half, c_cur + 1, Q);
                std::swap(seg_a, seg_b);
            }
            SAH_CHECK(hipMemcpyAsync(cnt, counters + 4 * cur, 8, hipMemcpyDeviceToHost, st));
            SAH_CHECK(hipStreamSynchronize(st));
            if (cnt[0] + cnt[1] == 0) break;
            huge_possible = cnt[0] > 0;
        }
        if (trace) {
            fprintf(stderr, "rt3 build:   SAH block levels (%d launched): %.3f ms\n", level, std::chrono::duration<double, std::milli>(tnow() - tl).count());
            tl = tnow();
        }
        SAH_CHECK(hipMemcpyAsync(&cnt[2], counters + 8, 4, hipMemcpyDeviceToHost, st));
        SAH_CHECK(hipStreamSynchronize(st));
        if (cnt[2]) hipLaunchKernelGGL(k_sah_small, dim3((cnt[2] + 15) / 16), dim3(256), 0, st, A, seg_small, cnt[2]);
        if (trace) {
            SAH_CHECK(hipStreamSynchronize(st));
            fprintf(stderr, "rt3 build:   SAH small: %u segments, %.3f ms (%u clusters)\n", cnt[2], std::chrono::duration<double, std::milli>(tnow() - tl).count(), nc);
        }
        const uint32_t no_parent = 0xFFFFFFFFu;
        SAH_CHECK(hipMemcpyAsync(pint, &no_parent, 4, hipMemcpyHostToDevice, st));
        SAH_CHECK(hipGetLastError());
        SAH_CHECK(hipStreamSynchronize(st));  // (no_parent / root live on this frame's stack)
        *relinked = true;
    }
sah_done:
#undef SAH_CHECK
    return err;
}

#define LB_CHECK(x)                  \
    do {                             \
        hipError_t e_ = (x);         \
        if (e_ != hipSuccess) {      \
            err = e_;                \
            goto done;               \
        }                            \
    } while (0)

hipError_t lbvh_build(hipStream_t st, const float* verts, const uint32_t* indices, const FlatGeomDev* geoms, const uint32_t* prim_geom,
                      const uint32_t* first_prim, uint32_t n, uint32_t leaf_max, uint32_t node_width, uint32_t node_quant, uint32_t collapse_mode,
                      uint32_t sah_top, uint32_t sah_device, BuildArena& arena, LbvhResult* out) {
    hipError_t err = hipSuccess;
    *out = LbvhResult{};
    out->n_tris = n;
    const int wide = node_width == 4, quant = wide ? (node_quant > 2 ? 2 : (int)node_quant) : 0, collapse = wide ? (collapse_mode > 2 ? 2 : (int)collapse_mode) : 0;
    const bool dp = collapse == 2;  // cost-driven collapse: tree order + bottom-up dynamic programme; the SAH top then runs on the device whatever sah_device says
    out->node_bytes = (wide && !quant) ? 128u : (quant == 2 ? 16u * kC48Stride : 64u);
    out->layout = !wide ? kLayoutBinary64 : (quant == 2 ? kLayoutWide48Q : (quant ? kLayoutWide64Q : kLayoutWide128));
    if (n == 0) return hipSuccess;
    const uint32_t nn = n > 1 ? n - 1 : 1;
    float *bmin = nullptr, *bmax = nullptr, *lmin = nullptr, *lmax = nullptr, *nbox = nullptr;
    uint32_t *live = nullptr, *dk = nullptr, *newpos = nullptr;
    float *dc = nullptr, *lmin2 = nullptr, *lmax2 = nullptr;
    float4* tris_dp = nullptr;  // cost-driven collapse: Morton-ordered triangle records before they move into tree order
    uint32_t *bounds = nullptr, *vals_in = nullptr, *vals_out = nullptr, *left = nullptr, *right = nullptr, *pint = nullptr, *pleaf = nullptr,
             *arrive = nullptr, *levels = nullptr, *rlo = nullptr, *rcnt = nullptr, *keep = nullptr, *newidx = nullptr, *n_int = nullptr,
             *n_ltri = nullptr, *cbase = nullptr, *tbase = nullptr;
    float4* tris_morton = nullptr;  // compact layout: Morton-ordered triangle records before they move into leaf order
    uint64_t *keys_in = nullptr, *keys_out = nullptr;
    void *temp = nullptr, *temp2 = nullptr;
    char* stage = nullptr;  // pinned host staging of the SAH top
    size_t temp_bytes = 0, temp2_bytes = 0;
    const unsigned grid = (unsigned)(((uint64_t)n + 255) / 256 > 4096 ? 4096 : ((uint64_t)n + 255) / 256);
    uint32_t init_bounds[12];
    uint32_t tail[2] = {0, 0};
    for (int k = 0; k < 3; k++) {
        init_bounds[k] = 0xFFFFFFFFu;
        init_bounds[3 + k] = 0u;
        init_bounds[6 + k] = 0xFFFFFFFFu;
        init_bounds[9 + k] = 0u;
    }
    {
        // everything below comes out of one block.  Per triangle: 4 x 12 (boxes) + 24 (node boxes) + 24 (sort keys / values) + 9 x 4
        // (links, ranges, flags) + 64 (compact layout only) + 8 (collapse frontier) + 16 + 44 + 12 (SAH top: marks, clusters,
        // segment queues) = 276 bytes, rounded up, plus the library scans' / sort's own scratch and the alignment of ~70 pieces
        size_t sort_bytes = 0, scan_bytes = 0;
        LB_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, keys_in, keys_out, vals_in, vals_out, (int)n, 0, 63, st));
        LB_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, keep, newidx, (int)nn, st));
        LB_CHECK(arena.reserve((size_t)n * 448 + sort_bytes + 2 * scan_bytes + ((size_t)1 << 20)));
    }
    LB_CHECK(arena.take(&bmin, (size_t)n * 12));
    LB_CHECK(arena.take(&bmax, (size_t)n * 12));
    LB_CHECK(arena.take(&lmin, (size_t)n * 12));
    LB_CHECK(arena.take(&lmax, (size_t)n * 12));
    LB_CHECK(arena.take(&nbox, (size_t)nn * 24));
    LB_CHECK(arena.take(&bounds, 64));
    LB_CHECK(arena.take(&keys_in, (size_t)n * 8));
    LB_CHECK(arena.take(&keys_out, (size_t)n * 8));
    LB_CHECK(arena.take(&vals_in, (size_t)n * 4));
    LB_CHECK(arena.take(&vals_out, (size_t)n * 4));
    LB_CHECK(arena.take(&left, (size_t)nn * 4));
    LB_CHECK(arena.take(&right, (size_t)nn * 4));
    LB_CHECK(arena.take(&pint, (size_t)nn * 4));
    LB_CHECK(arena.take(&pleaf, (size_t)n * 4));
    LB_CHECK(arena.take(&arrive, (size_t)nn * 4));
    LB_CHECK(arena.take(&rlo, (size_t)nn * 4));
    LB_CHECK(arena.take(&rcnt, (size_t)nn * 4));
    LB_CHECK(arena.take(&keep, (size_t)nn * 4));
    LB_CHECK(arena.take(&newidx, (size_t)nn * 4));
    LB_CHECK(arena.take(&levels, 4));
    LB_CHECK(arena.take(&live, (size_t)nn * 4));
    if (dp && n > 1) {
        LB_CHECK(arena.take(&dk, (size_t)nn * 4));
        LB_CHECK(arena.take(&dc, (size_t)nn * 12));
        LB_CHECK(arena.take(&newpos, (size_t)n * 4));
        LB_CHECK(arena.take(&lmin2, (size_t)n * 12));
        LB_CHECK(arena.take(&lmax2, (size_t)n * 12));
        LB_CHECK(arena.take(&tris_dp, (size_t)n * 48));
    }
    LB_CHECK(hipMalloc(&out->tris, (size_t)n * 48 + 128));  // + slack: the traversal fetch may over-read the last leaf by up to 128 B
    LB_CHECK(hipMemsetAsync((char*)out->tris + (size_t)n * 48, 0, 128, st));
    LB_CHECK(hipMalloc(&out->tri_shade, (size_t)n * 16));
    LB_CHECK(hipMalloc(&out->tri_uv, (size_t)n * 24));
    if (quant == 2 && n > 1) {
        LB_CHECK(arena.take(&tris_morton, (size_t)n * 48));
        LB_CHECK(arena.take(&n_int, (size_t)nn * 4));
        LB_CHECK(arena.take(&n_ltri, (size_t)nn * 4));
        LB_CHECK(arena.take(&cbase, (size_t)nn * 4));
        LB_CHECK(arena.take(&tbase, (size_t)nn * 4));
    }
    LB_CHECK(hipMemcpyAsync(bounds, init_bounds, sizeof(init_bounds), hipMemcpyHostToDevice, st));
    LB_CHECK(hipMemsetAsync(arrive, 0, (size_t)nn * 4, st));
    LB_CHECK(hipMemsetAsync(levels, 0, 4, st));
    hipLaunchKernelGGL(k_prim_bounds, dim3(grid > 512 ? 512 : grid), dim3(256), 0, st, verts, indices, geoms, prim_geom, first_prim, n, bmin, bmax, bounds);
    hipLaunchKernelGGL(k_tri_shade, dim3(grid), dim3(256), 0, st, verts, indices, geoms, prim_geom, first_prim, n, out->tri_shade, out->tri_uv);
    hipLaunchKernelGGL(k_morton, dim3(grid), dim3(256), 0, st, bmin, bmax, bounds, n, keys_in, vals_in);
    LB_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, keys_in, keys_out, vals_in, vals_out, (int)n, 0, 63, st));
    LB_CHECK(arena.take(&temp, temp_bytes ? temp_bytes : 16));
    LB_CHECK(hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (int)n, 0, 63, st));
    hipLaunchKernelGGL(k_leaves, dim3(grid), dim3(256), 0, st, verts, indices, geoms, prim_geom, first_prim, vals_out, bmin, bmax, bounds, n,
                       tris_dp ? tris_dp : (tris_morton ? tris_morton : out->tris), lmin, lmax);
    if (n == 1) {
        LB_CHECK(hipMalloc(&out->nodes, out->node_bytes));
        hipLaunchKernelGGL(k_single, dim3(1), dim3(1), 0, st, lmin, lmax, wide, quant, out->nodes);
        out->n_nodes = 1;
        out->max_depth = 2;
        LB_CHECK(hipGetLastError());
        LB_CHECK(hipStreamSynchronize(st));
    } else {
        hipLaunchKernelGGL(k_hierarchy, dim3(grid), dim3(256), 0, st, keys_out, (int)n, left, right, pint, pleaf, rlo, rcnt);
        // clusters of the SAH top: Karras subtrees of at most T triangles.  Without tree order a multi-triangle leaf must be a Morton range,
        // so T >= leaf_max; with it (cost-driven collapse) T goes down to single triangles
        const uint32_t T_sah = dp ? sah_top : (sah_top > leaf_max ? sah_top : leaf_max);
        if (dp) sah_device = 1;
        const bool lite = sah_top && sah_device && T_sah <= 64;  // the SAH top only reads the cluster boxes and writes every box above them itself
        if (lite) hipLaunchKernelGGL(k_refit_clusters, dim3(grid), dim3(256), 0, st, rlo, rcnt, lmin, lmax, nn, T_sah, nbox);
        else hipLaunchKernelGGL(k_refit, dim3(grid), dim3(256), 0, st, left, right, pint, pleaf, lmin, lmax, n, nbox, arrive);
        if (sah_top && sah_device) {  // re-link the upper tree by binned SAH on the GPU (bit-identical to the host path below)
            const uint32_t T = T_sah;
            bool relinked = false;
            const auto t0 = std::chrono::steady_clock::now();
            LB_CHECK(sah_top_relink_gpu(st, n, nn, left, right, rcnt, pint, pleaf, lmin, lmax, nbox, T, arena, &relinked));
            if (!relinked && lite) {  // fewer than three clusters: the Karras tree stands, and its upper boxes are still to come
                LB_CHECK(hipMemsetAsync(arrive, 0, (size_t)nn * 4, st));
                hipLaunchKernelGGL(k_refit, dim3(grid), dim3(256), 0, st, left, right, pint, pleaf, lmin, lmax, n, nbox, arrive);
            }
            if (getenv("RT3_TRACE_BUILD")) {
                LB_CHECK(hipStreamSynchronize(st));
                fprintf(stderr, "rt3 build: device SAH top (incl. GPU LBVH drain) %.2f ms\n",
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
            }
        } else if (sah_top) {  // the same re-link on the host (RT3_OPT_SAH_TOP_DEVICE = 0)
            const uint32_t T = sah_top > leaf_max ? sah_top : leaf_max;
            const bool trace = getenv("RT3_TRACE_BUILD") != nullptr;  // phase times of the host part to stderr
            auto now = [] { return std::chrono::steady_clock::now(); };
            auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            const auto t0 = now();
            // one pinned staging block for the eight arrays: pageable copies made the runtime pin and unpin the user pages around every
            // transfer, which cost more than the SAH itself (20+ ms of a 40 ms build)
            const size_t off_left = 0, off_right = off_left + (size_t)nn * 4, off_rcnt = off_right + (size_t)nn * 4, off_pint = off_rcnt + (size_t)nn * 4,
                         off_pleaf = off_pint + (size_t)nn * 4, off_lmin = off_pleaf + (size_t)n * 4, off_lmax = off_lmin + (size_t)n * 12,
                         off_nbox = off_lmax + (size_t)n * 12, stage_bytes = off_nbox + (size_t)nn * 24;
            LB_CHECK(hipHostMalloc((void**)&stage, stage_bytes, hipHostMallocDefault));
            uint32_t *h_left = (uint32_t*)(stage + off_left), *h_right = (uint32_t*)(stage + off_right), *h_rcnt = (uint32_t*)(stage + off_rcnt),
                     *h_pint = (uint32_t*)(stage + off_pint), *h_pleaf = (uint32_t*)(stage + off_pleaf);
            float *h_lmin = (float*)(stage + off_lmin), *h_lmax = (float*)(stage + off_lmax), *h_nbox = (float*)(stage + off_nbox);
            LB_CHECK(hipMemcpyAsync(h_left, left, (size_t)nn * 4, hipMemcpyDeviceToHost, st));
            LB_CHECK(hipMemcpyAsync(h_right, right, (size_t)nn * 4, hipMemcpyDeviceToHost, st));
            LB_CHECK(hipMemcpyAsync(h_rcnt, rcnt, (size_t)nn * 4, hipMemcpyDeviceToHost, st));
            LB_CHECK(hipMemcpyAsync(h_pint, pint, (size_t)nn * 4, hipMemcpyDeviceToHost, st));
            LB_CHECK(hipMemcpyAsync(h_pleaf, pleaf, (size_t)n * 4, hipMemcpyDeviceToHost, st));
            LB_CHECK(hipMemcpyAsync(h_lmin, lmin, (size_t)n * 12, hipMemcpyDeviceToHost, st));
            LB_CHECK(hipMemcpyAsync(h_lmax, lmax, (size_t)n * 12, hipMemcpyDeviceToHost, st));
            LB_CHECK(hipMemcpyAsync(h_nbox, nbox, (size_t)nn * 24, hipMemcpyDeviceToHost, st));
            out->bulk_copies += 8;  // the host SAH top (RT3_OPT_SAH_TOP_DEVICE = 0) is the one path that moves arrays
            LB_CHECK(hipStreamSynchronize(st));
            const auto t1 = now();
            const bool relinked = sah_top_relink(nn, h_left, h_right, h_rcnt, h_pint, h_pleaf, h_lmin, h_lmax, h_nbox, T);
            const auto t2 = now();
            if (trace) fprintf(stderr, "rt3 build: host SAH top: download %.2f ms (incl. GPU LBVH drain), relink %.2f ms\n", ms(t0, t1), ms(t1, t2));
            if (relinked) {
                LB_CHECK(hipMemcpyAsync(left, h_left, (size_t)nn * 4, hipMemcpyHostToDevice, st));
                LB_CHECK(hipMemcpyAsync(right, h_right, (size_t)nn * 4, hipMemcpyHostToDevice, st));
                LB_CHECK(hipMemcpyAsync(rcnt, h_rcnt, (size_t)nn * 4, hipMemcpyHostToDevice, st));
                LB_CHECK(hipMemcpyAsync(pint, h_pint, (size_t)nn * 4, hipMemcpyHostToDevice, st));
                LB_CHECK(hipMemcpyAsync(pleaf, h_pleaf, (size_t)n * 4, hipMemcpyHostToDevice, st));
                out->bulk_copies += 5;
                LB_CHECK(hipMemsetAsync(arrive, 0, (size_t)nn * 4, st));
                hipLaunchKernelGGL(k_refit, dim3(grid), dim3(256), 0, st, left, right, pint, pleaf, lmin, lmax, n, nbox, arrive);
                if (trace) {
                    LB_CHECK(hipStreamSynchronize(st));
                    fprintf(stderr, "rt3 build: upload + second refit %.2f ms\n", ms(t2, now()));
                }
            }
        }
        const auto t3 = std::chrono::steady_clock::now();
        if (dp) {
            // bottom-up: true triangle counts, the collapse costs and choices; then every triangle's / node's place in tree order, and the move
            LB_CHECK(hipMemsetAsync(arrive, 0, (size_t)nn * 4, st));
            hipLaunchKernelGGL(k_dp_up, dim3(grid), dim3(256), 0, st, left, right, pint, pleaf, lmin, lmax, nbox, n, leaf_max, rcnt, dc, dk, live, arrive);
            const unsigned g3 = (unsigned)(((uint64_t)n + nn + 255) / 256 > 4096 ? 4096 : ((uint64_t)n + nn + 255) / 256);
            hipLaunchKernelGGL(k_tree_order, dim3(g3), dim3(256), 0, st, left, right, pint, pleaf, rcnt, n, nn, newpos, rlo);
            float4* tris_to = quant == 2 ? tris_morton : out->tris;  // (the compact layout moves them once more, into leaf order, when it emits)
            hipLaunchKernelGGL(k_tree_reorder, dim3(g3), dim3(256), 0, st, newpos, n, nn, tris_dp, tris_to, lmin, lmax, lmin2, lmax2, left, right);
            lmin = lmin2;
            lmax = lmax2;
        } else {
            hipLaunchKernelGGL(k_live_flags, dim3(grid), dim3(256), 0, st, rcnt, nn, leaf_max, live);
        }
        if (collapse) {
            // top-down, one four-wide level per launch (the frontier of level l+1 is produced by level l); ~log4(n) launches, sixteen at a
            // time between looks at the frontier counters (fr_n[l] = size of level l's frontier)
            uint32_t *fr_a = nullptr, *fr_b = nullptr, *fr_n = nullptr, wide_levels = 0;
            const uint32_t root = 0, one = 1;
            const size_t n_cnt = (size_t)nn + 18;  // a level per node at most, plus one burst
            hipError_t e2 = arena.take(&fr_a, (size_t)nn * 4);
            if (e2 == hipSuccess) e2 = arena.take(&fr_b, (size_t)nn * 4);
            if (e2 == hipSuccess) e2 = arena.take(&fr_n, n_cnt * 4);
            if (e2 == hipSuccess) e2 = hipMemsetAsync(keep, 0, (size_t)nn * 4, st);
            if (e2 == hipSuccess) e2 = hipMemsetAsync(fr_n, 0, n_cnt * 4, st);
            if (e2 == hipSuccess) e2 = hipMemcpyAsync(fr_a, &root, 4, hipMemcpyHostToDevice, st);
            if (e2 == hipSuccess) e2 = hipMemcpyAsync(fr_n, &one, 4, hipMemcpyHostToDevice, st);
            const unsigned g2 = (unsigned)((nn + 255) / 256 > 1024 ? 1024 : (nn + 255) / 256);
            uint32_t level = 0, last = 1;
            while (e2 == hipSuccess && last > 0 && level + 16 < n_cnt) {
                for (int burst = 0; burst < 16; burst++, level++) {
                    hipLaunchKernelGGL(k_wide_level, dim3(g2), dim3(256), 0, st, left, right, live, nbox, dk, collapse, fr_a, fr_n + level, keep, fr_b, fr_n + level + 1);
                    std::swap(fr_a, fr_b);
                }
                e2 = hipMemcpyAsync(&last, fr_n + level, 4, hipMemcpyDeviceToHost, st);
                if (e2 == hipSuccess) e2 = hipStreamSynchronize(st);
            }
            if (e2 == hipSuccess) {  // the number of non-empty frontiers
                std::vector<uint32_t> h_cnt(level + 1);
                e2 = hipMemcpyAsync(h_cnt.data(), fr_n, (size_t)(level + 1) * 4, hipMemcpyDeviceToHost, st);
                if (e2 == hipSuccess) e2 = hipStreamSynchronize(st);
                while (wide_levels <= level && h_cnt[wide_levels] > 0) wide_levels++;
            }
            LB_CHECK(e2);
            const uint32_t lv = wide_levels + 1;  // levels from the root down to the deepest node's leaf slots
            LB_CHECK(hipMemcpyAsync(levels, &lv, 4, hipMemcpyHostToDevice, st));
            LB_CHECK(hipStreamSynchronize(st));
        } else {
            hipLaunchKernelGGL(k_keep_flags, dim3(grid), dim3(256), 0, st, pint, live, nn, wide, keep, levels);
        }
        LB_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, temp2_bytes, keep, newidx, (int)nn, st));
        LB_CHECK(arena.take(&temp2, temp2_bytes ? temp2_bytes : 16));
        LB_CHECK(hipcub::DeviceScan::ExclusiveSum(temp2, temp2_bytes, keep, newidx, (int)nn, st));
        LB_CHECK(hipMemcpyAsync(&tail[0], newidx + (nn - 1), 4, hipMemcpyDeviceToHost, st));
        LB_CHECK(hipMemcpyAsync(&tail[1], keep + (nn - 1), 4, hipMemcpyDeviceToHost, st));
        LB_CHECK(hipMemcpyAsync(&out->max_depth, levels, 4, hipMemcpyDeviceToHost, st));
        LB_CHECK(hipStreamSynchronize(st));
        out->n_nodes = tail[0] + tail[1];
        LB_CHECK(hipMalloc(&out->nodes, (size_t)out->n_nodes * out->node_bytes));
        if (quant == 2) {
            hipLaunchKernelGGL(k_child_counts, dim3(grid), dim3(256), 0, st, left, right, rcnt, nbox, keep, nn, live, dk, collapse, n_int, n_ltri);
            LB_CHECK(hipcub::DeviceScan::ExclusiveSum(temp2, temp2_bytes, n_int, cbase, (int)nn, st));
            LB_CHECK(hipcub::DeviceScan::ExclusiveSum(temp2, temp2_bytes, n_ltri, tbase, (int)nn, st));
            hipLaunchKernelGGL(k_assign_index, dim3(grid), dim3(256), 0, st, left, right, live, nbox, keep, nn, dk, collapse, cbase, newidx);
        }
        hipLaunchKernelGGL(k_emit_nodes, dim3(grid), dim3(256), 0, st, left, right, rlo, rcnt, keep, newidx, lmin, lmax, nbox, nn, live, dk, wide,
                           quant, collapse, out->nodes, cbase, tbase, tris_morton, out->tris);
        if (wide && quant == 1) {  // top-of-tree copy the traversal kernels keep in LDS
            uint32_t* d_ntop = nullptr;
            LB_CHECK(hipMalloc(&out->top, (size_t)kTopCacheNodes * 64));
            LB_CHECK(arena.take(&d_ntop, 4));
            hipLaunchKernelGGL(k_top_cache, dim3(1), dim3(1), 0, st, out->nodes, out->n_nodes, (uint32_t*)out->top, d_ntop);
            hipError_t e3 = hipMemcpyAsync(&out->n_top, d_ntop, 4, hipMemcpyDeviceToHost, st);
            if (e3 == hipSuccess) e3 = hipStreamSynchronize(st);
            LB_CHECK(e3);
        }
        LB_CHECK(hipGetLastError());
        LB_CHECK(hipStreamSynchronize(st));
        if (getenv("RT3_TRACE_BUILD"))
            fprintf(stderr, "rt3 build: collapse + emit %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t3).count());
    }
done:
    if (stage) (void)hipHostFree(stage);
    if (err != hipSuccess) {
        (void)hipFree(out->nodes);
        (void)hipFree(out->tris);
        (void)hipFree(out->tri_shade);
        (void)hipFree(out->tri_uv);
        (void)hipFree(out->top);
        out->top = nullptr;
        out->n_top = 0;
        out->nodes = nullptr;
        out->tris = nullptr;
        out->tri_shade = nullptr;
        out->tri_uv = nullptr;
    }
    return err;
}

}  // namespace rt3
