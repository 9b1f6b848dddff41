// rt3_probes.hip -- the probe-GI passes of the reference, restated for wave64:
//   "structured_importance_sampling"  k_sis                      shaders/old/structured_importance_sampling.slang:19-71
//   "trace_probes"                    k_probe_raygen -> k_extend -> k_probe_store         shaders/old/trace_probes.slang:15-77
//   "spherical_harmonic_conversion"   k_sh_conversion            shaders/old/spherical_harmonic_conversion.slang:9-33
//   "interpolate_probes"              k_interpolate (two phases) shaders/old/interpolate_probes.slang:11-103
// A probe is one 8x8-texel cell of the probe atlas and owns a 16x16 pixel block.  The reference dispatches 8x8 = 64 threads per
// probe; on CDNA4 that is exactly one wavefront, so its groupshared sort (math.slang:140-160) and WaveActiveSum become
// cross-lane exchanges with no barriers and no LDS round trips.  The shaders are restated as written (debug stores included);
// oracle/rt3_oracle_probes.c lists the [rule]s chosen where the text leaves a result open.
#include "rt3_internal.hpp"

namespace rt3 {
namespace {

constexpr float kShPi = 3.1415926536f;  // spherical_harmonics.slang:4

// octa_decode (packing.slang:77-86): rt3_device.hpp
// spherical_harmonics.slang:30-44 ; sh[r * 3 + c] = result[r][c]
RT3_DEV void sh3_evaluate(V3 d, float sh[9]) {
    sh[0] = 0.28209479177387814347403972578039f;
    sh[1] = -0.48860251190291992158638462283836f * d.y;
    sh[2] = 0.48860251190291992158638462283836f * d.z;
    sh[3] = -0.48860251190291992158638462283836f * d.x;
    sh[4] = 1.09254843059207907054338570580268f * d.x * d.y;
    sh[5] = 1.09254843059207907054338570580268f * d.y * d.z;
    sh[6] = 0.31539156525252000603089369029571f * (3.0f * d.z * d.z - 1.0f);
    sh[7] = 1.09254843059207907054338570580268f * d.x * d.z;
    sh[8] = 0.54627421529603953527169285290134f * (d.x * d.x - d.y * d.y);
}
// WaveActiveSum over the 64 lanes: butterfly with partner lane ^ 1, ^ 2, ... ^ 32 (every lane ends with the same bits)
RT3_DEV float wave_sum64(float v) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) v = v + __shfl_xor(v, off, 64);
    return v;
}
// math.slang:140-160: the bitonic network over a 64-entry groupshared array, one entry per lane; ascending
RT3_DEV void wave_sort64(float& key, uint32_t& idx) {
    const uint32_t i = __lane_id();
#pragma unroll
    for (uint32_t k = 2; k <= 64; k *= 2) {
#pragma unroll
        for (uint32_t j = k / 2; j > 0; j /= 2) {
            const float other_key = __shfl_xor(key, (int)j, 64);
            const uint32_t other_idx = (uint32_t)__shfl_xor((int)idx, (int)j, 64);
            const uint32_t l = i ^ j;
            const bool lower = i < l;  // the shader's `l > i` thread does the compare for the pair
            const float key_lo = lower ? key : other_key, key_hi = lower ? other_key : key;
            const uint32_t lo = lower ? i : l;
            const bool swap = ((lo & k) == 0) ? (key_lo > key_hi) : (key_lo < key_hi);
            if (swap) {
                key = other_key;
                idx = other_idx;
            }
        }
    }
}

RT3_DEV V3 world_pos_from_depth(const GConstDev& g, float depth, uint32_t px, uint32_t py) {  // gbuffer_helpers.slang:81-83
    V3 o, d;
    primary_ray(g, px, py, o, d);
    return o + d * depth;
}
RT3_DEV V3 gbuffer_normal(const uint4* gbuffer, size_t pi) {  // GbufferDataPacked::unpack_normal, gbuffer_helpers.slang:46-48
    return unpack_normal_11_10_11(reinterpret_cast<const uint32_t*>(gbuffer)[4 * pi + 1]);
}

// ------------------------------------------------------------------------------------------------ structured_importance_sampling
__global__ __launch_bounds__(64) void k_sis(uint32_t W, uint32_t AW, const uint4* __restrict__ gbuffer, uint16_t* __restrict__ out,
                                            float* __restrict__ debug) {
    __shared__ float s_n[256 * 3];
    const uint32_t ti = threadIdx.x, tx = blockIdx.x * 8 + (ti & 7), ty = blockIdx.y * 8 + (ti >> 3);
#pragma unroll
    for (uint32_t y = 0; y < 2; y++)
#pragma unroll
        for (uint32_t x = 0; x < 2; x++) {  // :24-30
            V3 n = gbuffer_normal(gbuffer, (size_t)(ty * 2 + y) * W + (tx * 2 + x));
            float* d = s_n + (ti * 4 + y * 2 + x) * 3;
            d[0] = n.x;
            d[1] = n.y;
            d[2] = n.z;
        }
    __syncthreads();
    const V3 dir = octa_decode(((float)(ti & 7) + 0.5f) / 8.0f, ((float)(ti >> 3) + 0.5f) / 8.0f);  // :33-34
    float pdf = 0.0f;
    for (int i = 0; i < 256; i++) pdf += fmax_sel(dot(v3(s_n[3 * i], s_n[3 * i + 1], s_n[3 * i + 2]), dir), 0.0f) / 256.0f;  // :36-39 (LDS broadcast reads)
    float key = pdf;
    uint32_t idx = ti;
    wave_sort64(key, idx);  // :45
    // :47-50 `brdf_pdf < 0` never holds, so index stays -1 and culled_rays 0: the threshold is the smallest pdf of the probe
    const float smallest = __shfl(key, 0, 64);
    const size_t ai = (size_t)ty * AW + tx;
    out[ai] = smallest < pdf ? (uint16_t)((1u << 15) | (ti * 4)) : (uint16_t)ti;  // :55-69
    debug[ai] = -1.0f;                                                             // :70
}

// ------------------------------------------------------------------------------------------------ trace_probes
__global__ void k_probe_raygen(GConstDev g, uint32_t W, uint32_t AW, uint32_t n, const float* __restrict__ depth,
                               const uint16_t* __restrict__ directions, float4* __restrict__ atlas, float* __restrict__ rays, size_t stride,
                               float2* __restrict__ d2) {
    for (uint32_t a = blockIdx.x * blockDim.x + threadIdx.x; a < n; a += gridDim.x * blockDim.x) {
        const uint32_t ax = a % AW, ay = a / AW, px = (ax / 8) * 16, py = (ay / 8) * 16;  // :17-24
        const float d0 = depth[(size_t)py * W + px];
        float4 ro = make_float4(0.0f, 0.0f, 0.0f, 0.0f), rd = make_float4(0.0f, 0.0f, 1.0f, -1.0f);  // inactive: TMax < TMin
        if (d0 == kBackgroundDepth) {  // :29-31
            atlas[a] = make_float4(0.0f, 0.0f, 0.0f, kBackgroundDepth);
            d2[a] = make_float2(-1.0f, -1.0f);
        } else {
            atlas[a] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // :33
            const uint32_t seed = rng_seed(ax, ay, g.frame);   // ray_rng, :21
            const uint32_t dw = directions[a], di = dw & 0x7FFFu, mip = dw >> 15, size = (1u << mip) * 8u;  // :41-44
            const float fx = (float)(di % size), fy = (float)(di / size), fs = (float)size;
            const float u0 = uniform_float(seed, 0), u1 = uniform_float(seed, 1);
            const V3 dir = octa_decode((fx + u0) / fs, (fy + u1) / fs);  // :47
            const V3 o = world_pos_from_depth(g, d0, px, py);           // :53
            ro = make_float4(o.x, o.y, o.z, 0.0005f);                   // :55
            rd = make_float4(dir.x, dir.y, dir.z, kBackgroundDepth);    // :56
            d2[a] = make_float2(fx / fs, fy / fs);
        }
        reinterpret_cast<float4*>(rays)[a] = ro;
        reinterpret_cast<float4*>(rays)[stride + a] = rd;
    }
}
// one wave per probe; lane = the probe's ray (local texel, row-major)
__global__ __launch_bounds__(64) void k_probe_store(SceneDev sc, uint32_t flags, float blend, uint32_t AW, const float* __restrict__ hits,
                                                    const float2* __restrict__ d2, const float4* __restrict__ prev, float4* __restrict__ atlas) {
    __shared__ int s_win[64];
    const uint32_t ti = threadIdx.x, ax = blockIdx.x * 8 + (ti & 7), ay = blockIdx.y * 8 + (ti >> 3);
    const size_t a = (size_t)ay * AW + ax;
    const float2 dd = d2[a];
    const bool active = dd.x >= 0.0f;
    const float4 h = reinterpret_cast<const float4*>(hits)[a];
    const uint32_t prim = __float_as_uint(h.w);
    const float probe_depth = prim == kMiss ? kBackgroundDepth : h.x;
    s_win[ti] = -1;
    __syncthreads();
    if (flags & RT3_FLAG_PROBE_RADIANCE) {  // the store trace_probes.slang:74 keeps in a comment
        if (active) {
            V3 rad = v3(0.0f, 0.0f, 0.0f);
            if (prim != kMiss) rad = hit_info(sc, prim, h.y, h.z).emissive;  // :59-62
            const float4 p = prev[a];
            atlas[a] = make_float4(p.x + (rad.x - p.x) * blend, p.y + (rad.y - p.y) * blend, p.z + (rad.z - p.z) * blend, probe_depth);
        }
        return;
    }
    // as written, :74: scatter to the texel the direction falls into; of several rays aiming at one texel the last thread wins
    const uint32_t lx = (uint32_t)(dd.x * 8.0f), ly = (uint32_t)(dd.y * 8.0f);
    const bool store = active && lx < 8u && ly < 8u;  // direction words outside the octahedral map (index >= size^2) store nothing
    const uint32_t slot = store ? ly * 8 + lx : 0;
    if (store) atomicMax(&s_win[slot], (int)ti);
    __syncthreads();
    if (store && s_win[slot] == (int)ti)
        atlas[(size_t)(blockIdx.y * 8 + ly) * AW + blockIdx.x * 8 + lx] = make_float4(dd.x, dd.y, 0.0f, probe_depth);
}

// ------------------------------------------------------------------------------------------------ spherical_harmonic_conversion
__global__ __launch_bounds__(64) void k_sh_conversion(uint32_t AW, const float4* __restrict__ atlas, float4* __restrict__ out) {
    const uint32_t ti = threadIdx.x;
    const V3 dir = octa_decode(((float)(ti & 7) + 0.5f) / 8.0f, ((float)(ti >> 3) + 0.5f) / 8.0f);  // :12-14
    const float4 col = atlas[(size_t)(blockIdx.y * 8 + (ti >> 3)) * AW + blockIdx.x * 8 + (ti & 7)];
    float sh[9];
    sh3_evaluate(dir, sh);
    const float factor = 4.0f * kShPi / 64.0f;  // :25
    const float c3[3] = {col.x, col.y, col.z};
#pragma unroll
    for (uint32_t c = 0; c < 3; c++) {
        float r[9];
#pragma unroll
        for (int k = 0; k < 9; k++) r[k] = wave_sum64(sh[k] * c3[c]) * factor;  // :20-22, :26-28
        if (ti == 0) {                                                           // WaveIsFirstLane, :24
            float4* o = out + 3 * (size_t)zcurve(blockIdx.x * 3 + c, blockIdx.y);  // :30-32 ; float3x3 = 3 rows padded to float4
            o[0] = make_float4(r[0], r[1], r[2], 0.0f);
            o[1] = make_float4(r[3], r[4], r[5], 0.0f);
            o[2] = make_float4(r[6], r[7], r[8], 0.0f);
        }
    }
}

// ------------------------------------------------------------------------------------------------ interpolate_probes
RT3_DEV float pow8(float x) {
    float x2 = x * x, x4 = x2 * x2;
    return x4 * x4;
}
// phase 1: the regular stores (:102); phase 2: the "interpolation failed" marks (:74), which land on the jittered pixel
__global__ void k_interpolate(GConstDev g, uint32_t W, uint32_t H, int phase, const uint4* __restrict__ gbuffer, const float* __restrict__ depth,
                              const float4* __restrict__ sh, float4* __restrict__ light) {
    const uint32_t n = W * H, NPX = W / 16, NPY = H / 16;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t px = i % W, py = i / W;
        const float pixel_depth = depth[i];
        if (pixel_depth == kBackgroundDepth) continue;  // :19-22
        const uint32_t seed = rng_seed(px, py, g.frame);
        const Surface surf = gbuffer_unpack(gbuffer[i]);
        const V3 pos = world_pos_from_depth(g, pixel_depth, px, py);
        const float u0 = uniform_float(seed, 0), u1 = uniform_float(seed, 1);
        const int jx = (int)((2.0f * u0 - 1.0f) * 16.0f), jy = (int)((2.0f * u1 - 1.0f) * 16.0f);  // :31
        int cx = (int)px + jx, cy = (int)py + jy;
        cx = cx < 0 ? 0 : (cx > (int)W - 1 ? (int)W - 1 : cx);
        cy = cy < 0 ? 0 : (cy > (int)H - 1 ? (int)H - 1 : cy);
        const V3 jpos = world_pos_from_depth(g, depth[(size_t)cy * W + cx], (uint32_t)cx, (uint32_t)cy);
        uint32_t qx = px, qy = py;
        if (fabsf(dot(normalize(jpos - pos), surf.normal)) < 0.01f) {  // :36-38
            qx = (uint32_t)cx;
            qy = (uint32_t)cy;
        }
        const uint32_t lpx = qx / 16, lpy = qy / 16;
        float w[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {  // :51-69
            const uint32_t cxp = lpx + (k & 1), cyp = lpy + (k >> 1);
            if (cxp >= NPX || cyp >= NPY) continue;
            const size_t ppi = (size_t)(cyp * 16) * W + cxp * 16;
            const float pd = depth[ppi];
            if (pd == kBackgroundDepth) continue;
            const V3 pp = world_pos_from_depth(g, pd, cxp * 16, cyp * 16);
            if (fabsf(dot(normalize(pp - pos), surf.normal)) > 0.01f) {
                w[k] = 0.0f;
            } else {
                float q = 1.0f - fabsf(pd - pixel_depth) / pixel_depth;
                q = fmin_sel(fmax_sel(q, 0.0f), 1.0f);
                q *= fmax_sel(dot(surf.normal, gbuffer_normal(gbuffer, ppi)), 0.0f);
                w[k] = pow8(q);
            }
        }
        if (w[0] * w[0] + w[1] * w[1] + w[2] * w[2] + w[3] * w[3] == 0.0f) {  // :72-76
            if (phase == 2) light[(size_t)qy * W + qx] = make_float4(1.0f, 0.0f, 0.0f, 1.0f);
            continue;
        }
        if (phase != 1) continue;
        const float wsum = w[0] + w[1] + w[2] + w[3];
        V3 acc = v3(0.0f, 0.0f, 0.0f);
        float lobe[9];
        sh3_evaluate(surf.normal, lobe);  // sh3TransformCosLobe, spherical_harmonics.slang:73-89
        lobe[0] *= kShPi;
        lobe[1] *= 2.0943951023931954923f;
        lobe[2] *= 2.0943951023931954923f;
        lobe[3] *= 2.0943951023931954923f;
#pragma unroll
        for (int k = 4; k < 9; k++) lobe[k] *= 0.7853981633974483096f;
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {  // :81-97
            const float wk = w[k] / wsum;
            if (wk == 0.0f) continue;
            const uint32_t cxp = lpx + (k & 1), cyp = lpy + (k >> 1);
            float pr[3];
            if (g.proberng == 1) {
                const V3 pn = gbuffer_normal(gbuffer, (size_t)(cyp * 16) * W + cxp * 16);
                pr[0] = (pn.x + 1.0f) / 2.0f;
                pr[1] = (pn.y + 1.0f) / 2.0f;
                pr[2] = (pn.z + 1.0f) / 2.0f;
            } else {
#pragma unroll
                for (uint32_t c = 0; c < 3; c++) {
                    const float4* e = sh + 3 * (size_t)zcurve(cxp * 3 + c, cyp);
                    const float4 r0 = e[0], r1 = e[1], r2 = e[2];
                    const float m[9] = {r0.x, r0.y, r0.z, r1.x, r1.y, r1.z, r2.x, r2.y, r2.z};
                    float s = m[0] * lobe[0];
#pragma unroll
                    for (int t = 1; t < 9; t++) s = s + m[t] * lobe[t];  // matrix_dot, spherical_harmonics.slang:59-63
                    pr[c] = s;
                }
            }
            acc.x += wk * fmax_sel(0.0f, pr[0]);
            acc.y += wk * fmax_sel(0.0f, pr[1]);
            acc.z += wk * fmax_sel(0.0f, pr[2]);
        }
        light[i] = make_float4(acc.x * (surf.albedo.x * kInvPi) + surf.emissive.x, acc.y * (surf.albedo.y * kInvPi) + surf.emissive.y,
                               acc.z * (surf.albedo.z * kInvPi) + surf.emissive.z, 1.0f);  // :99-102
    }
}

// ------------------------------------------------------------------------------------------------ self-test (ops 13..16)
__global__ void k_selftest_probe_scalar(int op, const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (op == 13) {
        V3 d = octa_decode(__uint_as_float(in[2 * i]), __uint_as_float(in[2 * i + 1]));
        out[3 * i] = __float_as_uint(d.x);
        out[3 * i + 1] = __float_as_uint(d.y);
        out[3 * i + 2] = __float_as_uint(d.z);
    } else {
        float sh[9];
        sh3_evaluate(v3(__uint_as_float(in[3 * i]), __uint_as_float(in[3 * i + 1]), __uint_as_float(in[3 * i + 2])), sh);
        for (int k = 0; k < 9; k++) out[9 * i + k] = __float_as_uint(sh[k]);
    }
}
__global__ __launch_bounds__(64) void k_selftest_probe_wave(int op, const uint32_t* __restrict__ in, uint32_t* __restrict__ out) {
    const uint32_t e = blockIdx.x, ti = threadIdx.x;
    float key = __uint_as_float(in[64 * e + ti]);
    if (op == 15) {
        uint32_t idx = ti;
        wave_sort64(key, idx);
        out[128 * e + ti] = __float_as_uint(key);
        out[128 * e + 64 + ti] = idx;
    } else {
        float s = wave_sum64(key);
        if (ti == 0) out[e] = __float_as_uint(s);
    }
}

inline unsigned blocks_for(uint64_t n, unsigned block, unsigned cap) {
    uint64_t b = (n + block - 1) / block;
    return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

void launch_sis(hipStream_t st, uint32_t W, uint32_t probes_x, uint32_t probes_y, const void* gbuffer, void* out, float* debug) {
    hipLaunchKernelGGL(k_sis, dim3(probes_x, probes_y), dim3(64), 0, st, W, probes_x * 8, (const uint4*)gbuffer, (uint16_t*)out, debug);
}
void launch_probe_raygen(hipStream_t st, const GConstDev& g, uint32_t W, uint32_t probes_x, uint32_t probes_y, const float* depth, const void* directions,
                         void* atlas, float* rays, size_t stride, void* d2) {
    const uint32_t n = probes_x * 8 * probes_y * 8;
    hipLaunchKernelGGL(k_probe_raygen, dim3(blocks_for(n, 256, 4096)), dim3(256), 0, st, g, W, probes_x * 8, n, depth, (const uint16_t*)directions,
                       (float4*)atlas, rays, stride, (float2*)d2);
}
void launch_probe_store(hipStream_t st, const SceneDev& sc, uint32_t flags, float blend, uint32_t probes_x, uint32_t probes_y, const float* hits,
                        const void* d2, const void* prev, void* atlas) {
    hipLaunchKernelGGL(k_probe_store, dim3(probes_x, probes_y), dim3(64), 0, st, sc, flags, blend, probes_x * 8, hits, (const float2*)d2,
                       (const float4*)prev, (float4*)atlas);
}
void launch_sh_conversion(hipStream_t st, uint32_t probes_x, uint32_t probes_y, const void* atlas, void* out) {
    hipLaunchKernelGGL(k_sh_conversion, dim3(probes_x, probes_y), dim3(64), 0, st, probes_x * 8, (const float4*)atlas, (float4*)out);
}
void launch_interpolate(hipStream_t st, const GConstDev& g, uint32_t W, uint32_t H, const void* gbuffer, const float* depth, const void* sh, void* light) {
    for (int phase = 1; phase <= 2; phase++)
        hipLaunchKernelGGL(k_interpolate, dim3(blocks_for((uint64_t)W * H, 256, 8192)), dim3(256), 0, st, g, W, H, phase, (const uint4*)gbuffer, depth,
                           (const float4*)sh, (float4*)light);
}
void launch_selftest_probes(hipStream_t st, int op, const uint32_t* in, uint32_t n, uint32_t* out) {
    if (op == 13 || op == 14) hipLaunchKernelGGL(k_selftest_probe_scalar, dim3((n + 255) / 256), dim3(256), 0, st, op, in, n, out);
    else hipLaunchKernelGGL(k_selftest_probe_wave, dim3(n), dim3(64), 0, st, op, in, out);
}

}  // namespace rt3
