// rt3_device.hpp -- device-side arithmetic of the gfx950 wavefront path tracer.
//
// Each function names the reference shader lines it implements (paths relative to DerEchteKarsten/RayTracer3).
// Arithmetic contract: fp32, compiled with -ffp-contract=off, IEEE divide/sqrt, min/max as explicit
// selects, transcendental functions only through the polynomials below -- so results are reproducible bit for bit
// on any IEEE-754 machine and can be checked exactly by the CPU oracle in tests.
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RT3_DEV __device__ __forceinline__

namespace rt3 {

constexpr float kPi = 3.14159265358979323846f;
constexpr float kTau = 6.28318530717958647692f;       // math.slang:3
constexpr float kInvPi = 0.318309886183790671538f;    // math.slang:4
constexpr float kHalfPi = 1.57079632679489661923f;
constexpr float kBackgroundDepth = 100000.0f;         // datatypes.slang:3
constexpr uint32_t kMiss = 0xFFFFFFFFu;
constexpr float kRayTMin = 0.001f;                    // refrence_mode.slang:31

struct V3 {
    float x, y, z;
};
RT3_DEV V3 v3(float x, float y, float z) { return V3{x, y, z}; }
RT3_DEV V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
RT3_DEV V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
RT3_DEV V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
RT3_DEV V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
RT3_DEV V3 neg(V3 a) { return V3{-a.x, -a.y, -a.z}; }
RT3_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT3_DEV V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
RT3_DEV V3 normalize(V3 a) {
    float inv = 1.0f / sqrtf(dot(a, a));
    return a * inv;
}
// exact n / d for 32-bit operands without a hardware divide (round-up magic number, Granlund-Montgomery / libdivide
// "branch-free"): q = umulhi(mul, n); n / d = (((n - q) >> 1) + q) >> shift.  A runtime integer division costs ~30 VALU.
struct FastDiv {
    uint32_t d, mul, shift;
};
__host__ __device__ inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f{d, 0u, 0u};
    if (d > 1u) {
        uint32_t l = 0;
        while (l < 32u && (1ull << l) < (unsigned long long)d) ++l;  // ceil(log2 d)
        f.mul = (uint32_t)((((1ull << l) - d) << 32) / d + 1ull);
        f.shift = l - 1u;
    }
    return f;
}
RT3_DEV uint32_t fast_div(const FastDiv& f, uint32_t n) {
    if (f.d <= 1u) return n;
    const uint32_t q = __umulhi(f.mul, n);
    return (((n - q) >> 1) + q) >> f.shift;
}
// ((x % W) + W) % W.  Texel neighbours of a coordinate in [0, 1] lie within one period of the image, where the wrap is a
// conditional add; the general form is kept for everything else.
RT3_DEV int wrap_index(int x, int W) {
    if ((uint32_t)(x + W) < 3u * (uint32_t)W) return x < 0 ? x + W : (x >= W ? x - W : x);
    return ((x % W) + W) % W;
}
RT3_DEV float fmin_sel(float a, float b) { return a < b ? a : b; }
RT3_DEV float fmax_sel(float a, float b) { return a > b ? a : b; }

// ------------------------------------------------------------------------------------------------ RNG
// random.slang:5-15
RT3_DEV uint32_t jenkins_hash(uint32_t a) {
    a = (a + 0x7ed55d16u) + (a << 12);
    a = (a ^ 0xc761c23cu) ^ (a >> 19);
    a = (a + 0x165667b1u) + (a << 5);
    a = (a + 0xd3a2646cu) ^ (a << 9);
    a = (a + 0xfd7046c5u) + (a << 3);
    a = (a ^ 0xb55a4f09u) ^ (a >> 16);
    return a;
}
// math.slang:105-117
RT3_DEV uint32_t integer_explode(uint32_t x) {
    x = (x | (x << 8)) & 0x00FF00FFu;
    x = (x | (x << 4)) & 0x0F0F0F0Fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
}
RT3_DEV uint32_t zcurve(uint32_t x, uint32_t y) { return integer_explode(x) | (integer_explode(y) << 1); }
// random.slang:42-46
RT3_DEV uint32_t rng_seed(uint32_t px, uint32_t py, uint32_t frame) { return jenkins_hash(zcurve(px, py)) + frame; }
// random.slang:49-79, counter passed explicitly (the stream is counter-based; see DESIGN.md for the index rule)
RT3_DEV uint32_t murmur3(uint32_t seed, uint32_t index) {
    uint32_t k = index * 0xcc9e2d51u;
    k = (k << 15) | (k >> 17);
    k *= 0x1b873593u;
    uint32_t h = seed ^ k;
    h = ((h << 13) | (h >> 19)) * 5u + 0xe6546b64u;
    h ^= 4u;
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}
// random.slang:82-89
RT3_DEV float uniform_float(uint32_t seed, uint32_t index) {
    return __uint_as_float((murmur3(seed, index) & 0x007FFFFFu) | 0x3F800000u) - 1.0f;
}
// Cranley-Patterson shift by one blue-noise byte (north_star): frac(u + c/256)
RT3_DEV float bluenoise_shift(float u, uint32_t c) {
    float r = u + (float)c * 0.00390625f;
    return r >= 1.0f ? r - 1.0f : r;
}

// ------------------------------------------------------------------------------------------------ packing.slang
RT3_DEV float unpack_unorm(uint32_t p, uint32_t bits) {  // :2-5
    uint32_t maxv = (1u << bits) - 1u;
    return (float)(p & maxv) / (float)maxv;
}
RT3_DEV uint32_t pack_unorm(float v, uint32_t bits) {  // :7-10
    uint32_t maxv = (1u << bits) - 1u;
    float c = fmin_sel(fmax_sel(v, 0.0f), 1.0f);
    return (uint32_t)(c * (float)maxv + 0.5f);
}
RT3_DEV uint32_t pack_normal_11_10_11(V3 n) {  // :12-18
    return pack_unorm(n.x * 0.5f + 0.5f, 11) + (pack_unorm(n.y * 0.5f + 0.5f, 10) << 11) + (pack_unorm(n.z * 0.5f + 0.5f, 11) << 21);
}
RT3_DEV V3 unpack_normal_11_10_11(uint32_t p) {  // :20-27
    return normalize(v3(unpack_unorm(p, 11) * 2.0f - 1.0f, unpack_unorm(p >> 11, 10) * 2.0f - 1.0f, unpack_unorm(p >> 21, 11) * 2.0f - 1.0f));
}
RT3_DEV uint32_t pack_color_888(V3 c) {  // :46-53
    return pack_unorm(sqrtf(c.x), 8) + (pack_unorm(sqrtf(c.y), 8) << 8) + (pack_unorm(sqrtf(c.z), 8) << 16);
}
RT3_DEV V3 unpack_color_888(uint32_t p) {  // :55-62
    V3 c = v3(unpack_unorm(p, 8), unpack_unorm(p >> 8, 8), unpack_unorm(p >> 16, 8));
    return c * c;
}
RT3_DEV uint32_t f32_to_f16_bits(float f) { return (uint32_t)__half_as_ushort(__float2half_rn(f)); }
RT3_DEV float f16_bits_to_f32(uint32_t h) { return __half2float(__ushort_as_half((unsigned short)h)); }
RT3_DEV uint32_t pack_2x16f(float a, float b) { return f32_to_f16_bits(a) | (f32_to_f16_bits(b) << 16); }  // :88-90
RT3_DEV float exp2_int(int e) { return __uint_as_float((uint32_t)(e + 127) << 23); }
RT3_DEV uint32_t float3_to_rgb9e5(V3 c) {  // :99-144
    const float max_rgb9e5 = (511.0f / 512.0f) * 65536.0f;
    float rc = fmin_sel(fmax_sel(c.x, 0.0f), max_rgb9e5), gc = fmin_sel(fmax_sel(c.y, 0.0f), max_rgb9e5), bc = fmin_sel(fmax_sel(c.z, 0.0f), max_rgb9e5);
    float maxrgb = fmax_sel(rc, fmax_sel(gc, bc));
    int fl2 = (int)((__float_as_uint(maxrgb) & 0x7F800000u) >> 23) - 127;
    int exp_shared = (fl2 > -16 ? fl2 : -16) + 1 + 15;
    float denom = exp2_int(exp_shared - 15 - 9);
    int maxm = (int)floorf(maxrgb / denom + 0.5f);
    if (maxm == 512) {
        denom *= 2.0f;
        exp_shared += 1;
    }
    int rm = (int)floorf(rc / denom + 0.5f), gm = (int)floorf(gc / denom + 0.5f), bm = (int)floorf(bc / denom + 0.5f);
    return ((uint32_t)rm << 23) | ((uint32_t)gm << 14) | ((uint32_t)bm << 5) | (uint32_t)exp_shared;
}
RT3_DEV V3 rgb9e5_to_float3(uint32_t v) {  // :146-162
    float scale = exp2_int((int)(v & 31u) - 24);
    return v3((float)((v >> 23) & 511u) * scale, (float)((v >> 14) & 511u) * scale, (float)((v >> 5) & 511u) * scale);
}

// gbuffer_helpers.slang:5-71
struct Surface {
    V3 albedo, emissive, normal;
    float roughness, metalness;
};
RT3_DEV uint4 gbuffer_pack(const Surface& s) {  // :22-34
    return make_uint4(pack_color_888(s.albedo), pack_normal_11_10_11(s.normal), pack_2x16f(sqrtf(s.roughness), s.metalness),
                      float3_to_rgb9e5(s.emissive));
}
RT3_DEV Surface gbuffer_unpack(uint4 p) {  // :59-70
    Surface s;
    s.albedo = unpack_color_888(p.x);
    s.normal = unpack_normal_11_10_11(p.y);
    float pr = f16_bits_to_f32(p.z & 0xFFFFu);
    s.roughness = pr * pr;
    s.metalness = f16_bits_to_f32((p.z >> 16) & 0xFFFFu);
    s.emissive = rgb9e5_to_float3(p.w);
    return s;
}

// ------------------------------------------------------------------------------------------------ math
// sin(2 pi u), cos(2 pi u), u in [0,1): exact quadrant split, odd/even Taylor polynomials on [0, pi/4]
RT3_DEV void sincos_2pi(float u, float& s_out, float& c_out) {
    float x = u * 4.0f;
    int q = (int)x;
    float r = x - (float)q;
    bool sw = r > 0.5f;
    if (sw) r = 1.0f - r;
    float a = r * kHalfPi, a2 = a * a;
    float s = a * (1.0f + a2 * (-1.6666667163e-01f + a2 * (8.3333337680e-03f + a2 * (-1.9841270114e-04f + a2 * 2.7557314297e-06f))));
    float c = 1.0f + a2 * (-0.5f + a2 * (4.1666667908e-02f + a2 * (-1.3888889225e-03f + a2 * (2.4801587642e-05f + a2 * -2.7557314297e-07f))));
    float ss = sw ? c : s, cc = sw ? s : c;
    q &= 3;
    s_out = q == 0 ? ss : (q == 1 ? cc : (q == 2 ? -ss : -cc));
    c_out = q == 0 ? cc : (q == 1 ? -ss : (q == 2 ? -cc : ss));
}
RT3_DEV float atan2_poly(float y, float x) {
    float ax = x < 0.0f ? -x : x, ay = y < 0.0f ? -y : y;
    float mx = fmax_sel(ax, ay), mn = fmin_sel(ax, ay);
    if (mx == 0.0f) return 0.0f;
    float a = mn / mx, s = a * a;
    float r = a * (0.99997726f + s * (-0.33262347f + s * (0.19354346f + s * (-0.11643287f + s * (0.05265332f + s * -0.01172120f)))));
    if (ay > ax) r = kHalfPi - r;
    if (x < 0.0f) r = kPi - r;
    if (y < 0.0f) r = -r;
    return r;
}
// math.slang:6-12
RT3_DEV void direction_to_equirect_uv(V3 d, float& u, float& v) {
    float as = atan2_poly(d.y, sqrtf(fmax_sel(0.0f, 1.0f - d.y * d.y)));
    u = 0.5f + atan2_poly(d.z, d.x) / kTau;
    v = 0.5f - as / kPi;
}
// math.slang:29-50 ; columns b1, b2 (third column is n)
RT3_DEV void build_orthonormal_basis(V3 n, V3& b1, V3& b2) {
    if (n.z < 0.0f) {
        const float a = 1.0f / (1.0f - n.z);
        const float b = n.x * n.y * a;
        b1 = v3(1.0f - n.x * n.x * a, -b, n.x);
        b2 = v3(b, n.y * n.y * a - 1.0f, -n.y);
    } else {
        const float a = 1.0f / (1.0f + n.z);
        const float b = -n.x * n.y * a;
        b1 = v3(1.0f - n.x * n.x * a, b, -n.x);
        b2 = v3(b, 1.0f - n.y * n.y * a, -n.y);
    }
}
// mul(tangent_to_world, wi), refrence_mode.slang:48
RT3_DEV V3 basis_apply(V3 b1, V3 b2, V3 n, V3 w) {
    return v3(b1.x * w.x + b2.x * w.y + n.x * w.z, b1.y * w.x + b2.y * w.y + n.y * w.z, b1.z * w.x + b2.z * w.y + n.z * w.z);
}
// brdf.slang:56-65 DiffuseBrdf::sample direction
RT3_DEV V3 diffuse_sample(float u0, float u1) {
    float sp, cp;
    sincos_2pi(u0, sp, cp);
    float cos_theta = sqrtf(fmax_sel(0.0f, 1.0f - u1));
    float sin_theta = sqrtf(fmax_sel(0.0f, 1.0f - cos_theta * cos_theta));
    return v3(cp * sin_theta, sp * sin_theta, cos_theta);
}

// ------------------------------------------------------------------------------------------------ layered BSDF
// DiffuseBrdf (brdf.slang:52-93) under a GGX SpecularBrdf (brdf.slang:141-311: VNDF sampling, height-correlated Smith,
// Schlick with f90 = 1).  f0 = lerp(0.04, albedo, metalness); diffuse albedo = albedo (1 - metalness), attenuated by the
// transmitted fraction (1 - F).  One lobe is picked with probability p_spec and the sample is weighted by f / pdf of the
// mixture.  pdfs are with respect to PROJECTED solid angle (brdf.slang:31); directions live in the tangent frame.
struct Bsdf {
    V3 da, f0;
    float alpha, p_spec;
};
RT3_DEV float luminance(V3 c) { return c.x * 0.299f + c.y * 0.587f + c.z * 0.114f; }  // math.slang:119-122
RT3_DEV float pow5(float x) {
    float x2 = x * x;
    return x2 * x2 * x;
}
RT3_DEV float g_smith_ggx1(float ndotv, float a2) {  // brdf.slang:111-114
    float tan2_v = (1.0f - ndotv * ndotv) / (ndotv * ndotv);
    return 2.0f / (1.0f + sqrtf(1.0f + a2 * tan2_v));
}
RT3_DEV float g_smith_ggx_correlated(float ndotv, float ndotl, float a2) {  // brdf.slang:104-109
    float lambda_v = ndotl * sqrtf((-ndotv * a2 + ndotv) * ndotv + a2);
    float lambda_l = ndotv * sqrtf((-ndotl * a2 + ndotl) * ndotl + a2);
    return 2.0f * ndotl * ndotv / (lambda_v + lambda_l);
}
RT3_DEV float ggx_ndf(float a2, float cos_theta) {  // brdf.slang:146-149
    float denom_sqrt = cos_theta * cos_theta * (a2 - 1.0f) + 1.0f;
    return a2 / (kPi * denom_sqrt * denom_sqrt);
}
RT3_DEV Bsdf bsdf_setup(V3 albedo, float roughness, float metalness) {
    Bsdf b;
    b.f0 = v3(0.04f + (albedo.x - 0.04f) * metalness, 0.04f + (albedo.y - 0.04f) * metalness, 0.04f + (albedo.z - 0.04f) * metalness);
    b.da = v3(albedo.x * (1.0f - metalness), albedo.y * (1.0f - metalness), albedo.z * (1.0f - metalness));
    b.alpha = fmax_sel(roughness, 0.05f);
    float ls = luminance(b.f0), ld = luminance(b.da);
    float p = (ls + ld) > 0.0f ? ls / (ls + ld) : 1.0f;
    b.p_spec = ld > 0.0f ? fmin_sel(fmax_sel(p, 0.1f), 0.9f) : 1.0f;
    return b;
}
// BRDF value (without the cosine) and mixture pdf (projected solid angle)
RT3_DEV void bsdf_eval(const Bsdf& b, V3 wo, V3 wi, V3& value, float& pdf_proj) {
    value = v3(0.0f, 0.0f, 0.0f);
    pdf_proj = 0.0f;
    if (!(wi.z > 0.0f)) return;
    if (!(wo.z > 1e-5f)) {  // grazing / back-facing view: diffuse only
        value = v3(b.da.x * kInvPi, b.da.y * kInvPi, b.da.z * kInvPi);
        pdf_proj = kInvPi;
        return;
    }
    const float a2 = b.alpha * b.alpha;
    V3 h = normalize(v3(wo.x + wi.x, wo.y + wi.y, wo.z + wi.z));  // brdf.slang:268
    float vh = dot(wi, h);
    float fr = pow5(fmax_sel(0.0f, 1.0f - vh));  // eval_fresnel_schlick, brdf.slang:95-97
    float G = g_smith_ggx_correlated(wo.z, wi.z, a2), D = ggx_ndf(a2, h.z);
    float pdf_h = g_smith_ggx1(wo.z, a2) * D * fmax_sel(0.0f, dot(wo, h)) / wo.z;  // pdf_ggx_vn, :161-165
    float pdf_spec = vh > 0.0f ? pdf_h * (1.0f / (4.0f * vh)) / wi.z : 0.0f;      // :278,286
    float spec_scale = G * D / (4.0f * wo.z * wi.z);                               // :302-306
    float Fx = b.f0.x + (1.0f - b.f0.x) * fr, Fy = b.f0.y + (1.0f - b.f0.y) * fr, Fz = b.f0.z + (1.0f - b.f0.z) * fr;
    value = v3(Fx * spec_scale + b.da.x * kInvPi * (1.0f - Fx), Fy * spec_scale + b.da.y * kInvPi * (1.0f - Fy),
               Fz * spec_scale + b.da.z * kInvPi * (1.0f - Fz));
    pdf_proj = b.p_spec * pdf_spec + (1.0f - b.p_spec) * kInvPi;
}
// sample_vndf (brdf.slang:187-216) -> half vector
RT3_DEV V3 sample_vndf(float alpha, V3 wo, float u0, float u1) {
    V3 Vh = normalize(v3(alpha * wo.x, alpha * wo.y, wo.z));
    V3 T1 = v3(1.0f, 0.0f, 0.0f);
    if (Vh.z < 0.9999f) T1 = normalize(v3(-Vh.y, Vh.x, 0.0f));  // cross((0,0,1), Vh)
    V3 T2 = cross(Vh, T1);
    float r = sqrtf(u0), sp, cp;
    sincos_2pi(u1, sp, cp);
    float t1 = r * cp, t2 = r * sp, sv = 0.5f * (1.0f + Vh.z);
    t2 = (1.0f - sv) * sqrtf(1.0f - t1 * t1) + sv * t2;
    float nz = sqrtf(fmax_sel(0.0f, 1.0f - t1 * t1 - t2 * t2));
    V3 Nh = v3(t1 * T1.x + t2 * T2.x + nz * Vh.x, t1 * T1.y + t2 * T2.y + nz * Vh.y, t1 * T1.z + t2 * T2.z + nz * Vh.z);
    return normalize(v3(alpha * Nh.x, alpha * Nh.y, fmax_sel(0.0f, Nh.z)));
}
// false if the sample is invalid (the path ends); else wi, value_over_pdf and the solid-angle pdf
RT3_DEV bool bsdf_sample(const Bsdf& b, V3 wo, float u0, float u1, float u2, V3& wi, V3& vop, float& pdf_solid) {
    if (wo.z > 1e-5f && u2 < b.p_spec) {
        V3 h = sample_vndf(b.alpha, wo, u0, u1);
        float s2 = 2.0f * dot(wo, h);
        wi = v3(s2 * h.x - wo.x, s2 * h.y - wo.y, s2 * h.z - wo.z);  // reflect(-wo, m)
        if (h.z <= 1e-5f || wi.z <= 1e-5f) return false;               // BRDF_SAMPLING_MIN_COS, brdf.slang:227
    } else {
        wi = diffuse_sample(u0, u1);
    }
    V3 value;
    float pdf;
    bsdf_eval(b, wo, wi, value, pdf);
    if (!(pdf > 0.0f)) return false;
    vop = v3(value.x / pdf, value.y / pdf, value.z / pdf);
    pdf_solid = pdf * wi.z;
    return true;
}

// ------------------------------------------------------------------------------------------------ camera / primary ray
struct GConstDev {  // == rt3_gconst (renderer/mod.rs:47-63)
    float proj[16], view[16], proj_inverse[16], view_inverse[16];
    float window_size[2];
    uint32_t frame;
    float blendfactor;
    uint32_t bounces, samples, proberng;
    float cell_size;
    uint32_t mouse[2], pad[2];
};
static_assert(sizeof(GConstDev) == 304, "GConst layout");

// gbuffer_helpers.slang:85-103 (view_dir + setupPrimaryRay); pixel (0,0) top-left, upright image (d.y flipped)
RT3_DEV void primary_ray(const GConstDev& g, uint32_t px, uint32_t py, V3& o, V3& d) {
    float cx = ((float)px + 0.5f) / g.window_size[0], cy = ((float)py + 0.5f) / g.window_size[1];
    float dx = cx * 2.0f - 1.0f, dy = -(cy * 2.0f - 1.0f);
    const float* m = g.proj_inverse;
    V3 target = v3(m[0] * dx + m[4] * dy + m[8] * 1.0f + m[12] * 1.0f, m[1] * dx + m[5] * dy + m[9] * 1.0f + m[13] * 1.0f,
                   m[2] * dx + m[6] * dy + m[10] * 1.0f + m[14] * 1.0f);
    V3 t = normalize(target);
    const float* w = g.view_inverse;
    d = v3(w[0] * t.x + w[4] * t.y + w[8] * t.z + w[12] * 0.0f, w[1] * t.x + w[5] * t.y + w[9] * t.z + w[13] * 0.0f,
           w[2] * t.x + w[6] * t.y + w[10] * t.z + w[14] * 0.0f);
    o = v3(w[12], w[13], w[14]);
}

// ------------------------------------------------------------------------------------------------ scene access
struct GeometryInfoDev {  // datatypes.slang:11-19 padded to 64 B
    float base_color[4];
    int32_t tex;
    float metallic;
    uint32_t index_offset, vertex_offset;
    float emission[4];
    float roughness;
    uint32_t pad[3];
};
static_assert(sizeof(GeometryInfoDev) == 64, "GeometryInfo layout");
// One entry per (instance, geometry) pair of the world (world/mod.rs:34-60: InstanceInfo{mesh_index, transform} + Transform{Mat4}):
// rt3_accel_build flattens the instances into world-space triangles (the 3 ms device build stands in for the TLAS), so the builder
// needs each pair's index / vertex offsets and its matrix, and hit_info its material and the matrix's upper 3 x 3 (hit_logic.slang:23).
struct FlatGeomDev {
    GeometryInfoDev g;
    float m[12];        // column-major 3 x 4: x_axis, y_axis, z_axis, w_axis (glam Mat4 columns without their last row)
    uint32_t identity;  // 1: the instance matrix is exactly the identity -- positions and normals are used as uploaded
    uint32_t geom, instance, pad;
};
static_assert(sizeof(FlatGeomDev) == 128, "FlatGeom layout");
// What hit_info reads of a flattened geometry, 80 bytes: k_shade keeps the whole table in LDS when it has at most kShadeGeomsLds entries
// ("material table in LDS when <= a few hundred"), so that a hit costs ONE dependent global gather (its shading record)
struct ShadeGeomDev {
    float base_color[3];
    int32_t tex;
    float metallic, roughness;
    float emission[3];
    uint32_t identity;
    float m[9];  // upper 3 x 3 of the instance matrix, column-major
    uint32_t pad;
};
static_assert(sizeof(ShadeGeomDev) == 80, "ShadeGeom layout");
constexpr uint32_t kShadeGeomsLds = 256;
RT3_DEV V3 transform_point(const float* m, V3 p) {  // glam Mat4::transform_point3: ((x_axis * x + y_axis * y) + z_axis * z) + w_axis
    return v3(m[9] + (m[6] * p.z + (m[3] * p.y + m[0] * p.x)), m[10] + (m[7] * p.z + (m[4] * p.y + m[1] * p.x)), m[11] + (m[8] * p.z + (m[5] * p.y + m[2] * p.x)));
}
RT3_DEV V3 transform_vector(const float* m9, V3 v) {  // mul(transform, float4(v, 0)).xyz, hit_logic.slang:23
    return v3(m9[6] * v.z + (m9[3] * v.y + m9[0] * v.x), m9[7] * v.z + (m9[4] * v.y + m9[1] * v.x), m9[8] * v.z + (m9[5] * v.y + m9[2] * v.x));
}

struct SceneDev {
    const float* verts;          // interleaved p n t (8 floats)
    const uint32_t* indices;
    const FlatGeomDev* geoms;    // one per (instance, geometry), in instance order
    const ShadeGeomDev* shade_geoms;  // the same table as hit_info needs it (80-byte entries)
    uint32_t n_geoms;
    const uint32_t* prim_geom;   // global primitive -> geometry
    const uint32_t* first_prim;  // geometry -> first global primitive
    const uint4* tri_shade;      // per global primitive, 16 B: the three vertex normals, octahedral 2 x 16 bit each, + the flattened geometry index
    const float2* tri_uv;        // per global primitive, 3 x float2: the vertex uvs (read only for textured geometries)
    const uint8_t* tex_pixels;   // all base-colour textures, RGBA8 (sRGB-encoded colour), back to back
    const uint4* tex_table;      // per texture {byte offset, width, height, -}
    const float* srgb_lut;       // 256 entries: sRGB EOTF
    uint32_t n_tex;
    // sky
    const uint2* sky;            // 8-byte texels {RGB9E5 radiance, pdf_uv}, in 4 x 4 texel tiles of 128 bytes = one cache line
    const uint32_t* sky_alias;   // sky_w x sky_h words, row-major: per-row alias tables, q16 | alias column << 16
    const float* cdf_marg;       // padded: {0, cdf[0..h-1], 2, 2, 2}
    const uint32_t* guide_marg;  // sky_h cells: lo | hi << 16 = search bounds of the cell's answers
    uint32_t sky_w, sky_h, sky_wt;  // sky_wt = tiles per tile row = ceil(sky_w / 4)
    const uint8_t* bluenoise;
    uint32_t bn_w, bn_h;
};

// hit_logic.slang:5-40 (transform = identity, vertex colour = 1).  The three index + three vertex gathers of
// :10-20 are folded at build time into one 64-byte shading record per primitive (same values, one cache line).
// Textures[i].SampleLevel(uv, 0).xyz (hit_logic.slang:32): sRGB decode per texel, bilinear, repeat addressing, mip 0
RT3_DEV V3 texture_sample(const SceneDev& sc, uint32_t index, float u, float v) {
    const uint4 t = sc.tex_table[index];
    const int W = (int)t.y, H = (int)t.z;
    const uint8_t* px = sc.tex_pixels + t.x;
    float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
    float xf = floorf(x), yf = floorf(y), fx = x - xf, fy = y - yf;
    int x0 = (int)xf, y0 = (int)yf, x1 = x0 + 1, y1 = y0 + 1;
    x0 = wrap_index(x0, W);
    x1 = wrap_index(x1, W);
    y0 = wrap_index(y0, H);
    y1 = wrap_index(y1, H);
    const uint32_t p00 = *reinterpret_cast<const uint32_t*>(px + 4 * ((size_t)y0 * W + x0)), p10 = *reinterpret_cast<const uint32_t*>(px + 4 * ((size_t)y0 * W + x1));
    const uint32_t p01 = *reinterpret_cast<const uint32_t*>(px + 4 * ((size_t)y1 * W + x0)), p11 = *reinterpret_cast<const uint32_t*>(px + 4 * ((size_t)y1 * W + x1));
    float o[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float top = sc.srgb_lut[(p00 >> (8 * k)) & 0xFFu] * (1.0f - fx) + sc.srgb_lut[(p10 >> (8 * k)) & 0xFFu] * fx;
        float bot = sc.srgb_lut[(p01 >> (8 * k)) & 0xFFu] * (1.0f - fx) + sc.srgb_lut[(p11 >> (8 * k)) & 0xFFu] * fx;
        o[k] = top * (1.0f - fy) + bot * fy;
    }
    return v3(o[0], o[1], o[2]);
}
// packing.slang:64-86: the reference's octahedral map.  Vertex normals live in the shading records through it, 16 bits per
// coordinate (the oracle's tri_shade defines the same representation: a normal IS octa_decode16(octa_encode16(n)) on both sides).
RT3_DEV V3 octa_decode(float fx, float fy) {  // :77-86
    fx = fx * 2.0f - 1.0f;
    fy = fy * 2.0f - 1.0f;
    V3 n = v3(fx, fy, 1.0f - fabsf(fx) - fabsf(fy));
    float t = fmin_sel(fmax_sel(-n.z, 0.0f), 1.0f);
    n.x -= ((n.x >= 0.0f ? 1.0f : 0.0f) * 2.0f - 1.0f) * t;
    n.y -= ((n.y >= 0.0f ? 1.0f : 0.0f) * 2.0f - 1.0f) * t;
    return normalize(n);
}
RT3_DEV uint32_t octa_encode16(V3 n) {  // :64-75, then 16-bit unorm per coordinate (round to nearest); a zero vector encodes +z
    const float s = fabsf(n.x) + fabsf(n.y) + fabsf(n.z);
    if (!(s > 0.0f) || !(s <= 3.4028234663852886e38f)) return 0x80008000u;  // (0.5, 0.5) -> +z
    float x = n.x / s, y = n.y / s;
    const float z = n.z / s;
    if (z < 0.0f) {  // octa_wrap
        const float wx = (1.0f - fabsf(y)) * ((x >= 0.0f ? 1.0f : 0.0f) * 2.0f - 1.0f);
        const float wy = (1.0f - fabsf(x)) * ((y >= 0.0f ? 1.0f : 0.0f) * 2.0f - 1.0f);
        x = wx;
        y = wy;
    }
    x = x * 0.5f + 0.5f;
    y = y * 0.5f + 0.5f;
    const uint32_t qx = (uint32_t)(fmin_sel(fmax_sel(x, 0.0f), 1.0f) * 65535.0f + 0.5f), qy = (uint32_t)(fmin_sel(fmax_sel(y, 0.0f), 1.0f) * 65535.0f + 0.5f);
    return qx | (qy << 16);
}
RT3_DEV V3 octa_decode16(uint32_t w) { return octa_decode((float)(w & 0xFFFFu) * (1.0f / 65535.0f), (float)(w >> 16) * (1.0f / 65535.0f)); }

// In two steps so that a caller can put independent work (the light sample's table gathers) between the issue of the
// shading-record load and its use.  The record is 16 bytes {n0, n1, n2, geometry}: a quarter of round 2's 64-byte record, i.e.
// a 4 MB table for 260 k triangles instead of 16.6 MB (an L2 miss costs a 128-byte line whatever the record's size).
struct HitRecord {
    uint4 rec;
    uint32_t prim;
};
RT3_DEV HitRecord hit_fetch(const SceneDev& sc, uint32_t prim) { return HitRecord{sc.tri_shade[prim], prim}; }
// `geoms`: the ShadeGeomDev table -- sc.shade_geoms, or the caller's LDS copy of it
RT3_DEV Surface hit_finish(const SceneDev& sc, const ShadeGeomDev* geoms, const HitRecord& h, float bu, float bv) {
    const ShadeGeomDev& gi = geoms[h.rec.w];
    const V3 n0 = octa_decode16(h.rec.x), n1 = octa_decode16(h.rec.y), n2 = octa_decode16(h.rec.z);
    float b0 = 1.0f - bu - bv;
    V3 n = v3(n0.x * b0 + n1.x * bu + n2.x * bv, n0.y * b0 + n1.y * bu + n2.y * bv, n0.z * b0 + n1.z * bu + n2.z * bv);
    n = normalize(n);                                   // :22
    if (!gi.identity) n = transform_vector(gi.m, n);    // :23 mul(geometryInfo.transform, float4(normal, 0.0)).xyz
    n = normalize(n);                                   // :23
    Surface s;
    s.albedo = v3(gi.base_color[0], gi.base_color[1], gi.base_color[2]);
    if (gi.tex > -1 && (uint32_t)gi.tex < sc.n_tex) {  // :27,31-33
        const float2 t0 = sc.tri_uv[3 * (size_t)h.prim], t1 = sc.tri_uv[3 * (size_t)h.prim + 1], t2 = sc.tri_uv[3 * (size_t)h.prim + 2];
        float uu = t0.x * b0 + t1.x * bu + t2.x * bv, vv = t0.y * b0 + t1.y * bu + t2.y * bv;
        s.albedo = s.albedo * texture_sample(sc, (uint32_t)gi.tex, uu, vv);
    }
    s.emissive = v3(gi.emission[0] * 12.0f, gi.emission[1] * 12.0f, gi.emission[2] * 12.0f);  // :36
    s.normal = n;
    s.roughness = gi.roughness;
    s.metalness = gi.metallic;
    return s;
}
RT3_DEV Surface hit_info(const SceneDev& sc, uint32_t prim, float bu, float bv) { return hit_finish(sc, sc.shade_geoms, hit_fetch(sc, prim), bu, bv); }

// ------------------------------------------------------------------------------------------------ sky (north_star)
// Texels are 8 bytes {RGB9E5 radiance, pdf_uv as f32}: the importance-sampling density of a texel travels with its colour (the
// texel is always one of the four bilinear corners), and a 4 x 4 texel tile is exactly one 128-byte line -- the unit the fabric
// moves whatever a lane asks for (profiles/r02_fetch_calibration.md): the 2 x 2 bilinear footprint costs 1.56 lines on average
// instead of 2.25 with row-major 16-byte texels.
RT3_DEV uint32_t sky_texel_index(const SceneDev& sc, int x, int y) {
    return (((uint32_t)y >> 2) * sc.sky_wt + ((uint32_t)x >> 2)) * 16u + ((((uint32_t)y & 3u) << 2) | ((uint32_t)x & 3u));
}
RT3_DEV V3 sky_eval_pdf(const SceneDev& sc, float u, float v, int tx, int ty, float& pdf_texel) {
    int W = (int)sc.sky_w, H = (int)sc.sky_h;
    float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
    float xf = floorf(x), yf = floorf(y);
    float fx = x - xf, fy = y - yf;
    int x0 = (int)xf, y0 = (int)yf, x1 = x0 + 1, y1 = y0 + 1;
    x0 = wrap_index(x0, W);
    x1 = wrap_index(x1, W);
    y0 = y0 < 0 ? 0 : (y0 > H - 1 ? H - 1 : y0);
    y1 = y1 < 0 ? 0 : (y1 > H - 1 ? H - 1 : y1);
    const uint2 t00 = sc.sky[sky_texel_index(sc, x0, y0)], t10 = sc.sky[sky_texel_index(sc, x1, y0)];
    const uint2 t01 = sc.sky[sky_texel_index(sc, x0, y1)], t11 = sc.sky[sky_texel_index(sc, x1, y1)];
    if (tx >= 0) {
        const bool in_x = tx == x0 || tx == x1, in_y = ty == y0 || ty == y1;
        pdf_texel = __uint_as_float(ty == y0 ? (tx == x0 ? t00.y : t10.y) : (tx == x0 ? t01.y : t11.y));
        if (!(in_x && in_y)) pdf_texel = __uint_as_float(sc.sky[sky_texel_index(sc, tx, ty)].y);  // not reached for (u, v) inside texel (tx, ty)
    }
    const V3 p00 = rgb9e5_to_float3(t00.x), p10 = rgb9e5_to_float3(t10.x), p01 = rgb9e5_to_float3(t01.x), p11 = rgb9e5_to_float3(t11.x);
    const float a[3] = {p00.x, p00.y, p00.z}, bq[3] = {p10.x, p10.y, p10.z}, c[3] = {p01.x, p01.y, p01.z}, dq[3] = {p11.x, p11.y, p11.z};
    float o[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float top = a[k] * (1.0f - fx) + bq[k] * fx, bot = c[k] * (1.0f - fx) + dq[k] * fx;
        o[k] = top * (1.0f - fy) + bot * fy;
    }
    return v3(o[0], o[1], o[2]);
}
RT3_DEV V3 sky_eval(const SceneDev& sc, float u, float v) {  // Skybox.SampleLevel(uv, 0): bilinear, wrap u, clamp v
    if (!sc.sky) return v3(0.0f, 0.0f, 0.0f);
    float unused;
    return sky_eval_pdf(sc, u, v, -1, -1, unused);
}
// radiance and solid-angle pdf of the sky sampler for a direction that left the scene (equirect coordinates u, v)
RT3_DEV V3 sky_eval_and_pdf(const SceneDev& sc, float u, float v, float& pdf) {
    int W = (int)sc.sky_w, H = (int)sc.sky_h;
    int ix = (int)(u * (float)W), iy = (int)(v * (float)H);
    ix = ix < 0 ? 0 : (ix > W - 1 ? W - 1 : ix);
    iy = iy < 0 ? 0 : (iy > H - 1 ? H - 1 : iy);
    float pt;
    V3 rad = sky_eval_pdf(sc, u, v, ix, iy, pt);
    float st, ct;
    sincos_2pi(v * 0.5f, st, ct);
    pdf = st > 0.0f ? pt / (2.0f * kPi * kPi * st) : 0.0f;
    return rad;
}
// The oracle's cdf_find -- first i with cdf[i] > u -- plus the bracket {cdf[i-1] (0 for i = 0), cdf[i]}, in two memory
// round trips: the guide cell of u gives bounds [lo, hi] around the answer; when they are at most two apart (81-93 % of
// the lookups on the bench sky) ONE unaligned 16-byte load {cdf[lo-1] .. cdf[lo+2]} of the padded CDF holds every
// candidate and the bracket.  Wider cells fall back to the binary search.  Used for the MARGINAL (row) table only, which
// k_shade stages in LDS; inside a row a light sample reads ONE word of the row's alias table instead of searching a CDF.
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
RT3_DEV uint32_t cdf_find_guided(const float* cdfp, const uint32_t* guide, uint32_t n, float u, float& lo_v, float& hi_v) {
    uint32_t k = (uint32_t)(u * (float)n);
    k = k > n - 1 ? n - 1 : k;
    const uint32_t g = guide[k];
    uint32_t lo = g & 0xFFFFu, hi = g >> 16;
    if (hi - lo <= 2u) {
        const f32x4_u c = *reinterpret_cast<const f32x4_u*>(cdfp + lo);
        const bool s1 = lo < hi && !(c.y > u);
        const bool s2 = s1 && lo + 1u < hi && !(c.z > u);
        lo_v = s2 ? c.z : (s1 ? c.y : c.x);
        hi_v = s2 ? c.w : (s1 ? c.z : c.y);
        return lo + (s1 ? 1u : 0u) + (s2 ? 1u : 0u);
    }
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (cdfp[mid + 1] > u) hi = mid;
        else lo = mid + 1;
    }
    lo_v = cdfp[lo];
    hi_v = cdfp[lo + 1];
    return lo;
}
// Light sample in two steps, so that the caller can drop samples below the surface's horizon (cos <= 0: nearly half of
// them) BEFORE paying for the radiance texels: (1) invert the CDFs -> texel, equirect coordinates, direction;
// (2) bilinear radiance + the texel's density.
struct SkyPick {
    float u, v, sin_theta;
    int x, y;
};
// cdf_marg / guide_marg: the marginal tables (the caller may have staged them in LDS)
RT3_DEV SkyPick sky_sample_direction(const SceneDev& sc, const float* cdf_marg, const uint32_t* guide_marg, float u0, float u1, V3& dir) {
    uint32_t W = sc.sky_w, H = sc.sky_h;
    float lo, hi;
    uint32_t y = cdf_find_guided(cdf_marg, guide_marg, H, u0, lo, hi);
    float dv = hi > lo ? (u0 - lo) / (hi - lo) : 0.5f;
    // the row's alias table (the oracle's sky_sample has the construction): cell k = floor(u1 W); xi = frac(u1 W) decides between
    // column k and its alias and is stretched back to [0, 1) as the position inside the chosen texel.  ONE gathered word.
    const float sx = u1 * (float)W;
    uint32_t k = (uint32_t)sx;
    k = k > W - 1 ? W - 1 : k;
    float xi = sx - (float)k;
    xi = xi < 0.99999994f ? xi : 0.99999994f;
    const uint32_t e = sc.sky_alias[(size_t)y * W + k];
    const float Q = (float)((e & 0xFFFFu) + 1u) * (1.0f / 65536.0f);
    const bool keep = xi < Q;
    const uint32_t x = keep ? k : (e >> 16);
    float du = keep ? xi / Q : (xi - Q) / (1.0f - Q);
    du = du < 0.99999994f ? du : 0.99999994f;
    SkyPick p;
    p.u = ((float)x + du) / (float)W;
    p.v = ((float)y + dv) / (float)H;
    p.x = (int)x;
    p.y = (int)y;
    float st, ct, s2, c2;
    sincos_2pi(p.v * 0.5f, st, ct);
    sincos_2pi(p.u, s2, c2);
    p.sin_theta = st;
    dir = v3((-c2) * st, ct, (-s2) * st);
    return p;
}
RT3_DEV void sky_sample_radiance(const SceneDev& sc, const SkyPick& p, V3& rad, float& pdf) {
    float pt;
    rad = sky_eval_pdf(sc, p.u, p.v, p.x, p.y, pt);
    pdf = p.sin_theta > 0.0f ? pt / (2.0f * kPi * kPi * p.sin_theta) : 0.0f;
}
RT3_DEV void sky_sample(const SceneDev& sc, float u0, float u1, V3& dir, V3& rad, float& pdf) {
    const SkyPick p = sky_sample_direction(sc, sc.cdf_marg, sc.guide_marg, u0, u1, dir);
    sky_sample_radiance(sc, p, rad, pdf);
}

// ------------------------------------------------------------------------------------------------ intersection (north_star)
struct Hit {
    float t, u, v;
    uint32_t prim;
};
// Triangle records: 3 x float4 {v0.xyz,v1.x} {v1.yz,v2.xy} {v2.z,prim,-,-}; the watertight two-sided test is tri_test_nb
// (rt3_kernels.hip).  Order-independent acceptance: t > tmin && (t < best.t || (t == best.t && prim < best.prim)).
// Dot products use explicit fused multiply-adds in a fixed order (the oracle mirrors them with fmaf).
RT3_DEV float dot_fma(V3 a, V3 b) { return __builtin_fmaf(a.x, b.x, __builtin_fmaf(a.y, b.y, a.z * b.z)); }
RT3_DEV V3 cross_fma(V3 a, V3 b) {
    return V3{__builtin_fmaf(a.y, b.z, -(a.z * b.y)), __builtin_fmaf(a.z, b.x, -(a.x * b.z)), __builtin_fmaf(a.x, b.y, -(a.y * b.x))};
}
RT3_DEV float guarded_inverse(float d) {
    float a = d < 0.0f ? -d : d;
    float g = a < 1e-20f ? (d < 0.0f ? -1e-20f : 1e-20f) : d;
    return 1.0f / g;
}
// slab test of one child box; returns hit and the entry distance
RT3_DEV bool slab_test(V3 bmin, V3 bmax, V3 o, V3 inv, float tmin, float tbest, float& tn_out) {
    float t0 = (bmin.x - o.x) * inv.x, t1 = (bmax.x - o.x) * inv.x;
    float tn = tmin, tf = tbest;
    float lo = t0 < t1 ? t0 : t1, hi = t0 < t1 ? t1 : t0;
    tn = lo > tn ? lo : tn;
    tf = hi < tf ? hi : tf;
    t0 = (bmin.y - o.y) * inv.y;
    t1 = (bmax.y - o.y) * inv.y;
    lo = t0 < t1 ? t0 : t1;
    hi = t0 < t1 ? t1 : t0;
    tn = lo > tn ? lo : tn;
    tf = hi < tf ? hi : tf;
    t0 = (bmin.z - o.z) * inv.z;
    t1 = (bmax.z - o.z) * inv.z;
    lo = t0 < t1 ? t0 : t1;
    hi = t0 < t1 ? t1 : t0;
    tn = lo > tn ? lo : tn;
    tf = hi < tf ? hi : tf;
    tn_out = tn;
    return tn <= tf;
}

// slab test on the hardware min / max / max3 / min3 instructions.  Same values as slab_test up to the sign of a zero
// (no NaN can occur: the inverse direction is guarded and finite), hence the same hit / order decisions.
RT3_DEV bool slab_test_hw(V3 bmin, V3 bmax, V3 o, V3 inv, float tmin, float tbest, float& tn_out) {
    float ax = (bmin.x - o.x) * inv.x, bx = (bmax.x - o.x) * inv.x;
    float ay = (bmin.y - o.y) * inv.y, by = (bmax.y - o.y) * inv.y;
    float az = (bmin.z - o.z) * inv.z, bz = (bmax.z - o.z) * inv.z;
    float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)), __builtin_fmaxf(__builtin_fminf(az, bz), tmin));
    float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)), __builtin_fminf(__builtin_fmaxf(az, bz), tbest));
    tn_out = tn;
    return tn <= tf;
}

// slab test on ray parameters that are already computed (quantised nodes: t = fma(q, step*inv, (org-o)*inv)); arguments are
// (lo, hi) per axis in x, y, z order
RT3_DEV bool slab_test_q(float ax, float bx, float ay, float by, float az, float bz, float tmin, float tbest, float& tn_out) {
    float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)), __builtin_fmaxf(__builtin_fminf(az, bz), tmin));
    float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)), __builtin_fminf(__builtin_fmaxf(az, bz), tbest));
    tn_out = tn;
    return tn <= tf;
}

}  // namespace rt3
