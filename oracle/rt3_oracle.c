/*
 * rt3_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).  See rt3_oracle.h.
 *
 * Every function cites the reference file:line (relative to /root/reference) it restates.  Functions marked
 * [north_star] have no reference counterpart (the reference uses the Vulkan driver's BVH and has no NEE / sky /
 * blue-noise code); they define the estimator the HIP product must reproduce.
 *
 * Arithmetic contract shared with the product (DESIGN.md "Arithmetic contract"): fp32 everywhere, no FMA
 * contraction (build with -ffp-contract=off), +,-,*,/,sqrt correctly rounded, min/max as explicit ternaries,
 * transcendental functions only through the polynomials defined here.  With that, product and oracle are
 * expected to agree bit for bit; tests state the tolerance they actually need.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fno-fast-math -pthread).
 */
#include "rt3_oracle.h"
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------ helpers */
#define F_PI 3.14159265358979323846f
#define F_TAU 6.28318530717958647692f       /* math.slang:3 */
#define F_FRAC_1_PI 0.318309886183790671538f /* math.slang:4 */
#define F_HALF_PI 1.57079632679489661923f

static inline float fminx(float a, float b) { return a < b ? a : b; }
static inline float fmaxx(float a, float b) { return a > b ? a : b; }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline float dot3(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void cross3(const float a[3], const float b[3], float o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
static inline void normalize3(float v[3]) {
    float inv = 1.0f / sqrtf(dot3(v, v));
    v[0] *= inv; v[1] *= inv; v[2] *= inv;
}

/* simple parallel-for over [0,n) in contiguous chunks */
typedef void (*pf_body)(void *ctx, uint32_t begin, uint32_t end, int tid);
typedef struct { pf_body fn; void *ctx; uint32_t begin, end; int tid; } pf_job;
static void *pf_thread(void *p) { pf_job *j = (pf_job *)p; j->fn(j->ctx, j->begin, j->end, j->tid); return NULL; }
static void parallel_for(uint32_t n, int n_threads, pf_body fn, void *ctx) {
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    if ((uint32_t)n_threads > n) n_threads = n ? (int)n : 1;
    if (n_threads == 1) { fn(ctx, 0, n, 0); return; }
    pthread_t th[256]; pf_job jobs[256];
    for (int i = 0; i < n_threads; i++) {
        jobs[i].fn = fn; jobs[i].ctx = ctx; jobs[i].tid = i;
        jobs[i].begin = (uint32_t)((uint64_t)n * i / n_threads);
        jobs[i].end = (uint32_t)((uint64_t)n * (i + 1) / n_threads);
        pthread_create(&th[i], NULL, pf_thread, &jobs[i]);
    }
    for (int i = 0; i < n_threads; i++) pthread_join(th[i], NULL);
}

/* ------------------------------------------------------------------------------------------------ RNG */
/* random.slang:5-15 (Bob Jenkins' integer hash) */
uint32_t orc_hash(uint32_t a) {
    a = (a + 0x7ed55d16u) + (a << 12);
    a = (a ^ 0xc761c23cu) ^ (a >> 19);
    a = (a + 0x165667b1u) + (a << 5);
    a = (a + 0xd3a2646cu) ^ (a << 9);
    a = (a + 0xfd7046c5u) + (a << 3);
    a = (a ^ 0xb55a4f09u) ^ (a >> 16);
    return a;
}
/* math.slang:105-112 */
static uint32_t integer_explode(uint32_t x) {
    x = (x | (x << 8)) & 0x00FF00FFu;
    x = (x | (x << 4)) & 0x0F0F0F0Fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
}
/* math.slang:114-117 */
uint32_t orc_zcurve(uint32_t x, uint32_t y) { return integer_explode(x) | (integer_explode(y) << 1); }
/* random.slang:42-46 */
uint32_t orc_rng_seed(uint32_t px, uint32_t py, uint32_t frame) { return orc_hash(orc_zcurve(px, py)) + frame; }
/* random.slang:49-79 : one-block murmur3 of the counter, len = 4 finaliser.  Counter-based: `index` is explicit. */
uint32_t orc_murmur3(uint32_t seed, uint32_t index) {
    const uint32_t c1 = 0xcc9e2d51u, c2 = 0x1b873593u;
    uint32_t h = seed, k = index;
    k *= c1;
    k = (k << 15) | (k >> 17);
    k *= c2;
    h ^= k;
    h = ((h << 13) | (h >> 19)) * 5u + 0xe6546b64u;
    h ^= 4u;
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}
/* random.slang:82-89 */
float orc_uniform_float(uint32_t seed, uint32_t index) {
    uint32_t v = orc_murmur3(seed, index);
    return u2f((v & 0x7FFFFFu) | 0x3F800000u) - 1.0f;
}
/* random.slang:17-24 (bit reversal part; the float scale is trivial) */
uint32_t orc_radical_inverse_bits(uint32_t bits) {
    bits = (bits << 16) | (bits >> 16);
    bits = ((bits & 0x55555555u) << 1) | ((bits & 0xAAAAAAAAu) >> 1);
    bits = ((bits & 0x33333333u) << 2) | ((bits & 0xCCCCCCCCu) >> 2);
    bits = ((bits & 0x0F0F0F0Fu) << 4) | ((bits & 0xF0F0F0F0u) >> 4);
    bits = ((bits & 0x00FF00FFu) << 8) | ((bits & 0xFF00FF00u) >> 8);
    return bits;
}

/* ------------------------------------------------------------------------------------------------ packing */
/* packing.slang:2-10 */
static float unpack_unorm(uint32_t p, uint32_t bits) {
    uint32_t maxv = (1u << bits) - 1u;
    return (float)(p & maxv) / (float)maxv;
}
static uint32_t pack_unorm(float v, uint32_t bits) {
    uint32_t maxv = (1u << bits) - 1u;
    float c = fminx(fmaxx(v, 0.0f), 1.0f);
    return (uint32_t)(c * (float)maxv + 0.5f);
}
/* packing.slang:12-18 */
uint32_t orc_pack_normal_11_10_11(const float n[3]) {
    uint32_t p = 0;
    p += pack_unorm(n[0] * 0.5f + 0.5f, 11);
    p += pack_unorm(n[1] * 0.5f + 0.5f, 10) << 11;
    p += pack_unorm(n[2] * 0.5f + 0.5f, 11) << 21;
    return p;
}
/* packing.slang:20-27 */
void orc_unpack_normal_11_10_11(uint32_t p, float n[3]) {
    n[0] = unpack_unorm(p, 11) * 2.0f - 1.0f;
    n[1] = unpack_unorm(p >> 11, 10) * 2.0f - 1.0f;
    n[2] = unpack_unorm(p >> 21, 11) * 2.0f - 1.0f;
    normalize3(n);
}
/* packing.slang:46-53 */
uint32_t orc_pack_color_888(const float c[3]) {
    uint32_t p = 0;
    p += pack_unorm(sqrtf(c[0]), 8);
    p += pack_unorm(sqrtf(c[1]), 8) << 8;
    p += pack_unorm(sqrtf(c[2]), 8) << 16;
    return p;
}
/* packing.slang:55-62 */
void orc_unpack_color_888(uint32_t p, float c[3]) {
    float r = unpack_unorm(p, 8), g = unpack_unorm(p >> 8, 8), b = unpack_unorm(p >> 16, 8);
    c[0] = r * r; c[1] = g * g; c[2] = b * b;
}
/* f32 -> f16 bits, round to nearest even (f32tof16, packing.slang:88-90) */
static uint32_t f32_to_f16(float f) {
    uint32_t x = f2u(f), sign = (x >> 16) & 0x8000u;
    uint32_t em = x & 0x7FFFFFFFu;
    if (em >= 0x7F800000u) return sign | (em > 0x7F800000u ? 0x7E00u : 0x7C00u);
    if (em >= 0x477FF000u) return sign | 0x7C00u; /* rounds to >= 65520 -> inf */
    if (em < 0x33000001u) return sign;            /* < 2^-25 (or == 2^-25 tie to even 0) -> 0 */
    int32_t e = (int32_t)(em >> 23) - 127;
    uint32_t m = (em & 0x7FFFFFu) | 0x800000u;
    uint32_t shift, h;
    if (e < -14) { /* subnormal half */
        shift = (uint32_t)(13 + (-14 - e));
        h = m >> shift;
        uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (h & 1u))) h++;
        return sign | h;
    }
    h = ((uint32_t)(e + 15) << 10) | ((m >> 13) & 0x3FFu);
    uint32_t rem = m & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
    return sign | h;
}
static float f16_to_f32(uint32_t h) {
    uint32_t sign = (h & 0x8000u) << 16, e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
    if (e == 0) {
        if (m == 0) return u2f(sign);
        float v = (float)m * 5.9604644775390625e-08f; /* 2^-24 */
        return (sign ? -v : v);
    }
    if (e == 31) return u2f(sign | 0x7F800000u | (m << 13));
    return u2f(sign | ((e + 112u) << 23) | (m << 13));
}
/* packing.slang:88-97 */
uint32_t orc_pack_2x16f(float a, float b) { return f32_to_f16(a) | (f32_to_f16(b) << 16); }
void orc_unpack_2x16f(uint32_t u, float out[2]) { out[0] = f16_to_f32(u & 0xFFFFu); out[1] = f16_to_f32((u >> 16) & 0xFFFFu); }

static float exp2i(int e) { return u2f((uint32_t)(e + 127) << 23); } /* exact 2^e, -126 <= e <= 127 */
/* packing.slang:99-144 */
uint32_t orc_float3_to_rgb9e5(const float c[3]) {
    const float max_rgb9e5 = (511.0f / 512.0f) * 65536.0f;
    float rc = fminx(fmaxx(c[0], 0.0f), max_rgb9e5);
    float gc = fminx(fmaxx(c[1], 0.0f), max_rgb9e5);
    float bc = fminx(fmaxx(c[2], 0.0f), max_rgb9e5);
    float maxrgb = fmaxx(rc, fmaxx(gc, bc));
    int fl2 = (int)((f2u(maxrgb) & 0x7F800000u) >> 23) - 127; /* floor_log2, packing.slang:115-119 */
    int exp_shared = (fl2 > -16 ? fl2 : -16) + 1 + 15;
    float denom = exp2i(exp_shared - 15 - 9);
    int maxm = (int)floorf(maxrgb / denom + 0.5f);
    if (maxm == 512) { denom *= 2.0f; exp_shared += 1; }
    int rm = (int)floorf(rc / denom + 0.5f);
    int gm = (int)floorf(gc / denom + 0.5f);
    int bm = (int)floorf(bc / denom + 0.5f);
    return ((uint32_t)rm << 23) | ((uint32_t)gm << 14) | ((uint32_t)bm << 5) | (uint32_t)exp_shared;
}
/* packing.slang:146-162 */
void orc_rgb9e5_to_float3(uint32_t v, float c[3]) {
    int e = (int)(v & 31u) - 15 - 9;
    float scale = exp2i(e);
    c[0] = (float)((v >> 23) & 511u) * scale;
    c[1] = (float)((v >> 14) & 511u) * scale;
    c[2] = (float)((v >> 5) & 511u) * scale;
}
/* gbuffer_helpers.slang:22-34 (roughness -> perceptual: sqrt, :73-75) */
void orc_gbuffer_pack(const float s[11], uint32_t out[4]) {
    out[0] = orc_pack_color_888(s + 0);
    out[1] = orc_pack_normal_11_10_11(s + 6);
    out[2] = orc_pack_2x16f(sqrtf(s[9]), s[10]);
    out[3] = orc_float3_to_rgb9e5(s + 3);
}
/* gbuffer_helpers.slang:59-70 (perceptual -> roughness: r*r, :77-79) */
void orc_gbuffer_unpack(const uint32_t in[4], float s[11]) {
    float rm[2];
    orc_unpack_color_888(in[0], s + 0);
    orc_unpack_normal_11_10_11(in[1], s + 6);
    orc_unpack_2x16f(in[2], rm);
    s[9] = rm[0] * rm[0];
    s[10] = rm[1];
    orc_rgb9e5_to_float3(in[3], s + 3);
}

/* ------------------------------------------------------------------------------------------------ math */
/* [arithmetic contract] sin(2 pi u), cos(2 pi u) for u in [0,1): exact quadrant reduction + Taylor/Horner on
 * [0, pi/4].  Stands in for sin()/cos() of brdf.slang:61-62 so that CPU and GPU agree bit for bit. */
void orc_sincos_2pi(float u, float *so, float *co) {
    float x = u * 4.0f;
    int q = (int)x;
    float r = x - (float)q;
    int swap = r > 0.5f;
    if (swap) r = 1.0f - r;
    float a = r * F_HALF_PI, a2 = a * a;
    float s = a * (1.0f + a2 * (-1.6666667163e-01f + a2 * (8.3333337680e-03f + a2 * (-1.9841270114e-04f + a2 * 2.7557314297e-06f))));
    float c = 1.0f + a2 * (-0.5f + a2 * (4.1666667908e-02f + a2 * (-1.3888889225e-03f + a2 * (2.4801587642e-05f + a2 * -2.7557314297e-07f))));
    if (swap) { float t = s; s = c; c = t; }
    switch (q & 3) {
        case 0: *so = s; *co = c; break;
        case 1: *so = c; *co = -s; break;
        case 2: *so = -s; *co = -c; break;
        default: *so = -c; *co = s; break;
    }
}
/* [arithmetic contract] atan2 through a degree-11 odd polynomial on [0,1] (max error ~1e-5 rad) */
float orc_atan2(float y, float x) {
    float ax = x < 0.0f ? -x : x, ay = y < 0.0f ? -y : y;
    float mx = fmaxx(ax, ay), mn = fminx(ax, ay);
    if (mx == 0.0f) return 0.0f;
    float a = mn / mx, s = a * a;
    float r = a * (0.99997726f + s * (-0.33262347f + s * (0.19354346f + s * (-0.11643287f + s * (0.05265332f + s * -0.01172120f)))));
    if (ay > ax) r = F_HALF_PI - r;
    if (x < 0.0f) r = F_PI - r;
    if (y < 0.0f) r = -r;
    return r;
}
/* math.slang:6-12 ; asin(y) := atan2(y, sqrt(1 - y^2)) */
void orc_dir_to_equirect_uv(const float d[3], float uv[2]) {
    float as = orc_atan2(d[1], sqrtf(fmaxx(0.0f, 1.0f - d[1] * d[1])));
    uv[0] = 0.5f + orc_atan2(d[2], d[0]) / F_TAU;
    uv[1] = 0.5f - as / F_PI;
}
/* math.slang:29-50 ; returns the basis columns b1,b2 (third column is n) */
void orc_onb(const float n[3], float b1[3], float b2[3]) {
    if (n[2] < 0.0f) {
        const float a = 1.0f / (1.0f - n[2]);
        const float b = n[0] * n[1] * a;
        b1[0] = 1.0f - n[0] * n[0] * a; b1[1] = -b; b1[2] = n[0];
        b2[0] = b; b2[1] = n[1] * n[1] * a - 1.0f; b2[2] = -n[1];
    } else {
        const float a = 1.0f / (1.0f + n[2]);
        const float b = -n[0] * n[1] * a;
        b1[0] = 1.0f - n[0] * n[0] * a; b1[1] = b; b1[2] = -n[0];
        b2[0] = b; b2[1] = 1.0f - n[1] * n[1] * a; b2[2] = -n[1];
    }
}
/* mul(tangent_to_world, wi), refrence_mode.slang:48 with the matrix of math.slang:45-49 */
static void onb_apply(const float b1[3], const float b2[3], const float n[3], const float w[3], float o[3]) {
    o[0] = b1[0] * w[0] + b2[0] * w[1] + n[0] * w[2];
    o[1] = b1[1] * w[0] + b2[1] * w[1] + n[1] * w[2];
    o[2] = b1[2] * w[0] + b2[2] * w[1] + n[2] * w[2];
}
/* brdf.slang:56-65 : DiffuseBrdf.sample direction (pdf = 1/pi projected-solid-angle, value_over_pdf = albedo) */
void orc_diffuse_sample(float u0, float u1, float wi[3]) {
    float sp, cp;
    orc_sincos_2pi(u0, &sp, &cp); /* phi = urand.x * TAU */
    float cos_theta = sqrtf(fmaxx(0.0f, 1.0f - u1));
    float sin_theta = sqrtf(fmaxx(0.0f, 1.0f - cos_theta * cos_theta));
    wi[0] = cp * sin_theta; wi[1] = sp * sin_theta; wi[2] = cos_theta;
}

/* postprocess.slang:13-88 (AGX_LOOK 2 "Punchy"; pow() of a negative base clamped to 0 -- documented deviation) */
static float agx_contrast(float x) { /* :13-25 */
    float x2 = x * x, x4 = x2 * x2;
    return 15.5f * x4 * x2 - 40.14f * x4 * x + 31.96f * x4 - 6.868f * x2 * x + 0.4298f * x2 + 0.1191f * x - 0.00232f;
}
void orc_agx_tonemap(const float in[3], float out[3]) {
    static const float m[9] = {0.842479062253094f, 0.0423282422610123f, 0.0423756549057051f,
                               0.0784335999999992f, 0.878468636469772f, 0.0784336f,
                               0.0792237451477643f, 0.0791661274605434f, 0.879142973793104f};
    static const float mi[9] = {1.19687900512017f, -0.0528968517574562f, -0.0529716355144438f,
                                -0.0980208811401368f, 1.15190312990417f, -0.0980434501171241f,
                                -0.0990297440797205f, -0.0989611768448433f, 1.15107367264116f};
    const float min_ev = -12.47393f, max_ev = 4.026069f;
    float v[3], w[3];
    for (int j = 0; j < 3; j++) v[j] = in[0] * m[0 * 3 + j] + in[1] * m[1 * 3 + j] + in[2] * m[2 * 3 + j]; /* mul(val, agx_mat) :35 */
    for (int j = 0; j < 3; j++) {
        float l = v[j] > 0.0f ? log2f(v[j]) : min_ev;
        l = fminx(fmaxx(l, min_ev), max_ev);
        l = (l - min_ev) / (max_ev - min_ev);
        v[j] = agx_contrast(l);
    }
    /* agxLook :62-88 */
    float luma = v[0] * 0.2126f + v[1] * 0.7152f + v[2] * 0.0722f;
    for (int j = 0; j < 3; j++) {
        float p = powf(fmaxx(v[j], 0.0f), 1.1f);
        w[j] = luma + 1.1f * (p - luma);
    }
    /* agxEotf :47-60 */
    for (int j = 0; j < 3; j++) out[j] = w[0] * mi[0 * 3 + j] + w[1] * mi[1 * 3 + j] + w[2] * mi[2 * 3 + j];
}

/* ------------------------------------------------------------------------------------------------ camera */
/* column-major 4x4: m[c*4+r] */
static void mat4_mul_vec(const float m[16], const float v[4], float o[4]) {
    for (int r = 0; r < 4; r++) o[r] = m[0 + r] * v[0] + m[4 + r] * v[1] + m[8 + r] * v[2] + m[12 + r] * v[3];
}
static void mat4_inverse(const float m[16], float inv[16]) { /* glam Mat4::inverse restated as a plain cofactor expansion */
    double a[16], o[16];
    for (int i = 0; i < 16; i++) a[i] = m[i];
    o[0] = a[5] * a[10] * a[15] - a[5] * a[11] * a[14] - a[9] * a[6] * a[15] + a[9] * a[7] * a[14] + a[13] * a[6] * a[11] - a[13] * a[7] * a[10];
    o[4] = -a[4] * a[10] * a[15] + a[4] * a[11] * a[14] + a[8] * a[6] * a[15] - a[8] * a[7] * a[14] - a[12] * a[6] * a[11] + a[12] * a[7] * a[10];
    o[8] = a[4] * a[9] * a[15] - a[4] * a[11] * a[13] - a[8] * a[5] * a[15] + a[8] * a[7] * a[13] + a[12] * a[5] * a[11] - a[12] * a[7] * a[9];
    o[12] = -a[4] * a[9] * a[14] + a[4] * a[10] * a[13] + a[8] * a[5] * a[14] - a[8] * a[6] * a[13] - a[12] * a[5] * a[10] + a[12] * a[6] * a[9];
    o[1] = -a[1] * a[10] * a[15] + a[1] * a[11] * a[14] + a[9] * a[2] * a[15] - a[9] * a[3] * a[14] - a[13] * a[2] * a[11] + a[13] * a[3] * a[10];
    o[5] = a[0] * a[10] * a[15] - a[0] * a[11] * a[14] - a[8] * a[2] * a[15] + a[8] * a[3] * a[14] + a[12] * a[2] * a[11] - a[12] * a[3] * a[10];
    o[9] = -a[0] * a[9] * a[15] + a[0] * a[11] * a[13] + a[8] * a[1] * a[15] - a[8] * a[3] * a[13] - a[12] * a[1] * a[11] + a[12] * a[3] * a[9];
    o[13] = a[0] * a[9] * a[14] - a[0] * a[10] * a[13] - a[8] * a[1] * a[14] + a[8] * a[2] * a[13] + a[12] * a[1] * a[10] - a[12] * a[2] * a[9];
    o[2] = a[1] * a[6] * a[15] - a[1] * a[7] * a[14] - a[5] * a[2] * a[15] + a[5] * a[3] * a[14] + a[13] * a[2] * a[7] - a[13] * a[3] * a[6];
    o[6] = -a[0] * a[6] * a[15] + a[0] * a[7] * a[14] + a[4] * a[2] * a[15] - a[4] * a[3] * a[14] - a[12] * a[2] * a[7] + a[12] * a[3] * a[6];
    o[10] = a[0] * a[5] * a[15] - a[0] * a[7] * a[13] - a[4] * a[1] * a[15] + a[4] * a[3] * a[13] + a[12] * a[1] * a[7] - a[12] * a[3] * a[5];
    o[14] = -a[0] * a[5] * a[14] + a[0] * a[6] * a[13] + a[4] * a[1] * a[14] - a[4] * a[2] * a[13] - a[12] * a[1] * a[6] + a[12] * a[2] * a[5];
    o[3] = -a[1] * a[6] * a[11] + a[1] * a[7] * a[10] + a[5] * a[2] * a[11] - a[5] * a[3] * a[10] - a[9] * a[2] * a[7] + a[9] * a[3] * a[6];
    o[7] = a[0] * a[6] * a[11] - a[0] * a[7] * a[10] - a[4] * a[2] * a[11] + a[4] * a[3] * a[10] + a[8] * a[2] * a[7] - a[8] * a[3] * a[6];
    o[11] = -a[0] * a[5] * a[11] + a[0] * a[7] * a[9] + a[4] * a[1] * a[11] - a[4] * a[3] * a[9] - a[8] * a[1] * a[7] + a[8] * a[3] * a[5];
    o[15] = a[0] * a[5] * a[10] - a[0] * a[6] * a[9] - a[4] * a[1] * a[10] + a[4] * a[2] * a[9] + a[8] * a[1] * a[6] - a[8] * a[2] * a[5];
    double det = a[0] * o[0] + a[1] * o[4] + a[2] * o[8] + a[3] * o[12];
    double id = 1.0 / det;
    for (int i = 0; i < 16; i++) inv[i] = (float)(o[i] * id);
}
/* camera.rs:52-58 (glam look_at_rh / perspective_rh, depth 0..1) + renderer/mod.rs:72-78.
 * "parity unpinned": glam is a third-party crate (0.29.3) absent from /root/reference; closed forms restated. */
void orc_camera_gconst(const float pos[3], const float dir_in[3], float fov_y, float aspect, float z_near,
                       float z_far, float width, float height, orc_gconst *g) {
    memset(g, 0, sizeof(*g));
    float f[3] = {dir_in[0], dir_in[1], dir_in[2]}, up[3] = {0.0f, 1.0f, 0.0f}, s[3], u[3];
    normalize3(f); /* Camera::new normalises direction, camera.rs:43 */
    cross3(f, up, s);
    normalize3(s);
    cross3(s, f, u);
    float *v = g->view;
    v[0] = s[0]; v[1] = u[0]; v[2] = -f[0]; v[3] = 0.0f;
    v[4] = s[1]; v[5] = u[1]; v[6] = -f[1]; v[7] = 0.0f;
    v[8] = s[2]; v[9] = u[2]; v[10] = -f[2]; v[11] = 0.0f;
    v[12] = -dot3(pos, s); v[13] = -dot3(pos, u); v[14] = dot3(pos, f); v[15] = 1.0f;
    float sf = (float)sin(0.5 * (double)fov_y), cf = (float)cos(0.5 * (double)fov_y);
    float h = cf / sf, w = h / aspect, r = z_far / (z_near - z_far);
    float *p = g->proj;
    p[0] = w; p[5] = h; p[10] = r; p[11] = -1.0f; p[14] = r * z_near;
    mat4_inverse(g->proj, g->proj_inverse);
    mat4_inverse(g->view, g->view_inverse);
    g->window_size[0] = width; g->window_size[1] = height;
    g->blendfactor = 1.0f;
}
/* gbuffer_helpers.slang:85-103 with PlanarViewConstants := {proj_inverse, view_inverse, window_size, view_inverse col 3}.
 * Pixel (0,0) is top-left and d.y is flipped so images are upright (documented deviation). */
void orc_primary_ray(const orc_gconst *g, uint32_t px, uint32_t py, float o[3], float d[3]) {
    float cx = ((float)px + 0.5f) / g->window_size[0], cy = ((float)py + 0.5f) / g->window_size[1];
    float clip[4] = {cx * 2.0f - 1.0f, -(cy * 2.0f - 1.0f), 1.0f, 1.0f}, target[4], dir4[4], w[4];
    mat4_mul_vec(g->proj_inverse, clip, target);
    float t3[3] = {target[0], target[1], target[2]};
    normalize3(t3);
    dir4[0] = t3[0]; dir4[1] = t3[1]; dir4[2] = t3[2]; dir4[3] = 0.0f;
    mat4_mul_vec(g->view_inverse, dir4, w);
    d[0] = w[0]; d[1] = w[1]; d[2] = w[2];
    o[0] = g->view_inverse[12]; o[1] = g->view_inverse[13]; o[2] = g->view_inverse[14];
}

/* orc_primary_ray for n pixels at once: 8 SoA arrays of n floats (ox oy oz dx dy dz tmin tmax) */
void orc_primary_rays(const orc_gconst *g, const uint32_t *xs, const uint32_t *ys, uint32_t n, float tmin, float tmax, float *rays) {
    for (uint32_t i = 0; i < n; i++) {
        float o[3], d[3];
        orc_primary_ray(g, xs[i], ys[i], o, d);
        for (int k = 0; k < 3; k++) { rays[(size_t)k * n + i] = o[k]; rays[(size_t)(3 + k) * n + i] = d[k]; }
        rays[(size_t)6 * n + i] = tmin; rays[(size_t)7 * n + i] = tmax;
    }
}

/* ------------------------------------------------------------------------------------------------ scene */
struct orc_scene {
    float *verts; uint32_t n_verts;      /* interleaved 8 floats */
    uint32_t *indices; uint32_t n_indices;
    orc_geometry_info *geoms; uint32_t *prim_counts, *first_prim; uint32_t n_geoms;
    uint32_t n_prims; uint32_t *prim_geom;
    /* [round 3] instances (world/mod.rs:34-60): the world is flattened into (instance, geometry) pairs, instance-major; prim_geom /
     * first_prim / n_prims describe the FLATTENED primitives, flat_geom / flat_inst name a pair's geometry and instance */
    orc_instance *inst; uint32_t n_inst;
    uint32_t n_flat; uint32_t *flat_geom, *flat_inst; uint8_t *flat_identity;
    /* accel */
    uint32_t n_tris, n_nodes, max_depth, leaf_max, node_width, node_quant, collapse, sah_top, tree_order;
    uint32_t recull; /* experiment (default 0; the product does not do it: measured slower): re-cull stacked node references on pop */
    uint32_t top_opt, top_opt_passes; /* experiment: insertion-based re-optimisation of the tree above subtrees of <= top_opt triangles */
    float dp_c_node, dp_c_tri; /* experiment knob of the cost-driven collapse (tests/experiments/tree_quality_gpu.py); both 1 = the product's rule */
    float *nodes;   /* 16 words per node */
    float *tris;    /* 12 words per tri (Morton order) */
    uint64_t *codes;
    /* sky */
    float *sky; uint32_t sky_w, sky_h;
    float *cdf_marg, *pdf_uv;      /* marginal CDF over rows; realised (u,v)-density per texel */
    uint32_t *sky_alias;           /* per texel: keep-threshold q16 | alias column << 16 (one alias table per row) */
    uint32_t *sky_q;               /* per texel: the RGB9E5 word the radiance is stored as */
    /* blue noise */
    uint8_t *bn; uint32_t bn_w, bn_h;
    /* base-colour textures (RGBA8, sRGB-encoded colour), hit_logic.slang:31-33 */
    uint8_t **tex; uint32_t *tex_w, *tex_h; uint32_t n_tex;
    float srgb_lut[256];
};

orc_scene *orc_scene_create(void) {
    orc_scene *s = (orc_scene *)calloc(1, sizeof(orc_scene));
    s->leaf_max = 2;
    s->node_width = 4;
    s->node_quant = 1;
    s->collapse = 2; /* cost-driven collapse over a binned-SAH tree down to single triangles (round 3) */
    s->sah_top = 1;
    s->dp_c_node = s->dp_c_tri = 1.0f;
    for (int i = 0; i < 256; i++) { /* sRGB EOTF, IEC 61966-2-1, in double */
        double c = i / 255.0;
        s->srgb_lut[i] = (float)(c <= 0.04045 ? c / 12.92 : pow((c + 0.055) / 1.055, 2.4));
    }
    return s;
}
/* leaf_max: 1..8 triangles per leaf; node_width: 2 (64 B nodes) or 4 (128 B nodes).  Call before orc_accel_build. */
void orc_accel_set_layout(orc_scene *s, uint32_t leaf_max, uint32_t node_width, uint32_t quantized) {
    s->leaf_max = leaf_max < 1 ? 1 : (leaf_max > 8 ? 8 : leaf_max);
    s->node_width = node_width == 2 ? 2 : 4;
    s->node_quant = s->node_width == 4 ? (quantized > 2 ? 2 : quantized) : 0; /* 0 fp32 128 B, 1 quantised 64 B, 2 compact 48 B */
}
void orc_accel_set_collapse(orc_scene *s, uint32_t mode) { s->collapse = mode > 2u ? 2u : mode; }
void orc_accel_set_tree_order(orc_scene *s, uint32_t on) { s->tree_order = on ? 1u : 0u; }
void orc_accel_set_dp_costs(orc_scene *s, float c_node, float c_tri) { s->dp_c_node = c_node; s->dp_c_tri = c_tri; }
void orc_accel_set_recull(orc_scene *s, uint32_t on) { s->recull = on; }
void orc_accel_set_top_opt(orc_scene *s, uint32_t k_top, uint32_t passes) { s->top_opt = k_top; s->top_opt_passes = passes; }
void orc_accel_set_sah_top(orc_scene *s, uint32_t cluster_size) { s->sah_top = cluster_size; }
uint32_t orc_accel_node_words(const orc_scene *s) { return s->node_width == 2 ? 16u : (s->node_quant == 2 ? 12u : (s->node_quant ? 16u : 32u)); }

/* [north_star] 64-byte four-wide node with quantised child boxes:
 *   words 0-2  origin = min corner of the union of the child boxes
 *   word  3    biased exponents ex | ey << 8 | ez << 16 of the per-axis power-of-two grid step (scale = 2^(e-127))
 *   words 4-9  24 bytes: child k at bytes 6k..6k+5 = qlo.xyz, qhi.xyz (8 bit each)
 *   words 10-13 the four child references; words 14-15 zero -- or, in the 64-byte node, the y and z steps as floats (word 3 then holds
 *   the x step as a float instead of the three exponent bytes)
 * decode: lo = origin + float(qlo) * scale, hi = origin + float(qhi) * scale.  Quantisation is conservative with
 * respect to exactly that decode expression (lo' <= lo, hi' >= hi), so no hit can be lost; empty slots keep ref
 * 0xFFFFFFFF and are skipped by reference. */
static void quantize_node(const float mn[4][3], const float mx[4][3], const uint32_t ref[4], uint32_t ns, int float_steps, uint32_t out[16]) {
    float org[3]; uint32_t eb[3]; uint8_t q[4][6];
    memset(q, 0, sizeof(q));
    for (int a = 0; a < 3; a++) {
        float lo = mn[0][a], hi = mx[0][a];
        for (uint32_t k = 1; k < ns; k++) { lo = fminx(lo, mn[k][a]); hi = fmaxx(hi, mx[k][a]); }
        org[a] = lo;
        float sdiv = (hi - lo) / 255.0f;
        uint32_t bits = f2u(sdiv), e = (bits >> 23) & 0xFFu;
        if (bits & 0x7FFFFFu) e += 1;
        if (e < 1) e = 1;
        for (;;) { /* grow the step until every child fits in 8 bits */
            float scale = u2f(e << 23);
            uint32_t worst = 0;
            for (uint32_t k = 0; k < ns; k++) {
                float fl = floorf((mn[k][a] - lo) / scale);
                uint32_t ql = fl > 255.0f ? 255u : (uint32_t)fl;
                while (ql > 0 && lo + (float)ql * scale > mn[k][a]) ql--;
                float fh = ceilf((mx[k][a] - lo) / scale);
                uint32_t qh = fh > 1024.0f ? 1024u : (uint32_t)fh;
                while (qh < 1024u && lo + (float)qh * scale < mx[k][a]) qh++;
                if (qh > worst) worst = qh;
                q[k][a] = (uint8_t)ql; q[k][3 + a] = (uint8_t)(qh > 255u ? 255u : qh);
            }
            if (worst <= 255u) break;
            e++;
        }
        eb[a] = e;
    }
    memset(out, 0, 64);
    out[0] = f2u(org[0]); out[1] = f2u(org[1]); out[2] = f2u(org[2]);
    out[3] = eb[0] | (eb[1] << 8) | (eb[2] << 16);
    uint8_t *bytes = (uint8_t *)(out + 4);
    for (uint32_t k = 0; k < 4; k++) for (int j = 0; j < 6; j++) bytes[6 * k + j] = k < ns ? q[k][j] : (j < 3 ? 255 : 0);
    for (uint32_t k = 0; k < 4; k++) out[10 + k] = k < ns ? ref[k] : 0xFFFFFFFFu;
    if (float_steps) { /* the 64-byte node: the three steps as floats in word 3 and the spare words 14, 15 (the kernels multiply them as they are) */
        out[3] = eb[0] << 23; out[14] = eb[1] << 23; out[15] = eb[2] << 23;
    }
}
/* Compact 48-byte node (node_quant 2): words 0..9 as above, but the four references are implied.  The internal children of
 * a node are numbered consecutively from node_base and the triangles of its leaf children are stored consecutively from
 * tri_base (both in slot order), so a child is described by one nibble: 0 = internal node, 8 | (count-1) = leaf of 1..8
 * triangles, 7 = empty slot.  Nibbles of slots 0,1 sit in bits 24..31 of word 3, slot 2 / 3 in the top nibble of
 * word 10 (node_base, 28 bits) / word 11 (tri_base, 28 bits).  Three 16-byte loads per node instead of four. */
static void compact_node(const float mn[4][3], const float mx[4][3], const uint32_t meta[4], uint32_t ns, uint32_t node_base,
                         uint32_t tri_base, uint32_t out[12]) {
    uint32_t tmp[16], ref[4] = {0, 0, 0, 0};
    quantize_node(mn, mx, ref, ns, 0, tmp);
    memcpy(out, tmp, 40);
    uint32_t m[4];
    for (uint32_t k = 0; k < 4; k++) m[k] = k < ns ? meta[k] : 7u;
    out[3] |= (m[0] << 24) | (m[1] << 28);
    out[10] = (node_base & 0x0FFFFFFFu) | (m[2] << 28);
    out[11] = (tri_base & 0x0FFFFFFFu) | (m[3] << 28);
}
static void compact_refs(const uint32_t *nd, uint32_t ref[4]) {
    uint32_t m[4] = {(nd[3] >> 24) & 15u, nd[3] >> 28, nd[10] >> 28, nd[11] >> 28};
    uint32_t nb = nd[10] & 0x0FFFFFFFu, tb = nd[11] & 0x0FFFFFFFu;
    for (int k = 0; k < 4; k++) {
        if (m[k] == 0u) ref[k] = nb++;
        else if (m[k] & 8u) { uint32_t c = (m[k] & 7u) + 1u; ref[k] = 0x80000000u | ((c - 1u) << 28) | tb; tb += c; }
        else ref[k] = 0xFFFFFFFFu;
    }
}
/* the decode expression quantize_node() is conservative to (documentation; traversal folds it into the ray, see slab_q) */
__attribute__((unused)) static void dequantize_slot(const uint32_t *nd, int k, float box[6]) {
    const uint8_t *bytes = (const uint8_t *)(nd + 4);
    for (int a = 0; a < 3; a++) {
        float scale = u2f(a == 0 ? nd[3] : nd[13 + a]), org = u2f(nd[a]); /* (64-byte node; the 48-byte one keeps exponent bytes in word 3) */
        box[a] = org + (float)bytes[6 * k + a] * scale;
        box[3 + a] = org + (float)bytes[6 * k + 3 + a] * scale;
    }
}
static void accel_free(orc_scene *s) {
    free(s->nodes); free(s->tris); free(s->codes);
    s->nodes = s->tris = NULL; s->codes = NULL; s->n_tris = s->n_nodes = 0;
}
void orc_scene_destroy(orc_scene *s) {
    if (!s) return;
    accel_free(s);
    free(s->verts); free(s->indices); free(s->geoms); free(s->prim_counts); free(s->first_prim); free(s->prim_geom);
    free(s->inst); free(s->flat_geom); free(s->flat_inst); free(s->flat_identity);
    free(s->sky); free(s->sky_alias); free(s->sky_q); free(s->cdf_marg); free(s->pdf_uv); free(s->bn);
    for (uint32_t i = 0; i < s->n_tex; i++) free(s->tex[i]);
    free(s->tex); free(s->tex_w); free(s->tex_h);
    free(s);
}
int orc_scene_set_vertices(orc_scene *s, const float *pnt, uint32_t n) {
    free(s->verts);
    s->verts = (float *)malloc((size_t)n * 32 + 4);
    memcpy(s->verts, pnt, (size_t)n * 32);
    s->n_verts = n;
    return 0;
}
int orc_scene_set_indices(orc_scene *s, const uint32_t *idx, uint32_t n) {
    free(s->indices);
    s->indices = (uint32_t *)malloc((size_t)n * 4 + 4);
    memcpy(s->indices, idx, (size_t)n * 4);
    s->n_indices = n;
    return 0;
}
/* One (instance, geometry) pair per flattened geometry, instance-major; no instances = one identity instance of everything.
 * The product's flatten_world (rt3_api.hip) makes the same tables. */
static const float k_identity16[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
static int flatten_world(orc_scene *s) {
    free(s->first_prim); free(s->prim_geom); free(s->flat_geom); free(s->flat_inst); free(s->flat_identity);
    s->first_prim = s->prim_geom = s->flat_geom = s->flat_inst = NULL; s->flat_identity = NULL;
    orc_instance whole; whole.geometry_first = 0; whole.geometry_count = s->n_geoms; memcpy(whole.transform, k_identity16, 64);
    const orc_instance *inst = s->n_inst ? s->inst : &whole;
    uint32_t ni = s->n_inst ? s->n_inst : 1u, nf = 0;
    for (uint32_t i = 0; i < ni; i++) {
        if ((uint64_t)inst[i].geometry_first + inst[i].geometry_count > s->n_geoms) return -1;
        nf += inst[i].geometry_count;
    }
    s->flat_geom = (uint32_t *)malloc((size_t)nf * 4 + 4); s->flat_inst = (uint32_t *)malloc((size_t)nf * 4 + 4);
    s->flat_identity = (uint8_t *)malloc((size_t)nf + 4); s->first_prim = (uint32_t *)malloc((size_t)nf * 4 + 4);
    uint32_t total = 0, j = 0;
    for (uint32_t i = 0; i < ni; i++)
        for (uint32_t k = 0; k < inst[i].geometry_count; k++, j++) {
            s->flat_geom[j] = inst[i].geometry_first + k; s->flat_inst[j] = i;
            s->flat_identity[j] = memcmp(inst[i].transform, k_identity16, 64) == 0;
            s->first_prim[j] = total; total += s->prim_counts[s->flat_geom[j]];
        }
    s->n_flat = nf; s->n_prims = total;
    s->prim_geom = (uint32_t *)malloc((size_t)total * 4 + 4);
    for (j = 0; j < nf; j++)
        for (uint32_t k = 0; k < s->prim_counts[s->flat_geom[j]]; k++) s->prim_geom[s->first_prim[j] + k] = j;
    return 0;
}
int orc_scene_set_geometry(orc_scene *s, const orc_geometry_info *g, const uint32_t *prim_counts, uint32_t n) {
    free(s->geoms); free(s->prim_counts);
    s->geoms = (orc_geometry_info *)malloc((size_t)n * sizeof(*g) + 4);
    memcpy(s->geoms, g, (size_t)n * sizeof(*g));
    s->prim_counts = (uint32_t *)malloc((size_t)n * 4 + 4);
    for (uint32_t i = 0; i < n; i++) s->prim_counts[i] = prim_counts[i];
    s->n_geoms = n;
    if (s->n_inst && flatten_world(s)) { s->n_inst = 0; } /* instances that no longer fit the geometries are dropped */
    return flatten_world(s);
}
/* Instance{model} + Transform{Mat4} (world/mod.rs:46-60): instance i places geometries [first, first + count) under its column-major
 * matrix; n = 0 restores the default (everything once, identity).  Takes effect at the next orc_accel_build. */
int orc_scene_set_instances(orc_scene *s, const orc_instance *inst, uint32_t n) {
    free(s->inst); s->inst = NULL; s->n_inst = 0;
    if (n) { s->inst = (orc_instance *)malloc((size_t)n * sizeof(*inst)); memcpy(s->inst, inst, (size_t)n * sizeof(*inst)); s->n_inst = n; }
    if (flatten_world(s)) { free(s->inst); s->inst = NULL; s->n_inst = 0; flatten_world(s); return -1; }
    return 0;
}
/* glam Mat4::transform_point3: ((x_axis * x + y_axis * y) + z_axis * z) + w_axis */
static void transform_point(const float *m, float p[3]) {
    float x = p[0], y = p[1], z = p[2];
    for (int r = 0; r < 3; r++) p[r] = m[12 + r] + (m[8 + r] * z + (m[4 + r] * y + m[r] * x));
}
int orc_scene_set_bluenoise(orc_scene *s, const uint8_t *rgba, uint32_t w, uint32_t h) {
    free(s->bn);
    s->bn = (uint8_t *)malloc((size_t)w * h * 4);
    memcpy(s->bn, rgba, (size_t)w * h * 4);
    s->bn_w = w; s->bn_h = h;
    return 0;
}
/* texture `index` = RGBA8 image with sRGB-encoded colour (what a glTF baseColorTexture is); indices must be set 0,1,2,... */
int orc_scene_set_texture(orc_scene *s, uint32_t index, const uint8_t *rgba, uint32_t w, uint32_t h) {
    if (index > s->n_tex || !w || !h) return -1;
    if (index == s->n_tex) {
        s->tex = (uint8_t **)realloc(s->tex, (size_t)(index + 1) * sizeof(uint8_t *));
        s->tex_w = (uint32_t *)realloc(s->tex_w, (size_t)(index + 1) * 4);
        s->tex_h = (uint32_t *)realloc(s->tex_h, (size_t)(index + 1) * 4);
        s->tex[index] = NULL;
        s->n_tex = index + 1;
    }
    free(s->tex[index]);
    s->tex[index] = (uint8_t *)malloc((size_t)w * h * 4);
    memcpy(s->tex[index], rgba, (size_t)w * h * 4);
    s->tex_w[index] = w; s->tex_h[index] = h;
    return 0;
}
/* Textures[i].SampleLevel(uv, 0).xyz (hit_logic.slang:32): sRGB decode per texel, bilinear, repeat addressing, mip 0 */
static void texture_sample(const orc_scene *s, uint32_t index, float u, float v, float out[3]) {
    int W = (int)s->tex_w[index], H = (int)s->tex_h[index];
    const uint8_t *px = s->tex[index];
    float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
    float xf = floorf(x), yf = floorf(y), fx = x - xf, fy = y - yf;
    int x0 = (int)xf, y0 = (int)yf, x1 = x0 + 1, y1 = y0 + 1;
    x0 = ((x0 % W) + W) % W; x1 = ((x1 % W) + W) % W; y0 = ((y0 % H) + H) % H; y1 = ((y1 % H) + H) % H;
    const uint8_t *p00 = px + 4 * ((size_t)y0 * W + x0), *p10 = px + 4 * ((size_t)y0 * W + x1);
    const uint8_t *p01 = px + 4 * ((size_t)y1 * W + x0), *p11 = px + 4 * ((size_t)y1 * W + x1);
    for (int k = 0; k < 3; k++) {
        float top = s->srgb_lut[p00[k]] * (1.0f - fx) + s->srgb_lut[p10[k]] * fx;
        float bot = s->srgb_lut[p01[k]] * (1.0f - fx) + s->srgb_lut[p11[k]] * fx;
        out[k] = top * (1.0f - fy) + bot * fy;
    }
}
/* math.slang:119-122 */
static float luminance3(const float c[3]) { return c[0] * 0.299f + c[1] * 0.587f + c[2] * 0.114f; }
/* [north_star] sky storage and importance tables.
 *  - Radiance is STORED as RGB9E5 (packing.slang:99-162, the format the reference's G-buffer keeps emissive in): every sky lookup
 *    -- background, escaped paths, light samples -- reads the de-quantised value, and the tables below are built from it.
 *  - f = (luminance + 1e-6) * sin(theta) per texel.  Rows: marginal CDF (inverted by search).  Inside a row: ONE alias table
 *    (Vose 1991) -- a light sample costs one table word instead of a CDF search: entry k = {q16, alias}: keep column k if
 *    xi < Q = (q16 + 1) / 65536, else take column `alias` (xi = frac(u * w)).  Q is 16-bit, so the probability a column REALLY
 *    has is (Q[x] + sum over j with alias[j] = x of (1 - Q[j])) / w; pdf_uv stores that realised density, which keeps the
 *    estimator exact for the quantised table (Q >= 2^-16: no column of a row is unreachable).
 *    All in double, fixed order; librt3 builds the identical tables (rt3_scene_set_sky). */
int orc_scene_set_sky(orc_scene *s, const float *rgb, uint32_t w, uint32_t h) {
    for (size_t i = 0; i < (size_t)w * h * 3; i++) /* same contract as rt3_scene_set_sky: finite, non-negative radiance */
        if (!(rgb[i] >= 0.0f && rgb[i] <= 3.4028234663852886e38f)) return -1;
    if (w > 65535u || h > 65535u) return -1;
    free(s->sky); free(s->sky_alias); free(s->sky_q); free(s->cdf_marg); free(s->pdf_uv);
    size_t n = (size_t)w * h;
    s->sky = (float *)malloc(n * 12);
    s->sky_q = (uint32_t *)malloc(n * 4);
    for (size_t i = 0; i < n; i++) {
        s->sky_q[i] = orc_float3_to_rgb9e5(rgb + 3 * i);
        orc_rgb9e5_to_float3(s->sky_q[i], s->sky + 3 * i);
    }
    s->sky_w = w; s->sky_h = h;
    s->sky_alias = (uint32_t *)malloc(n * 4);
    s->pdf_uv = (float *)malloc(n * 4);
    s->cdf_marg = (float *)malloc((size_t)h * 4);
    double *rowsum = (double *)malloc((size_t)h * 8), total = 0.0;
    double *f = (double *)malloc((size_t)w * 8), *sc = (double *)malloc((size_t)w * 8), *real = (double *)malloc((size_t)w * 8);
    uint32_t *small = (uint32_t *)malloc((size_t)w * 4), *large = (uint32_t *)malloc((size_t)w * 4);
    for (uint32_t y = 0; y < h; y++) {
        double st = sin(3.14159265358979323846 * ((double)y + 0.5) / (double)h), acc = 0.0;
        for (uint32_t x = 0; x < w; x++) {
            f[x] = ((double)luminance3(s->sky + 3 * ((size_t)y * w + x)) + 1e-6) * st;
            acc += f[x];
        }
        rowsum[y] = acc;
        total += acc;
        /* Vose's alias method, deterministic: columns enter the two stacks in ascending order, both are popped from the top */
        uint32_t ns = 0, nl = 0;
        uint32_t *al = s->sky_alias + (size_t)y * w;
        for (uint32_t x = 0; x < w; x++) {
            sc[x] = f[x] * (double)w / acc;
            if (sc[x] < 1.0) small[ns++] = x; else large[nl++] = x;
        }
        for (uint32_t x = 0; x < w; x++) al[x] = 65535u | (x << 16); /* default: always keep (also what is left on a stack at the end) */
        while (ns && nl) {
            uint32_t a = small[--ns], g = large[--nl];
            double q = sc[a] * 65536.0;
            int64_t q16 = (int64_t)floor(q + 0.5) - 1;
            q16 = q16 < 0 ? 0 : (q16 > 65535 ? 65535 : q16);
            al[a] = (uint32_t)q16 | (g << 16);
            sc[g] = (sc[g] + sc[a]) - 1.0;
            if (sc[g] < 1.0) small[ns++] = g; else large[nl++] = g;
        }
        /* realised column probabilities of the quantised table */
        for (uint32_t x = 0; x < w; x++) real[x] = 0.0;
        for (uint32_t x = 0; x < w; x++) {
            double Q = (double)((al[x] & 0xFFFFu) + 1u) / 65536.0;
            real[x] += Q;
            real[al[x] >> 16] += 1.0 - Q;
        }
        for (uint32_t x = 0; x < w; x++) s->pdf_uv[(size_t)y * w + x] = (float)real[x]; /* x row probability x h below */
    }
    double acc = 0.0;
    for (uint32_t y = 0; y < h; y++) {
        acc += rowsum[y];
        s->cdf_marg[y] = (float)(acc / total);
        const double rowp = rowsum[y] / total * (double)h;
        for (uint32_t x = 0; x < w; x++) s->pdf_uv[(size_t)y * w + x] = (float)((double)s->pdf_uv[(size_t)y * w + x] * rowp);
    }
    s->cdf_marg[h - 1] = 1.0f;
    free(rowsum); free(f); free(sc); free(real); free(small); free(large);
    return 0;
}
const uint32_t *orc_sky_alias(const orc_scene *s) { return s->sky_alias; }
const uint32_t *orc_sky_texels(const orc_scene *s) { return s->sky_q; }
const float *orc_sky_cdf_marg(const orc_scene *s) { return s->cdf_marg; }
const float *orc_sky_pdf_uv(const orc_scene *s) { return s->pdf_uv; }

/* ------------------------------------------------------------------------------------------------ LBVH [north_star] */
/* Replaces create_acceleration_structure (raytracing.rs:88-148): 63-bit Morton codes of the triangle-box centres,
 * sort by (code, primitive), Karras 2012 radix-tree topology, bottom-up boxes, 64 B two-child-box nodes. */
static uint64_t expand21(uint32_t v) {
    uint64_t x = v & 0x1FFFFFu;
    x = (x | (x << 32)) & 0x001F00000000FFFFull;
    x = (x | (x << 16)) & 0x001F0000FF0000FFull;
    x = (x | (x << 8)) & 0x100F00F00F00F00Full;
    x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}
static void tri_positions(const orc_scene *s, uint32_t prim, float a[3], float b[3], float c[3]) {
    uint32_t fg = s->prim_geom[prim], local = prim - s->first_prim[fg];
    const orc_geometry_info *gi = &s->geoms[s->flat_geom[fg]];
    uint32_t io = gi->index_offset + 3u * local;
    const float *v0 = s->verts + 8 * (size_t)(gi->vertex_offset + s->indices[io]);
    const float *v1 = s->verts + 8 * (size_t)(gi->vertex_offset + s->indices[io + 1]);
    const float *v2 = s->verts + 8 * (size_t)(gi->vertex_offset + s->indices[io + 2]);
    for (int k = 0; k < 3; k++) { a[k] = v0[k]; b[k] = v1[k]; c[k] = v2[k]; }
    if (!s->flat_identity[fg]) { /* world space: the instance's matrix (identity instances keep the uploaded bits) */
        const float *m = s->inst[s->flat_inst[fg]].transform;
        transform_point(m, a); transform_point(m, b); transform_point(m, c);
    }
}
typedef struct { uint64_t code; uint32_t prim; } code_prim;
static int cmp_code_prim(const void *pa, const void *pb) {
    const code_prim *a = (const code_prim *)pa, *b = (const code_prim *)pb;
    if (a->code != b->code) return a->code < b->code ? -1 : 1;
    return a->prim < b->prim ? -1 : (a->prim > b->prim ? 1 : 0);
}
static int delta_fn(const uint64_t *codes, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    uint64_t a = codes[i], b = codes[j];
    if (a != b) return __builtin_clzll(a ^ b);
    return 64 + __builtin_clz((uint32_t)i ^ (uint32_t)j);
}
/* ---- SAH top (hierarchical LBVH, after Pantaleoni & Luebke 2010 / Garanzha et al. 2011): the Karras tree is kept below
 * "cluster roots" (maximal subtrees of at most T triangles: contiguous Morton ranges); the C - 1 nodes above the C cluster roots
 * are re-linked into a tree built top-down by binned SAH (16 bins on the cluster centroids, cost = half area x triangle count)
 * over the cluster boxes.  Node indices are reused (the top of a binary tree with C leaves has C - 1 nodes), the root stays
 * node 0.  fp32 in a fixed order; the product's host code (rt3_lbvh.hip, sah_top_relink) runs the same algorithm. */
typedef struct { uint32_t ref, cnt; float mn[3], mx[3]; } sah_cluster;
static inline float half_area3(const float mn[3], const float mx[3]) {
    float ex = mx[0] - mn[0], ey = mx[1] - mn[1], ez = mx[2] - mn[2];
    return (ex * ey + ey * ez) + ez * ex;
}
static void sah_top_rebuild(uint32_t nn, uint32_t *left, uint32_t *right, uint32_t *rlo, uint32_t *rcnt, const float *lmin, const float *lmax,
                            const float *nmin, const float *nmax, uint32_t T) {
    uint32_t nc = 0, np = 0;
    sah_cluster *cl = (sah_cluster *)malloc(((size_t)nn + 2) * sizeof(sah_cluster));
    uint32_t *pool = (uint32_t *)malloc((size_t)nn * 4);
    for (uint32_t i = 0; i < nn; i++) {
        if (!(i == 0 || rcnt[i] > T)) continue;
        pool[np++] = i;
        uint32_t c2[2] = {left[i], right[i]};
        for (int c = 0; c < 2; c++) {
            uint32_t ch = c2[c];
            if (!(ch & 0x80000000u) && rcnt[ch] > T) continue;
            sah_cluster *k = &cl[nc++];
            k->ref = ch;
            if (ch & 0x80000000u) { uint32_t q = ch & 0x7FFFFFFFu; k->cnt = 1; memcpy(k->mn, lmin + 3 * q, 12); memcpy(k->mx, lmax + 3 * q, 12); }
            else { k->cnt = rcnt[ch]; memcpy(k->mn, nmin + 3 * ch, 12); memcpy(k->mx, nmax + 3 * ch, 12); }
        }
    }
    if (nc < 3 || np != nc - 1) { free(cl); free(pool); return; }
    uint32_t *idx = (uint32_t *)malloc((size_t)nc * 4), *tmp = (uint32_t *)malloc((size_t)nc * 4);
    for (uint32_t i = 0; i < nc; i++) idx[i] = i;
    typedef struct { uint32_t a, n, patch; } job;
    job *st = (job *)malloc((size_t)nc * 2 * sizeof(job) + 64);
    int sp = 0; uint32_t next_pool = 0;
    st[sp++] = (job){0, nc, 0xFFFFFFFFu};
    while (sp > 0) {
        job j = st[--sp];
        uint32_t ref;
        if (j.n == 1) ref = cl[idx[j.a]].ref;
        else {
            uint32_t node = pool[next_pool++];
            ref = node;
            float cmn[3] = {INFINITY, INFINITY, INFINITY}, cmx[3] = {-INFINITY, -INFINITY, -INFINITY};
            uint32_t total = 0;
            for (uint32_t k = 0; k < j.n; k++) {
                const sah_cluster *c = &cl[idx[j.a + k]];
                total += c->cnt;
                for (int a = 0; a < 3; a++) { float ce = (c->mn[a] + c->mx[a]) * 0.5f; cmn[a] = fminx(cmn[a], ce); cmx[a] = fmaxx(cmx[a], ce); }
            }
            float best_cost = INFINITY; int best_axis = -1, best_split = 0;
            for (int a = 0; a < 3; a++) {
                float ext = cmx[a] - cmn[a];
                if (!(ext > 0.0f)) continue;
                float bmn[16][3], bmx[16][3]; uint32_t bc[16];
                for (int b = 0; b < 16; b++) { bc[b] = 0; for (int q = 0; q < 3; q++) { bmn[b][q] = INFINITY; bmx[b][q] = -INFINITY; } }
                for (uint32_t k = 0; k < j.n; k++) {
                    const sah_cluster *c = &cl[idx[j.a + k]];
                    float ce = (c->mn[a] + c->mx[a]) * 0.5f;
                    int b = (int)(((ce - cmn[a]) / ext) * 16.0f); if (b > 15) b = 15;
                    bc[b] += c->cnt;
                    for (int q = 0; q < 3; q++) { bmn[b][q] = fminx(bmn[b][q], c->mn[q]); bmx[b][q] = fmaxx(bmx[b][q], c->mx[q]); }
                }
                float rmn[16][3], rmx[16][3]; uint32_t rc[16];
                for (int b = 15; b >= 0; b--) {
                    for (int q = 0; q < 3; q++) { rmn[b][q] = b == 15 ? bmn[b][q] : fminx(bmn[b][q], rmn[b + 1][q]); rmx[b][q] = b == 15 ? bmx[b][q] : fmaxx(bmx[b][q], rmx[b + 1][q]); }
                    rc[b] = bc[b] + (b == 15 ? 0u : rc[b + 1]);
                }
                float lmn[3] = {INFINITY, INFINITY, INFINITY}, lmx[3] = {-INFINITY, -INFINITY, -INFINITY}; uint32_t lc = 0;
                for (int sgl = 1; sgl < 16; sgl++) {
                    for (int q = 0; q < 3; q++) { lmn[q] = fminx(lmn[q], bmn[sgl - 1][q]); lmx[q] = fmaxx(lmx[q], bmx[sgl - 1][q]); }
                    lc += bc[sgl - 1];
                    if (lc == 0 || rc[sgl] == 0) continue;
                    float cost = half_area3(lmn, lmx) * (float)lc + half_area3(rmn[sgl], rmx[sgl]) * (float)rc[sgl];
                    if (cost < best_cost) { best_cost = cost; best_axis = a; best_split = sgl; }
                }
            }
            uint32_t nl = 0;
            if (best_axis < 0) nl = j.n / 2;
            else {
                float ext = cmx[best_axis] - cmn[best_axis]; uint32_t w = 0, r = 0;
                for (uint32_t k = 0; k < j.n; k++) {
                    const sah_cluster *c = &cl[idx[j.a + k]];
                    float ce = (c->mn[best_axis] + c->mx[best_axis]) * 0.5f;
                    int b = (int)(((ce - cmn[best_axis]) / ext) * 16.0f); if (b > 15) b = 15;
                    if (b < best_split) idx[j.a + w++] = idx[j.a + k]; else tmp[r++] = idx[j.a + k];
                }
                memcpy(idx + j.a + w, tmp, (size_t)r * 4);
                nl = w;
            }
            rcnt[node] = total > T ? total : T + 1u; rlo[node] = 0; /* a re-linked node is never a multi-triangle leaf: its triangles are not contiguous */
            st[sp++] = (job){j.a + nl, j.n - nl, (node << 1) | 1u};
            st[sp++] = (job){j.a, nl, (node << 1)};
        }
        if (j.patch != 0xFFFFFFFFu) { if (j.patch & 1u) right[j.patch >> 1] = ref; else left[j.patch >> 1] = ref; }
    }
    free(cl); free(pool); free(idx); free(tmp); free(st);
}
/* ---- [experiment, round 3] insertion-based optimisation of the TOP of the tree (after Bittner, Hapala, Havran 2013).  The treelet =
 * the internal nodes covering more than k_top triangles; its leaves = their children that do not (fixed subtrees).  Every treelet
 * element in turn is taken out (its parent goes with it, the sibling moves up) and put back where the summed half area of the treelet's
 * internal nodes grows least (exhaustive search with pruning; the old place is among the candidates, so the sum never grows).
 * Sequential and deterministic.  Node ids are reused (the freed parent becomes the new parent). */
typedef struct { uint32_t *left, *right, *par, *parleaf, *cnt; float *nmin, *nmax; const float *lmin, *lmax; uint32_t ktop; } topt;
static inline const float *tbox_mn(const topt *t, uint32_t r) { return (r & 0x80000000u) ? t->lmin + 3 * (size_t)(r & 0x7FFFFFFFu) : t->nmin + 3 * (size_t)r; }
static inline const float *tbox_mx(const topt *t, uint32_t r) { return (r & 0x80000000u) ? t->lmax + 3 * (size_t)(r & 0x7FFFFFFFu) : t->nmax + 3 * (size_t)r; }
static inline uint32_t tcount(const topt *t, uint32_t r) { return (r & 0x80000000u) ? 1u : t->cnt[r]; }
static inline int t_internal(const topt *t, uint32_t r) { return !(r & 0x80000000u) && t->cnt[r] > t->ktop; }
static inline void tset_parent(topt *t, uint32_t r, uint32_t p) { if (r & 0x80000000u) t->parleaf[r & 0x7FFFFFFFu] = p; else t->par[r] = p; }
static inline uint32_t tget_parent(const topt *t, uint32_t r) { return (r & 0x80000000u) ? t->parleaf[r & 0x7FFFFFFFu] : t->par[r]; }
static void trefit_up(topt *t, uint32_t i) {
    while (i != 0xFFFFFFFFu) {
        uint32_t l = t->left[i], r = t->right[i];
        for (int j = 0; j < 3; j++) { t->nmin[3 * (size_t)i + j] = fminx(tbox_mn(t, l)[j], tbox_mn(t, r)[j]); t->nmax[3 * (size_t)i + j] = fmaxx(tbox_mx(t, l)[j], tbox_mx(t, r)[j]); }
        t->cnt[i] = tcount(t, l) + tcount(t, r);
        i = t->par[i];
    }
}
static inline float union_area(const float *amn, const float *amx, const float *bmn, const float *bmx) {
    float mn[3], mx[3];
    for (int j = 0; j < 3; j++) { mn[j] = fminx(amn[j], bmn[j]); mx[j] = fmaxx(amx[j], bmx[j]); }
    return half_area3(mn, mx);
}
static void top_optimize(uint32_t nn, uint32_t n_leaves, uint32_t *left, uint32_t *right, uint32_t *rcnt, float *nmin, float *nmax, const float *lmin,
                         const float *lmax, uint32_t ktop, uint32_t passes) {
    topt t = {left, right, (uint32_t *)malloc((size_t)nn * 4), (uint32_t *)malloc((size_t)n_leaves * 4), rcnt, nmin, nmax, lmin, lmax, ktop};
    /* true counts + parents (pre-order list, then reversed) */
    uint32_t *order = (uint32_t *)malloc((size_t)nn * 4), no = 0, *stk = (uint32_t *)malloc((size_t)nn * 8 + 64);
    { int sp = 0; stk[sp++] = 0; t.par[0] = 0xFFFFFFFFu;
      while (sp > 0) { uint32_t i = stk[--sp]; order[no++] = i; uint32_t c2[2] = {left[i], right[i]};
          for (int c = 0; c < 2; c++) { tset_parent(&t, c2[c], i); if (!(c2[c] & 0x80000000u)) stk[sp++] = c2[c]; } } }
    for (uint32_t q = no; q-- > 0;) { uint32_t i = order[q]; rcnt[i] = tcount(&t, left[i]) + tcount(&t, right[i]); }
    /* the elements to move: every child of a treelet-internal node except the root's own children; largest parent box first would be Bittner's
     * priority -- a fixed pre-order sweep per pass is enough for a prototype */
    uint32_t *elems = (uint32_t *)malloc((size_t)nn * 8 + 64), ne;
    double before = 0.0, after = 0.0;
    for (uint32_t q = 0; q < no; q++) if (t_internal(&t, order[q])) before += half_area3(nmin + 3 * (size_t)order[q], nmax + 3 * (size_t)order[q]);
    for (uint32_t pass = 0; pass < passes; pass++) {
        ne = 0;
        { int sp = 0; stk[sp++] = 0;
          while (sp > 0) { uint32_t i = stk[--sp]; uint32_t c2[2] = {left[i], right[i]};
              for (int c = 0; c < 2; c++) { if (i != 0) elems[ne++] = c2[c]; if (t_internal(&t, c2[c])) stk[sp++] = c2[c]; } } }
        for (uint32_t e = 0; e < ne; e++) {
            const uint32_t nref = elems[e], p = tget_parent(&t, nref);
            if (p == 0xFFFFFFFFu || p == 0u) continue;            /* (moved under the root meanwhile) */
            const uint32_t g = t.par[p];
            if (g == 0xFFFFFFFFu) continue;
            const uint32_t sib = left[p] == nref ? right[p] : left[p];
            /* take n (and p) out */
            if (left[g] == p) left[g] = sib; else right[g] = sib;
            tset_parent(&t, sib, g);
            trefit_up(&t, g);
            if (!t_internal(&t, g) && g != 0) { /* g fell out of the treelet: keep it simple, undo */
                if (left[g] == sib) left[g] = p; else right[g] = p;
                tset_parent(&t, sib, p); t.par[p] = g; trefit_up(&t, p); continue;
            }
            const float *bmn = tbox_mn(&t, nref), *bmx = tbox_mx(&t, nref);
            /* best place: DFS over the treelet with the induced cost of the ancestors carried along */
            float best = INFINITY; uint32_t best_x = sib;
            typedef struct { uint32_t ref; float induced; } cand;
            cand *cs = (cand *)stk; int sp = 0;
            cs[sp++] = (cand){0u, 0.0f};
            const float an = half_area3(bmn, bmx);
            while (sp > 0) {
                cand c = cs[--sp];
                if (c.induced + an >= best) continue;
                const float *xmn = tbox_mn(&t, c.ref), *xmx = tbox_mx(&t, c.ref);
                const float ua = union_area(xmn, xmx, bmn, bmx);
                if (c.ref != 0u) { float cost = c.induced + ua; if (cost < best) { best = cost; best_x = c.ref; } }
                if (t_internal(&t, c.ref)) {
                    float ind = c.induced + (ua - half_area3(xmn, xmx));
                    cs[sp++] = (cand){right[c.ref], ind};
                    cs[sp++] = (cand){left[c.ref], ind};
                }
            }
            /* put it back: p becomes the parent of {best_x, n} in best_x's place */
            const uint32_t px = tget_parent(&t, best_x);
            if (left[px] == best_x) left[px] = p; else right[px] = p;
            t.par[p] = px; left[p] = best_x; right[p] = nref;
            tset_parent(&t, best_x, p); tset_parent(&t, nref, p);
            trefit_up(&t, p);
        }
    }
    { int sp = 0; stk[sp++] = 0; while (sp > 0) { uint32_t i = stk[--sp]; after += half_area3(nmin + 3 * (size_t)i, nmax + 3 * (size_t)i);
          if (t_internal(&t, left[i])) stk[sp++] = left[i];
          if (t_internal(&t, right[i])) stk[sp++] = right[i]; } }
    if (getenv("ORC_TRACE_BUILD")) fprintf(stderr, "orc top_optimize: k_top %u, %u passes, treelet area sum %.1f -> %.1f\n", ktop, passes, before, after);
    free(t.par); free(t.parleaf); free(order); free(stk); free(elems);
}
/* child slots of four-wide node i under the surface-area collapse (see orc_accel_build); half area = (ex*ey + ey*ez) + ez*ex */
static uint32_t sah_slots(uint32_t i, const uint32_t *left, const uint32_t *right, const uint32_t *rcnt, const float *nmin, const float *nmax,
                          uint32_t K, uint32_t sl[4]) {
    uint32_t ns = 2;
    sl[0] = left[i]; sl[1] = right[i];
    for (int it = 0; it < 2; it++) {
        int best = -1; float ba = -1.0f;
        for (uint32_t k = 0; k < ns; k++) {
            uint32_t ch = sl[k];
            if ((ch & 0x80000000u) || rcnt[ch] <= K) continue;
            float ex = nmax[3 * ch] - nmin[3 * ch], ey = nmax[3 * ch + 1] - nmin[3 * ch + 1], ez = nmax[3 * ch + 2] - nmin[3 * ch + 2];
            float a = (ex * ey + ey * ez) + ez * ex;
            if (a > ba) { ba = a; best = (int)k; }
        }
        if (best < 0) break;
        uint32_t ch = sl[best];
        for (int k = (int)ns; k > best + 1; k--) sl[k] = sl[k - 1];
        sl[best] = left[ch]; sl[best + 1] = right[ch];
        ns++;
    }
    return ns;
}
/* child slots of four-wide node i under the cost-driven collapse: the choices the bottom-up pass of orc_accel_build recorded in dk,
 * unfolded top-down, left subtree's slots first */
static uint32_t dp_slots(uint32_t i, const uint32_t *left, const uint32_t *right, const uint8_t *dk, uint32_t sl[4]) {
    uint32_t ns = 0, sn[8], sm[8]; int sp = 0;
    uint32_t k4 = dk[i] & 3u;
    sn[sp] = right[i]; sm[sp++] = 4u - k4;
    sn[sp] = left[i]; sm[sp++] = k4;
    while (sp > 0) {
        uint32_t nd = sn[--sp], m = sm[sp];
        if (nd & 0x80000000u) { sl[ns++] = nd; continue; }
        uint32_t f = dk[nd];
        if (m == 3u && !(f & 32u)) m = 2u;
        if (m == 2u && !(f & 16u)) m = 1u;
        if (m == 1u) { sl[ns++] = nd; continue; }
        uint32_t kl = m == 2u ? 1u : ((f >> 2) & 1u) + 1u;
        sn[sp] = right[nd]; sm[sp++] = m - kl;
        sn[sp] = left[nd]; sm[sp++] = kl;
    }
    return ns;
}
int orc_accel_build(orc_scene *s) {
    accel_free(s);
    uint32_t n = s->n_prims;
    s->n_tris = n;
    if (n == 0) return 0;
    float *bmin = (float *)malloc((size_t)n * 12), *bmax = (float *)malloc((size_t)n * 12);
    float cmin[3] = {INFINITY, INFINITY, INFINITY}, cmax[3] = {-INFINITY, -INFINITY, -INFINITY};
    float smin[3] = {INFINITY, INFINITY, INFINITY}, smax[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = 0; i < n; i++) {
        float a[3], b[3], c[3];
        tri_positions(s, i, a, b, c);
        for (int k = 0; k < 3; k++) {
            float lo = fminx(a[k], fminx(b[k], c[k])), hi = fmaxx(a[k], fmaxx(b[k], c[k]));
            bmin[3 * i + k] = lo; bmax[3 * i + k] = hi;
            float ce = (lo + hi) * 0.5f;
            cmin[k] = fminx(cmin[k], ce); cmax[k] = fmaxx(cmax[k], ce);
            smin[k] = fminx(smin[k], lo); smax[k] = fmaxx(smax[k], hi);
        }
    }
    /* conservative leaf padding so that the slab test never culls a triangle the fp32 triangle test accepts */
    float ext = fmaxx(smax[0] - smin[0], fmaxx(smax[1] - smin[1], smax[2] - smin[2]));
    float pad = ext * 1.0e-5f;
    code_prim *cp = (code_prim *)malloc((size_t)n * sizeof(code_prim));
    for (uint32_t i = 0; i < n; i++) {
        uint32_t q[3];
        for (int k = 0; k < 3; k++) {
            float e = cmax[k] - cmin[k];
            float ce = (bmin[3 * i + k] + bmax[3 * i + k]) * 0.5f;
            float nrm = e > 0.0f ? (ce - cmin[k]) / e : 0.0f;
            float qf = fminx(nrm * 2097152.0f, 2097151.0f);
            q[k] = (uint32_t)qf;
        }
        cp[i].code = (expand21(q[0]) << 2) | (expand21(q[1]) << 1) | expand21(q[2]);
        cp[i].prim = i;
    }
    qsort(cp, n, sizeof(code_prim), cmp_code_prim);
    s->codes = (uint64_t *)malloc((size_t)n * 8);
    s->tris = (float *)calloc((size_t)n * 12, 4);
    float *lmin = (float *)malloc((size_t)n * 12), *lmax = (float *)malloc((size_t)n * 12);
    for (uint32_t k = 0; k < n; k++) {
        uint32_t p = cp[k].prim;
        s->codes[k] = cp[k].code;
        float a[3], b[3], c[3];
        tri_positions(s, p, a, b, c);
        float *t = s->tris + 12 * (size_t)k;
        /* the three vertices exactly as uploaded (not v0 + edges): two triangles that share an edge must see bit-identical
         * end points, which is what makes the edge functions of tri_test watertight */
        t[0] = a[0]; t[1] = a[1]; t[2] = a[2];
        t[3] = b[0]; t[4] = b[1]; t[5] = b[2];
        t[6] = c[0]; t[7] = c[1]; t[8] = c[2];
        t[9] = u2f(p); t[10] = 0.0f; t[11] = 0.0f;
        for (int j = 0; j < 3; j++) { lmin[3 * k + j] = bmin[3 * p + j] - pad; lmax[3 * k + j] = bmax[3 * p + j] + pad; }
    }
    uint32_t nn = n > 1 ? n - 1 : 1;
    s->n_nodes = nn;
    if (n == 1) { /* root with the single leaf in slot 0, the other slots empty */
        const uint32_t W4 = s->node_width == 4, words = (W4 && !s->node_quant) ? 32u : 16u;
        s->nodes = (float *)calloc(words, 4);
        float *nd = s->nodes;
        if (s->node_quant) {
            float qmn[4][3], qmx[4][3]; uint32_t qref[4] = {0x80000000u, 0, 0, 0}, qmeta[4] = {8u, 7u, 7u, 7u};
            memcpy(qmn[0], lmin, 12); memcpy(qmx[0], lmax, 12);
            if (s->node_quant == 2) compact_node(qmn, qmx, qmeta, 1, 1u, 0u, (uint32_t *)nd);
            else quantize_node(qmn, qmx, qref, 1, 1, (uint32_t *)nd);
        } else
        for (uint32_t k = 0; k < (W4 ? 4u : 2u); k++) {
            float *mn = W4 ? nd + 8 * k : nd + 6 * k, *mx = mn + 3;
            for (int j = 0; j < 3; j++) { mn[j] = k ? INFINITY : lmin[j]; mx[j] = k ? -INFINITY : lmax[j]; }
            if (W4) nd[8 * k + 6] = u2f(k ? 0xFFFFFFFFu : 0x80000000u); else nd[12 + k] = u2f(k ? 0xFFFFFFFFu : 0x80000000u);
        }
        s->max_depth = 2;
    } else {
        uint32_t *left = (uint32_t *)malloc((size_t)nn * 4), *right = (uint32_t *)malloc((size_t)nn * 4);
        uint32_t *rlo = (uint32_t *)malloc((size_t)nn * 4), *rcnt = (uint32_t *)malloc((size_t)nn * 4);
        const uint64_t *codes = s->codes;
        int N = (int)n;
        for (int i = 0; i < N - 1; i++) { /* Karras 2012, Algorithm "construct internal node i" */
            int d = (delta_fn(codes, N, i, i + 1) - delta_fn(codes, N, i, i - 1)) >= 0 ? 1 : -1;
            int dmin = delta_fn(codes, N, i, i - d);
            int lmaxv = 2;
            while (delta_fn(codes, N, i, i + lmaxv * d) > dmin) lmaxv *= 2;
            int l = 0;
            for (int t = lmaxv / 2; t >= 1; t /= 2)
                if (delta_fn(codes, N, i, i + (l + t) * d) > dmin) l += t;
            int j = i + l * d;
            int dnode = delta_fn(codes, N, i, j);
            int sp = 0, t = l;
            do {
                t = (t + 1) >> 1;
                if (delta_fn(codes, N, i, i + (sp + t) * d) > dnode) sp += t;
            } while (t > 1);
            int gamma = i + sp * d + (d < 0 ? d : 0);
            int lo = i < j ? i : j, hi = i < j ? j : i;
            left[i] = (lo == gamma) ? (0x80000000u | (uint32_t)gamma) : (uint32_t)gamma;
            right[i] = (hi == gamma + 1) ? (0x80000000u | (uint32_t)(gamma + 1)) : (uint32_t)(gamma + 1);
            rlo[i] = (uint32_t)lo; rcnt[i] = (uint32_t)(hi - lo + 1);
        }
        /* boxes of every binary node (post-order from the root) and its binary depth (root = 0) */
        float *nmin = (float *)malloc((size_t)nn * 12), *nmax = (float *)malloc((size_t)nn * 12);
        uint32_t *stack = (uint32_t *)malloc((size_t)nn * 2 * 4 + 64);
        uint8_t *state = (uint8_t *)calloc(nn, 1);
        uint32_t *depth = (uint32_t *)calloc(nn, 4);
        for (int stage = 0;;) {
        memset(state, 0, nn); memset(depth, 0, (size_t)nn * 4);
        int sp = 0;
        stack[sp++] = 0;
        while (sp > 0) {
            uint32_t i = stack[sp - 1];
            uint32_t l = left[i], r = right[i];
            if (!state[i]) {
                state[i] = 1;
                if (!(l & 0x80000000u)) { depth[l] = depth[i] + 1; stack[sp++] = l; }
                if (!(r & 0x80000000u)) { depth[r] = depth[i] + 1; stack[sp++] = r; }
            } else {
                sp--;
                const float *amn = (l & 0x80000000u) ? lmin + 3 * (l & 0x7FFFFFFFu) : nmin + 3 * l, *amx = (l & 0x80000000u) ? lmax + 3 * (l & 0x7FFFFFFFu) : nmax + 3 * l;
                const float *bmn = (r & 0x80000000u) ? lmin + 3 * (r & 0x7FFFFFFFu) : nmin + 3 * r, *bmx = (r & 0x80000000u) ? lmax + 3 * (r & 0x7FFFFFFFu) : nmax + 3 * r;
                for (int j = 0; j < 3; j++) { nmin[3 * i + j] = fminx(amn[j], bmn[j]); nmax[3 * i + j] = fmaxx(amx[j], bmx[j]); }
            }
        }
        if (stage == 0) { stage = 1; if (s->sah_top) { sah_top_rebuild(nn, left, right, rlo, rcnt, lmin, lmax, nmin, nmax, (s->tree_order || (s->node_width == 4 && s->collapse == 2)) ? s->sah_top : (s->sah_top > s->leaf_max ? s->sah_top : s->leaf_max)); continue; } }
        if (stage == 1) { stage = 2; if (s->top_opt && (s->tree_order || (s->node_width == 4 && s->collapse == 2))) { top_optimize(nn, n, left, right, rcnt, nmin, nmax, lmin, lmax, s->top_opt, s->top_opt_passes ? s->top_opt_passes : 2u); continue; } }
        break;
        }
        /* Multi-triangle leaves: an internal node covering <= leaf_max triangles (contiguous in Morton order) is referenced
         * as a leaf {bit 31, count-1 in bits 30..28, first triangle in bits 27..0}; the root always stays a node.
         * Wide nodes (node_width 4): the nodes at even binary depth survive and absorb their internal children, so each
         * holds 2..4 child slots {min, max, ref, pad} of 32 B = one 128 B cache line; empty slots carry ref 0xFFFFFFFF.
         * Surviving nodes are renumbered densely in index order (an exclusive scan on the GPU). */
        const uint32_t K = s->leaf_max, W4 = s->node_width == 4;
        const int DP = W4 && s->collapse == 2;
        /* [round 3] TREE ORDER: the triangle records are re-ordered into the depth-first order of the final binary tree, so that EVERY
         * subtree (also one the SAH top re-linked) covers a contiguous range and can be referenced as a multi-triangle leaf.
         * order[] = internal nodes in pre-order (parents first); rcnt = true triangle count, rlo = first triangle of the subtree. */
        uint32_t *order = (uint32_t *)malloc((size_t)nn * 4), n_order = 0;
        {
            int sp = 0; stack[sp++] = 0;
            while (sp > 0) {
                uint32_t i = stack[--sp]; order[n_order++] = i;
                if (!(right[i] & 0x80000000u)) stack[sp++] = right[i];
                if (!(left[i] & 0x80000000u)) stack[sp++] = left[i];
            }
        }
        if (s->tree_order || DP) {
            for (uint32_t q = n_order; q-- > 0;) {
                uint32_t i = order[q], l = left[i], r = right[i];
                rcnt[i] = ((l & 0x80000000u) ? 1u : rcnt[l]) + ((r & 0x80000000u) ? 1u : rcnt[r]);
            }
            uint32_t *newpos = (uint32_t *)malloc((size_t)n * 4);
            rlo[0] = 0;
            for (uint32_t q = 0; q < n_order; q++) {
                uint32_t i = order[q], l = left[i], r = right[i], f = rlo[i];
                if (l & 0x80000000u) { newpos[l & 0x7FFFFFFFu] = f; f += 1; } else { rlo[l] = f; f += rcnt[l]; }
                if (r & 0x80000000u) newpos[r & 0x7FFFFFFFu] = f; else rlo[r] = f;
            }
            float *t2 = (float *)malloc((size_t)n * 48), *mn2 = (float *)malloc((size_t)n * 12), *mx2 = (float *)malloc((size_t)n * 12);
            for (uint32_t q = 0; q < n; q++) {
                memcpy(t2 + 12 * (size_t)newpos[q], s->tris + 12 * (size_t)q, 48);
                memcpy(mn2 + 3 * (size_t)newpos[q], lmin + 3 * (size_t)q, 12); memcpy(mx2 + 3 * (size_t)newpos[q], lmax + 3 * (size_t)q, 12);
            }
            free(s->tris); s->tris = t2; free(lmin); free(lmax); lmin = mn2; lmax = mx2;
            for (uint32_t i = 0; i < nn; i++) {
                if (left[i] & 0x80000000u) left[i] = 0x80000000u | newpos[left[i] & 0x7FFFFFFFu];
                if (right[i] & 0x80000000u) right[i] = 0x80000000u | newpos[right[i] & 0x7FFFFFFFu];
            }
            free(newpos);
        }
        /* [round 3] COST-DRIVEN COLLAPSE (collapse 2; after Ylitie, Karras, Laine 2017, section 3.1): C(n, m) = the least SAH cost of
         * representing the subtree of binary node n by at most m slots of a parent (m = 1..3),
         *     C(n,1) = min( A_n * cnt_n * C_TRI  [a leaf, cnt_n <= leaf_max],  A_n * C_NODE + D(n,4)  [a four-wide node] )
         *     C(n,m) = min( D(n,m), C(n,m-1) ),   D(n,j) = min over 0 < k < j of C(left,k) + C(right,j-k),   C(triangle,.) = A * C_TRI
         * bottom-up in fp32, fixed order, strict '<' so that ties keep the earlier (fewer slots / smaller k) choice.  A node step
         * and a triangle step of the walk cost about the same (one fetch round trip and a similar number of vector instructions),
         * hence C_NODE = C_TRI = 1.  dk bits: 0-1 k of D(n,4); 2 k of D(n,3) minus 1; 3 C(n,1) is a leaf; 4 C(n,2) = D(n,2); 5 C(n,3) = D(n,3). */
        float *dc = NULL; uint8_t *dk = NULL;
        if (DP) {
            dc = (float *)malloc((size_t)nn * 12); dk = (uint8_t *)calloc(nn, 1);
            for (uint32_t q = n_order; q-- > 0;) {
                uint32_t i = order[q], l = left[i], r = right[i];
                float Cl[3], Cr[3];
                if (l & 0x80000000u) { float a = half_area3(lmin + 3 * (l & 0x7FFFFFFFu), lmax + 3 * (l & 0x7FFFFFFFu)); if (s->dp_c_tri != 1.0f) a = a * s->dp_c_tri; Cl[0] = Cl[1] = Cl[2] = a; }
                else { Cl[0] = dc[3 * l]; Cl[1] = dc[3 * l + 1]; Cl[2] = dc[3 * l + 2]; }
                if (r & 0x80000000u) { float a = half_area3(lmin + 3 * (r & 0x7FFFFFFFu), lmax + 3 * (r & 0x7FFFFFFFu)); if (s->dp_c_tri != 1.0f) a = a * s->dp_c_tri; Cr[0] = Cr[1] = Cr[2] = a; }
                else { Cr[0] = dc[3 * r]; Cr[1] = dc[3 * r + 1]; Cr[2] = dc[3 * r + 2]; }
                float d2 = Cl[0] + Cr[0];
                float d3 = Cl[0] + Cr[1]; uint32_t k3 = 1; { float b = Cl[1] + Cr[0]; if (b < d3) { d3 = b; k3 = 2; } }
                float d4 = Cl[0] + Cr[2]; uint32_t k4 = 1; { float b = Cl[1] + Cr[1]; if (b < d4) { d4 = b; k4 = 2; } b = Cl[2] + Cr[0]; if (b < d4) { d4 = b; k4 = 3; } }
                float A = half_area3(nmin + 3 * i, nmax + 3 * i);
                float cint = A + d4, cleaf = (float)rcnt[i] * A;
                if (s->dp_c_node != 1.0f || s->dp_c_tri != 1.0f) { cint = A * s->dp_c_node + d4; cleaf = ((float)rcnt[i] * A) * s->dp_c_tri; }
                uint32_t leaf1 = i != 0 && rcnt[i] <= K && cleaf <= cint;
                float c1 = leaf1 ? cleaf : cint;
                uint32_t s2 = d2 < c1; float c2 = s2 ? d2 : c1;
                uint32_t s3 = d3 < c2; float c3 = s3 ? d3 : c2;
                dc[3 * i] = c1; dc[3 * i + 1] = c2; dc[3 * i + 2] = c3;
                dk[i] = (uint8_t)(k4 | ((k3 - 1u) << 2) | (leaf1 << 3) | (s2 << 4) | (s3 << 5));
            }
        }
#define ORC_LIVE(i) ((i) == 0 || (DP ? !(dk[i] & 8u) : rcnt[i] > K))
        /* which binary nodes survive as four-wide nodes.  collapse 0: the nodes at even binary depth (each absorbs its live
         * children).  collapse 1 (default): top-down from the root, a node's two child slots are grown to (up to) four by
         * repeatedly replacing the live internal slot of LARGEST SURFACE AREA by its two children (ties: first slot); the
         * live internal slots that remain are the next surviving nodes.  wlevel = four-wide level of a surviving node. */
        uint8_t *keepf = (uint8_t *)calloc(nn, 1);
        uint32_t *wlevel = (uint32_t *)calloc(nn, 4);
        if (W4 && s->collapse) {
            uint32_t *queue = (uint32_t *)malloc((size_t)nn * 4), qh = 0, qt = 0;
            queue[qt++] = 0; keepf[0] = 1;
            while (qh < qt) {
                uint32_t i = queue[qh++], sl[4], m = DP ? dp_slots(i, left, right, dk, sl) : sah_slots(i, left, right, rcnt, nmin, nmax, K, sl);
                for (uint32_t k = 0; k < m; k++)
                    if (!(sl[k] & 0x80000000u) && ORC_LIVE(sl[k])) { keepf[sl[k]] = 1; wlevel[sl[k]] = wlevel[i] + 1; queue[qt++] = sl[k]; }
            }
            free(queue);
        } else {
            for (uint32_t i = 0; i < nn; i++) { keepf[i] = ORC_LIVE(i) && (!W4 || (depth[i] & 1u) == 0); wlevel[i] = W4 ? depth[i] / 2u : depth[i]; }
        }
#define ORC_KEEP(i) (keepf[i])
#define ORC_SLOTS(i, sl, m) do { if (DP) m = dp_slots(i, left, right, dk, sl); else if (W4 && s->collapse) m = sah_slots(i, left, right, rcnt, nmin, nmax, K, sl); else { \
            uint32_t c2_[2] = {left[i], right[i]}; m = 0; \
            for (int c_ = 0; c_ < 2; c_++) { uint32_t ch_ = c2_[c_]; \
                if (W4 && !(ch_ & 0x80000000u) && ORC_LIVE(ch_)) { sl[m++] = left[ch_]; sl[m++] = right[ch_]; } else sl[m++] = ch_; } } } while (0)
        uint32_t *newidx = (uint32_t *)malloc((size_t)nn * 4), kept = 0, maxd = 0;
        for (uint32_t i = 0; i < nn; i++) { newidx[i] = kept; if (ORC_KEEP(i)) kept++; }
        const uint32_t QN = s->node_quant;
        const uint32_t words = (W4 && !QN) ? 32u : (QN == 2 ? 12u : 16u);
        free(s->nodes);
        s->nodes = (float *)calloc((size_t)kept * words, 4);
        /* compact layout: node_base / tri_base of every surviving node = exclusive sums (in index order) of its number of
         * internal child slots / of the triangles in its leaf slots; a child's own index is 1 + node_base(parent) + rank */
        uint32_t *cbase = NULL, *tbase = NULL; float *tris2 = NULL;
        if (QN == 2) {
            cbase = (uint32_t *)malloc((size_t)nn * 4); tbase = (uint32_t *)malloc((size_t)nn * 4);
            tris2 = (float *)calloc((size_t)n * 12, 4);
            uint32_t ci = 0, ti = 0;
            for (uint32_t i = 0; i < nn; i++) {
                cbase[i] = ci; tbase[i] = ti;
                if (!ORC_KEEP(i)) continue;
                uint32_t sl[4], m;
                ORC_SLOTS(i, sl, m);
                for (uint32_t k = 0; k < m; k++) {
                    if (sl[k] & 0x80000000u) ti += 1;
                    else if (ORC_LIVE(sl[k])) ci += 1;
                    else ti += rcnt[sl[k]];
                }
            }
            newidx[0] = 0;
            for (uint32_t i = 0; i < nn; i++) {
                if (!ORC_KEEP(i)) continue;
                uint32_t sl[4], m, rank = 0;
                ORC_SLOTS(i, sl, m);
                for (uint32_t k = 0; k < m; k++)
                    if (!(sl[k] & 0x80000000u) && ORC_LIVE(sl[k])) newidx[sl[k]] = 1u + cbase[i] + rank++;
            }
        }
        for (uint32_t i = 0; i < nn; i++) {
            if (!ORC_KEEP(i)) continue;
            uint32_t slots[4], ns;
            ORC_SLOTS(i, slots, ns);
            float *nd = s->nodes + (size_t)words * newidx[i];
            float qmn[4][3], qmx[4][3]; uint32_t qref[4], qmeta[4] = {7u, 7u, 7u, 7u}, tcur = QN == 2 ? tbase[i] : 0u;
            for (uint32_t k = 0; k < (W4 ? 4u : 2u); k++) {
                float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
                uint32_t ref = 0xFFFFFFFFu;
                if (k < ns) {
                    uint32_t ch = slots[k];
                    if (ch & 0x80000000u) { uint32_t q = ch & 0x7FFFFFFFu; ref = 0x80000000u | q; memcpy(mn, lmin + 3 * q, 12); memcpy(mx, lmax + 3 * q, 12); }
                    else {
                        memcpy(mn, nmin + 3 * ch, 12); memcpy(mx, nmax + 3 * ch, 12);
                        ref = ORC_LIVE(ch) ? newidx[ch] : (0x80000000u | ((rcnt[ch] - 1u) << 28) | rlo[ch]);
                    }
                }
                if (QN == 2 && k < ns) { /* meta nibble + move the leaf's triangles to their place behind tri_base */
                    if (ref & 0x80000000u) {
                        uint32_t first = ref & 0x0FFFFFFFu, cnt = ((ref >> 28) & 7u) + 1u;
                        qmeta[k] = 8u | (cnt - 1u);
                        memcpy(tris2 + 12 * (size_t)tcur, s->tris + 12 * (size_t)first, (size_t)cnt * 48);
                        tcur += cnt;
                    } else qmeta[k] = 0u;
                }
                if (QN) { memcpy(qmn[k], mn, 12); memcpy(qmx[k], mx, 12); qref[k] = ref; }
                else if (W4) { memcpy(nd + 8 * k, mn, 12); memcpy(nd + 8 * k + 3, mx, 12); nd[8 * k + 6] = u2f(ref); nd[8 * k + 7] = 0.0f; }
                else { memcpy(nd + 6 * k, mn, 12); memcpy(nd + 6 * k + 3, mx, 12); nd[12 + k] = u2f(ref); }
            }
            if (QN == 2) compact_node(qmn, qmx, qmeta, ns, 1u + cbase[i], tbase[i], (uint32_t *)nd);
            else if (QN) quantize_node(qmn, qmx, qref, ns, 1, (uint32_t *)nd);
            uint32_t lvl = wlevel[i] + 2u; /* levels from the root to this node's leaf slots */
            if (lvl > maxd) maxd = lvl;
        }
        s->max_depth = maxd;
        s->n_nodes = kept;
        if (QN == 2) { free(s->tris); s->tris = tris2; free(cbase); free(tbase); }
#undef ORC_LIVE
#undef ORC_KEEP
#undef ORC_SLOTS
        free(keepf); free(wlevel); free(order); free(dc); free(dk);
        free(left); free(right); free(rlo); free(rcnt); free(nmin); free(nmax); free(stack); free(state); free(depth); free(newidx);
    }
    free(bmin); free(bmax); free(cp); free(lmin); free(lmax);
    return 0;
}
uint32_t orc_accel_num_tris(const orc_scene *s) { return s->n_tris; }
uint32_t orc_accel_num_nodes(const orc_scene *s) { return s->n_nodes; }
const float *orc_accel_nodes(const orc_scene *s) { return s->nodes; }
const float *orc_accel_tris(const orc_scene *s) { return s->tris; }
const uint64_t *orc_accel_codes(const orc_scene *s) { return s->codes; }
uint32_t orc_accel_max_depth(const orc_scene *s) { return s->max_depth; }

/* ------------------------------------------------------------------------------------------------ traversal */
typedef struct { float t, u, v; uint32_t prim; } hit_t;

/* [north_star] Ray / triangle test: signed-volume edge functions (Pluecker form), watertight by construction.  Both faces hit
 * (no cull flags: pipeline_cache/mod.rs:326-333 registers a plain closest-hit group).
 *
 * The driver traversal the reference relies on is watertight by specification (VK_KHR_acceleration_structure).  Moeller-Trumbore
 * in fp32 is not: a ray through a shared edge can round to u < 0 in one triangle and u + v > 1 in its neighbour (146 of 2160 rays
 * aimed at the edge midpoints / vertices of the Cornell box slipped through; round 1 papered over it with a 2^-20 tolerance).
 * Here, with A, B, C the vertices relative to the ray origin (each computed from its own vertex only),
 *     U = d . (B x C)    V = d . (C x A)    W = d . (A x B)
 * and every cross-product component is (x*y) - (z*w) with two rounded products and one subtraction (NO fused multiply-add):
 * swapping the two vertices of an edge negates the result EXACTLY, and the dot product with d negates exactly with it.  The two
 * triangles on either side of an edge therefore evaluate the same number for it, up to the sign their winding gives it -- whatever the
 * rounding, a ray is on one side of the edge for both of them (on both sides when the value is exactly 0: both accept, the closest-hit
 * rule below picks one) and cannot pass between them.  The ray is inside when U, V, W share a sign (zeros allowed), i.e. when the
 * barycentrics w = U / det, u = V / det (vertex 1), v = W / det (vertex 2), det = U + (V + W), are all >= 0; t = sum of the vertices'
 * distances along the ray weighted by the barycentrics, divided by d.d (no unit-length assumption on d).
 * The test actually applied is w, u, v >= -2^-20: a SUPERSET of the sign test (so the shared-edge guarantee stands) that also closes
 * what no watertight algorithm covers -- T-junctions and edges of separate meshes that merely coincide (the walls of the Cornell box
 * meet the floor with different tessellations: 42 of 2160 rays aimed at those lines slipped through the exact sign test).
 * Acceptance: t > tmin and (t < best.t or (t == best.t and prim < best.prim)), which makes the closest hit independent of traversal
 * order.  Dot products use fused multiply-adds in a fixed order (the GPU's v_fma_f32; here __builtin_fmaf, one vfmadd in the "fma"
 * clones of the callers): a.b = fma(a0, b0, fma(a1, b1, a2*b2)). */
static inline float dot3f(const float a[3], const float b[3]) { return __builtin_fmaf(a[0], b[0], __builtin_fmaf(a[1], b[1], a[2] * b[2])); }
static inline void cross3x(const float a[3], const float b[3], float o[3]) { /* exactly antisymmetric: cross3x(b, a) == -cross3x(a, b) bit for bit */
    o[0] = (a[1] * b[2]) - (a[2] * b[1]);
    o[1] = (a[2] * b[0]) - (a[0] * b[2]);
    o[2] = (a[0] * b[1]) - (a[1] * b[0]);
}
#define ORC_EDGE_EPS 9.5367431640625e-07f
static inline void tri_test_dd(const float *tr, const float o[3], const float d[3], float inv_dd, float tmin, hit_t *best) {
    float A[3], B[3], C[3], bc[3], ca[3], ab[3];
    for (int k = 0; k < 3; k++) { A[k] = tr[k] - o[k]; B[k] = tr[3 + k] - o[k]; C[k] = tr[6 + k] - o[k]; }
    cross3x(B, C, bc); cross3x(C, A, ca); cross3x(A, B, ab);
    const float U = dot3f(d, bc), V = dot3f(d, ca), W = dot3f(d, ab);
    const float det = U + (V + W); /* the association of T below: equal vertex distances give t exactly */
    if (det == 0.0f) return;
    const float inv = 1.0f / det;
    const float w = U * inv, u = V * inv, v = W * inv;
    if (!(w >= -ORC_EDGE_EPS && u >= -ORC_EDGE_EPS && v >= -ORC_EDGE_EPS)) return;
    const float T = __builtin_fmaf(U, dot3f(A, d), __builtin_fmaf(V, dot3f(B, d), W * dot3f(C, d)));
    const float t = (T * inv) * inv_dd;
    uint32_t prim = f2u(tr[9]);
    if (t > tmin && (t < best->t || (t == best->t && prim < best->prim))) { best->t = t; best->u = u; best->v = v; best->prim = prim; }
}
static inline float ray_inv_dd(const float d[3]) { return 1.0f / dot3f(d, d); }
/* the three edge functions of tri_test_dd for a triangle given by its vertices (tests: exact antisymmetry of a shared edge) */
void orc_tri_edge_functions(const float *v0, const float *v1, const float *v2, const float *o, const float *d, float out[3]) {
    float A[3], B[3], C[3], bc[3], ca[3], ab[3];
    for (int k = 0; k < 3; k++) { A[k] = v0[k] - o[k]; B[k] = v1[k] - o[k]; C[k] = v2[k] - o[k]; }
    cross3x(B, C, bc); cross3x(C, A, ca); cross3x(A, B, ab);
    out[0] = dot3f(d, bc); out[1] = dot3f(d, ca); out[2] = dot3f(d, ab);
}
static inline void tri_test(const float *tr, const float o[3], const float d[3], float tmin, hit_t *best) { tri_test_dd(tr, o, d, ray_inv_dd(d), tmin, best); }
static inline float guard_inv(float d) {
    float a = d < 0.0f ? -d : d;
    float g = a < 1e-20f ? (d < 0.0f ? -1e-20f : 1e-20f) : d;
    return 1.0f / g;
}
static inline int slab(const float *bx, const float o[3], const float inv[3], float tmin, float tbest, float *tn_out) {
    float tn = tmin, tf = tbest;
    for (int k = 0; k < 3; k++) {
        float t0 = (bx[k] - o[k]) * inv[k], t1 = (bx[3 + k] - o[k]) * inv[k];
        float lo = t0 < t1 ? t0 : t1, hi = t0 < t1 ? t1 : t0;
        tn = lo > tn ? lo : tn;
        tf = hi < tf ? hi : tf;
    }
    *tn_out = tn;
    return tn <= tf;
}
#define ORC_STACK 256
static uint32_t *g_visit_hist; /* debug: per-node visit counters (tests/experiments/tree_quality.py) */
void orc_debug_set_visit_hist(uint32_t *h) { g_visit_hist = h; }
#define ORC_EMPTY 0xFFFFFFFFu
/* Closest / any hit.  Children are visited nearest first (four-wide any-hit: FARTHEST first; ties as the sorting network leaves them), the
 * others are pushed so that they pop in that order; no re-cull on pop.  A leaf reference holds 1..8 triangles, tested in order. */
/* Quantised nodes: the dequantisation is folded into the ray.  A child plane sits at org + q * step (q = 0..255), so its
 * ray parameter is ((org + q*step) - o) * inv = q * (step*inv) + (org - o)*inv: per node three products A = step*inv and three
 * B = (org - o)*inv, per plane ONE fused multiply-add fmaf(q, A, B) (single rounding, the GPU's v_fma_f32).  The rounding
 * differs from slab() by an amount that corresponds to ~1 ulp of |org - o| in space -- far inside the leaf padding. */
static inline int slab_q(const uint8_t *qb, const float A[3], const float B[3], float tmin, float tbest, float *tn_out) {
    float tn = tmin, tf = tbest;
    for (int k = 0; k < 3; k++) {
        float t0 = __builtin_fmaf((float)qb[k], A[k], B[k]), t1 = __builtin_fmaf((float)qb[3 + k], A[k], B[k]);
        float lo = t0 < t1 ? t0 : t1, hi = t0 < t1 ? t1 : t0;
        tn = lo > tn ? lo : tn;
        tf = hi < tf ? hi : tf;
    }
    *tn_out = tn;
    return tn <= tf;
}
/* target_clones: the "fma" clone inlines fmaf as one vfmadd instruction, the default clone calls libm (same result) */
__attribute__((target_clones("fma", "default")))
static void traverse(const orc_scene *s, const float o[3], const float d[3], float tmin, float tmax, int any,
                     hit_t *out, uint32_t *cn, uint32_t *ct) {
    hit_t best = {tmax, 0.0f, 0.0f, ORC_MISS};
    uint32_t nn = 0, nt = 0;
    int finite_ray = 1; /* [rule] a ray with a non-finite origin or direction misses (Vulkan leaves it undefined); same in the product */
    for (int k = 0; k < 3; k++) finite_ray &= (fabsf(o[k]) <= 3.4028234663852886e38f) & (fabsf(d[k]) <= 3.4028234663852886e38f);
    if (s->n_tris && finite_ray) {
        float inv[3] = {guard_inv(d[0]), guard_inv(d[1]), guard_inv(d[2])};
        const float inv_dd = ray_inv_dd(d);
        uint32_t stack[ORC_STACK]; int sp = 0;
        float sdist[ORC_STACK]; float dsel[4] = {0, 0, 0, 0}; /* experiment (recull): truncated entry distances of the stacked references */
        /* [round 3 experiment, off by default] re-cull on pop: closest hit, default node layout, trees of at most 2^18 nodes (a kernel would pack
         * the distance into 12 spare bits of a stacked node reference): a popped NODE reference whose entry distance -- truncated to the float's top 12 bits,
         * i.e. never above the distance itself -- lies beyond the hit found meanwhile is dropped without a visit; at most ONE reference per pop (the
         * one below it is then taken as it is), so that the kernel needs no loop */
        const int recull = s->recull && !any && s->node_width == 4 && s->node_quant == 1 && s->n_nodes <= (1u << 18);
        uint32_t cur = 0;
        const int W4 = s->node_width == 4;
        for (;;) {
            if (cur & 0x80000000u) {
                uint32_t first = cur & 0x0FFFFFFFu, cnt = ((cur >> 28) & 7u) + 1u, k;
                for (k = 0; k < cnt; k++) {
                    nt++;
                    tri_test_dd(s->tris + 12 * (size_t)(first + k), o, d, inv_dd, tmin, &best);
                    if (any && best.prim != ORC_MISS) break;
                }
                if (any && best.prim != ORC_MISS) break;
                if (recull && sp > 0 && !(stack[sp - 1] & 0x80000000u) && sdist[sp - 1] > best.t) sp--; /* ONE reference per pop (the kernel's rule) */
                if (sp == 0) break;
                cur = stack[--sp];
                continue;
            }
            nn++;
            if (g_visit_hist) __atomic_fetch_add(&g_visit_hist[cur], 1u, __ATOMIC_RELAXED);
            uint32_t ref[4]; int nh = 0;
            if (W4) {
                const int QN = (int)s->node_quant;
                const float *nd = s->nodes + (QN == 2 ? 12 : (QN ? 16 : 32)) * (size_t)cur;
                float ct[4]; uint32_t cr[4], r48[4];
                float A[3] = {0, 0, 0}, B[3] = {0, 0, 0};
                if (QN == 2) compact_refs((const uint32_t *)nd, r48);
                if (QN) for (int a = 0; a < 3; a++) {
                    const uint32_t *w = (const uint32_t *)nd;
                    float step = QN == 2 ? u2f(((w[3] >> (8 * a)) & 0xFFu) << 23) : u2f(a == 0 ? w[3] : w[13 + a]); /* exponent bytes (48 B) / floats (64 B) */
                    A[a] = step * inv[a];
                    B[a] = (nd[a] - o[a]) * inv[a];
                }
                for (int k = 0; k < 4; k++) {
                    uint32_t r = QN == 2 ? r48[k] : (QN ? ((const uint32_t *)nd)[10 + k] : f2u(nd[8 * k + 6])); float t;
                    int h = r != ORC_EMPTY && (QN ? slab_q((const uint8_t *)nd + 16 + 6 * k, A, B, tmin, best.t, &t) : slab(nd + 8 * k, o, inv, tmin, best.t, &t));
                    ct[k] = h ? t : INFINITY; cr[k] = h ? r : ORC_EMPTY;
                    nh += h;
                }
                /* the kernel's 5-comparator network on a key (strict <).  Closest hit: key = entry distance -> nearest child
                 * first.  Any hit: key = -entry distance -> FARTHEST child first: occlusion does not depend on the order, but
                 * a shadow ray starts on a surface, whose neighbourhood it grazes without hitting, and is usually blocked far
                 * away (ceiling, opposite wall): far-first finds that occluder in ~35 % fewer node visits than slot order and
                 * ~40 % fewer than near-first on the atrium.  Non-entered slots keep key +inf and sink to the end. */
                {
                    static const int net[5][2] = {{0, 1}, {2, 3}, {0, 2}, {1, 3}, {1, 2}};
                    if (any) for (int k = 0; k < 4; k++) if (cr[k] != ORC_EMPTY) ct[k] = -ct[k];
                    for (int c = 0; c < 5; c++) {
                        int a = net[c][0], b = net[c][1];
                        if (ct[b] < ct[a]) { float tt = ct[a]; ct[a] = ct[b]; ct[b] = tt; uint32_t rr = cr[a]; cr[a] = cr[b]; cr[b] = rr; }
                    }
                    for (int k = 0; k < nh; k++) { ref[k] = cr[k]; dsel[k] = u2f(f2u(ct[k]) & 0xFFF00000u); }
                }
            } else {
                const float *nd = s->nodes + 16 * (size_t)cur;
                float t0, t1;
                uint32_t r0 = f2u(nd[12]), r1 = f2u(nd[13]);
                int h0 = r0 != ORC_EMPTY && slab(nd, o, inv, tmin, best.t, &t0), h1 = r1 != ORC_EMPTY && slab(nd + 6, o, inv, tmin, best.t, &t1);
                if (h0 && h1) { int near1 = t1 < t0; ref[0] = near1 ? r1 : r0; ref[1] = near1 ? r0 : r1; nh = 2; }
                else if (h0) { ref[0] = r0; nh = 1; }
                else if (h1) { ref[0] = r1; nh = 1; }
            }
            if (nh == 0) {
                if (recull && sp > 0 && !(stack[sp - 1] & 0x80000000u) && sdist[sp - 1] > best.t) sp--;
                if (sp == 0) break;
                cur = stack[--sp]; continue;
            }
            for (int k = nh - 1; k >= 1; k--) { sdist[sp] = dsel[k]; stack[sp++] = ref[k]; }
            cur = ref[0];
        }
    }
    *out = best;
    if (cn) *cn = nn;
    if (ct) *ct = nt;
}

typedef struct {
    const orc_scene *s; const float *rays; uint32_t n; float *t, *u, *v; uint32_t *prim, *nn, *nt, *occ; int mode;
} trace_job;
static void trace_closest_body(void *c, uint32_t b, uint32_t e, int tid) {
    (void)tid;
    trace_job *j = (trace_job *)c; uint32_t n = j->n; const float *r = j->rays;
    for (uint32_t i = b; i < e; i++) {
        float o[3] = {r[i], r[n + i], r[2 * (size_t)n + i]}, d[3] = {r[3 * (size_t)n + i], r[4 * (size_t)n + i], r[5 * (size_t)n + i]};
        hit_t h; uint32_t cn, ct;
        traverse(j->s, o, d, r[6 * (size_t)n + i], r[7 * (size_t)n + i], 0, &h, &cn, &ct);
        j->t[i] = h.t; j->u[i] = h.u; j->v[i] = h.v; j->prim[i] = h.prim;
        if (j->nn) j->nn[i] = cn;
        if (j->nt) j->nt[i] = ct;
    }
}
void orc_trace_closest(const orc_scene *s, const float *rays, uint32_t n, float *t, float *u, float *v, uint32_t *prim,
                       uint32_t *n_nodes, uint32_t *n_tris, int n_threads) {
    trace_job j = {s, rays, n, t, u, v, prim, n_nodes, n_tris, NULL, 0};
    parallel_for(n, n_threads, trace_closest_body, &j);
}
static void trace_any_body(void *c, uint32_t b, uint32_t e, int tid) {
    (void)tid;
    trace_job *j = (trace_job *)c; uint32_t n = j->n; const float *r = j->rays;
    for (uint32_t i = b; i < e; i++) {
        float o[3] = {r[i], r[n + i], r[2 * (size_t)n + i]}, d[3] = {r[3 * (size_t)n + i], r[4 * (size_t)n + i], r[5 * (size_t)n + i]};
        hit_t h; uint32_t cn, ct;
        traverse(j->s, o, d, r[6 * (size_t)n + i], r[7 * (size_t)n + i], 1, &h, &cn, &ct);
        j->occ[i] = h.prim != ORC_MISS;
        if (j->nn) j->nn[i] = cn;
        if (j->nt) j->nt[i] = ct;
    }
}
void orc_trace_any(const orc_scene *s, const float *rays, uint32_t n, uint32_t *occluded, uint32_t *n_nodes,
                   uint32_t *n_tris, int n_threads) {
    trace_job j = {s, rays, n, NULL, NULL, NULL, NULL, n_nodes, n_tris, occluded, 0};
    parallel_for(n, n_threads, trace_any_body, &j);
}
/* double-precision Moeller-Trumbore, used only to pin the fp32 test */
static void tri_test_f64(const float *tr, const float of[3], const float df[3], double tmin, double *bt, double *bu, double *bv, uint32_t *bp) {
    double v0[3], e1[3], e2[3], o[3], d[3], pv[3], tv[3], qv[3];
    for (int k = 0; k < 3; k++) { v0[k] = tr[k]; e1[k] = (double)tr[3 + k] - tr[k]; e2[k] = (double)tr[6 + k] - tr[k]; o[k] = of[k]; d[k] = df[k]; }
    pv[0] = d[1] * e2[2] - d[2] * e2[1]; pv[1] = d[2] * e2[0] - d[0] * e2[2]; pv[2] = d[0] * e2[1] - d[1] * e2[0];
    double det = e1[0] * pv[0] + e1[1] * pv[1] + e1[2] * pv[2];
    if (det == 0.0) return;
    double inv = 1.0 / det;
    for (int k = 0; k < 3; k++) tv[k] = o[k] - v0[k];
    double u = (tv[0] * pv[0] + tv[1] * pv[1] + tv[2] * pv[2]) * inv;
    if (!(u >= 0.0 && u <= 1.0)) return;
    qv[0] = tv[1] * e1[2] - tv[2] * e1[1]; qv[1] = tv[2] * e1[0] - tv[0] * e1[2]; qv[2] = tv[0] * e1[1] - tv[1] * e1[0];
    double v = (d[0] * qv[0] + d[1] * qv[1] + d[2] * qv[2]) * inv;
    if (!(v >= 0.0 && u + v <= 1.0)) return;
    double t = (e2[0] * qv[0] + e2[1] * qv[1] + e2[2] * qv[2]) * inv;
    uint32_t prim = f2u(tr[9]);
    if (t > tmin && (t < *bt || (t == *bt && prim < *bp))) { *bt = t; *bu = u; *bv = v; *bp = prim; }
}
__attribute__((target_clones("fma", "default")))
static void trace_brute_body(void *c, uint32_t b, uint32_t e, int tid) {
    (void)tid;
    trace_job *j = (trace_job *)c; uint32_t n = j->n; const float *r = j->rays; const orc_scene *s = j->s;
    for (uint32_t i = b; i < e; i++) {
        float o[3] = {r[i], r[n + i], r[2 * (size_t)n + i]}, d[3] = {r[3 * (size_t)n + i], r[4 * (size_t)n + i], r[5 * (size_t)n + i]};
        float tmin = r[6 * (size_t)n + i], tmax = r[7 * (size_t)n + i];
        if (j->mode == 0) {
            hit_t best = {tmax, 0.0f, 0.0f, ORC_MISS};
            for (uint32_t k = 0; k < s->n_tris; k++) tri_test(s->tris + 12 * (size_t)k, o, d, tmin, &best);
            j->t[i] = best.t; j->u[i] = best.u; j->v[i] = best.v; j->prim[i] = best.prim;
        } else {
            double bt = tmax, bu = 0, bv = 0; uint32_t bp = ORC_MISS;
            for (uint32_t k = 0; k < s->n_tris; k++) tri_test_f64(s->tris + 12 * (size_t)k, o, d, tmin, &bt, &bu, &bv, &bp);
            j->t[i] = (float)bt; j->u[i] = (float)bu; j->v[i] = (float)bv; j->prim[i] = bp;
        }
    }
}
void orc_trace_brute(const orc_scene *s, const float *rays, uint32_t n, float *t, float *u, float *v, uint32_t *prim,
                     int mode, int n_threads) {
    trace_job j = {s, rays, n, t, u, v, prim, NULL, NULL, NULL, mode};
    parallel_for(n, n_threads, trace_brute_body, &j);
}

/* ------------------------------------------------------------------------------------------------ hit_info */
/* hit_logic.slang:5-40 with GeometryInfo.transform := identity, Vertex.color := 1;
 * surf = albedo[3] emissive[3] normal[3] roughness metalness */
/* [round 3] The vertex normals of the shading records are kept in the reference's octahedral map (packing.slang:64-86), 16 bits per
 * coordinate: a vertex normal IS octa_decode16(octa_encode16(n)) -- the scene representation on both sides of the ABI (the product's
 * 16-byte shading record {n0, n1, n2, geometry}).  A zero (or non-finite) normal encodes +z. */
void orc_octa_decode(float fx, float fy, float n[3]);
uint32_t orc_octa_encode16(const float nin[3]) {
    const float s = fabsf(nin[0]) + fabsf(nin[1]) + fabsf(nin[2]);
    if (!(s > 0.0f) || !(s <= 3.4028234663852886e38f)) return 0x80008000u;
    float x = nin[0] / s, y = nin[1] / s;
    const float z = nin[2] / s;
    if (z < 0.0f) { /* octa_wrap, :64-66 */
        const float wx = (1.0f - fabsf(y)) * ((x >= 0.0f ? 1.0f : 0.0f) * 2.0f - 1.0f);
        const float wy = (1.0f - fabsf(x)) * ((y >= 0.0f ? 1.0f : 0.0f) * 2.0f - 1.0f);
        x = wx; y = wy;
    }
    x = x * 0.5f + 0.5f; y = y * 0.5f + 0.5f;
    const uint32_t qx = (uint32_t)(fminx(fmaxx(x, 0.0f), 1.0f) * 65535.0f + 0.5f), qy = (uint32_t)(fminx(fmaxx(y, 0.0f), 1.0f) * 65535.0f + 0.5f);
    return qx | (qy << 16);
}
void orc_octa_decode16(uint32_t w, float n[3]) { orc_octa_decode((float)(w & 0xFFFFu) * (1.0f / 65535.0f), (float)(w >> 16) * (1.0f / 65535.0f), n); }
void orc_hit_info(const orc_scene *s, uint32_t prim, float bu, float bv, float surf[11]) {
    uint32_t fg = s->prim_geom[prim], local = prim - s->first_prim[fg];
    const orc_geometry_info *gi = &s->geoms[s->flat_geom[fg]];
    uint32_t io = gi->index_offset + 3u * local;
    const float *v0 = s->verts + 8 * (size_t)(gi->vertex_offset + s->indices[io]);
    const float *v1 = s->verts + 8 * (size_t)(gi->vertex_offset + s->indices[io + 1]);
    const float *v2 = s->verts + 8 * (size_t)(gi->vertex_offset + s->indices[io + 2]);
    float b0 = 1.0f - bu - bv, b1 = bu, b2 = bv;
    float n[3], n0[3], n1[3], n2[3];
    orc_octa_decode16(orc_octa_encode16(v0 + 3), n0); orc_octa_decode16(orc_octa_encode16(v1 + 3), n1); orc_octa_decode16(orc_octa_encode16(v2 + 3), n2);
    for (int k = 0; k < 3; k++) n[k] = n0[k] * b0 + n1[k] * b1 + n2[k] * b2;
    normalize3(n); /* :22 */
    if (!s->flat_identity[fg]) { /* :23 mul(geometryInfo.transform, float4(normal, 0.0)).xyz */
        const float *m = s->inst[s->flat_inst[fg]].transform;
        float x = n[0], y = n[1], z = n[2];
        for (int r = 0; r < 3; r++) n[r] = m[8 + r] * z + (m[4 + r] * y + m[r] * x);
    }
    normalize3(n); /* :23 */
    surf[0] = gi->base_color[0]; surf[1] = gi->base_color[1]; surf[2] = gi->base_color[2];
    if (gi->base_color_texture_index > -1 && (uint32_t)gi->base_color_texture_index < s->n_tex) { /* :27,31-33 */
        float uu = v0[6] * b0 + v1[6] * b1 + v2[6] * b2, vv = v0[7] * b0 + v1[7] * b1 + v2[7] * b2, tc[3];
        texture_sample(s, (uint32_t)gi->base_color_texture_index, uu, vv, tc);
        surf[0] = surf[0] * tc[0]; surf[1] = surf[1] * tc[1]; surf[2] = surf[2] * tc[2];
    }
    surf[3] = gi->emission[0] * 12.0f; surf[4] = gi->emission[1] * 12.0f; surf[5] = gi->emission[2] * 12.0f; /* :36 */
    surf[6] = n[0]; surf[7] = n[1]; surf[8] = n[2];
    surf[9] = gi->roughness; surf[10] = gi->metallic_factor;
}

/* ------------------------------------------------------------------------------------------------ sky [north_star] */
/* bilinear equirect lookup (Skybox.SampleLevel(uv, 0), postprocess.slang:102-104): wrap in u, clamp in v */
static void sky_eval(const orc_scene *s, float u, float v, float out[3]) {
    if (!s->sky) { out[0] = out[1] = out[2] = 0.0f; return; }
    int W = (int)s->sky_w, H = (int)s->sky_h;
    float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
    float xf = floorf(x), yf = floorf(y);
    float fx = x - xf, fy = y - yf;
    int x0 = (int)xf, y0 = (int)yf;
    int x1 = x0 + 1, y1 = y0 + 1;
    x0 = ((x0 % W) + W) % W; x1 = ((x1 % W) + W) % W;
    y0 = y0 < 0 ? 0 : (y0 > H - 1 ? H - 1 : y0);
    y1 = y1 < 0 ? 0 : (y1 > H - 1 ? H - 1 : y1);
    const float *p00 = s->sky + 3 * ((size_t)y0 * W + x0), *p10 = s->sky + 3 * ((size_t)y0 * W + x1);
    const float *p01 = s->sky + 3 * ((size_t)y1 * W + x0), *p11 = s->sky + 3 * ((size_t)y1 * W + x1);
    for (int k = 0; k < 3; k++) {
        float top = p00[k] * (1.0f - fx) + p10[k] * fx, bot = p01[k] * (1.0f - fx) + p11[k] * fx;
        out[k] = top * (1.0f - fy) + bot * fy;
    }
}
/* solid-angle pdf of the sky sampler for equirect coordinates (u,v) */
static float sky_pdf(const orc_scene *s, float u, float v) {
    if (!s->sky) return 0.0f;
    int W = (int)s->sky_w, H = (int)s->sky_h;
    int ix = (int)(u * (float)W), iy = (int)(v * (float)H);
    ix = ix < 0 ? 0 : (ix > W - 1 ? W - 1 : ix);
    iy = iy < 0 ? 0 : (iy > H - 1 ? H - 1 : iy);
    float st, ct;
    orc_sincos_2pi(v * 0.5f, &st, &ct);
    if (!(st > 0.0f)) return 0.0f;
    return s->pdf_uv[(size_t)iy * W + ix] / (2.0f * F_PI * F_PI * st);
}
static uint32_t cdf_find(const float *cdf, uint32_t n, float u) { /* first index with cdf[i] > u */
    uint32_t lo = 0, hi = n - 1;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (cdf[mid] > u) hi = mid; else lo = mid + 1;
    }
    return lo;
}
static void sky_sample(const orc_scene *s, float u0, float u1, float dir[3], float rad[3], float *pdf) {
    uint32_t W = s->sky_w, H = s->sky_h;
    uint32_t y = cdf_find(s->cdf_marg, H, u0);
    float lo = y > 0 ? s->cdf_marg[y - 1] : 0.0f, hi = s->cdf_marg[y];
    float dv = hi > lo ? (u0 - lo) / (hi - lo) : 0.5f;
    /* the row's alias table: cell k = floor(u1 w), xi = frac(u1 w) decides between column k and its alias and is then
     * stretched back to [0, 1) as the position inside the chosen texel */
    float sx = u1 * (float)W;
    uint32_t k = (uint32_t)sx;
    k = k > W - 1 ? W - 1 : k;
    float xi = sx - (float)k;
    xi = xi < 0.99999994f ? xi : 0.99999994f;
    const uint32_t e = s->sky_alias[(size_t)y * W + k];
    const float Q = (float)((e & 0xFFFFu) + 1u) * (1.0f / 65536.0f);
    const int keep = xi < Q;
    uint32_t x = keep ? k : (e >> 16);
    float du = keep ? xi / Q : (xi - Q) / (1.0f - Q);
    du = du < 0.99999994f ? du : 0.99999994f;
    float u = ((float)x + du) / (float)W, v = ((float)y + dv) / (float)H;
    float st, ct, s2, c2;
    orc_sincos_2pi(v * 0.5f, &st, &ct);
    orc_sincos_2pi(u, &s2, &c2);
    dir[0] = (-c2) * st; dir[1] = ct; dir[2] = (-s2) * st;
    sky_eval(s, u, v, rad);
    *pdf = st > 0.0f ? s->pdf_uv[(size_t)y * W + x] / (2.0f * F_PI * F_PI * st) : 0.0f;
}
/* [north_star] Cranley-Patterson shift by a blue-noise channel: frac(u + c/256) */
static inline float bn_shift(float u, uint32_t c) {
    float r = u + (float)c * 0.00390625f;
    return r >= 1.0f ? r - 1.0f : r;
}

/* ------------------------------------------------------------------------------------------------ layered BSDF */
/* [north_star "BSDF eval"] DiffuseBrdf (brdf.slang:52-93) under a GGX SpecularBrdf (brdf.slang:141-311: VNDF sampling,
 * height-correlated Smith, Schlick with f90 = 1), combined like the reference's source renderer: f0 = lerp(0.04, albedo,
 * metalness), diffuse albedo = albedo (1 - metalness), diffuse attenuated by the transmitted fraction (1 - F).  One
 * lobe is chosen with probability p_spec (luminance ratio, clamped to [0.1, 0.9]) and the sample is weighted by
 * f / pdf of the mixture (one-sample MIS).  All pdfs are with respect to PROJECTED solid angle like brdf.slang:31;
 * directions are in the tangent frame (z = shading normal). */
typedef struct { float da[3], f0[3], alpha, p_spec; } bsdf_t;
static inline float pow5(float x) { float x2 = x * x; return x2 * x2 * x; }
static float g_smith_ggx1(float ndotv, float a2) { /* brdf.slang:111-114 */
    float tan2_v = (1.0f - ndotv * ndotv) / (ndotv * ndotv);
    return 2.0f / (1.0f + sqrtf(1.0f + a2 * tan2_v));
}
static float g_smith_ggx_correlated(float ndotv, float ndotl, float a2) { /* brdf.slang:104-109 */
    float lambda_v = ndotl * sqrtf((-ndotv * a2 + ndotv) * ndotv + a2);
    float lambda_l = ndotv * sqrtf((-ndotl * a2 + ndotl) * ndotl + a2);
    return 2.0f * ndotl * ndotv / (lambda_v + lambda_l);
}
static float ggx_ndf(float a2, float cos_theta) { /* brdf.slang:146-149 */
    float denom_sqrt = cos_theta * cos_theta * (a2 - 1.0f) + 1.0f;
    return a2 / (F_PI * denom_sqrt * denom_sqrt);
}
static void bsdf_setup(const float surf[11], bsdf_t *b) {
    float m = surf[10];
    for (int k = 0; k < 3; k++) { b->f0[k] = 0.04f + (surf[k] - 0.04f) * m; b->da[k] = surf[k] * (1.0f - m); }
    b->alpha = fmaxx(surf[9], 0.05f);
    float ls = luminance3(b->f0), ld = luminance3(b->da);
    float p = (ls + ld) > 0.0f ? ls / (ls + ld) : 1.0f;
    b->p_spec = ld > 0.0f ? fminx(fmaxx(p, 0.1f), 0.9f) : 1.0f;
}
/* value (BRDF without the cosine) and mixture pdf (projected solid angle) for wo, wi in the tangent frame */
static void bsdf_eval(const bsdf_t *b, const float wo[3], const float wi[3], float value[3], float *pdf_proj) {
    value[0] = value[1] = value[2] = 0.0f; *pdf_proj = 0.0f;
    if (!(wi[2] > 0.0f)) return;
    if (!(wo[2] > 1e-5f)) { /* grazing / back-facing view: diffuse only */
        for (int k = 0; k < 3; k++) value[k] = b->da[k] * F_FRAC_1_PI;
        *pdf_proj = F_FRAC_1_PI;
        return;
    }
    const float a2 = b->alpha * b->alpha;
    float h[3] = {wo[0] + wi[0], wo[1] + wi[1], wo[2] + wi[2]};
    normalize3(h); /* brdf.slang:268 */
    float vh = dot3(wi, h);
    float fr = pow5(fmaxx(0.0f, 1.0f - vh)); /* eval_fresnel_schlick, brdf.slang:95-97 */
    float G = g_smith_ggx_correlated(wo[2], wi[2], a2), D = ggx_ndf(a2, h[2]);
    float pdf_h = g_smith_ggx1(wo[2], a2) * D * fmaxx(0.0f, dot3(wo, h)) / wo[2]; /* pdf_ggx_vn, :161-165 */
    float pdf_spec = vh > 0.0f ? pdf_h * (1.0f / (4.0f * vh)) / wi[2] : 0.0f;    /* :278,286 */
    float spec_scale = G * D / (4.0f * wo[2] * wi[2]);                            /* :302-306 */
    for (int k = 0; k < 3; k++) {
        float F = b->f0[k] + (1.0f - b->f0[k]) * fr;
        value[k] = F * spec_scale + b->da[k] * F_FRAC_1_PI * (1.0f - F);
    }
    *pdf_proj = b->p_spec * pdf_spec + (1.0f - b->p_spec) * F_FRAC_1_PI;
}
/* sample_vndf (brdf.slang:187-216, Heitz 2018 as in Falcor) -> half vector */
static void sample_vndf(float alpha, const float wo[3], float u0, float u1, float h[3]) {
    float Vh[3] = {alpha * wo[0], alpha * wo[1], wo[2]};
    normalize3(Vh);
    float T1[3] = {1.0f, 0.0f, 0.0f};
    if (Vh[2] < 0.9999f) { T1[0] = -Vh[1]; T1[1] = Vh[0]; T1[2] = 0.0f; normalize3(T1); } /* cross((0,0,1), Vh) */
    float T2[3]; cross3(Vh, T1, T2);
    float r = sqrtf(u0), sp, cp;
    orc_sincos_2pi(u1, &sp, &cp);
    float t1 = r * cp, t2 = r * sp, sv = 0.5f * (1.0f + Vh[2]);
    t2 = (1.0f - sv) * sqrtf(1.0f - t1 * t1) + sv * t2;
    float nz = sqrtf(fmaxx(0.0f, 1.0f - t1 * t1 - t2 * t2));
    float Nh[3];
    for (int k = 0; k < 3; k++) Nh[k] = t1 * T1[k] + t2 * T2[k] + nz * Vh[k];
    h[0] = alpha * Nh[0]; h[1] = alpha * Nh[1]; h[2] = fmaxx(0.0f, Nh[2]);
    normalize3(h);
}
/* returns 0 if the sample is invalid (path ends); else wi, value_over_pdf (throughput factor) and the solid-angle pdf */
static int bsdf_sample(const bsdf_t *b, const float wo[3], float u0, float u1, float u2, float wi[3], float vop[3], float *pdf_solid) {
    if (wo[2] > 1e-5f && u2 < b->p_spec) {
        float h[3];
        sample_vndf(b->alpha, wo, u0, u1, h);
        float s2 = 2.0f * dot3(wo, h);
        for (int k = 0; k < 3; k++) wi[k] = s2 * h[k] - wo[k]; /* reflect(-wo, m) */
        if (h[2] <= 1e-5f || wi[2] <= 1e-5f) return 0;        /* BRDF_SAMPLING_MIN_COS, brdf.slang:227 */
    } else {
        orc_diffuse_sample(u0, u1, wi);
    }
    float value[3], pdf;
    bsdf_eval(b, wo, wi, value, &pdf);
    if (!(pdf > 0.0f)) return 0;
    for (int k = 0; k < 3; k++) vop[k] = value[k] / pdf;
    *pdf_solid = pdf * wi[2];
    return 1;
}

/* ------------------------------------------------------------------------------------------------ passes */
typedef struct {
    const orc_scene *s; const orc_gconst *g; uint32_t x0, y0, x1, y1;
    uint32_t *gbuffer; float *depth; const uint32_t *gb_in; const float *depth_in; const float *prev; float *light;
    const float *in; float *out; uint64_t (*counts)[6];
} pass_job;

/* gbuffer.slang:8-21 */
static void gbuffer_body(void *c, uint32_t b, uint32_t e, int tid) {
    (void)tid;
    pass_job *j = (pass_job *)c;
    uint32_t W = (uint32_t)j->g->window_size[0], rw = j->x1 - j->x0;
    for (uint32_t i = b; i < e; i++) {
        uint32_t px = j->x0 + i % rw, py = j->y0 + i / rw;
        float o[3], d[3];
        orc_primary_ray(j->g, px, py, o, d);
        hit_t h;
        traverse(j->s, o, d, 0.0f, ORC_BACKGROUND_DEPTH, 0, &h, NULL, NULL); /* setupPrimaryRay: TMin 0, TMax 1e5 */
        size_t pi = (size_t)py * W + px;
        if (h.prim == ORC_MISS) {
            j->depth[pi] = ORC_BACKGROUND_DEPTH;
        } else {
            float surf[11];
            orc_hit_info(j->s, h.prim, h.u, h.v, surf);
            orc_gbuffer_pack(surf, j->gbuffer + 4 * pi);
            j->depth[pi] = h.t;
        }
    }
}
void orc_pass_gbuffer(const orc_scene *s, const orc_gconst *g, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                      uint32_t *gbuffer, float *depth, int n_threads) {
    pass_job j; memset(&j, 0, sizeof(j));
    j.s = s; j.g = g; j.x0 = x0; j.y0 = y0; j.x1 = x1; j.y1 = y1; j.gbuffer = gbuffer; j.depth = depth;
    parallel_for((x1 - x0) * (y1 - y0), n_threads, gbuffer_body, &j);
}

/* refrence_mode.slang:14-66.  Deviations (all documented in DESIGN.md):
 *  - RNG counter for (sample s, bounce b, dim d) = (s*B + b)*DIMS + d with DIMS = 2 (reference semantics) or 8
 *    (any feature flag set) instead of the sequential `index++`;
 *  - per-sample radiance L_s is summed over bounces first, then radiance = sum_s L_s in order;
 *  - hit_info is not evaluated on a miss (Appendix A (4));
 *  - [north_star] sky NEE with balance-heuristic MIS, blue-noise shift, face-forward normals behind flags. */
static void refmode_body(void *c, uint32_t bgn, uint32_t end, int tid) {
    pass_job *j = (pass_job *)c;
    const orc_scene *s = j->s; const orc_gconst *g = j->g;
    uint32_t W = (uint32_t)g->window_size[0], rw = j->x1 - j->x0;
    const uint32_t flags = g->pad[0], B = g->bounces, S = g->samples;
    const uint32_t dims = flags ? 8u : 2u;
    const int nee = (flags & ORC_F_NEE_SKY) && s->sky, bnz = (flags & ORC_F_BLUENOISE) && s->bn, spec = (flags & ORC_F_SPECULAR) != 0;
    uint64_t n_ext = 0, n_sh = 0, n_nodes = 0, n_tris = 0, n_sh_nodes = 0, n_sh_tris = 0;
    for (uint32_t i = bgn; i < end; i++) {
        uint32_t px = j->x0 + i % rw, py = j->y0 + i / rw;
        size_t pi = (size_t)py * W + px;
        float d0 = j->depth_in[pi];
        if (d0 == ORC_BACKGROUND_DEPTH) continue; /* :18-21 Light untouched */
        float surf0[11];
        orc_gbuffer_unpack(j->gb_in + 4 * pi, surf0); /* :23 */
        uint32_t seed = orc_rng_seed(px, py, g->frame); /* :25 */
        uint32_t bn[4] = {0, 0, 0, 0};
        if (bnz) { const uint8_t *p = s->bn + 4 * ((size_t)(py % s->bn_h) * s->bn_w + (px % s->bn_w)); bn[0] = p[0]; bn[1] = p[1]; bn[2] = p[2]; bn[3] = p[3]; }
        float radiance[3] = {0.0f, 0.0f, 0.0f};
        for (uint32_t sm = 0; sm < S; sm++) { /* :28 */
            float T[3] = {1.0f, 1.0f, 1.0f}, L[3] = {0.0f, 0.0f, 0.0f};
            float o[3], d[3], surf[11];
            orc_primary_ray(g, px, py, o, d); /* :30 */
            memcpy(surf, surf0, sizeof(surf));
            float t = d0; int hit = 1;
            float pdf_b = 0.0f;
            for (uint32_t b = 0; b < B; b++) { /* :36 */
                if (!hit) { /* :37-40 (sky term commented out in the reference; [north_star] MIS-weighted sky) */
                    if (nee) {
                        float uv[2], rad[3];
                        orc_dir_to_equirect_uv(d, uv);
                        sky_eval(s, uv[0], uv[1], rad);
                        float pl = sky_pdf(s, uv[0], uv[1]);
                        float w = pdf_b / (pdf_b + pl);
                        if (pdf_b > 0.0f) for (int k = 0; k < 3; k++) L[k] += T[k] * (rad[k] * w);
                    }
                    break;
                }
                uint32_t base = (sm * B + b) * dims;
                float u0 = orc_uniform_float(seed, base), u1 = orc_uniform_float(seed, base + 1); /* :43 */
                if (bnz) { u0 = bn_shift(u0, bn[0]); u1 = bn_shift(u1, bn[1]); }
                float *alb = surf, *emi = surf + 3, N[3] = {surf[6], surf[7], surf[8]};
                if ((flags & ORC_F_FACEFORWARD) && dot3(N, d) > 0.0f) { N[0] = -N[0]; N[1] = -N[1]; N[2] = -N[2]; }
                float b1[3], b2[3], wi[3], vop[3] = {alb[0], alb[1], alb[2]}, pdf_s = 0.0f, wo[3] = {0.0f, 0.0f, 1.0f};
                int valid = 1;
                bsdf_t bs;
                orc_onb(N, b1, b2);          /* :44 */
                if (spec) {
                    bsdf_setup(surf, &bs);
                    wo[0] = -(d[0] * b1[0] + d[1] * b1[1] + d[2] * b1[2]);
                    wo[1] = -(d[0] * b2[0] + d[1] * b2[1] + d[2] * b2[2]);
                    wo[2] = -(d[0] * N[0] + d[1] * N[1] + d[2] * N[2]);
                    float u2 = orc_uniform_float(seed, base + 2);
                    valid = bsdf_sample(&bs, wo, u0, u1, u2, wi, vop, &pdf_s);
                } else {
                    orc_diffuse_sample(u0, u1, wi); /* :45 */
                    pdf_s = wi[2] * F_FRAC_1_PI;
                }
                for (int k = 0; k < 3; k++) o[k] = o[k] + t * d[k]; /* :47 */
                for (int k = 0; k < 3; k++) L[k] += T[k] * emi[k]; /* :50 */
                if (nee) { /* [north_star] next-event estimation against the sky */
                    float ul0 = orc_uniform_float(seed, base + 3), ul1 = orc_uniform_float(seed, base + 4);
                    if (bnz) { ul0 = bn_shift(ul0, bn[2]); ul1 = bn_shift(ul1, bn[3]); }
                    float wl[3], rad[3], pl;
                    sky_sample(s, ul0, ul1, wl, rad, &pl);
                    float cosl = dot3(N, wl);
                    if (cosl > 0.0f && pl > 0.0f) {
                        float fv[3], scale;
                        if (spec) { /* evaluate the layered BSDF towards the light; f cos / (p_light + p_bsdf) */
                            float wlt[3] = {dot3(wl, b1), dot3(wl, b2), cosl}, pproj;
                            bsdf_eval(&bs, wo, wlt, fv, &pproj);
                            float pb = pproj * cosl;
                            scale = (b == B - 1) ? cosl / pl : cosl / (pl + pb);
                        } else { /* diffuse: f = albedo / pi folded into the scale */
                            float pb = cosl * F_FRAC_1_PI;
                            fv[0] = alb[0]; fv[1] = alb[1]; fv[2] = alb[2];
                            scale = (b == B - 1) ? (cosl * F_FRAC_1_PI) / pl : (cosl * F_FRAC_1_PI) / (pl + pb);
                        }
                        hit_t sh;
                        uint32_t cn, ct;
                        traverse(s, o, wl, 0.001f, ORC_BACKGROUND_DEPTH, 1, &sh, &cn, &ct);
                        n_sh++; n_nodes += cn; n_tris += ct; n_sh_nodes += cn; n_sh_tris += ct;
                        if (sh.prim == ORC_MISS) for (int k = 0; k < 3; k++) L[k] += (T[k] * fv[k]) * (rad[k] * scale);
                    }
                }
                if (!valid) break; /* invalid specular sample (brdf.slang:227-229): the path ends */
                float nd[3];
                onb_apply(b1, b2, N, wi, nd); /* :48 */
                pdf_b = pdf_s;
                for (int k = 0; k < 3; k++) { T[k] = T[k] * vop[k]; d[k] = nd[k]; } /* :51 value_over_pdf (= albedo for the diffuse BRDF) */
                if (b != B - 1) { /* :53-56 */
                    hit_t h; uint32_t cn, ct;
                    traverse(s, o, d, 0.001f, ORC_BACKGROUND_DEPTH, 0, &h, &cn, &ct); /* TMin 0.001 (:31) */
                    n_ext++; n_nodes += cn; n_tris += ct;
                    if (h.prim == ORC_MISS) hit = 0;
                    else { t = h.t; orc_hit_info(s, h.prim, h.u, h.v, surf); }
                }
            }
            for (int k = 0; k < 3; k++) radiance[k] += L[k];
        }
        for (int k = 0; k < 3; k++) radiance[k] = radiance[k] / (float)S; /* :59 */
        float *out = j->light + 4 * pi;
        if (g->blendfactor >= 1.0f) { /* :61-65 */
            out[0] = radiance[0]; out[1] = radiance[1]; out[2] = radiance[2]; out[3] = 0.0f;
        } else {
            const float *pv = j->prev + 4 * pi; float bf = g->blendfactor;
            for (int k = 0; k < 3; k++) out[k] = pv[k] + (radiance[k] - pv[k]) * bf; /* lerp(a,b,t) = a + (b-a)*t */
            out[3] = 0.0f;
        }
    }
    if (j->counts) { j->counts[tid][0] = n_ext; j->counts[tid][1] = n_sh; j->counts[tid][2] = n_nodes; j->counts[tid][3] = n_tris; j->counts[tid][4] = n_sh_nodes; j->counts[tid][5] = n_sh_tris; }
}
/* The ray batch the path tracer traces at bounce 1, sample 0 -- one extension ray per non-background pixel of the window, exactly as
 * refmode_body would emit it at (sm 0, b 0) for g's flags -- as 8 SoA arrays of `cap` floats (ox oy oz dx dy dz tmin tmax), row-major
 * pixel order, background pixels and ended paths skipped.  Returns the number of rays.  For the CPU-baseline leg of bench.py
 * ("the C2 primary batch + the bounce-1 batch"). */
uint32_t orc_bounce1_rays(const orc_scene *s, const orc_gconst *g, const uint32_t *gbuffer, const float *depth, float *rays, uint32_t cap) {
    const uint32_t W = (uint32_t)g->window_size[0], H = (uint32_t)g->window_size[1], flags = g->pad[0], B = g->bounces;
    const uint32_t dims = flags ? 8u : 2u;
    const int bnz = (flags & ORC_F_BLUENOISE) && s->bn, spec = (flags & ORC_F_SPECULAR) != 0;
    uint32_t n = 0;
    if (B < 2) return 0;
    for (uint32_t py = 0; py < H; py++)
        for (uint32_t px = 0; px < W; px++) {
            size_t pi = (size_t)py * W + px;
            float d0 = depth[pi];
            if (d0 == ORC_BACKGROUND_DEPTH || n >= cap) continue;
            float surf[11], o[3], d[3];
            orc_gbuffer_unpack(gbuffer + 4 * pi, surf);
            orc_primary_ray(g, px, py, o, d);
            uint32_t seed = orc_rng_seed(px, py, g->frame);
            float u0 = orc_uniform_float(seed, 0), u1 = orc_uniform_float(seed, 1);
            if (bnz) { const uint8_t *p = s->bn + 4 * ((size_t)(py % s->bn_h) * s->bn_w + (px % s->bn_w)); u0 = bn_shift(u0, p[0]); u1 = bn_shift(u1, p[1]); }
            float N[3] = {surf[6], surf[7], surf[8]};
            if ((flags & ORC_F_FACEFORWARD) && dot3(N, d) > 0.0f) { N[0] = -N[0]; N[1] = -N[1]; N[2] = -N[2]; }
            float b1[3], b2[3], wi[3], vop[3], pdf_s, nd[3];
            orc_onb(N, b1, b2);
            if (spec) {
                bsdf_t bs; bsdf_setup(surf, &bs);
                float wo[3] = {-(d[0] * b1[0] + d[1] * b1[1] + d[2] * b1[2]), -(d[0] * b2[0] + d[1] * b2[1] + d[2] * b2[2]), -(d[0] * N[0] + d[1] * N[1] + d[2] * N[2])};
                if (!bsdf_sample(&bs, wo, u0, u1, orc_uniform_float(seed, dims * 0u + 2u), wi, vop, &pdf_s)) continue;
            } else orc_diffuse_sample(u0, u1, wi);
            onb_apply(b1, b2, N, wi, nd);
            for (int k = 0; k < 3; k++) { rays[(size_t)k * cap + n] = o[k] + d0 * d[k]; rays[(size_t)(3 + k) * cap + n] = nd[k]; }
            rays[(size_t)6 * cap + n] = 0.001f; rays[(size_t)7 * cap + n] = ORC_BACKGROUND_DEPTH;
            n++;
        }
    return n;
}
void orc_pass_reference_mode(const orc_scene *s, const orc_gconst *g, uint32_t x0, uint32_t y0, uint32_t x1,
                             uint32_t y1, const uint32_t *gbuffer, const float *depth, const float *prev_light,
                             float *light, uint64_t *ray_counts, int n_threads) {
    pass_job j; memset(&j, 0, sizeof(j));
    uint64_t counts[256][6]; memset(counts, 0, sizeof(counts));
    j.s = s; j.g = g; j.x0 = x0; j.y0 = y0; j.x1 = x1; j.y1 = y1; j.gb_in = gbuffer; j.depth_in = depth;
    j.prev = prev_light; j.light = light; j.counts = counts;
    parallel_for((x1 - x0) * (y1 - y0), n_threads, refmode_body, &j);
    if (ray_counts) {
        for (int k = 0; k < 6; k++) ray_counts[k] = 0;
        for (int t = 0; t < 256; t++) for (int k = 0; k < 6; k++) ray_counts[k] += counts[t][k];
    }
}
/* postprocess.slang:90-112 */
static void post_body(void *c, uint32_t b, uint32_t e, int tid) {
    (void)tid;
    pass_job *j = (pass_job *)c;
    uint32_t W = (uint32_t)j->g->window_size[0], rw = j->x1 - j->x0;
    for (uint32_t i = b; i < e; i++) {
        uint32_t px = j->x0 + i % rw, py = j->y0 + i / rw;
        size_t pi = (size_t)py * W + px;
        float col[3], outc[3];
        if (j->depth_in[pi] != ORC_BACKGROUND_DEPTH) {
            col[0] = j->in[4 * pi]; col[1] = j->in[4 * pi + 1]; col[2] = j->in[4 * pi + 2];
        } else {
            float o[3], d[3], uv[2];
            orc_primary_ray(j->g, px, py, o, d);
            orc_dir_to_equirect_uv(d, uv);
            sky_eval(j->s, uv[0], uv[1], col);
        }
        orc_agx_tonemap(col, outc);
        j->out[4 * pi] = outc[0]; j->out[4 * pi + 1] = outc[1]; j->out[4 * pi + 2] = outc[2]; j->out[4 * pi + 3] = 1.0f;
    }
}
void orc_pass_postprocess(const orc_scene *s, const orc_gconst *g, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                          const float *depth, const float *in, float *out, int n_threads) {
    pass_job j; memset(&j, 0, sizeof(j));
    j.s = s; j.g = g; j.x0 = x0; j.y0 = y0; j.x1 = x1; j.y1 = y1; j.depth_in = depth; j.in = in; j.out = out;
    parallel_for((x1 - x0) * (y1 - y0), n_threads, post_body, &j);
}

/* ------------------------------------------------------------------------------------------------ tiles [north_star] */
static uint32_t compact1by1(uint32_t x) {
    x &= 0x55555555u;
    x = (x | (x >> 1)) & 0x33333333u;
    x = (x | (x >> 2)) & 0x0F0F0F0Fu;
    x = (x | (x >> 4)) & 0x00FF00FFu;
    x = (x | (x >> 8)) & 0x0000FFFFu;
    return x;
}
uint32_t orc_tile_pixels(uint32_t w, uint32_t h, uint32_t rank, uint32_t n_ranks, uint32_t *out_xy) {
    uint32_t tw = (w + 63) / 64, th = (h + 63) / 64, count = 0, tile_no = 0;
    uint32_t side = 1;
    while (side < tw || side < th) side *= 2;
    for (uint32_t z = 0; z < side * side; z++) {
        uint32_t tx = compact1by1(z), ty = compact1by1(z >> 1);
        if (tx >= tw || ty >= th) continue;
        uint32_t owner = tile_no % n_ranks;
        tile_no++;
        if (owner != rank) continue;
        for (uint32_t k = 0; k < 4096; k++) {
            uint32_t x = tx * 64 + compact1by1(k), y = ty * 64 + compact1by1(k >> 1);
            if (x >= w || y >= h) continue;
            if (out_xy) { out_xy[2 * count] = x; out_xy[2 * count + 1] = y; }
            count++;
        }
    }
    return count;
}
