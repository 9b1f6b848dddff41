/*
 * rt3_oracle_probes.c -- CPU ORACLE (test infrastructure, NOT the product) for the probe-GI passes of SURVEY.md 8f rank 4:
 * /root/reference/shaders/old/{structured_importance_sampling,trace_probes,spherical_harmonic_conversion,
 * interpolate_probes}.slang with shaders/include/{spherical_harmonics,packing,math}.slang.
 *
 * Parity unpinned: the reference has no host code left for these passes (only GConst.proberng / cell_size,
 * src/renderer/mod.rs:59-60), no tests and no outputs.  The shaders are restated AS WRITTEN, debug outputs included;
 * where the text leaves the result open the rule chosen here (and in the HIP product) is marked [rule]:
 *   - cross-lane float sums / the 64-element sort use one fixed network (butterfly over lane ^ 1, 2, 4, .. 32;
 *     bitonic network of math.slang:140-160 with `indecies` initialised to the lane id);
 *   - racing image stores resolve as "all initial stores first, then scatter stores in row-major thread order";
 *   - a probe ray that misses gets depth BACKGROUND_DEPTH and radiance 0 (default_miss.slang leaves depth unset);
 *   - out-of-range threads / probes do nothing (Vulkan robust access would drop those loads and stores), and so does a
 *     trace_probes scatter store whose direction word lies outside the octahedral map;
 *   - pow(x, 8) is three squarings; StructuredBuffer<float3x3> uses the std430 layout (3 rows x float4).
 * Same arithmetic contract as rt3_oracle.c (fp32, no contraction, left-to-right sums).
 */
#include "rt3_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define SH_PI 3.1415926536f /* spherical_harmonics.slang:4 */
#define F_FRAC_1_PI 0.318309886183790671538f

static inline float fminx(float a, float b) { return a < b ? a : b; }
static inline float fmaxx(float a, float b) { return a > b ? a : b; }
static inline float dot3(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void normalize3(float v[3]) {
    float inv = 1.0f / sqrtf(dot3(v, v));
    v[0] *= inv; v[1] *= inv; v[2] *= inv;
}

/* packing.slang:77-86 */
void orc_octa_decode(float fx, float fy, float n[3]) {
    fx = fx * 2.0f - 1.0f;
    fy = fy * 2.0f - 1.0f;
    n[0] = fx; n[1] = fy; n[2] = 1.0f - fabsf(fx) - fabsf(fy);
    float t = fminx(fmaxx(-n[2], 0.0f), 1.0f);
    n[0] -= ((n[0] >= 0.0f ? 1.0f : 0.0f) * 2.0f - 1.0f) * t;
    n[1] -= ((n[1] >= 0.0f ? 1.0f : 0.0f) * 2.0f - 1.0f) * t;
    normalize3(n);
}

/* spherical_harmonics.slang:30-44 ; sh[r*3+c] = result[r][c] */
void orc_sh3_evaluate(const float d[3], float sh[9]) {
    sh[0] = 0.28209479177387814347403972578039f;
    sh[1] = -0.48860251190291992158638462283836f * d[1];
    sh[2] = 0.48860251190291992158638462283836f * d[2];
    sh[3] = -0.48860251190291992158638462283836f * d[0];
    sh[4] = 1.09254843059207907054338570580268f * d[0] * d[1];
    sh[5] = 1.09254843059207907054338570580268f * d[1] * d[2];
    sh[6] = 0.31539156525252000603089369029571f * (3.0f * d[2] * d[2] - 1.0f);
    sh[7] = 1.09254843059207907054338570580268f * d[0] * d[2];
    sh[8] = 0.54627421529603953527169285290134f * (d[0] * d[0] - d[1] * d[1]);
}
/* spherical_harmonics.slang:73-89 */
static void sh3_transform_cos_lobe(const float n[3], float sh[9]) {
    orc_sh3_evaluate(n, sh);
    sh[0] *= SH_PI;
    sh[1] *= 2.0943951023931954923f; sh[2] *= 2.0943951023931954923f; sh[3] *= 2.0943951023931954923f;
    for (int i = 4; i < 9; i++) sh[i] *= 0.7853981633974483096f;
}
/* spherical_harmonics.slang:59-63 : nine products summed left to right */
static float matrix_dot(const float a[9], const float b[9]) {
    float s = a[0] * b[0];
    for (int i = 1; i < 9; i++) s = s + a[i] * b[i];
    return s;
}

/* math.slang:140-160 with `indecies[i] = i` [rule]; the 64 threads of a group are the 64 lanes of one wave */
void orc_wave_sort64(float keys[64], uint32_t idx[64]) {
    for (uint32_t i = 0; i < 64; i++) idx[i] = i;
    for (uint32_t k = 2; k <= 64; k *= 2)
        for (uint32_t j = k / 2; j > 0; j /= 2)
            for (uint32_t i = 0; i < 64; i++) {
                uint32_t l = i ^ j;
                if (l > i) {
                    if ((((i & k) == 0) && (keys[i] > keys[l])) || (((i & k) != 0) && (keys[i] < keys[l]))) {
                        float t = keys[i]; keys[i] = keys[l]; keys[l] = t;
                        uint32_t u = idx[i]; idx[i] = idx[l]; idx[l] = u;
                    }
                }
            }
}
/* WaveActiveSum (spherical_harmonic_conversion.slang:20-22) [rule]: butterfly, partner lane ^ 1, then ^ 2, ... ^ 32 */
float orc_wave_sum64(const float v[64]) {
    float a[64], b[64];
    memcpy(a, v, sizeof(a));
    for (uint32_t off = 1; off < 64; off *= 2) {
        for (uint32_t i = 0; i < 64; i++) b[i] = a[i] + a[i ^ off];
        memcpy(a, b, sizeof(a));
    }
    return a[0];
}

/* structured_importance_sampling.slang:19-71 */
void orc_pass_structured_importance_sampling(const orc_gconst *g, uint32_t probes_x, uint32_t probes_y, const uint32_t *gbuffer,
                                             uint16_t *out, float *debug) {
    uint32_t W = (uint32_t)g->window_size[0], AW = probes_x * 8;
    for (uint32_t gy = 0; gy < probes_y; gy++)
        for (uint32_t gx = 0; gx < probes_x; gx++) {
            float normals[256][3], pdfs[64], mine[64];
            uint32_t idx[64];
            for (uint32_t ti = 0; ti < 64; ti++) { /* :24-30 */
                uint32_t tx = gx * 8 + ti % 8, ty = gy * 8 + ti / 8;
                for (uint32_t y = 0; y < 2; y++)
                    for (uint32_t x = 0; x < 2; x++) {
                        size_t pi = (size_t)(ty * 2 + y) * W + (tx * 2 + x);
                        orc_unpack_normal_11_10_11(gbuffer[4 * pi + 1], normals[ti * 4 + y * 2 + x]);
                    }
            }
            for (uint32_t ti = 0; ti < 64; ti++) { /* :33-39 */
                float dir[3];
                orc_octa_decode(((float)(ti % 8) + 0.5f) / 8.0f, ((float)(ti / 8) + 0.5f) / 8.0f, dir);
                float pdf = 0.0f;
                for (int i = 0; i < 256; i++) pdf += fmaxx(dot3(normals[i], dir), 0.0f) / 256.0f;
                pdfs[ti] = mine[ti] = pdf;
            }
            orc_wave_sort64(pdfs, idx); /* :45 */
            /* :47-50: brdf_pdf < 0 never holds (a sum of max(.., 0); a NaN compares false) -> index = -1, culled_rays = 0 */
            for (uint32_t ti = 0; ti < 64; ti++) { /* :55-70 */
                size_t ai = (size_t)(gy * 8 + ti / 8) * AW + (gx * 8 + ti % 8);
                if (pdfs[0] < mine[ti]) out[ai] = (uint16_t)((1u << 15) | (ti * 4));
                else out[ai] = (uint16_t)ti;
                debug[ai] = -1.0f; /* :70 debug[..] = index */
            }
        }
}

/* trace_probes.slang:15-77 */
void orc_pass_trace_probes(const orc_scene *s, const orc_gconst *g, uint32_t probes_x, uint32_t probes_y, const uint32_t *gbuffer,
                           const float *depth, const uint16_t *directions, const float *prev_atlas, float *atlas, int n_threads) {
    (void)gbuffer; /* primary_brdf (:37-38) is only used by the commented-out estimator */
    uint32_t W = (uint32_t)g->window_size[0], AW = probes_x * 8, AH = probes_y * 8;
    size_t n = (size_t)AW * AH;
    float *rays = (float *)calloc(n * 8 + 1, sizeof(float));
    float *ht = (float *)malloc(n * 3 * sizeof(float));
    uint32_t *hp = (uint32_t *)malloc(n * sizeof(uint32_t));
    float *d2 = (float *)malloc(n * 2 * sizeof(float));
    for (uint32_t ay = 0; ay < AH; ay++)
        for (uint32_t ax = 0; ax < AW; ax++) {
            size_t a = (size_t)ay * AW + ax;
            uint32_t px = (ax / 8) * 16, py = (ay / 8) * 16; /* :24 */
            float d0 = depth[(size_t)py * W + px];
            float o[3] = {0, 0, 0}, d[3] = {0, 0, 1}, tmin = 0.0f, tmax = -1.0f; /* inactive: can hit nothing */
            if (d0 == ORC_BACKGROUND_DEPTH) { /* :29-31 */
                atlas[4 * a + 0] = atlas[4 * a + 1] = atlas[4 * a + 2] = 0.0f;
                atlas[4 * a + 3] = ORC_BACKGROUND_DEPTH;
                d2[2 * a] = -1.0f;
            } else {
                atlas[4 * a + 0] = atlas[4 * a + 1] = atlas[4 * a + 2] = atlas[4 * a + 3] = 0.0f; /* :33 */
                uint32_t seed = orc_rng_seed(ax, ay, g->frame);                                 /* :21 ray_rng */
                uint32_t dw = directions[a], di = dw & 0x7FFFu, mip = dw >> 15, size = (1u << mip) * 8u; /* :41-44 */
                float fx = (float)(di % size), fy = (float)(di / size), fs = (float)size;               /* :45 */
                float u0 = orc_uniform_float(seed, 0), u1 = orc_uniform_float(seed, 1);
                orc_octa_decode((fx + u0) / fs, (fy + u1) / fs, d); /* :47 */
                float po[3], pd[3];
                orc_primary_ray(g, px, py, po, pd); /* world_pos_from_depth, gbuffer_helpers.slang:81-83 */
                for (int k = 0; k < 3; k++) o[k] = po[k] + pd[k] * d0;
                tmin = 0.0005f; tmax = ORC_BACKGROUND_DEPTH; /* :55-56 */
                d2[2 * a] = fx / fs; d2[2 * a + 1] = fy / fs;
            }
            rays[0 * n + a] = o[0]; rays[1 * n + a] = o[1]; rays[2 * n + a] = o[2];
            rays[3 * n + a] = d[0]; rays[4 * n + a] = d[1]; rays[5 * n + a] = d[2];
            rays[6 * n + a] = tmin; rays[7 * n + a] = tmax;
        }
    orc_trace_closest(s, rays, (uint32_t)n, ht, ht + n, ht + 2 * n, hp, NULL, NULL, n_threads);
    for (uint32_t ay = 0; ay < AH; ay++) /* [rule] scatter stores after all initial stores, row-major thread order */
        for (uint32_t ax = 0; ax < AW; ax++) {
            size_t a = (size_t)ay * AW + ax;
            if (d2[2 * a] < 0.0f) continue;
            float probe_depth = hp[a] == ORC_MISS ? ORC_BACKGROUND_DEPTH : ht[a]; /* [rule] miss */
            if (g->pad[0] & ORC_F_PROBE_RADIANCE) { /* the store the shader keeps in a comment, :74 */
                float rad[3] = {0, 0, 0}, surf[11];
                if (hp[a] != ORC_MISS) {
                    orc_hit_info(s, hp[a], ht[n + a], ht[2 * n + a], surf);
                    rad[0] = surf[3]; rad[1] = surf[4]; rad[2] = surf[5]; /* :62 radiance = emissive */
                }
                for (int k = 0; k < 3; k++) {
                    float p = prev_atlas[4 * a + k];
                    atlas[4 * a + k] = p + (rad[k] - p) * g->blendfactor; /* lerp(prev, radiance, blendfactor) */
                }
                atlas[4 * a + 3] = probe_depth;
            } else { /* as written, :74 */
                uint32_t lx = (uint32_t)(d2[2 * a] * 8.0f), ly = (uint32_t)(d2[2 * a + 1] * 8.0f);
                if (lx >= 8 || ly >= 8) continue; /* [rule] a direction word outside the octahedral map (index >= size^2) stores nothing */
                size_t t = (size_t)((ay / 8) * 8 + ly) * AW + (ax / 8) * 8 + lx;
                atlas[4 * t + 0] = d2[2 * a]; atlas[4 * t + 1] = d2[2 * a + 1]; atlas[4 * t + 2] = 0.0f;
                atlas[4 * t + 3] = probe_depth;
            }
        }
    free(rays); free(ht); free(hp); free(d2);
}

/* spherical_harmonic_conversion.slang:9-33 ; sh_out element e = 12 floats (rows padded to float4), e = zcurve(3*gx + c, gy) */
void orc_pass_sh_conversion(uint32_t probes_x, uint32_t probes_y, const float *atlas, float *sh_out) {
    uint32_t AW = probes_x * 8;
    const float factor = 4.0f * SH_PI / 64.0f; /* :25 */
    for (uint32_t gy = 0; gy < probes_y; gy++)
        for (uint32_t gx = 0; gx < probes_x; gx++) {
            float term[3][9][64];
            for (uint32_t ti = 0; ti < 64; ti++) {
                float dir[3], sh[9];
                orc_octa_decode(((float)(ti % 8) + 0.5f) / 8.0f, ((float)(ti / 8) + 0.5f) / 8.0f, dir);
                orc_sh3_evaluate(dir, sh);
                const float *col = atlas + 4 * ((size_t)(gy * 8 + ti / 8) * AW + (gx * 8 + ti % 8));
                for (int c = 0; c < 3; c++)
                    for (int k = 0; k < 9; k++) term[c][k][ti] = sh[k] * col[c];
            }
            for (uint32_t c = 0; c < 3; c++) {
                float *o = sh_out + 12 * (size_t)orc_zcurve(gx * 3 + c, gy);
                for (int k = 0; k < 9; k++) o[(k / 3) * 4 + k % 3] = orc_wave_sum64(term[c][k]) * factor;
                o[3] = o[7] = o[11] = 0.0f;
            }
        }
}

/* interpolate_probes.slang:11-103 */
static void world_pos(const orc_gconst *g, float depth, uint32_t px, uint32_t py, float out[3]) {
    float o[3], d[3];
    orc_primary_ray(g, px, py, o, d);
    for (int k = 0; k < 3; k++) out[k] = o[k] + d[k] * depth;
}
static inline float pow8(float x) { float x2 = x * x, x4 = x2 * x2; return x4 * x4; }

/* returns 0 = nothing written, 1 = radiance for (px,py) in rad, 2 = interpolation failed: red at (*fx,*fy) */
static int interpolate_pixel(const orc_gconst *g, const uint32_t *gbuffer, const float *depth, const float *sh, uint32_t px, uint32_t py,
                             float rad[3], uint32_t *fx, uint32_t *fy) {
    uint32_t W = (uint32_t)g->window_size[0], H = (uint32_t)g->window_size[1];
    float pixel_depth = depth[(size_t)py * W + px];
    if (pixel_depth == ORC_BACKGROUND_DEPTH) return 0; /* :19-22 */
    uint32_t seed = orc_rng_seed(px, py, g->frame);
    float surf[11], pos[3], jpos[3], v[3];
    orc_gbuffer_unpack(gbuffer + 4 * ((size_t)py * W + px), surf);
    const float *nrm = surf + 6;
    world_pos(g, pixel_depth, px, py, pos);
    float u0 = orc_uniform_float(seed, 0), u1 = orc_uniform_float(seed, 1);
    int jx = (int)((2.0f * u0 - 1.0f) * 16.0f), jy = (int)((2.0f * u1 - 1.0f) * 16.0f); /* :31 */
    int cx = (int)px + jx, cy = (int)py + jy;
    cx = cx < 0 ? 0 : (cx > (int)W - 1 ? (int)W - 1 : cx);
    cy = cy < 0 ? 0 : (cy > (int)H - 1 ? (int)H - 1 : cy);
    world_pos(g, depth[(size_t)cy * W + cx], (uint32_t)cx, (uint32_t)cy, jpos);
    for (int k = 0; k < 3; k++) v[k] = jpos[k] - pos[k];
    normalize3(v);
    uint32_t qx = px, qy = py;
    if (fabsf(dot3(v, nrm)) < 0.01f) { qx = (uint32_t)cx; qy = (uint32_t)cy; } /* :36-38 */
    uint32_t lpx = qx / 16, lpy = qy / 16, NPX = W / 16, NPY = H / 16;
    float w[4] = {0, 0, 0, 0};
    for (uint32_t i = 0; i < 4; i++) { /* :51-69 */
        uint32_t cxp = lpx + (i & 1), cyp = lpy + (i >> 1);
        if (cxp >= NPX || cyp >= NPY) continue;
        float pd = depth[(size_t)(cyp * 16) * W + cxp * 16];
        if (pd == ORC_BACKGROUND_DEPTH) continue;
        float pp[3], pn[3];
        world_pos(g, pd, cxp * 16, cyp * 16, pp);
        for (int k = 0; k < 3; k++) v[k] = pp[k] - pos[k];
        normalize3(v);
        if (fabsf(dot3(v, nrm)) > 0.01f) {
            w[i] = 0.0f;
        } else {
            float q = 1.0f - fabsf(pd - pixel_depth) / pixel_depth;
            q = fminx(fmaxx(q, 0.0f), 1.0f);
            orc_unpack_normal_11_10_11(gbuffer[4 * ((size_t)(cyp * 16) * W + cxp * 16) + 1], pn);
            q *= fmaxx(dot3(nrm, pn), 0.0f);
            w[i] = pow8(q);
        }
    }
    if (w[0] * w[0] + w[1] * w[1] + w[2] * w[2] + w[3] * w[3] == 0.0f) { /* :72-76 */
        *fx = qx; *fy = qy;
        return 2;
    }
    float wsum = w[0] + w[1] + w[2] + w[3];
    for (int i = 0; i < 4; i++) w[i] /= wsum;
    float acc[3] = {0, 0, 0};
    for (uint32_t i = 0; i < 4; i++) { /* :81-97 */
        if (w[i] == 0.0f) continue; /* [rule] adds nothing; also keeps out-of-range probes unread */
        uint32_t cxp = lpx + (i & 1), cyp = lpy + (i >> 1);
        float pr[3];
        if (g->proberng == 1) {
            float pn[3];
            orc_unpack_normal_11_10_11(gbuffer[4 * ((size_t)(cyp * 16) * W + cxp * 16) + 1], pn);
            for (int k = 0; k < 3; k++) pr[k] = (pn[k] + 1.0f) / 2.0f;
        } else {
            float lobe[9];
            sh3_transform_cos_lobe(nrm, lobe);
            for (uint32_t c = 0; c < 3; c++) {
                const float *e = sh + 12 * (size_t)orc_zcurve(cxp * 3 + c, cyp);
                float m[9] = {e[0], e[1], e[2], e[4], e[5], e[6], e[8], e[9], e[10]};
                pr[c] = matrix_dot(m, lobe);
            }
        }
        for (int k = 0; k < 3; k++) acc[k] += w[i] * fmaxx(0.0f, pr[k]);
    }
    for (int k = 0; k < 3; k++) rad[k] = acc[k] * (surf[k] * F_FRAC_1_PI) + surf[3 + k]; /* :99-100 */
    return 1;
}
void orc_pass_interpolate_probes(const orc_gconst *g, const uint32_t *gbuffer, const float *depth, const float *sh, float *light) {
    uint32_t W = (uint32_t)g->window_size[0], H = (uint32_t)g->window_size[1];
    for (int phase = 1; phase <= 2; phase++) /* [rule] the "interpolation failed" marks (:74) land after the regular stores (:102) */
        for (uint32_t py = 0; py < H; py++)
            for (uint32_t px = 0; px < W; px++) {
                float rad[3];
                uint32_t fx = 0, fy = 0;
                int r = interpolate_pixel(g, gbuffer, depth, sh, px, py, rad, &fx, &fy);
                if (r != phase) continue;
                float *o = light + 4 * (r == 1 ? (size_t)py * W + px : (size_t)fy * W + fx);
                o[0] = r == 1 ? rad[0] : 1.0f; o[1] = r == 1 ? rad[1] : 0.0f; o[2] = r == 1 ? rad[2] : 0.0f; o[3] = 1.0f;
            }
}
