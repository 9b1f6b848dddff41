/*
 * rt3_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Plain-C restatement of the reference path tracer semantics written in
 * /root/reference/shaders/old/{gbuffer,refrence_mode,default_hit,default_miss,postprocess}.slang and
 * /root/reference/shaders/include/{random,math,packing,brdf,gbuffer_helpers,hit_logic,datatypes}.slang,
 * plus the north_star additions (LBVH, Moeller-Trumbore, sky NEE, blue-noise shift) in the exact arithmetic
 * the HIP product uses, so GPU-vs-oracle comparisons are meaningful sample by sample.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Parity pinning: the reference has NO tests / golden vectors.  Integer functions are pinned by the
 * known-answer values of SURVEY.md section 8c (tests/test_oracle_kat.py); the BVH / triangle test is pinned against the
 * double-precision brute-force intersector below; everything that rests on the closed Vulkan driver
 * (VK_KHR_acceleration_structure) or on glam is "parity unpinned" (see DESIGN.md).
 */
#ifndef RT3_ORACLE_H
#define RT3_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MISS 0xFFFFFFFFu
#define ORC_BACKGROUND_DEPTH 100000.0f /* datatypes.slang:3 */

/* feature flags carried in GConst.pad[0]; 0 == reference semantics (diffuse, emissive-only, 2 draws/bounce) */
#define ORC_F_NEE_SKY 1u
#define ORC_F_BLUENOISE 2u
#define ORC_F_SPECULAR 4u
#define ORC_F_FACEFORWARD 8u
#define ORC_F_PROBE_RADIANCE 16u /* trace_probes stores the blended radiance its line 74 keeps in a comment (0: the debug store as written) */

/* src/renderer/mod.rs:47-63 == datatypes.slang:28-43; 304 bytes, column-major matrices */
typedef struct orc_gconst {
    float proj[16], view[16], proj_inverse[16], view_inverse[16];
    float window_size[2];
    uint32_t frame;
    float blendfactor;
    uint32_t bounces, samples, proberng;
    float cell_size;
    uint32_t mouse[2];
    uint32_t pad[2]; /* pad[0] = feature flags (build extension), pad[1] reserved */
} orc_gconst;

/* datatypes.slang:11-19 (52 B of payload, padded to 64) */
typedef struct orc_geometry_info {
    float base_color[4];
    int32_t base_color_texture_index;
    float metallic_factor;
    uint32_t index_offset;
    uint32_t vertex_offset;
    float emission[4];
    float roughness;
    uint32_t _pad[3];
} orc_geometry_info;

typedef struct orc_scene orc_scene;

/* ---- integer KATs: random.slang:5-15,49-89, math.slang:105-117 ---- */
uint32_t orc_hash(uint32_t a);
uint32_t orc_zcurve(uint32_t x, uint32_t y);
uint32_t orc_rng_seed(uint32_t px, uint32_t py, uint32_t frame);
uint32_t orc_murmur3(uint32_t seed, uint32_t index);
float orc_uniform_float(uint32_t seed, uint32_t index);
uint32_t orc_radical_inverse_bits(uint32_t bits);

/* ---- packing.slang ---- */
uint32_t orc_pack_color_888(const float c[3]);
void orc_unpack_color_888(uint32_t p, float c[3]);
uint32_t orc_pack_normal_11_10_11(const float n[3]);
void orc_unpack_normal_11_10_11(uint32_t p, float n[3]);
uint32_t orc_pack_2x16f(float a, float b);
void orc_unpack_2x16f(uint32_t u, float out[2]);
uint32_t orc_float3_to_rgb9e5(const float c[3]);
void orc_rgb9e5_to_float3(uint32_t v, float c[3]);
/* gbuffer_helpers.slang:22-34,59-70 : in/out = albedo[3] emissive[3] normal[3] roughness metalness (11 floats) */
void orc_gbuffer_pack(const float surf[11], uint32_t out[4]);
void orc_gbuffer_unpack(const uint32_t in[4], float surf[11]);

/* ---- math ---- */
void orc_sincos_2pi(float u, float *s, float *c);
float orc_atan2(float y, float x);
void orc_onb(const float n[3], float b1[3], float b2[3]);           /* math.slang:29-50 */
void orc_diffuse_sample(float u0, float u1, float wi[3]);           /* brdf.slang:56-65 */
void orc_dir_to_equirect_uv(const float d[3], float uv[2]);         /* math.slang:6-12 */
void orc_agx_tonemap(const float in[3], float out[3]);              /* postprocess.slang:13-88 */

/* ---- camera: camera.rs:52-58, renderer/mod.rs:72-78 ---- */
void orc_camera_gconst(const float pos[3], const float dir[3], float fov_y, float aspect, float z_near, float z_far,
                       float width, float height, orc_gconst *out);
void orc_primary_ray(const orc_gconst *g, uint32_t px, uint32_t py, float o[3], float d[3]); /* gbuffer_helpers.slang:85-103 */
void orc_primary_rays(const orc_gconst *g, const uint32_t *xs, const uint32_t *ys, uint32_t n, float tmin, float tmax, float *rays); /* the same, n pixels: 8 SoA arrays */

/* Instance{model} + Transform{Mat4} of the reference's world (src/renderer/world/mod.rs:34-60); == rt3_instance */
typedef struct orc_instance {
    uint32_t geometry_first, geometry_count;
    float transform[16]; /* column-major, object -> world, last row (0, 0, 0, 1) */
} orc_instance;

/* ---- scene ---- */
orc_scene *orc_scene_create(void);
void orc_scene_destroy(orc_scene *s);
/* interleaved p(3) n(3) t(2) == assets/mod.rs:127-133 */
int orc_scene_set_vertices(orc_scene *s, const float *pnt, uint32_t n_vertices);
int orc_scene_set_indices(orc_scene *s, const uint32_t *idx, uint32_t n_indices);
int orc_scene_set_geometry(orc_scene *s, const orc_geometry_info *g, const uint32_t *prim_counts, uint32_t n);
int orc_scene_set_instances(orc_scene *s, const orc_instance *inst, uint32_t n); /* n = 0: every geometry once, identity */
int orc_scene_set_sky(orc_scene *s, const float *rgb, uint32_t w, uint32_t h);
int orc_scene_set_bluenoise(orc_scene *s, const uint8_t *rgba, uint32_t w, uint32_t h);
int orc_scene_set_texture(orc_scene *s, uint32_t index, const uint8_t *rgba_srgb, uint32_t w, uint32_t h); /* hit_logic.slang:31-33 */
/* LBVH (Karras 2012) over all triangles; replaces raytracing.rs:88-148 */
/* leaf_max 1..8 triangles per leaf (default 2); node_width 2 = 64 B binary nodes, 4 = four-wide nodes (default);
 * quantized (width 4 only): 0 = 128 B fp32 boxes, 1 = 64 B nodes with 8-bit conservative child boxes and explicit
 * references, 2 = compact 48 B nodes (same boxes; references implied by node_base / tri_base + one nibble per child) */
void orc_accel_set_layout(orc_scene *s, uint32_t leaf_max, uint32_t node_width, uint32_t quantized);
/* four-wide collapse rule: 0 = even binary depth, 1 = surface-area greedy, 2 = cost-driven (bottom-up SAH dynamic programme that
 * also decides which subtrees become multi-triangle leaves; implies tree order) */
void orc_accel_set_collapse(orc_scene *s, uint32_t mode);
/* tree order: triangle records re-ordered depth-first along the final binary tree, so that every subtree is a contiguous range
 * (lets the SAH top go down to single triangles, cluster_size 1, with multi-triangle leaves formed above them) */
void orc_accel_set_tree_order(orc_scene *s, uint32_t on);
/* experiments only: weights of a node step / a triangle step in the cost-driven collapse (the product uses 1, 1) */
void orc_accel_set_dp_costs(orc_scene *s, float c_node, float c_tri);
/* experiments only (needs tree order / collapse 2): insertion-based re-optimisation of the tree above subtrees of <= k_top triangles */
void orc_accel_set_top_opt(orc_scene *s, uint32_t k_top, uint32_t passes);
void orc_accel_set_recull(orc_scene *s, uint32_t on); /* experiments only (default 0): closest-hit walks drop a popped node reference that lies beyond the current hit */
/* SAH top: the tree above Karras subtrees of at most cluster_size triangles is re-linked by binned SAH (0 = plain LBVH; default 2) */
void orc_accel_set_sah_top(orc_scene *s, uint32_t cluster_size);
uint32_t orc_accel_node_words(const orc_scene *s);
int orc_accel_build(orc_scene *s);
uint32_t orc_accel_num_tris(const orc_scene *s);
uint32_t orc_accel_num_nodes(const orc_scene *s);
const float *orc_accel_nodes(const orc_scene *s);  /* n_nodes x orc_accel_node_words() 32-bit words */
const float *orc_accel_tris(const orc_scene *s);   /* n_tris  x 12 words (48 B); Morton order, compact layout: leaf order */
const uint64_t *orc_accel_codes(const orc_scene *s);
uint32_t orc_accel_max_depth(const orc_scene *s);
/* sky tables (for parity checks against the product) */
void orc_tri_edge_functions(const float *v0, const float *v1, const float *v2, const float *o, const float *d, float out[3]);
const uint32_t *orc_sky_alias(const orc_scene *s);  /* w*h words: q16 | alias column << 16 */
const uint32_t *orc_sky_texels(const orc_scene *s); /* w*h RGB9E5 words */
const float *orc_sky_cdf_marg(const orc_scene *s);
const float *orc_sky_pdf_uv(const orc_scene *s);

/* ---- ray queries: rays SoA ox,oy,oz,dx,dy,dz,tmin,tmax (8 arrays of n floats, one block rays[8*n]) ---- */
/* hits: t,u,v (float) + prim (u32, global primitive id, ORC_MISS on miss). n_nodes / n_tris may be NULL. */
void orc_trace_closest(const orc_scene *s, const float *rays, uint32_t n, float *t, float *u, float *v, uint32_t *prim,
                       uint32_t *n_nodes, uint32_t *n_tris, int n_threads);
void orc_trace_any(const orc_scene *s, const float *rays, uint32_t n, uint32_t *occluded, uint32_t *n_nodes,
                   uint32_t *n_tris, int n_threads);
/* brute force over all triangles: fp32 test identical to the BVH path (mode 0) or double precision (mode 1) */
void orc_trace_brute(const orc_scene *s, const float *rays, uint32_t n, float *t, float *u, float *v, uint32_t *prim,
                     int mode, int n_threads);

/* hit_logic.slang:5-40 : out = albedo emissive normal roughness metalness (11 floats) */
void orc_hit_info(const orc_scene *s, uint32_t prim, float bu, float bv, float surf[11]);

/* ---- passes over a pixel rectangle [x0,x1) x [y0,y1) of the window in g->window_size.
 * All images are full-window row-major arrays. ---- */
/* gbuffer.slang:8-21 */
void orc_pass_gbuffer(const orc_scene *s, const orc_gconst *g, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                      uint32_t *gbuffer /*W*H*4*/, float *depth /*W*H*/, int n_threads);
/* refrence_mode.slang:14-66 ; light/prev_light RGBA32F. ray_counts[0]=extension rays, [1]=shadow rays,
 * [2]=BVH nodes visited, [3]=triangles tested (all rays), [4], [5] = the shadow rays' share of [2], [3] (6 words, may be NULL) */
void orc_pass_reference_mode(const orc_scene *s, const orc_gconst *g, uint32_t x0, uint32_t y0, uint32_t x1,
                             uint32_t y1, const uint32_t *gbuffer, const float *depth, const float *prev_light,
                             float *light, uint64_t *ray_counts, int n_threads);
/* the extension rays of (sample 0, bounce 0 -> 1) for every non-background pixel, 8 SoA arrays of `cap` floats; returns their number */
uint32_t orc_bounce1_rays(const orc_scene *s, const orc_gconst *g, const uint32_t *gbuffer, const float *depth, float *rays, uint32_t cap);
/* postprocess.slang:90-112 ; out RGBA32F display-referred */
void orc_pass_postprocess(const orc_scene *s, const orc_gconst *g, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                          const float *depth, const float *in, float *out, int n_threads);

/* ---- probe-GI passes (rt3_oracle_probes.c).  probes = 8x8-texel cells of the probe atlas, one per
 * 16x16 pixel block; atlas images are (8*probes_x) x (8*probes_y); W, H come from g->window_size. ---- */
void orc_octa_decode(float fx, float fy, float n[3]);           /* packing.slang:77-86 */
uint32_t orc_octa_encode16(const float n[3]);                   /* packing.slang:64-75 + 16-bit unorm per coordinate */
void orc_octa_decode16(uint32_t w, float n[3]);
void orc_sh3_evaluate(const float d[3], float sh[9]);           /* spherical_harmonics.slang:30-44, sh[r*3+c] */
void orc_wave_sort64(float keys[64], uint32_t idx[64]);         /* math.slang:140-160 over the 64 lanes of a wave */
float orc_wave_sum64(const float v[64]);                        /* WaveActiveSum, fixed butterfly order */
/* structured_importance_sampling.slang:19-71 : out = R16_UINT atlas, debug = R32F atlas */
void orc_pass_structured_importance_sampling(const orc_gconst *g, uint32_t probes_x, uint32_t probes_y, const uint32_t *gbuffer,
                                             uint16_t *out, float *debug);
/* trace_probes.slang:15-77 : atlas / prev_atlas RGBA32F */
void orc_pass_trace_probes(const orc_scene *s, const orc_gconst *g, uint32_t probes_x, uint32_t probes_y, const uint32_t *gbuffer,
                           const float *depth, const uint16_t *directions, const float *prev_atlas, float *atlas, int n_threads);
/* spherical_harmonic_conversion.slang:9-33 : sh_out[zcurve(3*gx + c, gy)] = 3 rows x float4 (std430 float3x3) */
void orc_pass_sh_conversion(uint32_t probes_x, uint32_t probes_y, const float *atlas, float *sh_out);
/* interpolate_probes.slang:11-103 */
void orc_pass_interpolate_probes(const orc_gconst *g, const uint32_t *gbuffer, const float *depth, const float *sh, float *light);

/* tile map: 64x64 tiles, Z-order over the tile grid, tile i -> rank i % n_ranks.
 * Returns number of pixels owned by `rank`; if out_xy != NULL writes (x,y) pairs in render order. */
uint32_t orc_tile_pixels(uint32_t w, uint32_t h, uint32_t rank, uint32_t n_ranks, uint32_t *out_xy);

#ifdef __cplusplus
}
#endif
#endif
