/*
 * rt3.h -- C ABI of librt3.so: the MI355X (gfx950) wavefront path tracer that stands in for the reference's
 * Vulkan ray-tracing passes.  Plain pointers and sizes only; every entry point cites the reference interface it
 * replaces (paths relative to DerEchteKarsten/RayTracer3).  INTEGRATION.md shows the Rust `extern "C"` block a
 * maintainer would add on the reference side.
 *
 * Conventions:
 *  - every call returns 0 on success or a negative RT3_E_* code; rt3_last_error() gives the text; nothing aborts or
 *    throws across the ABI (the reference `.unwrap()`s on its frame path, render_graph/mod.rs:601-610);
 *  - a context is NOT re-entrant: call it from one thread at a time (the reference's renderer systems are chained on
 *    the main thread, renderer/mod.rs:108-116); one context drives one GPU on one HIP stream (multi-GPU: one context
 *    and one process per GPU, joined only by rt3_gather_tiles);
 *  - host pointers are borrowed for the duration of the call and copied synchronously (like DynamicBuffer::push,
 *    vulkan/buffer.rs:406-420); device memory is owned by the context and released by rt3_destroy();
 *  - resource handles are u32 `tag << 30 | index` exactly like DescriptorResourceHandle (bindless/mod.rs:67-77):
 *    tag 0 = storage buffer, 1 = storage image, 3 = acceleration structure.
 */
#ifndef RT3_H
#define RT3_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT3_OK 0
#define RT3_E_INVALID (-1)     /* bad argument / unknown pass / wrong binding count */
#define RT3_E_HIP (-2)         /* a HIP runtime call failed (text in rt3_last_error) */
#define RT3_E_NO_DEVICE (-3)   /* no gfx950 device visible */
#define RT3_E_STATE (-4)       /* call order (e.g. pass launched before rt3_accel_build) */
#define RT3_E_UNSUPPORTED (-5) /* a feature this build does not implement */
#define RT3_E_DEPTH (-6)       /* BVH deeper than the traversal stack supports */
#define RT3_E_COMM (-7)        /* an RCCL call of the frame-end gather failed (text in rt3_last_error) */

#define RT3_INVALID_HANDLE 0xFFFFFFFFu
#define RT3_TAG_BUFFER 0u
#define RT3_TAG_IMAGE 1u
#define RT3_TAG_ACCEL 3u
#define RT3_MISS 0xFFFFFFFFu
#define RT3_BACKGROUND_DEPTH 100000.0f /* shaders/include/datatypes.slang:3 */

/* image formats (subset of the vk::Format values the reference passes use) */
#define RT3_FORMAT_R32_SFLOAT 100u          /* depth / gbuffer_depth (renderer/mod.rs:80, gbuffer.slang:6) */
#define RT3_FORMAT_R32G32B32A32_SFLOAT 109u /* Light / PrevLight / color (refrence_mode.slang:10-11) */
#define RT3_FORMAT_R32G32B32A32_UINT 107u   /* packed G-buffer (gbuffer.slang:5) */
#define RT3_FORMAT_R8G8B8A8_UNORM 37u       /* display image */
#define RT3_FORMAT_R16_UINT 74u             /* probe ray directions (trace_probes.slang:10, structured_importance_sampling.slang:10) */

/* feature flags carried in GConst.pad[0]; 0 = the reference's estimator (diffuse BSDF, emissive-only transport, 2 random draws
 * per bounce, refrence_mode.slang:36-57) with ONE documented difference: the reference advances its RNG counter sequentially
 * (random.slang:49-79), a wavefront needs a closed form and uses counter = (sample * bounces + bounce) * 2 + dim -- the same
 * numbers whenever no path of the pixel ends early (a closed box), other samples of the same distribution otherwise
 * (DESIGN.md section 4 item 1).  The other flags are north_star additions. */
#define RT3_F_NEE_SKY 1u     /* next-event estimation + MIS against the equirect sky */
#define RT3_F_BLUENOISE 2u   /* Cranley-Patterson shift by resources/bluenoise.png */
#define RT3_F_SPECULAR 4u    /* layered BSDF: DiffuseBrdf under the GGX SpecularBrdf of brdf.slang:141-311 */
#define RT3_F_FACEFORWARD 8u /* flip the shading normal towards the incoming ray */
#define RT3_F_PROBE_RADIANCE 16u /* trace_probes: store lerp(prev, radiance, blendfactor) per ray, the store its line 74 keeps in a comment */

/* src/renderer/mod.rs:47-63 == shaders/include/datatypes.slang:28-43.  304 bytes, 16-byte aligned, column-major
 * matrices.  Offsets 0,64,128,192,256,264,268,272,276,280,284,288,296. */
typedef struct rt3_gconst {
    float proj[16];
    float view[16];
    float proj_inverse[16];
    float view_inverse[16];
    float window_size[2];
    uint32_t frame;
    float blendfactor;
    uint32_t bounces;
    uint32_t samples;
    uint32_t proberng;
    float cell_size;
    uint32_t mouse[2];
    uint32_t pad[2]; /* pad[0] = RT3_F_* flags, pad[1] reserved (0) */
} rt3_gconst;

/* shaders/include/datatypes.slang:11-19 (52 bytes of fields, padded to 64 for float4 alignment) */
typedef struct rt3_geometry_info {
    float base_color[4];
    int32_t base_color_texture_index; /* -1 = none, else an index set with rt3_scene_set_texture (hit_logic.slang:31-33) */
    float metallic_factor;
    uint32_t index_offset;
    uint32_t vertex_offset;
    float emission[4];
    float roughness;
    uint32_t _pad[3];
} rt3_geometry_info;

/* One placed mesh of the world: `Instance{model}` + `Transform{Mat4}` of an entity (src/renderer/world/mod.rs:46-60), the rows
 * `InstanceInfo{mesh_index, transform}` of the global instance / transform buffers (world/mod.rs:34-38,104-125).  A mesh here is
 * a run of geometries [geometry_first, geometry_first + geometry_count) of rt3_scene_set_geometry; `transform` is column-major
 * like glam's Mat4 (object -> world), last row (0, 0, 0, 1) -- the 3 x 4 a VkAccelerationStructureInstanceKHR can hold. */
typedef struct rt3_instance {
    uint32_t geometry_first, geometry_count;
    float transform[16];
} rt3_instance;

/* frame statistics (no reference equivalent: the reference has no counters) */
typedef struct rt3_stats {
    uint64_t extension_rays; /* closest-hit rays traced since rt3_stats_reset (primary + bounce) */
    uint64_t shadow_rays;    /* any-hit rays traced */
    uint64_t nodes_visited;  /* by k_extend; only when counting is enabled (RT3_OPT_COUNT_TRAVERSAL) */
    uint64_t tris_tested;
    uint64_t shadow_nodes_visited; /* by k_shadow */
    uint64_t shadow_tris_tested;
    uint64_t extend_launches; /* k_extend launches */
    double extend_ms;         /* sum of HIP-event durations of k_extend launches (RT3_OPT_PROFILE) */
    uint64_t shadow_launches;
    double shadow_ms;
    double shade_ms;
    double other_ms;
    /* k_trace = one launch per bounce that drains the extension queue and then the shadow queue (the frame's dominant
     * kernel); its rays are ALSO included in extension_rays / shadow_rays and its visits in the four totals above */
    uint64_t trace_launches;
    double trace_ms;
    uint64_t trace_rays[2];  /* counting mode: [0] closest-hit, [1] any-hit rays traced by k_trace launches */
    uint64_t trace_nodes[2];
    uint64_t trace_tris[2];
    double gather_ms; /* RCCL send / grouped receives of rt3_gather_tiles (the pack / untile kernels are in other_ms) */
    uint64_t nodes_visited_lds;        /* of nodes_visited: served by the traversal kernels' LDS copy of the top of the tree, i.e. NOT */
    uint64_t shadow_nodes_visited_lds; /* requested from the vector-memory path (counting mode; default node layout only) */
    double accel_build_ms;       /* host wall clock of the last rt3_accel_build, stream synchronised on both sides */
    uint64_t accel_bulk_copies;  /* host <-> device copies of array size (> 64 KiB) made by rt3_accel_build calls since rt3_stats_reset:
                                    0 on the default path (the build stays on the GPU; a few KiB of per-geometry tables go up) */
} rt3_stats;

typedef struct rt3_ctx rt3_ctx;

/* ---- context: Context::new + RayTracingContext::new + BindlessDescriptorHeap::new + RenderGraph::new
 *      (renderer/mod.rs:32-45, vulkan/mod.rs:86-231) -> hipSetDevice + one stream ---- */
int rt3_create(int device, rt3_ctx **out);
void rt3_destroy(rt3_ctx *ctx);
const char *rt3_last_error(rt3_ctx *ctx); /* ctx may be NULL: last creation error */
int rt3_device_name(rt3_ctx *ctx, char *buf, size_t buf_size);

#define RT3_OPT_BATCH_SPP 1       /* samples per wavefront batch (0 = auto) */
#define RT3_OPT_PROFILE 2         /* 1: bracket kernels with HIP events on the context's stream */
#define RT3_OPT_COUNT_TRAVERSAL 3 /* 1: k_extend/k_shadow also count nodes / triangles (slower; for roofline bytes) */
#define RT3_OPT_EXTEND_VARIANT 4  /* traversal tuning: idle lanes of a wave before it refills them from its ray pool (default 12) */
#define RT3_OPT_LEAF_SIZE 5       /* 1..8 triangles per BVH leaf (default 2); takes effect at the next rt3_accel_build */
#define RT3_OPT_NODE_WIDTH 6      /* 2 = binary nodes, 4 = four-wide nodes (default); next rt3_accel_build */
#define RT3_OPT_NODE_QUANT 7      /* width 4 only: 1 = 64 B nodes with 8-bit conservative child boxes (default), 0 = 128 B fp32 boxes, 2 = compact 48 B nodes (implied references) */
#define RT3_OPT_WIDE_COLLAPSE 8   /* width 4 only: how the binary tree becomes four-wide nodes: 2 = cost-driven (default since round 3: a bottom-up SAH dynamic
                                     programme also decides which subtrees become multi-triangle leaves; triangle records in tree order), 1 = greedy by surface
                                     area, 0 = even binary depth */
#define RT3_OPT_POOL_CHUNK 9      /* traversal tuning: rays a wave takes from the launch's ray pool per grab (default 256) */
#define RT3_OPT_FUSED_TRACE 10    /* 1: one k_trace launch per bounce walks the extension queue and then the shadow queue; 0 (default): separate k_shadow and k_extend launches */
#define RT3_OPT_SAH_TOP 11       /* T > 0 (default 1 = binned SAH down to single triangles; collapse 0 / 1 use max(T, leaf size)): the tree above Karras subtrees of at most T triangles is re-linked by binned SAH
                                     (the reference asks its driver for PREFER_FAST_TRACE builds, raytracing.rs:103,131); 0 = plain LBVH */
#define RT3_OPT_TRACE_BLOCKS 12   /* traversal tuning: persistent workgroups (256 threads) per traversal launch (default 2048 = 8 per CU) */
#define RT3_OPT_SAH_TOP_DEVICE 13  /* 1 (default): the SAH top is built on the GPU, no bulk copies; 0: on the host (same tree, bit for bit) */
int rt3_set_option(rt3_ctx *ctx, int option, int64_t value);

/* ---- scene upload: DynamicBuffer::push (vulkan/buffer.rs:406-420) into the world buffers of
 *      world/mod.rs:103-125 (vertex / index / geometry), plus the two north_star inputs ---- */
int rt3_scene_set_vertices(rt3_ctx *ctx, const float *interleaved_p_n_t, uint32_t n_vertices); /* Vertex, assets/mod.rs:127-133 */
int rt3_scene_set_indices(rt3_ctx *ctx, const uint32_t *indices, uint32_t n_indices);
int rt3_scene_set_geometry(rt3_ctx *ctx, const rt3_geometry_info *infos, const uint32_t *prim_counts, uint32_t n);
/* equirect sky (main.rs:94, the commented skybox2.exr): finite, non-negative radiance.  Stored as RGB9E5 (packing.slang:99-162, the
 * format the reference's G-buffer keeps emissive in; values above 65408 clamp): every sky lookup reads the de-quantised texel */
int rt3_scene_set_sky(rt3_ctx *ctx, const float *rgb, uint32_t width, uint32_t height);
int rt3_scene_set_bluenoise(rt3_ctx *ctx, const uint8_t *rgba, uint32_t width, uint32_t height); /* resources/bluenoise.png */
/* base-colour texture `index` (dense indices 0..n-1): RGBA8 with sRGB-encoded colour, sampled bilinearly with repeat
 * addressing at mip 0 like Textures[i].SampleLevel(uvs, 0.0) (hit_logic.slang:31-33; bindless set 2, bindless/mod.rs:38-77) */
int rt3_scene_set_texture(rt3_ctx *ctx, uint32_t index, const uint8_t *rgba_srgb, uint32_t width, uint32_t height);

/* ---- instances: the reference's world is a list of placed meshes (add_instance / loaded_assets, world/mod.rs:50-101) under a
 *      top-level acceleration structure (create_acceleration_structure(.., level, ..), vulkan/raytracing.rs:88-148), and hit_info
 *      turns the shading normal by the instance matrix (hit_logic.slang:23).  Here: n = 0 (the default) places every geometry once
 *      under the identity.  Otherwise instance i places geometries [first, first + count) under its matrix; the same geometry may be
 *      placed many times.  There is no separate top level: rt3_accel_build FLATTENS the instances into world-space triangles
 *      (p' = ((x_axis x + y_axis y) + z_axis z) + w_axis in fp32, glam's transform_point3; identity matrices leave positions
 *      untouched) and builds one tree over them -- the build takes ~3 ms for 260 k triangles and stays on the GPU, so re-building
 *      after an instance moved IS the TLAS update (rt3_stats.accel_build_ms).  Primitive ids reported by hits (gbuffer, rt3_trace_rays)
 *      count through the placed geometries in instance order.  Normals: normalize(M3 * normalize(interpolated)), M3 = upper 3 x 3,
 *      as hit_logic.slang:22-23 writes it (no inverse transpose).  Call before rt3_accel_build; borrowed for the call. ---- */
int rt3_scene_set_instances(rt3_ctx *ctx, const rt3_instance *instances, uint32_t n);

/* ---- acceleration structure: create_acceleration_structure (vulkan/raytracing.rs:88-148) -> GPU LBVH.
 *      Returns the handle (tag 3) in *out_handle, like the TLAS registered at bindless/mod.rs:314-337 ---- */
int rt3_accel_build(rt3_ctx *ctx, uint32_t *out_handle);
/* introspection for parity tests: copy the BVH to the host (nodes: n_nodes x node_bytes (64 | 128), tris: n_tris x 48 B) */
int rt3_accel_info(rt3_ctx *ctx, uint32_t *n_nodes, uint32_t *n_tris, uint32_t *max_depth, uint32_t *node_bytes);
int rt3_accel_download(rt3_ctx *ctx, void *nodes, size_t nodes_bytes, void *tris, size_t tris_bytes);
/* the way back: install a tree built elsewhere over the same (flattened) triangles -- a better offline builder, a cache of an earlier run
 * (what vkCmdCopyMemoryToAccelerationStructureKHR is to the reference's driver).  Default layout only (64-byte nodes, 48-byte triangle
 * records, rt3_accel_download's format); call rt3_accel_build first (it makes the shading records).  Every reference is validated on
 * the host (range, no node reachable twice, depth) before a kernel may follow it.  Triangle records may repeat a primitive (spatial
 * splits): the closest hit is decided by (t, prim), not by the record.  Used by tests/experiments/tree_quality_gpu.py. */
int rt3_accel_import(rt3_ctx *ctx, const void *nodes, size_t nodes_bytes, const void *tris, size_t tris_bytes);
/* sky tables for parity tests (any pointer may be NULL): per-row alias words q16 | alias << 16 (w*h), RGB9E5 texels (w*h),
 * marginal CDF (h), realised (u,v) density (w*h) */
int rt3_sky_download(rt3_ctx *ctx, uint32_t *alias, uint32_t *texels_rgb9e5, float *cdf_marg, float *pdf_uv);

/* ---- resources: RenderGraph::image / buffer / import (render_graph/mod.rs:422-483) ---- */
int rt3_buffer_create(rt3_ctx *ctx, size_t bytes, uint32_t *out_handle);
int rt3_image_create(rt3_ctx *ctx, uint32_t width, uint32_t height, uint32_t format, uint32_t *out_handle);
int rt3_image_import(rt3_ctx *ctx, void *device_ptr, uint32_t width, uint32_t height, uint32_t format, uint32_t *out_handle);
int rt3_resource_upload(rt3_ctx *ctx, uint32_t handle, const void *src, size_t bytes);
int rt3_resource_download(rt3_ctx *ctx, uint32_t handle, void *dst, size_t bytes); /* headless stand-in for present */
int rt3_resource_device_ptr(rt3_ctx *ctx, uint32_t handle, void **out_ptr, size_t *out_bytes);

/* ---- framebuffer partition (north_star): 64x64 tiles, Z-order over the tile grid, tile i -> rank i % n_ranks.
 *      Passes only touch the pixels owned by `rank`.  Default: rank 0 of 1. ---- */
int rt3_set_tile_partition(rt3_ctx *ctx, uint32_t width, uint32_t height, uint32_t rank, uint32_t n_ranks);
int rt3_tile_pixel_count(rt3_ctx *ctx, uint32_t rank, uint32_t n_ranks, uint32_t *out_count);
/* gather support: image (full window) <-> contiguous per-rank tile buffer (count x 16 bytes, device memory) */
int rt3_image_pack_tiles(rt3_ctx *ctx, uint32_t image, uint32_t rank, uint32_t n_ranks, void *dst_device);
int rt3_image_unpack_tiles(rt3_ctx *ctx, uint32_t image, uint32_t rank, uint32_t n_ranks, const void *src_device);

/* ---- frame-end gather (north_star: "the framebuffer is tile-partitioned across the 8 GPUs of one node with a single RCCL gather
 *      over xGMI at frame end").  No reference counterpart: the reference is single-device.  One context =
 *      one rank = one GPU = one process.  Rank 0 makes an id with rt3_comm_unique_id and the HOST carries its 128 bytes to the
 *      other ranks over whatever channel it already has (the ABI opens no sockets); then every rank calls rt3_comm_init
 *      (collective: ncclCommInitRank on the context's device) with the rank / n_ranks it gave rt3_set_tile_partition.
 *      rt3_gather_tiles is the one collective of a frame: enqueued on the context's stream behind the passes, no host
 *      synchronisation.  Non-root ranks pack their tiles of `image` and send them; `root` receives every rank's tiles at its exact
 *      offset of ONE contiguous buffer (all receives in one RCCL group: the root's inbound xGMI links run concurrently, nothing is
 *      forwarded) and scatters them into its `image` with ONE untile launch.  The root's own tiles never move.
 *      rt3_gather_layout / rt3_gather_unpack expose the root's half without the exchange (hosts that move the bytes themselves --
 *      the gloo rehearsal on a one-GPU box -- and the layout tests): offsets[r] .. offsets[r+1] is rank r's pixel range in the
 *      receive buffer (16 bytes per pixel, pixels in rt3_image_pack_tiles order, the root's range empty), n_ranks + 1 entries. ---- */
#define RT3_COMM_ID_BYTES 128
int rt3_comm_version(int *out); /* ncclGetVersion of the RCCL this library is linked against: major * 10000 + minor * 100 + patch */
int rt3_comm_unique_id(void *id_out /* RT3_COMM_ID_BYTES */);
int rt3_comm_init(rt3_ctx *ctx, const void *id /* RT3_COMM_ID_BYTES */, uint32_t rank, uint32_t n_ranks);
int rt3_comm_destroy(rt3_ctx *ctx);
int rt3_gather_tiles(rt3_ctx *ctx, uint32_t image, uint32_t root);
int rt3_gather_layout(rt3_ctx *ctx, uint32_t image, uint32_t root, uint32_t n_ranks, uint64_t *offsets /* n_ranks + 1 */);
int rt3_gather_unpack(rt3_ctx *ctx, uint32_t image, uint32_t root, uint32_t n_ranks, const void *recv_device);

/* ---- pass launch: ExecutionTrait::execute (render_graph/mod.rs:80-91) of a RayTracingPass / ComputePass node
 *      (render_graph/executions.rs:15-55,80-121) -> RayTracingPipelineHandle::launch(x, y) /
 *      ComputePipelineHandle::dispatch(x, y, z) (pipeline_cache/mod.rs:24-76).
 *      `pass_name` is the shader path the reference would load from ./shaders/bin/{path}.slang.spv
 *      (pipeline_cache/mod.rs:278-279); `constants` is the raw GConst blob (build.rs:66-94); `bindings` is the ordered
 *      handle list of the node's non-attachment edges (bake.rs:51-83):
 *        "gbuffer"       (x,y)=window   bindings {gbuffer RGBA32UI, gbuffer_depth R32F}                (gbuffer.slang:5-6)
 *        "refrence_mode" (x,y)=window   bindings {gbuffer, gbuffer_depth, Light, PrevLight}            (refrence_mode.slang:8-11)
 *        "postprocess"   (x,y,z)=groups of 8x8  bindings {Depth, Out RGBA32F, In RGBA32F}             (postprocess.slang:5-7)
 *      and the probe-GI passes (restated as written, debug stores included; rules for what the text leaves open are listed in
 *      DESIGN.md section 11).  A probe owns 16x16 pixels and an 8x8-texel cell of the atlas images; bindings are ordered by
 *      (descriptor set, binding) as the shaders declare them:
 *        "structured_importance_sampling" (x,y,1)=probes  {gbuffer, gbuffer_depth, out R16UI, debug R32F, probe_atlas RGBA32F}
 *        "trace_probes"                   (x,y)=atlas size {gbuffer, gbuffer_depth, directions R16UI, probe_atlas, prev_probe_atlas}
 *        "spherical_harmonic_conversion"  (x,y,1)=probes  {out buffer of float3x3 (48 B, rows padded to float4), probe_atlas}
 *        "interpolate_probes"             (x,y,1)=groups of 8x8 over the window  {gbuffer, gbuffer_depth, sh_coeficents buffer, Light}
 *      Work is enqueued on the context's stream and returns immediately. ---- */
int rt3_pass_launch(rt3_ctx *ctx, const char *pass_name, const char *entry, uint32_t x, uint32_t y, uint32_t z,
                    const void *constants, size_t constants_size, const uint32_t *bindings, uint32_t n_bindings);
/* timeline-semaphore wait of begin_frame (render_graph/mod.rs:656-665) -> hipStreamSynchronize */
int rt3_frame_wait(rt3_ctx *ctx);

/* ---- traversal on a caller-supplied ray batch (`trace()` call sites gbuffer.slang:13, refrence_mode.slang:54):
 *      rays = 8 SoA arrays of n floats (ox,oy,oz,dx,dy,dz,tmin,tmax) in HOST memory; results to host.
 *      any_hit = 0: closest hit -> t,u,v,prim ; any_hit = 1: prim[i] = 1 if occluded else 0 (t,u,v untouched).
 *      n_nodes / n_tris (may be NULL) receive per-ray traversal counts.  `repeat` > 1 re-launches the kernel for timing;
 *      *kernel_ms (may be NULL) receives the average HIP-event duration of one launch. ---- */
int rt3_trace_rays(rt3_ctx *ctx, const float *rays, uint32_t n, int any_hit, float *t, float *u, float *v, uint32_t *prim,
                   uint32_t *n_nodes, uint32_t *n_tris, int repeat, double *kernel_ms);

/* ---- device self-test: evaluates one device function per element so known-answer tests can pin the GPU arithmetic.
 *      op: 0 hash(u32) 1 zcurve(x,y) 2 murmur3(seed,index) 3 uniform_float(seed,index) 4 gbuffer pack (11 f32 -> 4 u32)
 *      5 gbuffer unpack (4 u32 -> 11 f32) 6 diffuse sample (u0,u1 -> wi) 7 orthonormal basis (n -> b1,b2) 8 AgX (rgb -> rgb)
 *      9 sincos_2pi (u -> sin,cos) 10 atan2 (y,x) 11 rng_seed(px,py,frame)
 *      12 division-free integer helpers (n,d -> n/d, n%d, wrap(int(n), (d & 0xFFFF)+1))
 *      13 octa_decode (fx,fy -> n) 14 sh3Evaluate (dir -> 9 coefficients) 15 64-lane bitonic sort (64 keys -> 64 keys, 64 lane ids)
 *      16 64-lane sum (64 floats -> 1) 17 octa_encode16 (n -> the 2 x 16-bit word of a shading-record normal) 18 octa_decode16 (word -> n).
 *      in/out: host arrays of 32-bit words. ---- */
int rt3_selftest_eval(rt3_ctx *ctx, int op, const void *in, uint32_t n, void *out);

int rt3_stats_reset(rt3_ctx *ctx);
int rt3_stats_get(rt3_ctx *ctx, rt3_stats *out); /* synchronises the stream */

/* ---- host helpers mirroring Camera::view_matrix / projection_matrix (components/camera.rs:52-58) and the GConst fill
 *      of renderer::commands (renderer/mod.rs:72-78) ---- */
void rt3_camera_gconst(const float position[3], const float direction[3], float fov_y_radians, float aspect,
                       float z_near, float z_far, float width, float height, rt3_gconst *out);

#ifdef __cplusplus
}
#endif
#endif
