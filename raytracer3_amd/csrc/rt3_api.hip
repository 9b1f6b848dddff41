// rt3_api.hip -- host layer + C ABI of librt3.so (include/rt3.h).
//
// One context = one GPU + one HIP stream.  It owns the world buffers (world/mod.rs:103-125), the LBVH
// (raytracing.rs:88-148), the name-less resource table with bindless-style handles (bindless/mod.rs:67-77) and the
// wavefront work queues.  rt3_pass_launch() is the drop-in for executing one pass node of the reference's frame graph
// (render_graph/mod.rs:80-107): the pass name selects a HIP kernel sequence instead of a SPIR-V pipeline.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <string>
#include <vector>

#include "../../include/rt3.h"
#include "rt3_internal.hpp"

using namespace rt3;

static_assert(sizeof(rt3_gconst) == 304 && sizeof(GConstDev) == 304, "GConst is 304 bytes (renderer/mod.rs:47-63)");
static_assert(sizeof(rt3_geometry_info) == 64, "geometry info is 64 bytes");
static_assert(RT3_F_NEE_SKY == RT3_FLAG_NEE_SKY && RT3_F_BLUENOISE == RT3_FLAG_BLUENOISE && RT3_F_FACEFORWARD == RT3_FLAG_FACEFORWARD && RT3_F_SPECULAR == RT3_FLAG_SPECULAR && RT3_F_PROBE_RADIANCE == RT3_FLAG_PROBE_RADIANCE, "flags");

namespace {

thread_local std::string g_create_error;

struct Resource {
    uint32_t tag = 0;
    void* ptr = nullptr;
    size_t bytes = 0;
    uint32_t w = 0, h = 0, format = 0;
    bool owned = true;
};
struct PixelList {
    uint32_t w, h, rank, n_ranks, count;
    uint32_t* dev;
    uint2* dev_bn = nullptr;      // {x | y << 16, blue-noise word of that pixel}: one load instead of two dependent ones in k_shade
    uint64_t bn_stamp = ~0ull;    // which blue-noise upload dev_bn was built from
};
// Frame-end gather (north_star: "a single RCCL gather over xGMI at frame end").  The root receives every other rank's tiles
// into ONE contiguous buffer -- rank r's count[r] pixels at pixel offset off[r], ranks in ascending order, the root itself
// contributing nothing (its tiles are already in its image) -- and scatters all of them with ONE untile launch over `dev`,
// the concatenation of those ranks' pixel lists.
struct GatherLayout {
    uint32_t w, h, root, n_ranks;
    std::vector<uint64_t> off;  // n_ranks + 1 entries, in pixels
    uint32_t* dev = nullptr;    // off[n_ranks] pixel words (x | y << 16)
};
enum Cat { CAT_EXTEND = 0, CAT_SHADOW = 1, CAT_SHADE = 2, CAT_OTHER = 3, CAT_TRACE = 4, CAT_GATHER = 5 };
struct Timed {
    hipEvent_t a, b;
    int cat;
};
struct CounterBlock {  // device counters of one refrence_mode launch, harvested lazily
    uint32_t first, n_pairs;  // n_pairs x {extension-queue size, shadow-queue size}: one 8-byte pair per bounce (k_shade bumps both with ONE 64-bit atomic)
};

}  // namespace

struct rt3_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    char name[256] = {0};
    // scene
    float* d_verts = nullptr;
    uint32_t n_verts = 0;
    uint32_t* d_indices = nullptr;
    uint32_t n_indices = 0;
    FlatGeomDev* d_geoms = nullptr;          // one entry per (instance, geometry): built by rt3_accel_build (flatten_world)
    ShadeGeomDev* d_shade_geoms = nullptr;   // the same table as hit_info reads it
    uint32_t n_geoms = 0, n_prims = 0;       // uploaded geometries / their primitives (one instance of each)
    uint32_t n_flat_geoms = 0, n_flat_prims = 0;  // after flattening: what the acceleration structure and the shading records cover
    uint32_t *d_prim_geom = nullptr, *d_first_prim = nullptr;
    std::vector<rt3_instance> h_instances;   // empty = one identity instance of every geometry
    uint64_t bulk_copies = 0;                // host <-> device copies of more than 64 KiB made by rt3_accel_build (rt3_stats.accel_bulk_copies)
    uint2* d_sky = nullptr;  // 8-byte texels {RGB9E5, pdf_uv} in 4 x 4 tiles
    float* d_cdf_marg = nullptr;
    uint32_t *d_sky_alias = nullptr, *d_guide_marg = nullptr;
    uint32_t sky_w = 0, sky_h = 0, sky_wt = 0;
    uint8_t* d_bn = nullptr;
    uint32_t bn_w = 0, bn_h = 0;
    uint64_t bn_stamp = 0;  // bumped by every rt3_scene_set_bluenoise
    // base-colour textures: host staging (RGBA8) + device atlas rebuilt lazily
    std::vector<std::vector<uint8_t>> h_tex;
    std::vector<uint32_t> tex_w, tex_h;
    uint8_t* d_tex_pixels = nullptr;
    uint4* d_tex_table = nullptr;
    float* d_srgb_lut = nullptr;
    bool tex_dirty = false;
    LbvhResult bvh;
    BuildArena build_arena;
    bool accel_built = false;
    std::vector<uint32_t> h_indices;  // host copies, only for range validation (rt3_scene_set_geometry, again in rt3_accel_build)
    std::vector<rt3_geometry_info> h_geoms;
    std::vector<uint32_t> h_prim_counts;
    int64_t max_tex_index = -1;
    // resources
    std::vector<Resource> resources;
    std::vector<PixelList> pixlists;
    std::vector<GatherLayout> gather_layouts;
    uint32_t rank = 0, n_ranks = 1, part_w = 0, part_h = 0;
    // communicator of the frame-end gather (RCCL): one rank per context / GPU / process
    ncclComm_t comm = nullptr;
    uint32_t comm_rank = 0, comm_size = 0;
    void* gather_buf = nullptr;  // non-root: this rank's packed tiles; root: the receive buffer of all other ranks' tiles
    size_t gather_buf_bytes = 0;
    // work queues (capacity in paths)
    size_t cap = 0, cap_pix = 0;
    float *rays[2] = {nullptr, nullptr}, *hits = nullptr, *T[2] = {nullptr, nullptr};
    float *sh_rays = nullptr, *sh_contrib = nullptr, *lacc = nullptr, *radsum = nullptr;
    uint32_t* d_counters = nullptr;
    uint32_t counters_cap = 1 << 16, counters_next = 0;
    unsigned long long* d_totals = nullptr;
    std::vector<CounterBlock> pending_counters;
    // options / stats
    int64_t opt_batch_spp = 0;
    bool opt_profile = false, opt_count = false;
    int opt_variant = 0;  // RT3_OPT_EXTEND_VARIANT: reserved for traversal experiments
    uint32_t opt_leaf_size = 2, opt_node_width = 4, opt_node_quant = 1, opt_collapse = 2, opt_sah_top = 1, opt_sah_device = 1;
    int opt_fused_trace = 0;  // 1: k_trace (extension + shadow queue in one launch per bounce)
    rt3_stats stats;
    uint64_t primary_rays_pending = 0;
    std::vector<Timed> pending_events;
    std::vector<Timed> free_events;
};

namespace {

int fail(rt3_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    else g_create_error = msg;
    return code;
}
#define HIPC(ctx, call)                                                                                              \
    do {                                                                                                             \
        hipError_t e_ = (call);                                                                                      \
        if (e_ != hipSuccess) return fail(ctx, RT3_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_));        \
    } while (0)

template <typename T>
int dev_alloc(rt3_ctx* c, T** p, size_t count) {
    if (*p) {
        (void)hipFree(*p);
        *p = nullptr;
    }
    HIPC(c, hipMalloc((void**)p, (count ? count : 1) * sizeof(T)));
    return RT3_OK;
}
template <typename T>
void dev_free(T*& p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}

uint32_t spread1by1(uint32_t x) {  // math.slang:105-112 integer_explode
    x = (x | (x << 8)) & 0x00FF00FFu;
    x = (x | (x << 4)) & 0x0F0F0F0Fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
}
uint32_t zcurve_host(uint32_t x, uint32_t y) { return spread1by1(x) | (spread1by1(y) << 1); }  // math.slang:114-117
uint32_t compact1by1(uint32_t x) {
    x &= 0x55555555u;
    x = (x | (x >> 1)) & 0x33333333u;
    x = (x | (x >> 2)) & 0x0F0F0F0Fu;
    x = (x | (x >> 4)) & 0x00FF00FFu;
    x = (x | (x >> 8)) & 0x0000FFFFu;
    return x;
}
// 64x64 tiles, Z-order over the tile grid, tile i -> rank i % n_ranks; Z-order inside a tile (primary-ray coherence)
void tile_pixels(uint32_t w, uint32_t h, uint32_t rank, uint32_t n_ranks, std::vector<uint32_t>& out) {
    out.clear();
    uint32_t tw = (w + 63) / 64, th = (h + 63) / 64, side = 1, tile_no = 0;
    while (side < tw || side < th) side *= 2;
    for (uint32_t z = 0; z < side * side; z++) {
        uint32_t tx = compact1by1(z), ty = compact1by1(z >> 1);
        if (tx >= tw || ty >= th) continue;
        uint32_t owner = tile_no++ % n_ranks;
        if (owner != rank) continue;
        for (uint32_t k = 0; k < 4096; k++) {
            uint32_t x = tx * 64 + compact1by1(k), y = ty * 64 + compact1by1(k >> 1);
            if (x < w && y < h) out.push_back(x | (y << 16));
        }
    }
}
int get_pixlist(rt3_ctx* c, uint32_t w, uint32_t h, uint32_t rank, uint32_t n_ranks, PixelList** out) {
    for (auto& p : c->pixlists)
        if (p.w == w && p.h == h && p.rank == rank && p.n_ranks == n_ranks) {
            *out = &p;
            return RT3_OK;
        }
    if (w == 0 || h == 0 || w > 65535 || h > 65535 || n_ranks == 0 || rank >= n_ranks) return fail(c, RT3_E_INVALID, "bad window / rank for tile partition");
    std::vector<uint32_t> px;
    tile_pixels(w, h, rank, n_ranks, px);
    PixelList pl;
    pl.w = w; pl.h = h; pl.rank = rank; pl.n_ranks = n_ranks; pl.count = (uint32_t)px.size(); pl.dev = nullptr;
    HIPC(c, hipMalloc((void**)&pl.dev, (px.size() ? px.size() : 1) * 4));
    if (!px.empty()) HIPC(c, hipMemcpy(pl.dev, px.data(), px.size() * 4, hipMemcpyHostToDevice));
    c->pixlists.push_back(pl);
    *out = &c->pixlists.back();
    return RT3_OK;
}

Resource* get_res(rt3_ctx* c, uint32_t handle, uint32_t want_tag) {
    uint32_t tag = handle >> 30, idx = handle & 0x3FFFFFFFu;
    if (tag != want_tag || idx >= c->resources.size()) return nullptr;
    Resource* r = &c->resources[idx];
    return r->tag == want_tag && r->ptr ? r : nullptr;
}
size_t format_bytes(uint32_t f) {
    switch (f) {
        case RT3_FORMAT_R32_SFLOAT: return 4;
        case RT3_FORMAT_R32G32B32A32_SFLOAT: return 16;
        case RT3_FORMAT_R32G32B32A32_UINT: return 16;
        case RT3_FORMAT_R8G8B8A8_UNORM: return 4;
        case RT3_FORMAT_R16_UINT: return 2;
        default: return 0;
    }
}

SceneDev scene_dev(const rt3_ctx* c) {
    SceneDev s;
    s.verts = c->d_verts;
    s.indices = c->d_indices;
    s.geoms = c->d_geoms;
    s.shade_geoms = c->d_shade_geoms;
    s.n_geoms = c->n_flat_geoms;
    s.prim_geom = c->d_prim_geom;
    s.first_prim = c->d_first_prim;
    s.tri_shade = c->bvh.tri_shade;
    s.tri_uv = c->bvh.tri_uv;
    s.guide_marg = c->d_guide_marg;
    s.sky = c->d_sky;
    s.sky_alias = c->d_sky_alias;
    s.cdf_marg = c->d_cdf_marg;
    s.sky_w = c->sky_w;
    s.sky_h = c->sky_h;
    s.sky_wt = c->sky_wt;
    s.bluenoise = c->d_bn;
    s.bn_w = c->bn_w;
    s.bn_h = c->bn_h;
    s.tex_pixels = c->d_tex_pixels;
    s.tex_table = c->d_tex_table;
    s.srgb_lut = c->d_srgb_lut;
    s.n_tex = c->d_tex_pixels ? (uint32_t)c->h_tex.size() : 0u;
    return s;
}

// (re)build the device texture atlas after rt3_scene_set_texture calls
int sync_textures(rt3_ctx* c) {
    if (!c->tex_dirty) return RT3_OK;
    std::vector<uint4> table(c->h_tex.size());
    size_t total = 0;
    for (size_t i = 0; i < c->h_tex.size(); i++) {
        if (c->h_tex[i].empty()) return fail(c, RT3_E_STATE, "texture " + std::to_string(i) + " was never set (indices must be dense)");
        table[i] = make_uint4((uint32_t)total, c->tex_w[i], c->tex_h[i], 0u);
        total += c->h_tex[i].size();
    }
    if (total > 0xFFFFFFF0ull) return fail(c, RT3_E_INVALID, "textures exceed 4 GiB");
    std::vector<uint8_t> all(total);
    for (size_t i = 0; i < c->h_tex.size(); i++) memcpy(all.data() + table[i].x, c->h_tex[i].data(), c->h_tex[i].size());
    if (int r = dev_alloc(c, &c->d_tex_pixels, total)) return r;
    if (int r = dev_alloc(c, &c->d_tex_table, table.size())) return r;
    HIPC(c, hipMemcpy(c->d_tex_pixels, all.data(), total, hipMemcpyHostToDevice));
    HIPC(c, hipMemcpy(c->d_tex_table, table.data(), table.size() * sizeof(uint4), hipMemcpyHostToDevice));
    if (!c->d_srgb_lut) {
        float lut[256];
        for (int i = 0; i < 256; i++) {  // sRGB EOTF (IEC 61966-2-1), evaluated in double
            double v = i / 255.0;
            lut[i] = (float)(v <= 0.04045 ? v / 12.92 : std::pow((v + 0.055) / 1.055, 2.4));
        }
        if (int r = dev_alloc(c, &c->d_srgb_lut, (size_t)256)) return r;
        HIPC(c, hipMemcpy(c->d_srgb_lut, lut, sizeof(lut), hipMemcpyHostToDevice));
    }
    c->tex_dirty = false;
    return RT3_OK;
}

// Failure-atomic: if any allocation fails the whole queue set is released and the capacities drop to 0, so the next pass
// re-allocates (or reports the error again) instead of launching kernels on a half-resized set.
void free_work(rt3_ctx* c) {
    for (int k = 0; k < 2; k++) { dev_free(c->rays[k]); dev_free(c->T[k]); }
    dev_free(c->hits); dev_free(c->sh_rays); dev_free(c->sh_contrib); dev_free(c->lacc); dev_free(c->radsum);
    c->cap = 0;
    c->cap_pix = 0;
}
int ensure_work(rt3_ctx* c, size_t paths, size_t npix) {
    int r = RT3_OK;
    if (paths > c->cap) {
        size_t P = (paths + 255) & ~(size_t)255;
        c->cap = 0;
        for (int k = 0; k < 2 && !r; k++) {
            if (!r) r = dev_alloc(c, &c->rays[k], 8 * P);
            if (!r) r = dev_alloc(c, &c->T[k], 3 * P);  // throughput planes (the path's pdf and id ride in the ray records)
        }
        if (!r) r = dev_alloc(c, &c->hits, 4 * P);
        if (!r) r = dev_alloc(c, &c->sh_rays, 8 * P);
        if (!r) r = dev_alloc(c, &c->sh_contrib, 2 * P);  // {blue contribution, path id} records (red / green ride with the ray)
        if (!r) r = dev_alloc(c, &c->lacc, 4 * P);        // float4 per path
        if (!r) c->cap = P;
    }
    if (!r && npix > c->cap_pix) {
        c->cap_pix = 0;
        r = dev_alloc(c, &c->radsum, 3 * npix);
        if (!r) c->cap_pix = npix;
    }
    if (r) free_work(c);
    return r;
}

int harvest(rt3_ctx* c) {  // stream must be idle
    if (!c->pending_counters.empty()) {
        std::vector<uint32_t> h(c->counters_next);
        HIPC(c, hipMemcpy(h.data(), c->d_counters, (size_t)c->counters_next * 4, hipMemcpyDeviceToHost));
        for (auto& b : c->pending_counters) {
            for (uint32_t k = 0; k < b.n_pairs; k++) {
                c->stats.extension_rays += h[b.first + 2 * k];
                c->stats.shadow_rays += h[b.first + 2 * k + 1];
            }
        }
        c->pending_counters.clear();
    }
    c->counters_next = 0;
    c->stats.extension_rays += c->primary_rays_pending;
    c->primary_rays_pending = 0;
    if (c->opt_count) {
        unsigned long long t[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // [0..3] k_extend / k_shadow, [4..9] k_trace {rays, nodes, tris} x 2, [10..11] node visits served by the LDS top-of-tree copy {closest, any}
        HIPC(c, hipMemcpy(t, c->d_totals, sizeof(t), hipMemcpyDeviceToHost));
        c->stats.nodes_visited += t[0] + t[5];
        c->stats.tris_tested += t[1] + t[6];
        c->stats.shadow_nodes_visited += t[2] + t[8];
        c->stats.shadow_tris_tested += t[3] + t[9];
        c->stats.trace_rays[0] += t[4];
        c->stats.trace_nodes[0] += t[5];
        c->stats.trace_tris[0] += t[6];
        c->stats.trace_rays[1] += t[7];
        c->stats.trace_nodes[1] += t[8];
        c->stats.trace_tris[1] += t[9];
        c->stats.nodes_visited_lds += t[10];
        c->stats.shadow_nodes_visited_lds += t[11];
        HIPC(c, hipMemset(c->d_totals, 0, sizeof(t)));
    }
    for (auto& t : c->pending_events) {
        float ms = 0.0f;
        HIPC(c, hipEventElapsedTime(&ms, t.a, t.b));
        switch (t.cat) {
            case CAT_EXTEND: c->stats.extend_ms += ms; c->stats.extend_launches++; break;
            case CAT_SHADOW: c->stats.shadow_ms += ms; c->stats.shadow_launches++; break;
            case CAT_SHADE: c->stats.shade_ms += ms; break;
            case CAT_TRACE: c->stats.trace_ms += ms; c->stats.trace_launches++; break;
            case CAT_GATHER: c->stats.gather_ms += ms; break;
            default: c->stats.other_ms += ms; break;
        }
        c->free_events.push_back(t);
    }
    c->pending_events.clear();
    return RT3_OK;
}

struct ScopedTimer {  // brackets one kernel launch with HIP events on the context's stream when profiling is on
    rt3_ctx* c;
    Timed t;
    bool on;
    ScopedTimer(rt3_ctx* ctx, int cat) : c(ctx), on(ctx->opt_profile) {
        if (!on) return;
        if (!c->free_events.empty()) {
            t = c->free_events.back();
            c->free_events.pop_back();
        } else if (hipEventCreate(&t.a) != hipSuccess || hipEventCreate(&t.b) != hipSuccess) {
            on = false;
            return;
        }
        t.cat = cat;
        (void)hipEventRecord(t.a, c->stream);
    }
    ~ScopedTimer() {
        if (!on) return;
        (void)hipEventRecord(t.b, c->stream);
        c->pending_events.push_back(t);
    }
};

int reserve_counters(rt3_ctx* c, uint32_t n, uint32_t* first) {
    if (c->counters_next + n > c->counters_cap) {
        HIPC(c, hipStreamSynchronize(c->stream));
        if (int r = harvest(c)) return r;
    }
    if (n > c->counters_cap) return fail(c, RT3_E_INVALID, "too many bounces x batches for the counter block");
    *first = c->counters_next;
    c->counters_next += n;
    HIPC(c, hipMemsetAsync(c->d_counters + *first, 0, (size_t)n * 4, c->stream));
    return RT3_OK;
}

// ---------------------------------------------------------------------------------------------- passes
int check_window(rt3_ctx* c, const rt3_gconst* g, uint32_t* W, uint32_t* H) {
    float fw = g->window_size[0], fh = g->window_size[1];
    if (!(fw >= 1.0f && fh >= 1.0f && fw <= 65535.0f && fh <= 65535.0f) || fw != std::floor(fw) || fh != std::floor(fh))
        return fail(c, RT3_E_INVALID, "GConst.window_size must hold integral pixel counts in [1, 65535]");
    *W = (uint32_t)fw;
    *H = (uint32_t)fh;
    return RT3_OK;
}
Resource* image_checked(rt3_ctx* c, uint32_t handle, uint32_t W, uint32_t H, uint32_t format, const char* what) {
    Resource* r = get_res(c, handle, RT3_TAG_IMAGE);
    if (!r || r->w != W || r->h != H || r->format != format) {
        c->err = std::string("binding '") + what + "' is not a " + std::to_string(W) + "x" + std::to_string(H) + " image of the expected format";
        return nullptr;
    }
    return r;
}

// gbuffer.slang:8-21
int pass_gbuffer(rt3_ctx* c, const rt3_gconst* g, uint32_t x, uint32_t y, const uint32_t* b, uint32_t nb) {
    uint32_t W, H;
    if (int r = check_window(c, g, &W, &H)) return r;
    if (x != W || y != H) return fail(c, RT3_E_INVALID, "gbuffer: launch size must be the window size (WorkSize2D::FullScreen, executions.rs:73)");
    if (nb != 2) return fail(c, RT3_E_INVALID, "gbuffer expects 2 bindings {gbuffer, gbuffer_depth}");
    Resource* gb = image_checked(c, b[0], W, H, RT3_FORMAT_R32G32B32A32_UINT, "gbuffer");
    Resource* dp = image_checked(c, b[1], W, H, RT3_FORMAT_R32_SFLOAT, "gbuffer_depth");
    if (!gb || !dp) return RT3_E_INVALID;
    PixelList* pl;
    if (int r = get_pixlist(c, W, H, c->rank, c->n_ranks, &pl)) return r;
    if (pl->count == 0) return RT3_OK;
    if (int r = ensure_work(c, pl->count, pl->count)) return r;
    GConstDev gd;
    memcpy(&gd, g, sizeof(gd));
    size_t S = c->cap;
    {
        ScopedTimer t(c, CAT_OTHER);
        launch_raygen(c->stream, gd, pl->dev, pl->count, c->rays[0], S);
    }
    uint32_t wc_slot;
    if (int r = reserve_counters(c, 1, &wc_slot)) return r;  // ray-pool cursor of the launch
    {
        ScopedTimer t(c, CAT_EXTEND);
        launch_extend(c->stream, c->opt_count, c->bvh.layout, c->bvh.nodes, c->bvh.tris, c->bvh.top, c->bvh.n_top, c->rays[0], S, nullptr, pl->count, pl->count, c->hits, nullptr, nullptr,
                      c->opt_count ? c->d_totals : nullptr, c->d_counters + wc_slot);
    }
    c->primary_rays_pending += pl->count;
    {
        ScopedTimer t(c, CAT_OTHER);
        launch_gbuffer(c->stream, scene_dev(c), pl->dev, pl->count, W, c->hits, S, gb->ptr, (float*)dp->ptr);
    }
    HIPC(c, hipGetLastError());
    return RT3_OK;
}

// refrence_mode.slang:14-66 as a wavefront loop
int pass_reference_mode(rt3_ctx* c, const rt3_gconst* g, uint32_t x, uint32_t y, const uint32_t* b, uint32_t nb) {
    uint32_t W, H;
    if (int r = check_window(c, g, &W, &H)) return r;
    if (x != W || y != H) return fail(c, RT3_E_INVALID, "refrence_mode: launch size must be the window size");
    if (nb != 4) return fail(c, RT3_E_INVALID, "refrence_mode expects 4 bindings {gbuffer, gbuffer_depth, Light, PrevLight}");
    Resource* gb = image_checked(c, b[0], W, H, RT3_FORMAT_R32G32B32A32_UINT, "gbuffer");
    Resource* dp = image_checked(c, b[1], W, H, RT3_FORMAT_R32_SFLOAT, "gbuffer_depth");
    Resource* li = image_checked(c, b[2], W, H, RT3_FORMAT_R32G32B32A32_SFLOAT, "Light");
    Resource* pv = image_checked(c, b[3], W, H, RT3_FORMAT_R32G32B32A32_SFLOAT, "PrevLight");
    if (!gb || !dp || !li || !pv) return RT3_E_INVALID;
    const uint32_t Sspp = g->samples, B = g->bounces;
    if (Sspp == 0 || B == 0) return RT3_OK;  // GConst::default() leaves samples = bounces = 0 (renderer/mod.rs:47-63): nothing to trace
    if (B > 64) return fail(c, RT3_E_INVALID, "bounces > 64");
    PixelList* pl;
    if (int r = get_pixlist(c, W, H, c->rank, c->n_ranks, &pl)) return r;
    const uint32_t npix = pl->count;
    if (npix == 0) return RT3_OK;
    if (pl->bn_stamp != c->bn_stamp || !pl->dev_bn) {  // (re)build the {pixel, blue-noise word} list of this window / rank
        if (!pl->dev_bn) HIPC(c, hipMalloc((void**)&pl->dev_bn, (size_t)npix * 8));
        launch_pixbn(c->stream, pl->dev, npix, c->d_bn, c->bn_w, c->bn_h, pl->dev_bn);
        pl->bn_stamp = c->bn_stamp;
    }
    // paths per wavefront batch: 160 B of queue state each, so 2^28 paths = 43 GB of the 288 GB; the C3 frame (132.7 M paths)
    // is ONE batch.  Larger launches amortise the ramp / tail of the persistent traversal kernels: 16 -> 64 spp per batch = -6.5 % frame time.
    uint64_t max_paths = 1ull << 28;
    uint32_t sb = c->opt_batch_spp > 0 ? (uint32_t)c->opt_batch_spp : (uint32_t)std::max<uint64_t>(1, max_paths / npix);
    if (sb > Sspp) sb = Sspp;
    if ((uint64_t)sb * npix > 0xFFFFFF00ull) return fail(c, RT3_E_INVALID, "batch too large");
    if (int r = ensure_work(c, (size_t)sb * npix, npix)) return r;
    const size_t S = c->cap;
    GConstDev gd;
    memcpy(&gd, g, sizeof(gd));
    const bool nee = (g->pad[0] & RT3_F_NEE_SKY) && c->d_sky;
    SceneDev sc = scene_dev(c);
    for (uint32_t s0 = 0; s0 < Sspp; s0 += sb) {
        const uint32_t nsb = std::min(sb, Sspp - s0);
        const uint32_t n_first = nsb * npix;
        uint32_t first;
        if (int r = reserve_counters(c, 4 * B + 1, &first)) return r;
        first += first & 1u;  // 8-byte aligned pairs
        // pair b = {extension rays emitted at bounce b (b < B-1), shadow rays emitted at bounce b}; then the ray-pool cursors
        uint32_t* pairs = c->d_counters + first;
        uint32_t* pool_cur = c->d_counters + first + 2 * B;  // [b], [B + b]: ray-pool cursors of the k_extend / k_shadow launch of bounce b
        c->pending_counters.push_back(CounterBlock{first, B});
        auto ext_cnt_at = [pairs](uint32_t b) { return pairs + 2 * b; };
        auto sh_cnt_at = [pairs](uint32_t b) { return pairs + 2 * b + 1; };
        int cur = 0;
        for (uint32_t bn = 0; bn < B; bn++) {
            ShadeLaunch L;
            L.g = gd; L.sc = sc; L.pixels = pl->dev; L.pixbn = pl->dev_bn; L.npix = npix; L.width = W; L.s0 = s0; L.bounce = bn;
            L.gbuffer = gb->ptr; L.depth = (const float*)dp->ptr;
            L.in_rays = c->rays[cur]; L.in_hits = c->hits; L.in_T = c->T[cur];
            L.in_count = bn ? ext_cnt_at(bn - 1) : nullptr; L.n_first = n_first;
            L.out_rays = c->rays[cur ^ 1]; L.out_T = c->T[cur ^ 1]; L.out_count = ext_cnt_at(bn);
            L.sh_rays = c->sh_rays; L.sh_contrib = c->sh_contrib; L.sh_count = sh_cnt_at(bn);
            L.lacc = c->lacc; L.stride = S; L.max_n = n_first;
            {
                ScopedTimer t(c, CAT_SHADE);
                launch_shade(c->stream, bn == 0, L);
            }
            cur ^= 1;
            // both queues in one launch (k_trace, RT3_OPT_FUSED_TRACE = 1): one end-of-launch drain less per bounce (the waves of a
            // persistent walk finish spread over the time their longest last ray takes, ~0.15 ms), against the cost of mixing the
            // two ray kinds in a wave.  Which side wins moved with every change of the shadow walk (+3 % at N = 8 before the SAH top,
            // -3 % after it), so the default is the simple one: separate launches.
            const bool fuse = c->opt_fused_trace == 1;
            if (nee && bn != B - 1 && fuse) {
                ScopedTimer t(c, CAT_TRACE);
                launch_trace(c->stream, c->opt_count, c->bvh.layout, c->bvh.nodes, c->bvh.tris, c->bvh.top, c->bvh.n_top, c->rays[cur], c->sh_rays, S, ext_cnt_at(bn), sh_cnt_at(bn), n_first,
                             c->hits, c->sh_contrib, c->lacc, c->opt_count ? c->d_totals + 4 : nullptr, pool_cur + bn, pool_cur + B + bn);
            } else {
                if (nee) {
                    ScopedTimer t(c, CAT_SHADOW);
                    launch_shadow(c->stream, c->opt_count, c->bvh.layout, c->bvh.nodes, c->bvh.tris, c->bvh.top, c->bvh.n_top, c->sh_rays, S, sh_cnt_at(bn), 0, n_first, c->sh_contrib, nullptr,
                                  c->lacc, S, nullptr, nullptr, nullptr, c->opt_count ? c->d_totals + 2 : nullptr, pool_cur + B + bn);
                }
                if (bn != B - 1) {
                    ScopedTimer t(c, CAT_EXTEND);
                    launch_extend(c->stream, c->opt_count, c->bvh.layout, c->bvh.nodes, c->bvh.tris, c->bvh.top, c->bvh.n_top, c->rays[cur], S, ext_cnt_at(bn), 0, n_first, c->hits, nullptr, nullptr,
                                  c->opt_count ? c->d_totals : nullptr, pool_cur + bn, true);
                }
            }
        }
        {
            ScopedTimer t(c, CAT_OTHER);
            launch_accumulate(c->stream, gd, pl->dev, npix, W, (const float*)dp->ptr, c->lacc, S, nsb, s0 == 0, s0 + nsb >= Sspp, c->radsum, li->ptr,
                              pv->ptr);
        }
    }
    HIPC(c, hipGetLastError());
    return RT3_OK;
}

// postprocess.slang:90-112
int pass_postprocess(rt3_ctx* c, const rt3_gconst* g, uint32_t x, uint32_t y, uint32_t z, const uint32_t* b, uint32_t nb) {
    uint32_t W, H;
    if (int r = check_window(c, g, &W, &H)) return r;
    if (x != (W + 7) / 8 || y != (H + 7) / 8 || z != 1)
        return fail(c, RT3_E_INVALID, "postprocess: dispatch must be ceil(W/8) x ceil(H/8) x 1 groups (DispatchSize::FullScreen, build.rs:254-258)");
    if (nb != 3) return fail(c, RT3_E_INVALID, "postprocess expects 3 bindings {Depth, Out, In}");
    Resource* dp = image_checked(c, b[0], W, H, RT3_FORMAT_R32_SFLOAT, "Depth");
    Resource* out = image_checked(c, b[1], W, H, RT3_FORMAT_R32G32B32A32_SFLOAT, "Out");
    Resource* in = image_checked(c, b[2], W, H, RT3_FORMAT_R32G32B32A32_SFLOAT, "In");
    if (!dp || !out || !in) return RT3_E_INVALID;
    PixelList* pl;
    if (int r = get_pixlist(c, W, H, c->rank, c->n_ranks, &pl)) return r;
    if (pl->count == 0) return RT3_OK;
    GConstDev gd;
    memcpy(&gd, g, sizeof(gd));
    ScopedTimer t(c, CAT_OTHER);
    launch_postprocess(c->stream, gd, scene_dev(c), pl->dev, pl->count, W, (const float*)dp->ptr, in->ptr, out->ptr);
    HIPC(c, hipGetLastError());
    return RT3_OK;
}

// ---- probe-GI passes.  A probe owns a 16x16 pixel block and an 8x8-texel cell of the probe atlas; the passes
//      run on the whole window on every rank (they are not part of the tile-partitioned path).
int probe_grid(rt3_ctx* c, const char* pass, uint32_t W, uint32_t H, uint32_t px, uint32_t py) {
    if (px == 0 || py == 0 || px > W / 16 || py > H / 16)
        return fail(c, RT3_E_INVALID, std::string(pass) + ": the probe grid must be between 1x1 and floor(W/16) x floor(H/16) probes");
    return RT3_OK;
}
// structured_importance_sampling.slang:7-11 : set 1 {gbuffer, gbuffer_depth, out, debug}, set 2 {probe_atlas}
int pass_sis(rt3_ctx* c, const rt3_gconst* g, uint32_t x, uint32_t y, uint32_t z, const uint32_t* b, uint32_t nb) {
    uint32_t W, H;
    if (int r = check_window(c, g, &W, &H)) return r;
    if (z != 1) return fail(c, RT3_E_INVALID, "structured_importance_sampling: dispatch is probes_x x probes_y x 1 groups of 8x8 threads");
    if (int r = probe_grid(c, "structured_importance_sampling", W, H, x, y)) return r;
    if (nb != 5) return fail(c, RT3_E_INVALID, "structured_importance_sampling expects 5 bindings {gbuffer, gbuffer_depth, out, debug, probe_atlas}");
    Resource* gb = image_checked(c, b[0], W, H, RT3_FORMAT_R32G32B32A32_UINT, "gbuffer");
    Resource* dp = image_checked(c, b[1], W, H, RT3_FORMAT_R32_SFLOAT, "gbuffer_depth");
    Resource* out = image_checked(c, b[2], x * 8, y * 8, RT3_FORMAT_R16_UINT, "out");
    Resource* dbg = image_checked(c, b[3], x * 8, y * 8, RT3_FORMAT_R32_SFLOAT, "debug");
    Resource* at = image_checked(c, b[4], x * 8, y * 8, RT3_FORMAT_R32G32B32A32_SFLOAT, "probe_atlas");
    if (!gb || !dp || !out || !dbg || !at) return RT3_E_INVALID;
    ScopedTimer t(c, CAT_OTHER);
    launch_sis(c->stream, W, x, y, gb->ptr, out->ptr, (float*)dbg->ptr);
    HIPC(c, hipGetLastError());
    return RT3_OK;
}
// trace_probes.slang:8-12 : set 1 {gbuffer, gbuffer_depth, directions}, set 2 {probe_atlas}, set 3 {prev_probe_atlas}
int pass_trace_probes(rt3_ctx* c, const rt3_gconst* g, uint32_t x, uint32_t y, const uint32_t* b, uint32_t nb) {
    uint32_t W, H;
    if (int r = check_window(c, g, &W, &H)) return r;
    if (x % 8 || y % 8) return fail(c, RT3_E_INVALID, "trace_probes: launch size is the probe atlas, 8 x 8 texels per probe");
    if (int r = probe_grid(c, "trace_probes", W, H, x / 8, y / 8)) return r;
    if (nb != 5) return fail(c, RT3_E_INVALID, "trace_probes expects 5 bindings {gbuffer, gbuffer_depth, directions, probe_atlas, prev_probe_atlas}");
    Resource* gb = image_checked(c, b[0], W, H, RT3_FORMAT_R32G32B32A32_UINT, "gbuffer");
    Resource* dp = image_checked(c, b[1], W, H, RT3_FORMAT_R32_SFLOAT, "gbuffer_depth");
    Resource* dir = image_checked(c, b[2], x, y, RT3_FORMAT_R16_UINT, "directions");
    Resource* at = image_checked(c, b[3], x, y, RT3_FORMAT_R32G32B32A32_SFLOAT, "probe_atlas");
    Resource* pv = image_checked(c, b[4], x, y, RT3_FORMAT_R32G32B32A32_SFLOAT, "prev_probe_atlas");
    if (!gb || !dp || !dir || !at || !pv) return RT3_E_INVALID;
    if (at->ptr == pv->ptr) return fail(c, RT3_E_INVALID, "trace_probes: probe_atlas and prev_probe_atlas must be different images");
    const uint32_t n = x * y;
    if (int r = ensure_work(c, n, 0)) return r;
    GConstDev gd;
    memcpy(&gd, g, sizeof(gd));
    const size_t S = c->cap;
    {
        ScopedTimer t(c, CAT_OTHER);
        launch_probe_raygen(c->stream, gd, W, x / 8, y / 8, (const float*)dp->ptr, dir->ptr, at->ptr, c->rays[0], S, c->T[0]);
    }
    uint32_t wc_slot;
    if (int r = reserve_counters(c, 1, &wc_slot)) return r;
    {
        ScopedTimer t(c, CAT_EXTEND);
        launch_extend(c->stream, c->opt_count, c->bvh.layout, c->bvh.nodes, c->bvh.tris, c->bvh.top, c->bvh.n_top, c->rays[0], S, nullptr, n, n, c->hits, nullptr, nullptr,
                      c->opt_count ? c->d_totals : nullptr, c->d_counters + wc_slot);
    }
    c->primary_rays_pending += n;
    {
        ScopedTimer t(c, CAT_OTHER);
        launch_probe_store(c->stream, scene_dev(c), g->pad[0], g->blendfactor, x / 8, y / 8, c->hits, c->T[0], pv->ptr, at->ptr);
    }
    HIPC(c, hipGetLastError());
    return RT3_OK;
}
// spherical_harmonic_conversion.slang:6-7 : set 0 {out}, set 1 {probe_atlas}
int pass_sh_conversion(rt3_ctx* c, uint32_t x, uint32_t y, uint32_t z, const uint32_t* b, uint32_t nb) {
    if (z != 1 || x == 0 || y == 0 || x > 8191 || y > 8191)
        return fail(c, RT3_E_INVALID, "spherical_harmonic_conversion: dispatch is probes_x x probes_y x 1 groups of 8x8 threads");
    if (nb != 2) return fail(c, RT3_E_INVALID, "spherical_harmonic_conversion expects 2 bindings {out, probe_atlas}");
    Resource* out = get_res(c, b[0], RT3_TAG_BUFFER);
    Resource* at = image_checked(c, b[1], x * 8, y * 8, RT3_FORMAT_R32G32B32A32_SFLOAT, "probe_atlas");
    if (!at) return RT3_E_INVALID;
    const size_t need = ((size_t)zcurve_host(x * 3 - 1, y - 1) + 1) * 48;  // float3x3 elements at Z-curve indices (:30-32)
    if (!out || out->bytes < need) return fail(c, RT3_E_INVALID, "spherical_harmonic_conversion: 'out' must be a buffer of at least " + std::to_string(need) + " bytes");
    ScopedTimer t(c, CAT_OTHER);
    launch_sh_conversion(c->stream, x, y, at->ptr, out->ptr);
    HIPC(c, hipGetLastError());
    return RT3_OK;
}
// interpolate_probes.slang:6-9 : set 1 {gbuffer, gbuffer_depth, sh_coeficents}, set 2 {Light}
int pass_interpolate_probes(rt3_ctx* c, const rt3_gconst* g, uint32_t x, uint32_t y, uint32_t z, const uint32_t* b, uint32_t nb) {
    uint32_t W, H;
    if (int r = check_window(c, g, &W, &H)) return r;
    if (x != (W + 7) / 8 || y != (H + 7) / 8 || z != 1) return fail(c, RT3_E_INVALID, "interpolate_probes: dispatch must be ceil(W/8) x ceil(H/8) x 1 groups");
    if (nb != 4) return fail(c, RT3_E_INVALID, "interpolate_probes expects 4 bindings {gbuffer, gbuffer_depth, sh_coeficents, Light}");
    Resource* gb = image_checked(c, b[0], W, H, RT3_FORMAT_R32G32B32A32_UINT, "gbuffer");
    Resource* dp = image_checked(c, b[1], W, H, RT3_FORMAT_R32_SFLOAT, "gbuffer_depth");
    Resource* sh = get_res(c, b[2], RT3_TAG_BUFFER);
    Resource* li = image_checked(c, b[3], W, H, RT3_FORMAT_R32G32B32A32_SFLOAT, "Light");
    if (!gb || !dp || !li) return RT3_E_INVALID;
    const uint32_t npx = W / 16, npy = H / 16;
    const size_t need = npx && npy ? ((size_t)zcurve_host(npx * 3 - 1, npy - 1) + 1) * 48 : 0;
    if (!sh || sh->bytes < need) return fail(c, RT3_E_INVALID, "interpolate_probes: 'sh_coeficents' must be a buffer of at least " + std::to_string(need) + " bytes");
    GConstDev gd;
    memcpy(&gd, g, sizeof(gd));
    ScopedTimer t(c, CAT_OTHER);
    launch_interpolate(c->stream, gd, W, H, gb->ptr, (const float*)dp->ptr, sh->ptr, li->ptr);
    HIPC(c, hipGetLastError());
    return RT3_OK;
}

}  // namespace

// ================================================================================================== C ABI
extern "C" {

int rt3_create(int device, rt3_ctx** out) {
    if (!out) return fail(nullptr, RT3_E_INVALID, "out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(nullptr, RT3_E_NO_DEVICE, "no HIP device visible (librt3 has no CPU fallback)");
    if (device < 0 || device >= n) return fail(nullptr, RT3_E_INVALID, "device index out of range");
    hipDeviceProp_t prop;
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) return fail(nullptr, RT3_E_HIP, "hipSetDevice failed");
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(nullptr, RT3_E_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", librt3 is built for gfx950 only");
    rt3_ctx* c = new rt3_ctx();
    c->device = device;
    snprintf(c->name, sizeof(c->name), "%s (%s)", prop.name, prop.gcnArchName);
    memset(&c->stats, 0, sizeof(c->stats));
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || hipMalloc((void**)&c->d_counters, (size_t)c->counters_cap * 4) != hipSuccess ||
        hipMalloc((void**)&c->d_totals, 96) != hipSuccess || hipMemset(c->d_totals, 0, 96) != hipSuccess) {
        delete c;
        return fail(nullptr, RT3_E_HIP, "stream / counter allocation failed");
    }
    *out = c;
    return RT3_OK;
}

void rt3_destroy(rt3_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    dev_free(c->d_verts); dev_free(c->d_indices); dev_free(c->d_geoms); dev_free(c->d_shade_geoms); dev_free(c->d_prim_geom); dev_free(c->d_first_prim);
    dev_free(c->d_tex_pixels); dev_free(c->d_tex_table); dev_free(c->d_srgb_lut);
    dev_free(c->d_sky); dev_free(c->d_sky_alias); dev_free(c->d_cdf_marg); dev_free(c->d_bn);
    dev_free(c->bvh.nodes); dev_free(c->bvh.tris); dev_free(c->bvh.tri_shade); dev_free(c->bvh.tri_uv); dev_free(c->bvh.top); dev_free(c->d_guide_marg);
    c->build_arena.release();
    for (auto& r : c->resources)
        if (r.owned && r.ptr) (void)hipFree(r.ptr);
    for (auto& p : c->pixlists) {
        (void)hipFree(p.dev);
        (void)hipFree(p.dev_bn);
    }
    free_work(c);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    dev_free(c->gather_buf);
    for (auto& gl : c->gather_layouts) (void)hipFree(gl.dev);
    dev_free(c->d_counters); dev_free(c->d_totals);
    for (auto& t : c->pending_events) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
    for (auto& t : c->free_events) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* rt3_last_error(rt3_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int rt3_device_name(rt3_ctx* c, char* buf, size_t n) {
    if (!c || !buf || !n) return RT3_E_INVALID;
    snprintf(buf, n, "%s", c->name);
    return RT3_OK;
}

int rt3_set_option(rt3_ctx* c, int option, int64_t value) {
    if (!c) return RT3_E_INVALID;
    switch (option) {
        case RT3_OPT_BATCH_SPP: c->opt_batch_spp = value; return RT3_OK;
        case RT3_OPT_PROFILE: c->opt_profile = value != 0; return RT3_OK;
        case RT3_OPT_COUNT_TRAVERSAL: c->opt_count = value != 0; return RT3_OK;
        case RT3_OPT_EXTEND_VARIANT:
            c->opt_variant = (int)value;
            HIPC(c, hipSetDevice(c->device));  // the traversal knobs are __constant__ words of the device the context runs on
            set_refill_lanes((uint32_t)value);
            return RT3_OK;
        case RT3_OPT_LEAF_SIZE:
            if (value < 1 || value > 8) return fail(c, RT3_E_INVALID, "leaf size must be 1..8");
            c->opt_leaf_size = (uint32_t)value;
            c->accel_built = false;
            return RT3_OK;
        case RT3_OPT_NODE_QUANT:
            if (value < 0 || value > 2) return fail(c, RT3_E_INVALID, "node quantisation must be 0 (fp32), 1 (64 B) or 2 (compact 48 B)");
            c->opt_node_quant = (uint32_t)value;
            c->accel_built = false;
            return RT3_OK;
        case RT3_OPT_SAH_TOP:
            if (value < 0 || value > 65536) return fail(c, RT3_E_INVALID, "SAH-top cluster size must be 0 (off) .. 65536");
            c->opt_sah_top = (uint32_t)value;
            c->accel_built = false;
            return RT3_OK;
        case RT3_OPT_SAH_TOP_DEVICE:
            if (value != 0 && value != 1) return fail(c, RT3_E_INVALID, "SAH-top device must be 0 (host) or 1 (GPU)");
            c->opt_sah_device = (uint32_t)value;
            c->accel_built = false;
            return RT3_OK;
        case RT3_OPT_FUSED_TRACE:
            if (value < 0 || value > 1) return fail(c, RT3_E_INVALID, "fused trace must be 0 or 1");
            c->opt_fused_trace = (int)value;
            return RT3_OK;
        case RT3_OPT_POOL_CHUNK:
            if (value < 64 || value > 65536 || (value & 63)) return fail(c, RT3_E_INVALID, "pool chunk must be a multiple of 64 in [64, 65536]");
            HIPC(c, hipSetDevice(c->device));
            set_pool_chunk((uint32_t)value);
            return RT3_OK;
        case RT3_OPT_TRACE_BLOCKS:
            if (value < 1 || value > 65535) return fail(c, RT3_E_INVALID, "trace blocks must be in [1, 65535]");
            set_trace_blocks((uint32_t)value);
            return RT3_OK;
        case RT3_OPT_WIDE_COLLAPSE:
            if (value < 0 || value > 2) return fail(c, RT3_E_INVALID, "wide collapse must be 0 (even depth), 1 (surface area) or 2 (cost-driven)");
            c->opt_collapse = (uint32_t)value;
            c->accel_built = false;
            return RT3_OK;
        case RT3_OPT_NODE_WIDTH:
            if (value != 2 && value != 4) return fail(c, RT3_E_INVALID, "node width must be 2 or 4");
            c->opt_node_width = (uint32_t)value;
            c->accel_built = false;
            return RT3_OK;
        default: return fail(c, RT3_E_INVALID, "unknown option");
    }
}

// ---- scene
int rt3_scene_set_vertices(rt3_ctx* c, const float* v, uint32_t n) {
    if (!c || (!v && n)) return fail(c, RT3_E_INVALID, "vertices NULL");
    // a NaN / infinite position would poison the scene bounds, the Morton codes and every box above it: reject it here
    // (bounded magnitude too, so that box extents and the quantisation grid cannot overflow to infinity)
    for (size_t i = 0; i < (size_t)n; i++)
        for (int k = 0; k < 3; k++)
            if (!(std::fabs(v[8 * i + k]) <= 1.0e18f)) return fail(c, RT3_E_INVALID, "vertex " + std::to_string(i) + ": position is not finite (or beyond 1e18)");
    HIPC(c, hipSetDevice(c->device));
    if (int r = dev_alloc(c, &c->d_verts, (size_t)n * 8)) return r;
    if (n) HIPC(c, hipMemcpy(c->d_verts, v, (size_t)n * 32, hipMemcpyHostToDevice));
    c->n_verts = n;
    c->accel_built = false;
    return RT3_OK;
}
int rt3_scene_set_indices(rt3_ctx* c, const uint32_t* idx, uint32_t n) {
    if (!c || (!idx && n)) return fail(c, RT3_E_INVALID, "indices NULL");
    HIPC(c, hipSetDevice(c->device));
    if (int r = dev_alloc(c, &c->d_indices, (size_t)n)) return r;
    if (n) HIPC(c, hipMemcpy(c->d_indices, idx, (size_t)n * 4, hipMemcpyHostToDevice));
    c->n_indices = n;
    c->h_indices.assign(idx, idx + n);
    c->accel_built = false;
    return RT3_OK;
}
// bounds of every geometry's index / vertex range against the world buffers as they are NOW: the kernels index them without
// checks (a GPU fault would take the node down).  Run by rt3_scene_set_geometry and again by rt3_accel_build, because the vertex
// and index buffers may be replaced (by smaller ones) after the geometry was set.
static int validate_geometry(rt3_ctx* c, const rt3_geometry_info* g, const uint32_t* prim_counts, uint32_t n) {
    for (uint32_t i = 0; i < n; i++) {
        if ((uint64_t)g[i].index_offset + 3ull * prim_counts[i] > c->n_indices)
            return fail(c, RT3_E_INVALID, "geometry " + std::to_string(i) + ": index range exceeds the index buffer");
        uint32_t mx = 0;
        for (uint64_t k = 0; k < 3ull * prim_counts[i]; k++) {
            uint32_t v = c->h_indices[g[i].index_offset + k];
            mx = v > mx ? v : mx;
        }
        if (prim_counts[i] && (uint64_t)g[i].vertex_offset + mx >= (uint64_t)c->n_verts)
            return fail(c, RT3_E_INVALID, "geometry " + std::to_string(i) + ": vertex range exceeds the vertex buffer (set vertices and indices before geometry)");
    }
    return RT3_OK;
}
int rt3_scene_set_geometry(rt3_ctx* c, const rt3_geometry_info* g, const uint32_t* prim_counts, uint32_t n) {
    if (!c || ((!g || !prim_counts) && n)) return fail(c, RT3_E_INVALID, "geometry NULL");
    HIPC(c, hipSetDevice(c->device));
    if (int r = validate_geometry(c, g, prim_counts, n)) return r;
    uint64_t total = 0;
    int64_t max_tex = -1;
    for (uint32_t i = 0; i < n; i++) {
        if (g[i].base_color_texture_index > max_tex) max_tex = g[i].base_color_texture_index;
        total += prim_counts[i];
    }
    if (total > 0x7FFFFFFFull) return fail(c, RT3_E_INVALID, "too many primitives");
    // (the device tables -- one entry per (instance, geometry) -- are made by rt3_accel_build, which knows the instances)
    c->h_geoms.assign(g, g + n);
    c->h_prim_counts.assign(prim_counts, prim_counts + n);
    c->n_geoms = n;
    c->max_tex_index = max_tex;
    c->n_prims = (uint32_t)total;
    c->accel_built = false;
    return RT3_OK;
}
// Sky storage and importance tables (north_star; the oracle's orc_scene_set_sky has the definitions and is built by the same
// arithmetic, in double, in the same order): radiance stored as RGB9E5 (packing.slang:99-162), marginal CDF over rows, one alias
// table per row with 16-bit keep-thresholds, pdf_uv = the density the quantised tables really realise.
static uint32_t host_rgb9e5(const float* c) {  // packing.slang:99-144 == rt3_device.hpp float3_to_rgb9e5
    auto bits = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
    auto from_bits = [](uint32_t u) { float f; memcpy(&f, &u, 4); return f; };
    const float mx = (511.0f / 512.0f) * 65536.0f;
    auto clampf = [&](float v) { v = v > 0.0f ? v : 0.0f; return v < mx ? v : mx; };
    const float rc = clampf(c[0]), gc = clampf(c[1]), bc = clampf(c[2]);
    const float m1 = gc > bc ? gc : bc, maxrgb = rc > m1 ? rc : m1;
    const int fl2 = (int)((bits(maxrgb) & 0x7F800000u) >> 23) - 127;
    int exp_shared = (fl2 > -16 ? fl2 : -16) + 1 + 15;
    float denom = from_bits((uint32_t)(exp_shared - 15 - 9 + 127) << 23);
    const int maxm = (int)std::floor(maxrgb / denom + 0.5f);
    if (maxm == 512) {
        denom *= 2.0f;
        exp_shared += 1;
    }
    const int rm = (int)std::floor(rc / denom + 0.5f), gm = (int)std::floor(gc / denom + 0.5f), bm = (int)std::floor(bc / denom + 0.5f);
    return ((uint32_t)rm << 23) | ((uint32_t)gm << 14) | ((uint32_t)bm << 5) | (uint32_t)exp_shared;
}
static void host_rgb9e5_decode(uint32_t v, float* c) {  // packing.slang:146-162
    const uint32_t sb = (uint32_t)((int)(v & 31u) - 24 + 127) << 23;
    float scale;
    memcpy(&scale, &sb, 4);
    c[0] = (float)((v >> 23) & 511u) * scale;
    c[1] = (float)((v >> 14) & 511u) * scale;
    c[2] = (float)((v >> 5) & 511u) * scale;
}
int rt3_scene_set_sky(rt3_ctx* c, const float* rgb, uint32_t w, uint32_t h) {
    if (!c || !rgb || !w || !h) return fail(c, RT3_E_INVALID, "sky NULL / empty");
    if (w > 65535 || h > 65535) return fail(c, RT3_E_INVALID, "sky larger than 65535 texels per side");
    HIPC(c, hipSetDevice(c->device));
    const size_t n = (size_t)w * h;
    for (size_t i = 0; i < 3 * n; i++)  // a NaN or negative texel would poison the sampling tables
        if (!(rgb[i] >= 0.0f && rgb[i] <= 3.4028234663852886e38f))
            return fail(c, RT3_E_INVALID, "sky texel " + std::to_string(i / 3) + " is negative or not finite (clamp the image before uploading it)");
    std::vector<uint32_t> texq(n), alias(n);
    std::vector<float> pdf(n), marg(h);
    std::vector<double> rows(h), f(w), sc(w), real(w);
    std::vector<uint32_t> small(w), large(w);
    double total = 0.0;
    for (uint32_t y = 0; y < h; y++) {
        const double st = std::sin(3.14159265358979323846 * ((double)y + 0.5) / (double)h);
        double acc = 0.0;
        for (uint32_t x = 0; x < w; x++) {
            const size_t i = (size_t)y * w + x;
            texq[i] = host_rgb9e5(rgb + 3 * i);
            float p[3];
            host_rgb9e5_decode(texq[i], p);
            const float lum = p[0] * 0.299f + p[1] * 0.587f + p[2] * 0.114f;  // luminance(), math.slang:119-122
            f[x] = ((double)lum + 1e-6) * st;
            acc += f[x];
        }
        rows[y] = acc;
        total += acc;
        uint32_t ns = 0, nl = 0;
        uint32_t* al = alias.data() + (size_t)y * w;
        for (uint32_t x = 0; x < w; x++) {
            sc[x] = f[x] * (double)w / acc;
            if (sc[x] < 1.0) small[ns++] = x;
            else large[nl++] = x;
        }
        for (uint32_t x = 0; x < w; x++) al[x] = 65535u | (x << 16);
        while (ns && nl) {  // Vose's alias method; both stacks filled in ascending column order and popped from the top
            const uint32_t a = small[--ns], g = large[--nl];
            const double q = sc[a] * 65536.0;
            int64_t q16 = (int64_t)std::floor(q + 0.5) - 1;
            q16 = q16 < 0 ? 0 : (q16 > 65535 ? 65535 : q16);
            al[a] = (uint32_t)q16 | (g << 16);
            sc[g] = (sc[g] + sc[a]) - 1.0;
            if (sc[g] < 1.0) small[ns++] = g;
            else large[nl++] = g;
        }
        for (uint32_t x = 0; x < w; x++) real[x] = 0.0;
        for (uint32_t x = 0; x < w; x++) {
            const double Q = (double)((al[x] & 0xFFFFu) + 1u) / 65536.0;
            real[x] += Q;
            real[al[x] >> 16] += 1.0 - Q;
        }
        for (uint32_t x = 0; x < w; x++) pdf[(size_t)y * w + x] = (float)real[x];
    }
    double run = 0.0;
    for (uint32_t y = 0; y < h; y++) {
        run += rows[y];
        marg[y] = (float)(run / total);
        const double rowp = rows[y] / total * (double)h;
        for (uint32_t x = 0; x < w; x++) pdf[(size_t)y * w + x] = (float)((double)pdf[(size_t)y * w + x] * rowp);
    }
    marg[h - 1] = 1.0f;
    // guide table of the marginal CDF: guide[k] = first index with cdf > k / n, so a lookup of u (cell k = floor(u n)) starts inside
    // [guide[k-1], guide[k+1]].  Stored per cell as one word lo | hi << 16 (hi clamped to n-1): one load instead of two.
    std::vector<uint32_t> gmarg(h);
    {
        std::vector<uint32_t> g(h + 1);
        uint32_t i = 0;
        for (uint32_t k = 0; k <= h; k++) {
            const float thr = (float)k / (float)h;
            while (i < h - 1 && !(marg[i] > thr)) i++;
            g[k] = i;
        }
        for (uint32_t k = 0; k < h; k++) {
            const uint32_t lo = g[k > 0 ? k - 1 : 0], hi = g[k + 1] > h - 1 ? h - 1 : g[k + 1];
            gmarg[k] = lo | (hi << 16);
        }
    }
    // the marginal CDF is stored with one leading 0 and three trailing pads (2.0 > any u): cdfp[i + 1] = cdf[i], so that
    // {cdf[i-1], cdf[i], cdf[i+1], cdf[i+2]} is ONE 16-byte load at cdfp + i for every i
    std::vector<float> margp((size_t)h + 4);
    margp[0] = 0.0f;
    std::memcpy(margp.data() + 1, marg.data(), (size_t)h * 4);
    margp[h + 1] = margp[h + 2] = margp[h + 3] = 2.0f;
    // texels in 4 x 4 tiles of 128 bytes; ragged edges are padded (never addressed: lookups wrap / clamp to [0, w) x [0, h))
    const uint32_t wt = (w + 3) / 4, ht = (h + 3) / 4;
    std::vector<uint2> tiled((size_t)wt * ht * 16, make_uint2(0u, 0u));
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            uint32_t pb;
            memcpy(&pb, &pdf[(size_t)y * w + x], 4);
            tiled[((size_t)(y >> 2) * wt + (x >> 2)) * 16 + (((y & 3u) << 2) | (x & 3u))] = make_uint2(texq[(size_t)y * w + x], pb);
        }
    if (int r = dev_alloc(c, &c->d_guide_marg, gmarg.size())) return r;
    if (int r = dev_alloc(c, &c->d_sky_alias, alias.size())) return r;
    if (int r = dev_alloc(c, &c->d_sky, tiled.size())) return r;
    if (int r = dev_alloc(c, &c->d_cdf_marg, margp.size())) return r;
    HIPC(c, hipMemcpy(c->d_guide_marg, gmarg.data(), gmarg.size() * 4, hipMemcpyHostToDevice));
    HIPC(c, hipMemcpy(c->d_sky_alias, alias.data(), alias.size() * 4, hipMemcpyHostToDevice));
    HIPC(c, hipMemcpy(c->d_sky, tiled.data(), tiled.size() * 8, hipMemcpyHostToDevice));
    HIPC(c, hipMemcpy(c->d_cdf_marg, margp.data(), margp.size() * 4, hipMemcpyHostToDevice));
    c->sky_w = w;
    c->sky_h = h;
    c->sky_wt = wt;
    return RT3_OK;
}
int rt3_scene_set_bluenoise(rt3_ctx* c, const uint8_t* rgba, uint32_t w, uint32_t h) {
    if (!c || !rgba || !w || !h) return fail(c, RT3_E_INVALID, "bluenoise NULL / empty");
    HIPC(c, hipSetDevice(c->device));
    if (int r = dev_alloc(c, &c->d_bn, (size_t)w * h * 4)) return r;
    HIPC(c, hipMemcpy(c->d_bn, rgba, (size_t)w * h * 4, hipMemcpyHostToDevice));
    c->bn_w = w;
    c->bn_h = h;
    c->bn_stamp++;
    return RT3_OK;
}
// base-colour texture `index` (RGBA8, sRGB-encoded colour), sampled by hit_info when GeometryInfo.baseColorTextureIndex == index
int rt3_scene_set_texture(rt3_ctx* c, uint32_t index, const uint8_t* rgba, uint32_t w, uint32_t h) {
    if (!c || !rgba || !w || !h || w > 16384 || h > 16384 || index > 4096) return fail(c, RT3_E_INVALID, "texture: NULL / bad size / index");
    if (index >= c->h_tex.size()) {
        c->h_tex.resize(index + 1);
        c->tex_w.resize(index + 1, 0);
        c->tex_h.resize(index + 1, 0);
    }
    c->h_tex[index].assign(rgba, rgba + (size_t)w * h * 4);
    c->tex_w[index] = w;
    c->tex_h[index] = h;
    c->tex_dirty = true;
    return RT3_OK;
}
int rt3_sky_download(rt3_ctx* c, uint32_t* alias, uint32_t* texels, float* marg, float* pdf) {
    if (!c || !c->d_sky) return fail(c, RT3_E_STATE, "no sky set");
    const uint32_t w = c->sky_w, h = c->sky_h, wt = c->sky_wt, ht = (h + 3) / 4;
    if (alias) HIPC(c, hipMemcpy(alias, c->d_sky_alias, (size_t)w * h * 4, hipMemcpyDeviceToHost));
    if (marg) HIPC(c, hipMemcpy(marg, c->d_cdf_marg + 1, (size_t)h * 4, hipMemcpyDeviceToHost));  // strip the padding
    if (texels || pdf) {  // un-tile
        std::vector<uint2> tiled((size_t)wt * ht * 16);
        HIPC(c, hipMemcpy(tiled.data(), c->d_sky, tiled.size() * 8, hipMemcpyDeviceToHost));
        for (uint32_t y = 0; y < h; y++)
            for (uint32_t x = 0; x < w; x++) {
                const uint2 t = tiled[((size_t)(y >> 2) * wt + (x >> 2)) * 16 + (((y & 3u) << 2) | (x & 3u))];
                if (texels) texels[(size_t)y * w + x] = t.x;
                if (pdf) memcpy(&pdf[(size_t)y * w + x], &t.y, 4);
            }
    }
    return RT3_OK;
}

// world/mod.rs:34-60,104-125: InstanceInfo{mesh_index, transform} + Transform{Mat4}, global instance / transform buffers
int rt3_scene_set_instances(rt3_ctx* c, const rt3_instance* inst, uint32_t n) {
    if (!c || (!inst && n)) return fail(c, RT3_E_INVALID, "instances NULL");
    for (uint32_t i = 0; i < n; i++) {
        for (int k = 0; k < 16; k++)
            if (!(std::fabs(inst[i].transform[k]) <= 1.0e18f)) return fail(c, RT3_E_INVALID, "instance " + std::to_string(i) + ": transform is not finite (or beyond 1e18)");
        const float* m = inst[i].transform;
        if (m[3] != 0.0f || m[7] != 0.0f || m[11] != 0.0f || m[15] != 1.0f)
            return fail(c, RT3_E_INVALID, "instance " + std::to_string(i) + ": the last row of the transform must be (0, 0, 0, 1) (VkTransformMatrixKHR is 3 x 4 too)");
    }
    c->h_instances.assign(inst, inst + n);
    c->accel_built = false;
    return RT3_OK;
}
// One (instance, geometry) pair per entry, instance-major; no instances = one identity instance of everything.  A few KiB of tables
// go up; primitive -> entry is filled in on the device (k_prim_geom), so a rebuild after a moved instance copies nothing big.
static int flatten_world(rt3_ctx* c) {
    static const float kIdentity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    rt3_instance whole;
    whole.geometry_first = 0;
    whole.geometry_count = c->n_geoms;
    memcpy(whole.transform, kIdentity, sizeof(kIdentity));
    const rt3_instance* inst = c->h_instances.empty() ? &whole : c->h_instances.data();
    const size_t n_inst = c->h_instances.empty() ? 1 : c->h_instances.size();
    std::vector<FlatGeomDev> flat;
    std::vector<ShadeGeomDev> shade;
    std::vector<uint32_t> first;
    uint64_t total = 0;
    for (size_t i = 0; i < n_inst; i++) {
        if ((uint64_t)inst[i].geometry_first + inst[i].geometry_count > c->n_geoms)
            return fail(c, RT3_E_INVALID, "instance " + std::to_string(i) + ": geometry range exceeds the geometries set (rt3_scene_set_geometry)");
        const float* m = inst[i].transform;
        const bool identity = memcmp(m, kIdentity, sizeof(kIdentity)) == 0;
        for (uint32_t k = 0; k < inst[i].geometry_count; k++) {
            const uint32_t g = inst[i].geometry_first + k;
            FlatGeomDev f;
            memset(&f, 0, sizeof(f));
            static_assert(sizeof(rt3_geometry_info) == sizeof(GeometryInfoDev), "geometry info layouts");
            memcpy(&f.g, &c->h_geoms[g], sizeof(f.g));
            for (int col = 0; col < 4; col++)
                for (int row = 0; row < 3; row++) f.m[3 * col + row] = m[4 * col + row];
            f.identity = identity ? 1u : 0u;
            f.geom = g;
            f.instance = (uint32_t)i;
            ShadeGeomDev sg;
            memset(&sg, 0, sizeof(sg));
            for (int q = 0; q < 3; q++) { sg.base_color[q] = f.g.base_color[q]; sg.emission[q] = f.g.emission[q]; }
            sg.tex = f.g.tex;
            sg.metallic = f.g.metallic;
            sg.roughness = f.g.roughness;
            sg.identity = f.identity;
            memcpy(sg.m, f.m, 9 * sizeof(float));
            flat.push_back(f);
            shade.push_back(sg);
            first.push_back((uint32_t)total);
            total += c->h_prim_counts[g];
            if (total > (1ull << 28)) return fail(c, RT3_E_UNSUPPORTED, "more than 2^28 triangles after instancing (leaf references hold 28 bits)");
        }
    }
    const size_t nf = flat.size();
    if (int r = dev_alloc(c, &c->d_geoms, nf)) return r;
    if (int r = dev_alloc(c, &c->d_shade_geoms, nf)) return r;
    if (int r = dev_alloc(c, &c->d_first_prim, nf)) return r;
    if (int r = dev_alloc(c, &c->d_prim_geom, (size_t)total)) return r;
    if (nf) {
        HIPC(c, hipMemcpy(c->d_geoms, flat.data(), nf * sizeof(FlatGeomDev), hipMemcpyHostToDevice));
        HIPC(c, hipMemcpy(c->d_shade_geoms, shade.data(), nf * sizeof(ShadeGeomDev), hipMemcpyHostToDevice));
        HIPC(c, hipMemcpy(c->d_first_prim, first.data(), nf * 4, hipMemcpyHostToDevice));
        if (nf * sizeof(FlatGeomDev) > (64u << 10)) c->bulk_copies += 3;
        launch_prim_geom(c->stream, c->d_first_prim, (uint32_t)nf, (uint32_t)total, c->d_prim_geom);
        HIPC(c, hipGetLastError());
    }
    c->n_flat_geoms = (uint32_t)nf;
    c->n_flat_prims = (uint32_t)total;
    return RT3_OK;
}

// ---- acceleration structure
int rt3_accel_build(rt3_ctx* c, uint32_t* out_handle) {
    if (!c) return RT3_E_INVALID;
    HIPC(c, hipSetDevice(c->device));
    if (c->n_prims && (!c->d_verts || !c->d_indices)) return fail(c, RT3_E_STATE, "set vertices, indices and geometry before rt3_accel_build");
    // the vertex / index buffers may have been replaced since rt3_scene_set_geometry checked its ranges against them
    if (int r = validate_geometry(c, c->h_geoms.data(), c->h_prim_counts.data(), (uint32_t)c->h_geoms.size())) return r;
    HIPC(c, hipStreamSynchronize(c->stream));
    const auto t_build0 = std::chrono::steady_clock::now();
    if (int r = flatten_world(c)) return r;
    c->accel_built = false;  // until the rebuild has succeeded: a failed one must leave RT3_E_STATE behind, not an empty tree
    dev_free(c->bvh.nodes);
    dev_free(c->bvh.tris);
    dev_free(c->bvh.tri_shade);
    dev_free(c->bvh.tri_uv);
    dev_free(c->bvh.top);
    hipError_t e = lbvh_build(c->stream, c->d_verts, c->d_indices, c->d_geoms, c->d_prim_geom, c->d_first_prim, c->n_flat_prims, c->opt_leaf_size,
                              c->opt_node_width, c->opt_node_quant, c->opt_collapse, c->opt_sah_top, c->opt_sah_device, c->build_arena, &c->bvh);
    if (c->build_arena.cap > ((size_t)1 << 30)) c->build_arena.release();  // a big scene's scratch is not worth keeping resident
    if (e != hipSuccess) return fail(c, RT3_E_HIP, std::string("lbvh_build: ") + hipGetErrorString(e));
    // worst-case stack use of the near-first walk: (children per node - 1) entries per level above the leaves
    const uint32_t stack_need = c->bvh.max_depth > 1 ? (c->opt_node_width - 1) * (c->bvh.max_depth - 1) : 0;
    if (stack_need > kMaxStack) {
        dev_free(c->bvh.nodes);
        dev_free(c->bvh.tris);
        dev_free(c->bvh.tri_shade);
        dev_free(c->bvh.tri_uv);
        dev_free(c->bvh.top);
        return fail(c, RT3_E_DEPTH, "LBVH with " + std::to_string(c->bvh.max_depth) + " levels needs " + std::to_string(stack_need) +
                                        " stack entries, the traversal kernels hold " + std::to_string(kMaxStack));
    }
    HIPC(c, hipStreamSynchronize(c->stream));
    c->stats.accel_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_build0).count();
    c->stats.accel_bulk_copies += c->bulk_copies + c->bvh.bulk_copies;
    c->bulk_copies = 0;
    c->accel_built = true;
    if (out_handle) *out_handle = (RT3_TAG_ACCEL << 30) | 0u;
    return RT3_OK;
}
int rt3_accel_info(rt3_ctx* c, uint32_t* n_nodes, uint32_t* n_tris, uint32_t* max_depth, uint32_t* node_bytes) {
    if (!c || !c->accel_built) return fail(c, RT3_E_STATE, "no acceleration structure built");
    if (n_nodes) *n_nodes = c->bvh.n_nodes;
    if (n_tris) *n_tris = c->bvh.n_tris;
    if (max_depth) *max_depth = c->bvh.max_depth;
    if (node_bytes) *node_bytes = c->bvh.node_bytes;
    return RT3_OK;
}
int rt3_accel_download(rt3_ctx* c, void* nodes, size_t nodes_bytes, void* tris, size_t tris_bytes) {
    if (!c || !c->accel_built) return fail(c, RT3_E_STATE, "no acceleration structure built");
    if (nodes) {
        if (nodes_bytes != (size_t)c->bvh.n_nodes * c->bvh.node_bytes) return fail(c, RT3_E_INVALID, "nodes_bytes mismatch");
        if (nodes_bytes) HIPC(c, hipMemcpy(nodes, c->bvh.nodes, nodes_bytes, hipMemcpyDeviceToHost));
    }
    if (tris) {
        if (tris_bytes != (size_t)c->bvh.n_tris * 48) return fail(c, RT3_E_INVALID, "tris_bytes mismatch");
        if (tris_bytes) HIPC(c, hipMemcpy(tris, c->bvh.tris, tris_bytes, hipMemcpyDeviceToHost));
    }
    return RT3_OK;
}
// The counterpart of rt3_accel_download: install a tree somebody else built over the SAME flattened triangles (an offline builder,
// a cache of an earlier run; Vulkan's vkCmdCopyMemoryToAccelerationStructureKHR plays this role for the reference's driver).  Default
// layout only (64-byte quantised four-wide nodes, 48-byte triangle records).  Every reference is checked on the host before the
// kernels may follow it: in range, no node reachable twice (so the walk terminates), depth within the traversal stack.
int rt3_accel_import(rt3_ctx* c, const void* nodes, size_t nodes_bytes, const void* tris, size_t tris_bytes) {
    if (!c || !nodes || !tris) return fail(c, RT3_E_INVALID, "accel_import: NULL argument");
    if (!c->accel_built) return fail(c, RT3_E_STATE, "accel_import: build the scene's own structure first (rt3_accel_build makes the shading records)");
    if (c->bvh.layout != kLayoutWide64Q) return fail(c, RT3_E_UNSUPPORTED, "accel_import: default node layout only");
    if (nodes_bytes == 0 || nodes_bytes % 64 || tris_bytes % 48 || nodes_bytes / 64 > 0x3FFFFFFFull) return fail(c, RT3_E_INVALID, "accel_import: sizes must be multiples of 64 / 48 bytes");
    const uint32_t nn = (uint32_t)(nodes_bytes / 64), nt = (uint32_t)(tris_bytes / 48);
    const uint32_t* w = static_cast<const uint32_t*>(nodes);
    const uint32_t* tw = static_cast<const uint32_t*>(tris);
    for (uint32_t k = 0; k < nt; k++)
        if (tw[12 * (size_t)k + 9] >= c->n_flat_prims) return fail(c, RT3_E_INVALID, "accel_import: triangle record " + std::to_string(k) + " names a primitive the scene does not have");
    std::vector<uint8_t> seen(nn, 0);
    std::vector<std::pair<uint32_t, uint32_t>> st;  // (node, level)
    st.emplace_back(0u, 1u);
    seen[0] = 1;
    uint32_t max_level = 1;
    while (!st.empty()) {
        const auto [node, level] = st.back();
        st.pop_back();
        max_level = level > max_level ? level : max_level;
        for (int k = 0; k < 4; k++) {
            const uint32_t ref = w[16 * (size_t)node + 10 + k];
            if (ref == 0xFFFFFFFFu) continue;
            if (ref & 0x80000000u) {
                const uint64_t first = ref & 0x0FFFFFFFu, cnt = ((ref >> 28) & 7u) + 1u;
                if (first + cnt > nt) return fail(c, RT3_E_INVALID, "accel_import: node " + std::to_string(node) + " references triangles beyond the array");
            } else {
                if (ref >= nn || seen[ref]) return fail(c, RT3_E_INVALID, "accel_import: node " + std::to_string(node) + " references a node out of range or reachable twice");
                seen[ref] = 1;
                st.emplace_back(ref, level + 1);
            }
        }
    }
    const uint32_t depth = max_level + 1;  // levels from the root to the leaf slots, as lbvh_build counts them
    if (3u * (depth - 1) > kMaxStack) return fail(c, RT3_E_DEPTH, "accel_import: the tree is deeper than the traversal stack supports");
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipStreamSynchronize(c->stream));
    float4 *d_nodes = nullptr, *d_tris = nullptr;
    HIPC(c, hipMalloc(&d_nodes, nodes_bytes));
    hipError_t e = hipMalloc(&d_tris, tris_bytes + 128);  // (the walk over-reads a leaf's last record by up to 128 bytes)
    if (e == hipSuccess) e = hipMemset(d_tris, 0, tris_bytes + 128);
    if (e == hipSuccess) e = hipMemcpy(d_nodes, nodes, nodes_bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess && tris_bytes) e = hipMemcpy(d_tris, tris, tris_bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(d_nodes);
        (void)hipFree(d_tris);
        return fail(c, RT3_E_HIP, std::string("accel_import: ") + hipGetErrorString(e));
    }
    dev_free(c->bvh.nodes);
    dev_free(c->bvh.tris);
    c->bvh.nodes = d_nodes;
    c->bvh.tris = d_tris;
    c->bvh.n_nodes = nn;
    c->bvh.n_tris = nt;
    c->bvh.max_depth = depth;
    e = lbvh_make_top(c->stream, c->bvh.nodes, nn, &c->bvh.top, &c->bvh.n_top);
    if (e != hipSuccess) {
        c->accel_built = false;
        return fail(c, RT3_E_HIP, std::string("accel_import: top-of-tree copy: ") + hipGetErrorString(e));
    }
    return RT3_OK;
}

// ---- resources
int rt3_buffer_create(rt3_ctx* c, size_t bytes, uint32_t* out) {
    if (!c || !out || !bytes) return fail(c, RT3_E_INVALID, "bad buffer size");
    HIPC(c, hipSetDevice(c->device));
    Resource r;
    r.tag = RT3_TAG_BUFFER;
    r.bytes = bytes;
    HIPC(c, hipMalloc(&r.ptr, bytes));
    HIPC(c, hipMemset(r.ptr, 0, bytes));
    c->resources.push_back(r);
    *out = (RT3_TAG_BUFFER << 30) | (uint32_t)(c->resources.size() - 1);
    return RT3_OK;
}
int rt3_image_create(rt3_ctx* c, uint32_t w, uint32_t h, uint32_t format, uint32_t* out) {
    size_t px = format_bytes(format);
    if (!c || !out || !w || !h || !px) return fail(c, RT3_E_INVALID, "bad image size / format");
    HIPC(c, hipSetDevice(c->device));
    Resource r;
    r.tag = RT3_TAG_IMAGE;
    r.w = w; r.h = h; r.format = format;
    r.bytes = (size_t)w * h * px;
    HIPC(c, hipMalloc(&r.ptr, r.bytes));
    HIPC(c, hipMemset(r.ptr, 0, r.bytes));
    c->resources.push_back(r);
    *out = (RT3_TAG_IMAGE << 30) | (uint32_t)(c->resources.size() - 1);
    return RT3_OK;
}
int rt3_image_import(rt3_ctx* c, void* device_ptr, uint32_t w, uint32_t h, uint32_t format, uint32_t* out) {
    size_t px = format_bytes(format);
    if (!c || !out || !device_ptr || !w || !h || !px) return fail(c, RT3_E_INVALID, "bad import");
    Resource r;
    r.tag = RT3_TAG_IMAGE;
    r.w = w; r.h = h; r.format = format;
    r.bytes = (size_t)w * h * px;
    r.ptr = device_ptr;
    r.owned = false;
    c->resources.push_back(r);
    *out = (RT3_TAG_IMAGE << 30) | (uint32_t)(c->resources.size() - 1);
    return RT3_OK;
}
static Resource* any_res(rt3_ctx* c, uint32_t handle) {
    Resource* r = get_res(c, handle, RT3_TAG_IMAGE);
    return r ? r : get_res(c, handle, RT3_TAG_BUFFER);
}
int rt3_resource_upload(rt3_ctx* c, uint32_t handle, const void* src, size_t bytes) {
    if (!c || !src) return RT3_E_INVALID;
    Resource* r = any_res(c, handle);
    if (!r || bytes != r->bytes) return fail(c, RT3_E_INVALID, "upload: bad handle or size");
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipStreamSynchronize(c->stream));
    HIPC(c, hipMemcpy(r->ptr, src, bytes, hipMemcpyHostToDevice));
    return RT3_OK;
}
int rt3_resource_download(rt3_ctx* c, uint32_t handle, void* dst, size_t bytes) {
    if (!c || !dst) return RT3_E_INVALID;
    Resource* r = any_res(c, handle);
    if (!r || bytes != r->bytes) return fail(c, RT3_E_INVALID, "download: bad handle or size");
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipStreamSynchronize(c->stream));
    HIPC(c, hipMemcpy(dst, r->ptr, bytes, hipMemcpyDeviceToHost));
    return RT3_OK;
}
int rt3_resource_device_ptr(rt3_ctx* c, uint32_t handle, void** out_ptr, size_t* out_bytes) {
    if (!c || !out_ptr) return RT3_E_INVALID;
    Resource* r = any_res(c, handle);
    if (!r) return fail(c, RT3_E_INVALID, "bad handle");
    *out_ptr = r->ptr;
    if (out_bytes) *out_bytes = r->bytes;
    return RT3_OK;
}

// ---- tiles
int rt3_set_tile_partition(rt3_ctx* c, uint32_t w, uint32_t h, uint32_t rank, uint32_t n_ranks) {
    if (!c) return RT3_E_INVALID;
    HIPC(c, hipSetDevice(c->device));
    PixelList* pl;
    if (int r = get_pixlist(c, w, h, rank, n_ranks, &pl)) return r;
    c->rank = rank;
    c->n_ranks = n_ranks;
    c->part_w = w;
    c->part_h = h;
    return RT3_OK;
}
int rt3_tile_pixel_count(rt3_ctx* c, uint32_t rank, uint32_t n_ranks, uint32_t* out) {
    if (!c || !out || !c->part_w) return fail(c, RT3_E_STATE, "call rt3_set_tile_partition first");
    PixelList* pl;
    if (int r = get_pixlist(c, c->part_w, c->part_h, rank, n_ranks, &pl)) return r;
    *out = pl->count;
    return RT3_OK;
}
int rt3_image_pack_tiles(rt3_ctx* c, uint32_t image, uint32_t rank, uint32_t n_ranks, void* dst) {
    if (!c || !dst) return RT3_E_INVALID;
    Resource* r = get_res(c, image, RT3_TAG_IMAGE);
    if (!r || format_bytes(r->format) != 16) return fail(c, RT3_E_INVALID, "pack_tiles needs a 16-byte-per-pixel image");
    HIPC(c, hipSetDevice(c->device));
    PixelList* pl;
    if (int e = get_pixlist(c, r->w, r->h, rank, n_ranks, &pl)) return e;
    if (pl->count) launch_pack_tiles(c->stream, pl->dev, pl->count, r->w, r->ptr, dst);
    HIPC(c, hipGetLastError());
    return RT3_OK;
}
int rt3_image_unpack_tiles(rt3_ctx* c, uint32_t image, uint32_t rank, uint32_t n_ranks, const void* src) {
    if (!c || !src) return RT3_E_INVALID;
    Resource* r = get_res(c, image, RT3_TAG_IMAGE);
    if (!r || format_bytes(r->format) != 16) return fail(c, RT3_E_INVALID, "unpack_tiles needs a 16-byte-per-pixel image");
    HIPC(c, hipSetDevice(c->device));
    PixelList* pl;
    if (int e = get_pixlist(c, r->w, r->h, rank, n_ranks, &pl)) return e;
    if (pl->count) launch_unpack_tiles(c->stream, pl->dev, pl->count, r->w, src, r->ptr);
    HIPC(c, hipGetLastError());
    return RT3_OK;
}

// ---- frame-end gather over RCCL (north_star).  One rank per context: ncclCommInitRank from a unique id the host
//      application carries from rank 0 to the others over whatever channel it has (the ABI never opens a socket itself).
#define NCCLC(ctx, call)                                                                                          \
    do {                                                                                                          \
        ncclResult_t e_ = (call);                                                                                 \
        if (e_ != ncclSuccess) return fail(ctx, RT3_E_COMM, std::string(#call) + ": " + ncclGetErrorString(e_));  \
    } while (0)
static_assert(sizeof(ncclUniqueId) == RT3_COMM_ID_BYTES, "RT3_COMM_ID_BYTES must be sizeof(ncclUniqueId)");

// A failed send / receive leaves a half-posted exchange behind: peers would block on operations that are never matched and the next
// gather on this communicator would hang with them.  Abort it (ncclCommAbort also ends an open group) and drop it, so that the next
// call answers RT3_E_STATE instead; the host then decides (bench.py: the run fails, nothing is reported as measured).
static int comm_abort(rt3_ctx* c, const std::string& what) {
    if (c->comm) {
        (void)ncclCommAbort(c->comm);
        c->comm = nullptr;
        c->comm_size = 0;
    }
    return fail(c, RT3_E_COMM, what + " (communicator aborted)");
}
static int get_gather_layout(rt3_ctx* c, uint32_t w, uint32_t h, uint32_t root, uint32_t n_ranks, GatherLayout** out) {
    for (auto& g : c->gather_layouts)
        if (g.w == w && g.h == h && g.root == root && g.n_ranks == n_ranks) {
            *out = &g;
            return RT3_OK;
        }
    if (w == 0 || h == 0 || w > 65535 || h > 65535 || n_ranks == 0 || root >= n_ranks) return fail(c, RT3_E_INVALID, "bad window / root / rank count for the gather");
    GatherLayout gl;
    gl.w = w; gl.h = h; gl.root = root; gl.n_ranks = n_ranks;
    gl.off.assign((size_t)n_ranks + 1, 0);
    std::vector<uint32_t> all, px;
    for (uint32_t r = 0; r < n_ranks; r++) {
        gl.off[r] = all.size();
        if (r == root) continue;  // the root's tiles never leave its image
        tile_pixels(w, h, r, n_ranks, px);
        all.insert(all.end(), px.begin(), px.end());
    }
    gl.off[n_ranks] = all.size();
    HIPC(c, hipMalloc((void**)&gl.dev, (all.size() ? all.size() : 1) * 4));
    if (!all.empty()) {
        hipError_t e = hipMemcpy(gl.dev, all.data(), all.size() * 4, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(gl.dev);  // not yet owned by the context's layout cache
            return fail(c, RT3_E_HIP, std::string("gather layout upload: ") + hipGetErrorString(e));
        }
    }
    c->gather_layouts.push_back(gl);
    *out = &c->gather_layouts.back();
    return RT3_OK;
}
static int ensure_gather_buf(rt3_ctx* c, size_t bytes) {
    if (bytes <= c->gather_buf_bytes) return RT3_OK;
    HIPC(c, hipStreamSynchronize(c->stream));  // an earlier gather may still be reading the old buffer
    dev_free(c->gather_buf);
    c->gather_buf_bytes = 0;
    HIPC(c, hipMalloc(&c->gather_buf, bytes));
    c->gather_buf_bytes = bytes;
    return RT3_OK;
}

int rt3_comm_unique_id(void* id_out) {
    if (!id_out) return fail(nullptr, RT3_E_INVALID, "id_out is NULL");
    ncclUniqueId id;
    ncclResult_t e = ncclGetUniqueId(&id);
    if (e != ncclSuccess) return fail(nullptr, RT3_E_COMM, std::string("ncclGetUniqueId: ") + ncclGetErrorString(e));
    memcpy(id_out, &id, sizeof(id));
    return RT3_OK;
}
int rt3_comm_version(int* out) {  // ncclGetVersion: major * 10000 + minor * 100 + patch (RCCL reports the NCCL API level it implements)
    if (!out) return RT3_E_INVALID;
    ncclResult_t e = ncclGetVersion(out);
    return e == ncclSuccess ? RT3_OK : fail(nullptr, RT3_E_COMM, std::string("ncclGetVersion: ") + ncclGetErrorString(e));
}
int rt3_comm_init(rt3_ctx* c, const void* id, uint32_t rank, uint32_t n_ranks) {
    if (!c || !id) return fail(c, RT3_E_INVALID, "comm_init: NULL argument");
    if (n_ranks == 0 || rank >= n_ranks) return fail(c, RT3_E_INVALID, "comm_init: rank must be < n_ranks");
    if (c->comm) return fail(c, RT3_E_STATE, "comm_init: this context already has a communicator (rt3_comm_destroy first)");
    HIPC(c, hipSetDevice(c->device));
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    NCCLC(c, ncclCommInitRank(&c->comm, (int)n_ranks, uid, (int)rank));
    c->comm_rank = rank;
    c->comm_size = n_ranks;
    return RT3_OK;
}
int rt3_comm_destroy(rt3_ctx* c) {
    if (!c) return RT3_E_INVALID;
    if (!c->comm) return RT3_OK;
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipStreamSynchronize(c->stream));
    ncclComm_t comm = c->comm;
    c->comm = nullptr;
    c->comm_size = 0;
    NCCLC(c, ncclCommDestroy(comm));
    return RT3_OK;
}
int rt3_gather_layout(rt3_ctx* c, uint32_t image, uint32_t root, uint32_t n_ranks, uint64_t* offsets) {
    if (!c || !offsets) return fail(c, RT3_E_INVALID, "gather_layout: NULL argument");
    Resource* r = get_res(c, image, RT3_TAG_IMAGE);
    if (!r || format_bytes(r->format) != 16) return fail(c, RT3_E_INVALID, "the gather needs a 16-byte-per-pixel image");
    HIPC(c, hipSetDevice(c->device));
    GatherLayout* gl;
    if (int e = get_gather_layout(c, r->w, r->h, root, n_ranks, &gl)) return e;
    memcpy(offsets, gl->off.data(), ((size_t)n_ranks + 1) * sizeof(uint64_t));
    return RT3_OK;
}
// the root's half of the gather without the exchange: `recv_device` is laid out as rt3_gather_layout says
int rt3_gather_unpack(rt3_ctx* c, uint32_t image, uint32_t root, uint32_t n_ranks, const void* recv_device) {
    if (!c || !recv_device) return fail(c, RT3_E_INVALID, "gather_unpack: NULL argument");
    Resource* r = get_res(c, image, RT3_TAG_IMAGE);
    if (!r || format_bytes(r->format) != 16) return fail(c, RT3_E_INVALID, "the gather needs a 16-byte-per-pixel image");
    HIPC(c, hipSetDevice(c->device));
    GatherLayout* gl;
    if (int e = get_gather_layout(c, r->w, r->h, root, n_ranks, &gl)) return e;
    const uint64_t total = gl->off[n_ranks];
    if (total) launch_unpack_tiles(c->stream, gl->dev, (uint32_t)total, r->w, recv_device, r->ptr);  // ONE launch for all ranks
    HIPC(c, hipGetLastError());
    return RT3_OK;
}
int rt3_gather_tiles(rt3_ctx* c, uint32_t image, uint32_t root) {
    if (!c) return RT3_E_INVALID;
    if (!c->comm) return fail(c, RT3_E_STATE, "gather_tiles: call rt3_comm_init first");
    Resource* r = get_res(c, image, RT3_TAG_IMAGE);
    if (!r || format_bytes(r->format) != 16) return fail(c, RT3_E_INVALID, "the gather needs a 16-byte-per-pixel image");
    const uint32_t n = c->comm_size, me = c->comm_rank;
    if (root >= n) return fail(c, RT3_E_INVALID, "gather_tiles: root must be < n_ranks");
    if (c->n_ranks != n || c->rank != me)
        return fail(c, RT3_E_STATE, "gather_tiles: the tile partition (rt3_set_tile_partition) and the communicator disagree on rank / n_ranks");
    if (n == 1) return RT3_OK;  // the frame is already whole
    HIPC(c, hipSetDevice(c->device));
    if (me != root) {
        PixelList* pl;
        if (int e = get_pixlist(c, r->w, r->h, me, n, &pl)) return e;
        if (pl->count == 0) return RT3_OK;  // (the root skips empty ranks too)
        if (int e = ensure_gather_buf(c, (size_t)pl->count * 16)) return e;
        {
            ScopedTimer t(c, CAT_OTHER);
            launch_pack_tiles(c->stream, pl->dev, pl->count, r->w, r->ptr, c->gather_buf);
        }
        HIPC(c, hipGetLastError());
        ScopedTimer t(c, CAT_GATHER);
        ncclResult_t se = ncclSend(c->gather_buf, (size_t)pl->count * 4, ncclFloat, (int)root, c->comm, c->stream);
        if (se != ncclSuccess) return comm_abort(c, std::string("ncclSend: ") + ncclGetErrorString(se));
        return RT3_OK;
    }
    GatherLayout* gl;
    if (int e = get_gather_layout(c, r->w, r->h, root, n, &gl)) return e;
    const uint64_t total = gl->off[n];
    if (total == 0) return RT3_OK;
    if (int e = ensure_gather_buf(c, (size_t)total * 16)) return e;
    {
        // exact per-rank counts at exact offsets, every peer's recv in ONE group = one gather; xGMI is point to point, so the
        // root's inbound links run concurrently and nothing is forwarded (a ring would move (n-1) x the bytes)
        ScopedTimer t(c, CAT_GATHER);
        NCCLC(c, ncclGroupStart());
        for (uint32_t p = 0; p < n; p++) {
            const uint64_t cnt = gl->off[p + 1] - gl->off[p];
            if (p == root || cnt == 0) continue;
            ncclResult_t e = ncclRecv((char*)c->gather_buf + gl->off[p] * 16, (size_t)cnt * 4, ncclFloat, (int)p, c->comm, c->stream);
            if (e != ncclSuccess) return comm_abort(c, std::string("ncclRecv: ") + ncclGetErrorString(e));
        }
        ncclResult_t ge = ncclGroupEnd();
        if (ge != ncclSuccess) return comm_abort(c, std::string("ncclGroupEnd: ") + ncclGetErrorString(ge));
    }
    ScopedTimer t(c, CAT_OTHER);
    launch_unpack_tiles(c->stream, gl->dev, (uint32_t)total, r->w, c->gather_buf, r->ptr);  // stream-ordered behind the receives
    HIPC(c, hipGetLastError());
    return RT3_OK;
}

// ---- pass launch
int rt3_pass_launch(rt3_ctx* c, const char* pass_name, const char* entry, uint32_t x, uint32_t y, uint32_t z, const void* constants,
                    size_t constants_size, const uint32_t* bindings, uint32_t n_bindings) {
    if (!c || !pass_name) return fail(c, RT3_E_INVALID, "pass_name NULL");
    if (entry && strcmp(entry, "main") != 0) return fail(c, RT3_E_INVALID, std::string("unknown entry point '") + entry + "' (the reference passes use \"main\")");
    if (!constants || constants_size != sizeof(rt3_gconst)) return fail(c, RT3_E_INVALID, "constants must be the 304-byte GConst block");
    if (!bindings && n_bindings) return fail(c, RT3_E_INVALID, "bindings NULL");
    if (!c->accel_built) return fail(c, RT3_E_STATE, "rt3_accel_build has not been called for the current scene");
    HIPC(c, hipSetDevice(c->device));
    if (int r = sync_textures(c)) return r;
    if (c->max_tex_index >= (int64_t)c->h_tex.size())
        return fail(c, RT3_E_STATE, "a geometry references base-colour texture " + std::to_string(c->max_tex_index) + " but only " +
                                        std::to_string(c->h_tex.size()) + " texture(s) were set (rt3_scene_set_texture)");
    rt3_gconst g;
    memcpy(&g, constants, sizeof(g));
    if (!strcmp(pass_name, "gbuffer")) return pass_gbuffer(c, &g, x, y, bindings, n_bindings);
    if (!strcmp(pass_name, "refrence_mode")) return pass_reference_mode(c, &g, x, y, bindings, n_bindings);
    if (!strcmp(pass_name, "postprocess")) return pass_postprocess(c, &g, x, y, z, bindings, n_bindings);
    if (!strcmp(pass_name, "structured_importance_sampling")) return pass_sis(c, &g, x, y, z, bindings, n_bindings);
    if (!strcmp(pass_name, "trace_probes")) return pass_trace_probes(c, &g, x, y, bindings, n_bindings);
    if (!strcmp(pass_name, "spherical_harmonic_conversion")) return pass_sh_conversion(c, x, y, z, bindings, n_bindings);
    if (!strcmp(pass_name, "interpolate_probes")) return pass_interpolate_probes(c, &g, x, y, z, bindings, n_bindings);
    return fail(c, RT3_E_INVALID, std::string("unknown pass '") + pass_name +
                                      "' (known: gbuffer, refrence_mode, postprocess, structured_importance_sampling, trace_probes, "
                                      "spherical_harmonic_conversion, interpolate_probes)");
}
int rt3_frame_wait(rt3_ctx* c) {
    if (!c) return RT3_E_INVALID;
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipStreamSynchronize(c->stream));
    return RT3_OK;
}

// ---- ray batches
int rt3_trace_rays(rt3_ctx* c, const float* rays, uint32_t n, int any_hit, float* t, float* u, float* v, uint32_t* prim, uint32_t* n_nodes,
                   uint32_t* n_tris, int repeat, double* kernel_ms) {
    if (!c || !rays || !prim || (!any_hit && (!t || !u || !v))) return fail(c, RT3_E_INVALID, "trace_rays: NULL argument");
    if (!c->accel_built) return fail(c, RT3_E_STATE, "rt3_accel_build has not been called for the current scene");
    if (n == 0) return RT3_OK;
    HIPC(c, hipSetDevice(c->device));
    float *d_rays = nullptr, *d_hits = nullptr;
    uint32_t *d_cn = nullptr, *d_ct = nullptr, *d_occ = nullptr, *d_cur = nullptr;
    const bool count = n_nodes || n_tris;
    int rc = RT3_OK;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto cleanup = [&]() {
        (void)hipFree(d_rays); (void)hipFree(d_hits); (void)hipFree(d_cn); (void)hipFree(d_ct); (void)hipFree(d_occ); (void)hipFree(d_cur);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    };
#define TR(call)                                                                            \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            rc = fail(c, RT3_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_));     \
            cleanup();                                                                      \
            return rc;                                                                      \
        }                                                                                   \
    } while (0)
    TR(hipMalloc((void**)&d_rays, (size_t)n * 32));
    TR(hipMalloc((void**)&d_hits, (size_t)n * 16));
    TR(hipMalloc((void**)&d_occ, (size_t)n * 4));
    TR(hipMalloc((void**)&d_cur, 4));
    if (count) {
        TR(hipMalloc((void**)&d_cn, (size_t)n * 4));
        TR(hipMalloc((void**)&d_ct, (size_t)n * 4));
    }
    {  // host SoA (ox..tmax) -> device records {o.xyz, tmin} x n, {d.xyz, tmax} x n
        std::vector<float> rec((size_t)n * 8);
        for (uint32_t i = 0; i < n; i++) {
            float* a = &rec[4 * (size_t)i];
            float* b = &rec[4 * ((size_t)n + i)];
            a[0] = rays[i]; a[1] = rays[(size_t)n + i]; a[2] = rays[2 * (size_t)n + i]; a[3] = rays[6 * (size_t)n + i];
            b[0] = rays[3 * (size_t)n + i]; b[1] = rays[4 * (size_t)n + i]; b[2] = rays[5 * (size_t)n + i]; b[3] = rays[7 * (size_t)n + i];
        }
        TR(hipMemcpy(d_rays, rec.data(), (size_t)n * 32, hipMemcpyHostToDevice));
    }
    TR(hipEventCreate(&e0));
    TR(hipEventCreate(&e1));
    if (repeat < 1) repeat = 1;
    auto launch = [&]() {
        (void)hipMemsetAsync(d_cur, 0, 4, c->stream);  // ray-pool cursor
        if (any_hit)
            launch_shadow(c->stream, count, c->bvh.layout, c->bvh.nodes, c->bvh.tris, c->bvh.top, c->bvh.n_top, d_rays, n, nullptr, n, n, nullptr, nullptr, nullptr, 0, d_occ, d_cn, d_ct, nullptr, d_cur);
        else
            launch_extend(c->stream, count, c->bvh.layout, c->bvh.nodes, c->bvh.tris, c->bvh.top, c->bvh.n_top, d_rays, n, nullptr, n, n, d_hits, d_cn, d_ct, nullptr, d_cur);
    };
    launch();  // warm-up (also the result-producing launch)
    TR(hipEventRecord(e0, c->stream));
    for (int k = 0; k < repeat; k++) launch();
    TR(hipEventRecord(e1, c->stream));
    TR(hipGetLastError());
    TR(hipStreamSynchronize(c->stream));
    float ms = 0.0f;
    TR(hipEventElapsedTime(&ms, e0, e1));
    if (kernel_ms) *kernel_ms = (double)ms / repeat;
    if (any_hit) {
        TR(hipMemcpy(prim, d_occ, (size_t)n * 4, hipMemcpyDeviceToHost));
    } else {
        std::vector<float> rec((size_t)n * 4);  // device hits are {t, u, v, prim} records
        TR(hipMemcpy(rec.data(), d_hits, (size_t)n * 16, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < n; i++) {
            t[i] = rec[4 * (size_t)i];
            u[i] = rec[4 * (size_t)i + 1];
            v[i] = rec[4 * (size_t)i + 2];
            memcpy(&prim[i], &rec[4 * (size_t)i + 3], 4);
        }
    }
    if (n_nodes) TR(hipMemcpy(n_nodes, d_cn, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (n_tris) TR(hipMemcpy(n_tris, d_ct, (size_t)n * 4, hipMemcpyDeviceToHost));
#undef TR
    cleanup();
    return RT3_OK;
}

int rt3_selftest_eval(rt3_ctx* c, int op, const void* in, uint32_t n, void* out) {
    uint32_t iw, ow;
    if (!c || !in || !out || !selftest_widths(op, &iw, &ow)) return fail(c, RT3_E_INVALID, "selftest: bad op / NULL");
    if (n == 0) return RT3_OK;
    HIPC(c, hipSetDevice(c->device));
    uint32_t *d_in = nullptr, *d_out = nullptr;
    HIPC(c, hipMalloc((void**)&d_in, (size_t)n * iw * 4));
    if (hipMalloc((void**)&d_out, (size_t)n * ow * 4) != hipSuccess) {
        (void)hipFree(d_in);
        return fail(c, RT3_E_HIP, "selftest: hipMalloc failed");
    }
    hipError_t e = hipMemcpy(d_in, in, (size_t)n * iw * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        launch_selftest(c->stream, op, d_in, n, d_out);
        e = hipStreamSynchronize(c->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(out, d_out, (size_t)n * ow * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(c, RT3_E_HIP, std::string("selftest: ") + hipGetErrorString(e));
    return RT3_OK;
}

int rt3_stats_reset(rt3_ctx* c) {
    if (!c) return RT3_E_INVALID;
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipStreamSynchronize(c->stream));
    if (int r = harvest(c)) return r;
    memset(&c->stats, 0, sizeof(c->stats));
    return RT3_OK;
}
int rt3_stats_get(rt3_ctx* c, rt3_stats* out) {
    if (!c || !out) return RT3_E_INVALID;
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipStreamSynchronize(c->stream));
    if (int r = harvest(c)) return r;
    *out = c->stats;
    return RT3_OK;
}

// ---- camera: components/camera.rs:52-58 (glam look_at_rh / perspective_rh with depth 0..1) + renderer/mod.rs:72-78
static void invert4(const float* m, float* out) {
    double a[4][4], inv[4][4];
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++) a[r][c] = m[c * 4 + r];
    // adjugate through 3x3 minors
    auto minor3 = [&](int rr, int cc) {
        double s[3][3];
        int ri = 0;
        for (int r = 0; r < 4; r++) {
            if (r == rr) continue;
            int ci = 0;
            for (int c = 0; c < 4; c++) {
                if (c == cc) continue;
                s[ri][ci++] = a[r][c];
            }
            ri++;
        }
        return s[0][0] * (s[1][1] * s[2][2] - s[1][2] * s[2][1]) - s[0][1] * (s[1][0] * s[2][2] - s[1][2] * s[2][0]) +
               s[0][2] * (s[1][0] * s[2][1] - s[1][1] * s[2][0]);
    };
    double det = 0.0;
    for (int c = 0; c < 4; c++) det += ((c & 1) ? -1.0 : 1.0) * a[0][c] * minor3(0, c);
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) inv[c][r] = (((r + c) & 1) ? -1.0 : 1.0) * minor3(r, c) / det;
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++) out[c * 4 + r] = (float)inv[r][c];
}
void rt3_camera_gconst(const float position[3], const float direction[3], float fov_y, float aspect, float z_near, float z_far, float width,
                       float height, rt3_gconst* g) {
    memset(g, 0, sizeof(*g));
    float len = std::sqrt(direction[0] * direction[0] + direction[1] * direction[1] + direction[2] * direction[2]);
    float f[3] = {direction[0] / len, direction[1] / len, direction[2] / len};  // Camera::new normalises, camera.rs:43
    // look_to_rh(eye, dir, up=+Y): s = normalize(f x up), u = s x f
    float s[3] = {f[1] * 0.0f - f[2] * 1.0f, f[2] * 0.0f - f[0] * 0.0f, f[0] * 1.0f - f[1] * 0.0f};
    float sl = std::sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
    s[0] /= sl; s[1] /= sl; s[2] /= sl;
    float u[3] = {s[1] * f[2] - s[2] * f[1], s[2] * f[0] - s[0] * f[2], s[0] * f[1] - s[1] * f[0]};
    float* v = g->view;
    v[0] = s[0]; v[1] = u[0]; v[2] = -f[0];
    v[4] = s[1]; v[5] = u[1]; v[6] = -f[1];
    v[8] = s[2]; v[9] = u[2]; v[10] = -f[2];
    v[12] = -(position[0] * s[0] + position[1] * s[1] + position[2] * s[2]);
    v[13] = -(position[0] * u[0] + position[1] * u[1] + position[2] * u[2]);
    v[14] = position[0] * f[0] + position[1] * f[1] + position[2] * f[2];
    v[15] = 1.0f;
    float sf = (float)std::sin(0.5 * (double)fov_y), cf = (float)std::cos(0.5 * (double)fov_y);
    float hh = cf / sf, ww = hh / aspect, r = z_far / (z_near - z_far);
    g->proj[0] = ww;
    g->proj[5] = hh;
    g->proj[10] = r;
    g->proj[11] = -1.0f;
    g->proj[14] = r * z_near;
    invert4(g->proj, g->proj_inverse);
    invert4(g->view, g->view_inverse);
    g->window_size[0] = width;
    g->window_size[1] = height;
    g->blendfactor = 1.0f;
}

}  // extern "C"
