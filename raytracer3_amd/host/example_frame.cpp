// example_frame.cpp -- a compiled host driving librt3.so through the C++ mirror of the reference pass graph.
// It is the analogue of `renderer::commands` (src/renderer/mod.rs:65-106) for the three path-tracing passes:
// reads a scene dump, describes one frame with the builder chain, runs it and writes Light (RGBA32F) + colour.
//
//   example_frame scene.bin W H spp bounces flags frame out.bin
// scene.bin: u32 n_verts, n_idx, n_geoms, sky_w, sky_h, bn_w, bn_h, pad | verts (n*8 f32) | indices (u32) |
//            geometry infos (64 B each) | prim counts (u32) | sky rgb f32 | blue noise rgba8 | camera: pos[3] dir[3] fov aspect (f32)
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "render_graph.hpp"

template <class T>
static std::vector<T> read_vec(FILE* f, size_t n) {
    std::vector<T> v(n);
    if (n && fread(v.data(), sizeof(T), n, f) != n) throw std::runtime_error("short read");
    return v;
}

int main(int argc, char** argv) {
    if (argc != 9) {
        fprintf(stderr, "usage: %s scene.bin W H spp bounces flags frame out.bin\n", argv[0]);
        return 2;
    }
    try {
        FILE* f = fopen(argv[1], "rb");
        if (!f) throw std::runtime_error("cannot open scene");
        auto hdr = read_vec<uint32_t>(f, 8);
        auto verts = read_vec<float>(f, (size_t)hdr[0] * 8);
        auto idx = read_vec<uint32_t>(f, hdr[1]);
        auto geoms = read_vec<rt3_geometry_info>(f, hdr[2]);
        auto counts = read_vec<uint32_t>(f, hdr[2]);
        auto sky = read_vec<float>(f, (size_t)hdr[3] * hdr[4] * 3);
        auto bn = read_vec<uint8_t>(f, (size_t)hdr[5] * hdr[6] * 4);
        auto cam = read_vec<float>(f, 8);
        fclose(f);
        const uint32_t W = atoi(argv[2]), H = atoi(argv[3]);

        rt3::Context ctx(0);
        ctx.check(rt3_set_tile_partition(ctx.raw(), W, H, 0, 1), "partition");
        ctx.check(rt3_scene_set_vertices(ctx.raw(), verts.data(), hdr[0]), "vertices");  // DynamicBuffer::push x3, world/mod.rs:83-101
        ctx.check(rt3_scene_set_indices(ctx.raw(), idx.data(), hdr[1]), "indices");
        ctx.check(rt3_scene_set_geometry(ctx.raw(), geoms.data(), counts.data(), hdr[2]), "geometry");
        if (!sky.empty()) ctx.check(rt3_scene_set_sky(ctx.raw(), sky.data(), hdr[3], hdr[4]), "sky");
        if (!bn.empty()) ctx.check(rt3_scene_set_bluenoise(ctx.raw(), bn.data(), hdr[5], hdr[6]), "bluenoise");
        uint32_t tlas = 0;
        ctx.check(rt3_accel_build(ctx.raw(), &tlas), "accel");  // create_acceleration_structure, raytracing.rs:88-148

        rt3::Camera camera{{cam[0], cam[1], cam[2]}, {cam[3], cam[4], cam[5]}, cam[6], cam[7]};
        rt3_gconst gconst = camera.gconst(W, H);
        gconst.samples = atoi(argv[4]);
        gconst.bounces = atoi(argv[5]);
        gconst.pad[0] = (uint32_t)atoi(argv[6]);
        gconst.frame = (uint32_t)atoi(argv[7]);
        gconst.blendfactor = 1.0f;

        rt3::RenderGraph rg(ctx, W, H);
        rg.begin_frame();
        auto gbuffer = rg.image(rt3::ImageSize::FullScreen(), RT3_FORMAT_R32G32B32A32_UINT, "gbuffer");
        auto depth = rg.image(rt3::ImageSize::FullScreen(), RT3_FORMAT_R32_SFLOAT, "gbuffer_depth");
        auto light = rg.image(rt3::ImageSize::FullScreen(), RT3_FORMAT_R32G32B32A32_SFLOAT, "Light");
        auto prev = rg.image(rt3::ImageSize::FullScreen(), RT3_FORMAT_R32G32B32A32_SFLOAT, "PrevLight");
        auto color = rg.image(rt3::ImageSize::FullScreen(), RT3_FORMAT_R32G32B32A32_SFLOAT, "color");
        auto gb = rt3::RayTracingPass::New(rg, "gbuffer").shader("gbuffer").constants(gconst)
                      .write(rt3::IMPORTED, gbuffer).write(rt3::IMPORTED, depth).launch(rt3::WorkSize2D::FullScreen());
        auto pt = rt3::RayTracingPass::New(rg, "refrence_mode").shader("refrence_mode").constants(gconst)
                      .read(gb, gbuffer).read(gb, depth).write(rt3::IMPORTED, light).read(rt3::IMPORTED, prev).launch(rt3::WorkSize2D::FullScreen());
        rt3::ComputePass::New(rg, "postprocess").shader("postprocess").constants(gconst)
            .read(gb, depth).write(rt3::IMPORTED, color).read(pt, light).dispatch(rt3::DispatchSize::FullScreen());
        rg.draw_frame(color);

        std::vector<float> out((size_t)W * H * 4), col((size_t)W * H * 4);
        ctx.check(rt3_resource_download(ctx.raw(), light, out.data(), out.size() * 4), "download");
        ctx.check(rt3_resource_download(ctx.raw(), color, col.data(), col.size() * 4), "download");
        FILE* o = fopen(argv[8], "wb");
        fwrite(out.data(), 4, out.size(), o);
        fwrite(col.data(), 4, col.size(), o);
        fclose(o);
        rt3_stats st;
        ctx.check(rt3_stats_get(ctx.raw(), &st), "stats");
        printf("example_frame: %ux%u, %llu extension + %llu shadow rays, %llu frames drawn\n", W, H, (unsigned long long)st.extension_rays,
               (unsigned long long)st.shadow_rays, (unsigned long long)rg.frame_number);
    } catch (const std::exception& e) {
        fprintf(stderr, "example_frame: %s\n", e.what());
        return 1;
    }
    return 0;
}
