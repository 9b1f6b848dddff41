// example_frame.cpp -- a compiled host driving librt3.so through the C++ mirror of the reference pass graph.
// It is the analogue of `renderer::commands` (src/renderer/mod.rs:65-106) for the three path-tracing passes:
// reads a scene dump, describes one frame with the builder chain, runs it and writes Light (RGBA32F) + colour.
//
//   example_frame scene.bin W H spp bounces flags frame out.bin [probes]
// With the trailing word `probes` the frame is the probe-GI chain of the old shaders instead (gbuffer ->
// structured_importance_sampling -> trace_probes -> spherical_harmonic_conversion -> interpolate_probes) and
// out.bin holds Light followed by the probe atlas.
// Multi-GPU (one process per GPU, like `bench.py --gpus N`): with RT3_RANKS = n > 0 in the environment this process is rank RT3_RANK of n on
// device RT3_DEVICE (default: the rank), renders its 64x64 tiles and joins the frame's ONE collective, rt3_gather_tiles; rank 0 creates the
// RCCL id and hands it to the others through the file RT3_UID_FILE (the C ABI opens no channel of its own) and writes out.bin.
// scene.bin: u32 n_verts, n_idx, n_geoms, sky_w, sky_h, bn_w, bn_h, pad | verts (n*8 f32) | indices (u32) |
//            geometry infos (64 B each) | prim counts (u32) | sky rgb f32 | blue noise rgba8 | camera: pos[3] dir[3] fov aspect (f32)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "render_graph.hpp"

template <class T>
static std::vector<T> read_vec(FILE* f, size_t n) {
    std::vector<T> v(n);
    if (n && fread(v.data(), sizeof(T), n, f) != n) throw std::runtime_error("short read");
    return v;
}

int main(int argc, char** argv) {
    const bool probes = argc == 10 && std::string(argv[9]) == "probes";
    if (argc != 9 && !probes) {
        fprintf(stderr, "usage: %s scene.bin W H spp bounces flags frame out.bin [probes]\n", argv[0]);
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

        const uint32_t n_ranks = getenv("RT3_RANKS") ? (uint32_t)atoi(getenv("RT3_RANKS")) : 0u;  // 0: single process, no communicator
        const uint32_t rank = n_ranks && getenv("RT3_RANK") ? (uint32_t)atoi(getenv("RT3_RANK")) : 0u;
        rt3::Context ctx(n_ranks ? (getenv("RT3_DEVICE") ? atoi(getenv("RT3_DEVICE")) : (int)rank) : 0);
        ctx.check(rt3_set_tile_partition(ctx.raw(), W, H, rank, n_ranks ? n_ranks : 1u), "partition");
        if (n_ranks) {  // rank 0's id travels over whatever channel the host has -- here a file
            const char* uid_file = getenv("RT3_UID_FILE");
            if (!uid_file) throw std::runtime_error("RT3_UID_FILE is not set");
            unsigned char id[RT3_COMM_ID_BYTES];
            // The file starts with a nonce the launcher gives every rank of ONE run (RT3_UID_NONCE): a file left behind by an earlier run
            // carries another nonce and is ignored, instead of handing the ranks a dead ncclUniqueId to hang on.
            const char* nonce_s = getenv("RT3_UID_NONCE");
            const unsigned long long nonce = nonce_s ? strtoull(nonce_s, nullptr, 10) : 0ull;
            if (rank == 0) {
                if (rt3_comm_unique_id(id)) throw std::runtime_error(std::string("rt3_comm_unique_id: ") + rt3_last_error(nullptr));
                const std::string tmp = std::string(uid_file) + ".tmp";
                FILE* u = fopen(tmp.c_str(), "wb");
                if (!u || fwrite(&nonce, 1, sizeof(nonce), u) != sizeof(nonce) || fwrite(id, 1, sizeof(id), u) != sizeof(id)) throw std::runtime_error("cannot write the id file");
                fclose(u);
                if (rename(tmp.c_str(), uid_file)) throw std::runtime_error("cannot publish the id file");
            } else {
                bool got = false;
                for (int tries = 0; tries < 1200 && !got; tries++) {
                    unsigned long long seen = ~nonce;
                    if (FILE* u = fopen(uid_file, "rb")) {
                        got = fread(&seen, 1, sizeof(seen), u) == sizeof(seen) && seen == nonce && fread(id, 1, sizeof(id), u) == sizeof(id);
                        fclose(u);
                    }
                    if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(50));
                }
                if (!got) throw std::runtime_error("no id file with this run's nonce appeared within 60 s");
            }
            ctx.check(rt3_comm_init(ctx.raw(), id, rank, n_ranks), "comm init");  // collective: ncclCommInitRank on this context's device
        }
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
        if (probes) {
            const uint32_t px = W / 16, py = H / 16;
            auto explode = [](uint32_t v) {  // ZCurveToLinearIndex, math.slang:105-117
                v = (v | (v << 8)) & 0x00FF00FFu;
                v = (v | (v << 4)) & 0x0F0F0F0Fu;
                v = (v | (v << 2)) & 0x33333333u;
                return (v | (v << 1)) & 0x55555555u;
            };
            const size_t sh_bytes = 48 * (size_t)((explode(px * 3 - 1) | (explode(py - 1) << 1)) + 1);
            const auto asize = rt3::ImageSize::XY(px * 8, py * 8);
            auto gbuffer = rg.image(rt3::ImageSize::FullScreen(), RT3_FORMAT_R32G32B32A32_UINT, "gbuffer");
            auto depth = rg.image(rt3::ImageSize::FullScreen(), RT3_FORMAT_R32_SFLOAT, "gbuffer_depth");
            auto light = rg.image(rt3::ImageSize::FullScreen(), RT3_FORMAT_R32G32B32A32_SFLOAT, "Light");
            auto directions = rg.image(asize, RT3_FORMAT_R16_UINT, "probe_directions");
            auto debug = rg.image(asize, RT3_FORMAT_R32_SFLOAT, "probe_debug");
            auto atlas = rg.image(asize, RT3_FORMAT_R32G32B32A32_SFLOAT, "probe_atlas");
            auto prev_atlas = rg.image(asize, RT3_FORMAT_R32G32B32A32_SFLOAT, "prev_probe_atlas");
            auto sh = rg.buffer(sh_bytes, "sh_coeficents");
            auto gb = rt3::RayTracingPass::New(rg, "gbuffer").shader("gbuffer").constants(gconst)
                          .write(rt3::IMPORTED, gbuffer).write(rt3::IMPORTED, depth).launch(rt3::WorkSize2D::FullScreen());
            auto sis = rt3::ComputePass::New(rg, "structured_importance_sampling").shader("structured_importance_sampling").constants(gconst)
                           .read(gb, gbuffer).read(gb, depth).write(rt3::IMPORTED, directions).write(rt3::IMPORTED, debug).read(rt3::IMPORTED, atlas)
                           .dispatch(rt3::DispatchSize::XY(px, py));
            auto tp = rt3::RayTracingPass::New(rg, "trace_probes").shader("trace_probes").constants(gconst)
                          .read(gb, gbuffer).read(gb, depth).read(sis, directions).write(rt3::IMPORTED, atlas).read(rt3::IMPORTED, prev_atlas)
                          .launch(rt3::WorkSize2D::XY(px * 8, py * 8));
            auto shc = rt3::ComputePass::New(rg, "spherical_harmonic_conversion").shader("spherical_harmonic_conversion").constants(gconst)
                           .write(rt3::IMPORTED, sh).read(tp, atlas).dispatch(rt3::DispatchSize::XY(px, py));
            rt3::ComputePass::New(rg, "interpolate_probes").shader("interpolate_probes").constants(gconst)
                .read(gb, gbuffer).read(gb, depth).read(shc, sh).write(rt3::IMPORTED, light).dispatch(rt3::DispatchSize::FullScreen());
            // under a tile partition the probe chain runs replicated: every rank renders it for the whole window (it reads the whole G-buffer and
            // is launch-bound well under a millisecond), then the frame-end gather of each rank's own tiles assembles the root's image
            if (n_ranks > 1) ctx.check(rt3_set_tile_partition(ctx.raw(), W, H, 0, 1), "partition off");
            rg.draw_frame(light);
            if (n_ranks > 1) {
                ctx.check(rt3_set_tile_partition(ctx.raw(), W, H, rank, n_ranks), "partition on");
                ctx.check(rt3_gather_tiles(ctx.raw(), light, 0), "gather");
                if (rank != 0) return 0;
            }
            std::vector<float> out((size_t)W * H * 4), at((size_t)px * 8 * py * 8 * 4);
            ctx.check(rt3_resource_download(ctx.raw(), light, out.data(), out.size() * 4), "download");
            ctx.check(rt3_resource_download(ctx.raw(), atlas, at.data(), at.size() * 4), "download");
            FILE* o = fopen(argv[8], "wb");
            fwrite(out.data(), 4, out.size(), o);
            fwrite(at.data(), 4, at.size(), o);
            fclose(o);
            printf("example_frame: probe-GI frame %ux%u, %ux%u probes\n", W, H, px, py);
            return 0;
        }
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
        if (n_ranks) {  // the frame's collective(s): enqueued behind the passes on the context's stream, no host synchronisation
            ctx.check(rt3_gather_tiles(ctx.raw(), light, 0), "gather Light");
            ctx.check(rt3_gather_tiles(ctx.raw(), color, 0), "gather color");
            if (rank != 0) {
                ctx.check(rt3_frame_wait(ctx.raw()), "wait");
                printf("example_frame: rank %u of %u sent its tiles\n", rank, n_ranks);
                return 0;
            }
        }

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
