// asset_tool.cpp -- command-line front end of the native asset pipeline (assets.hpp).
//
//   asset_tool glb <scene.glb> <outdir>               dump the flattened Mesh (what `loaded_assets` uploads, world/mod.rs:83-101)
//   asset_tool exr <sky.exr> <outdir>                 dump the decoded equirect image
//   asset_tool png <image.png|.jpg> <outdir>          dump the decoded RGBA8 image (PNG or baseline JPEG)
//   asset_tool bincode <file> <current|old> <outdir>  dump a processed-asset cache file (assets/mod.rs:118-137)
//   asset_tool render <scene.glb> <sky.exr|-> <bluenoise.png|-> W H spp bounces flags px py pz dx dy dz fov_deg <out.bin>
//                                                     load -> upload -> gbuffer / refrence_mode / postprocess on the GPU;
//                                                     out.bin = Light RGBA32F then colour RGBA32F
// Dumps are raw little-endian arrays + manifest.txt; tests/test_host_assets.py compares them with the Python loaders.
// Only `render` needs librt3.so to find a GPU.
#include <cstdio>
#include <cstdlib>

#include "assets.hpp"
#include "render_graph.hpp"

namespace A = rt3::assets;

template <class T>
static void dump(const std::string& path, const std::vector<T>& v) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error(path + ": cannot write");
    if (!v.empty()) fwrite(v.data(), sizeof(T), v.size(), f);
    fclose(f);
}

int main(int argc, char** argv) {
    try {
        const std::string cmd = argc > 1 ? argv[1] : "";
        if (cmd == "glb" && argc == 4) {
            A::Mesh m = A::load_glb(argv[2]);
            const std::string out = argv[3];
            dump(out + "/vertices.bin", m.vertices);
            dump(out + "/indices.bin", m.indices);
            dump(out + "/geometries.bin", m.geometries);
            dump(out + "/prim_counts.bin", m.prim_counts);
            FILE* f = fopen((out + "/manifest.txt").c_str(), "w");
            fprintf(f, "vertices %zu\nindices %zu\ngeometries %zu\ntextures %zu\n", m.n_vertices(), m.indices.size(), m.geometries.size(), m.textures.size());
            for (size_t i = 0; i < m.textures.size(); i++) {
                fprintf(f, "texture %zu %u %u\n", i, m.textures[i].w, m.textures[i].h);
                dump(out + "/texture_" + std::to_string(i) + ".bin", m.textures[i].rgba);
            }
            for (auto& n : m.names) fprintf(f, "name %s\n", n.c_str());
            fclose(f);
            printf("asset_tool: %zu vertices, %zu triangles, %zu geometries, %zu textures\n", m.n_vertices(), m.n_triangles(), m.geometries.size(), m.textures.size());
        } else if (cmd == "exr" && argc == 4) {
            A::SkyImage s = A::read_exr(argv[2]);
            dump(std::string(argv[3]) + "/sky.bin", s.rgb);
            FILE* f = fopen((std::string(argv[3]) + "/manifest.txt").c_str(), "w");
            fprintf(f, "sky %u %u\n", s.w, s.h);
            fclose(f);
        } else if (cmd == "png" && argc == 4) {
            std::vector<uint8_t> raw = A::read_file(argv[2]);
            A::Image im = A::decode_image(raw.data(), raw.size());
            dump(std::string(argv[3]) + "/image.bin", im.rgba);
            FILE* f = fopen((std::string(argv[3]) + "/manifest.txt").c_str(), "w");
            fprintf(f, "image %u %u\n", im.w, im.h);
            fclose(f);
        } else if (cmd == "bincode" && argc == 5) {
            A::ProcessedMesh pm = A::read_processed_mesh(argv[2], std::string(argv[3]) == "old");
            const std::string out = argv[4];
            std::vector<uint32_t> ml;
            for (auto& m : pm.meshlets) ml.insert(ml.end(), {m.vertex_offset, m.triangle_offset, m.vertex_count, m.triangle_count});
            std::vector<float> mats;
            for (auto& m : pm.materials) mats.insert(mats.end(), {m.color[0], m.color[1], m.color[2], m.metalic_factor, m.roughness_factor, (float)m.texture_offset});
            dump(out + "/meshlets.bin", ml);
            dump(out + "/materials.bin", mats);
            dump(out + "/vertices.bin", pm.vertices);
            dump(out + "/indices.bin", pm.indices);
            FILE* f = fopen((out + "/manifest.txt").c_str(), "w");
            fprintf(f, "meshlets %zu\nmaterials %zu\nvertices %zu\nindices %zu\nuploaded %d\n", pm.meshlets.size(), pm.materials.size(), pm.vertices.size() / 8,
                    pm.indices.size(), pm.uploaded ? 1 : 0);
            fclose(f);
        } else if (cmd == "render" && argc == 18) {
            A::Mesh mesh = A::load_glb(argv[2]);
            const uint32_t W = (uint32_t)atoi(argv[5]), H = (uint32_t)atoi(argv[6]);
            rt3::Context ctx(0);
            ctx.check(rt3_set_tile_partition(ctx.raw(), W, H, 0, 1), "partition");
            A::upload(ctx.raw(), mesh);
            if (std::string(argv[3]) != "-") A::upload_sky(ctx.raw(), A::read_exr(argv[3]));
            if (std::string(argv[4]) != "-") {
                std::vector<uint8_t> raw = A::read_file(argv[4]);
                A::Image bn = A::decode_png(raw.data(), raw.size());
                ctx.check(rt3_scene_set_bluenoise(ctx.raw(), bn.rgba.data(), bn.w, bn.h), "bluenoise");
            }
            rt3::Camera camera{{(float)atof(argv[10]), (float)atof(argv[11]), (float)atof(argv[12])},
                               {(float)atof(argv[13]), (float)atof(argv[14]), (float)atof(argv[15])},
                               (float)(atof(argv[16]) * 3.14159265358979323846 / 180.0), (float)W / (float)H};
            rt3_gconst gconst = camera.gconst(W, H);
            gconst.samples = (uint32_t)atoi(argv[7]);
            gconst.bounces = (uint32_t)atoi(argv[8]);
            gconst.pad[0] = (uint32_t)atoi(argv[9]);
            gconst.frame = 0;
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
            FILE* o = fopen(argv[17], "wb");
            if (!o) throw std::runtime_error("cannot write the output file");
            fwrite(out.data(), 4, out.size(), o);
            fwrite(col.data(), 4, col.size(), o);
            fclose(o);
            printf("asset_tool: rendered %zu triangles at %ux%u\n", mesh.n_triangles(), W, H);
        } else {
            fprintf(stderr, "usage: asset_tool glb|exr|png|bincode|render ... (see the header of asset_tool.cpp)\n");
            return 2;
        }
    } catch (const std::exception& e) {
        fprintf(stderr, "asset_tool: %s\n", e.what());
        return 1;
    }
    return 0;
}
