// render_graph.hpp -- C++ mirror of the reference's host-side pass-graph API on top of the C ABI (include/rt3.h).
//
// The reference host is Rust (src/renderer/render_graph/*, src/components/camera.rs); no Rust toolchain exists in the
// build image, so this header is what a compiled host links against.  Names and call shapes follow the reference:
//
//   let gb = RayTracingPass::new(&mut rg, "gbuffer").shader("gbuffer").constants(&gconst)
//                .write(IMPORTED, gbuffer).write(IMPORTED, depth).launch(WorkSize2D::FullScreen);          (Rust)
//   auto gb = rt3::RayTracingPass::New(rg, "gbuffer").shader("gbuffer").constants(gconst)
//                .write(rt3::IMPORTED, gbuffer).write(rt3::IMPORTED, depth).launch(rt3::WorkSize2D::FullScreen());   (here)
//
//   NodeBuilder::{constants, read, write, read_write}      render_graph/build.rs:66-209
//   RayTracingPass / ComputePass + launch / dispatch       render_graph/executions.rs:15-121, build.rs:371-399
//   WorkSize2D / DispatchSize / ImageSize                  executions.rs:57-78, build.rs:211-263
//   RenderGraph::{image, buffer, bake, draw_frame}         render_graph/mod.rs:422-483,534-655, bake.rs:29-83
//   Camera::{view_matrix, projection_matrix}               components/camera.rs:23-59
// Builder misuse panics in the reference; here it throws std::runtime_error (never across the C ABI).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/rt3.h"

namespace rt3 {

using NodeHandle = size_t;
using ResourceHandle = uint32_t;
constexpr NodeHandle IMPORTED = ~static_cast<NodeHandle>(0);  // render_graph/mod.rs:38

struct WorkSize2D {  // executions.rs:57-78
    enum Kind { kFullScreen, kFractionalFullScreen, kX, kXY } kind = kFullScreen;
    uint32_t a = 0, b = 0;
    static WorkSize2D FullScreen() { return {kFullScreen, 0, 0}; }
    static WorkSize2D FractionalFullScreen(uint32_t x, uint32_t y) { return {kFractionalFullScreen, x, y}; }
    static WorkSize2D X(uint32_t x) { return {kX, x, 0}; }
    static WorkSize2D XY(uint32_t x, uint32_t y) { return {kXY, x, y}; }
    std::pair<uint32_t, uint32_t> size(uint32_t w, uint32_t h) const {
        switch (kind) {
            case kFractionalFullScreen: return {(w + a - 1) / a, (h + b - 1) / b};
            case kX: return {a, 1};
            case kXY: return {a, b};
            default: return {w, h};  // WINDOW_SIZE exactly: ray-gen grids are not rounded to groups
        }
    }
};
struct DispatchSize {  // build.rs:231-263
    enum Kind { kFullScreen, kXYZ } kind = kFullScreen;
    uint32_t x = 0, y = 0, z = 0;
    static DispatchSize FullScreen() { return {kFullScreen, 0, 0, 0}; }
    static DispatchSize X(uint32_t a) { return {kXYZ, a, 1, 1}; }
    static DispatchSize XY(uint32_t a, uint32_t b) { return {kXYZ, a, b, 1}; }
    static DispatchSize XYZ(uint32_t a, uint32_t b, uint32_t c) { return {kXYZ, a, b, c}; }
    void size(uint32_t w, uint32_t h, uint32_t out[3]) const {
        if (kind == kFullScreen) {
            out[0] = (w + 7) / 8;  // groups of 8 x 8 (build.rs:254-258)
            out[1] = (h + 7) / 8;
            out[2] = 1;
        } else {
            out[0] = x; out[1] = y; out[2] = z;
        }
    }
};
struct ImageSize {  // build.rs:211-229
    enum Kind { kFullScreen, kFractionalFullScreen, kXY } kind = kFullScreen;
    uint32_t a = 0, b = 0;
    static ImageSize FullScreen() { return {kFullScreen, 0, 0}; }
    static ImageSize FractionalFullScreen(uint32_t dx, uint32_t dy) { return {kFractionalFullScreen, dx, dy}; }
    static ImageSize XY(uint32_t x, uint32_t y) { return {kXY, x, y}; }
    std::pair<uint32_t, uint32_t> size(uint32_t w, uint32_t h) const {
        if (kind == kFractionalFullScreen) return {(w + a - 1) / a, (h + b - 1) / b};
        if (kind == kXY) return {a, b};
        return {w, h};
    }
};

// Context::new + RayTracingContext::new (renderer/mod.rs:32-45): owns the rt3_ctx
class Context {
   public:
    explicit Context(int device = 0) {
        if (int rc = rt3_create(device, &ctx_)) throw std::runtime_error(std::string("rt3_create: ") + rt3_last_error(nullptr) + " (" + std::to_string(rc) + ")");
    }
    ~Context() { rt3_destroy(ctx_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    rt3_ctx* raw() const { return ctx_; }
    void check(int rc, const char* what) const {
        if (rc) throw std::runtime_error(std::string(what) + ": " + rt3_last_error(ctx_) + " (" + std::to_string(rc) + ")");
    }

   private:
    rt3_ctx* ctx_ = nullptr;
};

enum class EdgeType { ShaderRead, ShaderWrite, ShaderReadWrite };  // render_graph/mod.rs:109-126 (no attachments: raster is out of scope)
struct NodeEdge {
    bool has_origin;
    NodeHandle origin;
    EdgeType edge_type;
    ResourceHandle resource;
};
struct Node {  // render_graph/mod.rs:101-107
    std::string name, path, entry = "main";
    bool raytracing = true;
    WorkSize2D launch;
    DispatchSize dispatch;
    std::vector<uint8_t> constants;  // copied like build.rs:66-94 copies into the mapped constants buffer
    std::vector<NodeEdge> edges;
};

class RenderGraph {
   public:
    RenderGraph(Context& ctx, uint32_t width, uint32_t height) : ctx_(ctx), w_(width), h_(height) {}

    // mod.rs:440-483: created on first use, looked up by name afterwards
    ResourceHandle image(ImageSize size, uint32_t format, const std::string& name) {
        auto it = named_.find(name);
        if (it != named_.end()) return it->second;
        auto wh = size.size(w_, h_);
        ResourceHandle h = 0;
        ctx_.check(rt3_image_create(ctx_.raw(), wh.first, wh.second, format, &h), "rt3_image_create");
        return named_[name] = h;
    }
    ResourceHandle buffer(size_t bytes, const std::string& name) {
        auto it = named_.find(name);
        if (it != named_.end()) return it->second;
        ResourceHandle h = 0;
        ctx_.check(rt3_buffer_create(ctx_.raw(), bytes, &h), "rt3_buffer_create");
        return named_[name] = h;
    }
    void begin_frame() { nodes.clear(); }  // mod.rs:656-686

    std::vector<NodeHandle> bake(NodeHandle root) const {  // bake.rs:29-49
        std::vector<NodeHandle> order, out;
        flatten(root, order);
        std::reverse(order.begin(), order.end());
        for (NodeHandle n : order)
            if (std::find(out.begin(), out.end(), n) == out.end()) out.push_back(n);
        return out;
    }
    // mod.rs:534-655: root = the node touching `output` (:553-562); bindings = handles of the edges in builder order (bake.rs:51-83)
    void draw_frame(ResourceHandle output, bool wait = true) {
        NodeHandle root = IMPORTED;
        for (size_t i = 0; i < nodes.size(); i++)
            for (const auto& e : nodes[i].edges)
                if (e.resource == output) root = i;
        if (root == IMPORTED) throw std::runtime_error("no node touches the output resource");
        for (NodeHandle ni : bake(root)) {
            const Node& n = nodes[ni];
            uint32_t xyz[3] = {0, 0, 1};
            if (n.raytracing) {
                auto s = n.launch.size(w_, h_);
                xyz[0] = s.first;
                xyz[1] = s.second;
            } else {
                n.dispatch.size(w_, h_, xyz);
            }
            std::vector<uint32_t> bindings;
            for (const auto& e : n.edges) bindings.push_back(e.resource);
            ctx_.check(rt3_pass_launch(ctx_.raw(), n.path.c_str(), n.entry.c_str(), xyz[0], xyz[1], xyz[2], n.constants.data(), n.constants.size(),
                                       bindings.data(), (uint32_t)bindings.size()),
                       ("pass " + n.name).c_str());
        }
        frame_number++;
        if (wait) ctx_.check(rt3_frame_wait(ctx_.raw()), "rt3_frame_wait");
    }

    Context& ctx() { return ctx_; }
    std::vector<Node> nodes;
    uint64_t frame_number = 0;

   private:
    void flatten(NodeHandle root, std::vector<NodeHandle>& order) const {
        order.push_back(root);
        for (const auto& e : nodes[root].edges)
            if (e.has_origin) flatten(e.origin, order);
    }
    Context& ctx_;
    uint32_t w_, h_;
    std::map<std::string, ResourceHandle> named_;
};

class NodeBuilder {  // build.rs:32-209
   public:
    NodeBuilder(RenderGraph& rg, const std::string& name, bool raytracing) : rg_(rg) {
        for (const auto& n : rg.nodes)
            if (n.name == name) throw std::runtime_error("Node name " + name + " allready used");  // build.rs:57-59
        node_.name = name;
        node_.raytracing = raytracing;
    }
    NodeBuilder& shader(const std::string& path) { node_.path = path; return *this; }
    NodeBuilder& entry(const std::string& e) { node_.entry = e; return *this; }
    template <class C>
    NodeBuilder& constants(const C& c) {
        node_.constants.resize(sizeof(C));
        std::memcpy(node_.constants.data(), &c, sizeof(C));
        return *this;
    }
    NodeBuilder& read(NodeHandle origin, ResourceHandle h) {  // build.rs:96-141
        if (origin != IMPORTED) {
            const NodeEdge* prev = nullptr;
            for (const auto& e : rg_.nodes.at(origin).edges)
                if (e.resource == h) { prev = &e; break; }
            if (!prev) throw std::runtime_error("Origin doesnt write to handle");
            if (prev->edge_type == EdgeType::ShaderRead) throw std::runtime_error("Origin contains handle but does not write to it");
        }
        return edge(origin, h, EdgeType::ShaderRead);
    }
    NodeBuilder& write(NodeHandle last_read, ResourceHandle h) { return edge(last_read, h, EdgeType::ShaderWrite); }
    NodeBuilder& read_write(NodeHandle origin, ResourceHandle h) { return edge(origin, h, EdgeType::ShaderReadWrite); }
    NodeHandle launch(WorkSize2D size = WorkSize2D::FullScreen()) {  // build.rs:380-383
        node_.launch = size;
        return build();
    }
    NodeHandle dispatch(DispatchSize size = DispatchSize::FullScreen()) {  // build.rs:395-398
        node_.dispatch = size;
        return build();
    }

   private:
    NodeBuilder& edge(NodeHandle origin, ResourceHandle h, EdgeType t) {
        node_.edges.push_back(NodeEdge{origin != IMPORTED, origin, t, h});
        return *this;
    }
    NodeHandle build() {  // build.rs:194-208
        for (size_t i = 0; i < node_.edges.size(); i++)
            for (size_t j = i + 1; j < node_.edges.size(); j++)
                if (node_.edges[i].resource == node_.edges[j].resource) throw std::runtime_error("resource is duplicate");
        rg_.nodes.push_back(node_);
        return rg_.nodes.size() - 1;
    }
    RenderGraph& rg_;
    Node node_;
};
struct RayTracingPass {  // executions.rs:86-101 (`new` is a C++ keyword)
    static NodeBuilder New(RenderGraph& rg, const std::string& name) { return NodeBuilder(rg, name, true); }
};
struct ComputePass {  // executions.rs:21-36
    static NodeBuilder New(RenderGraph& rg, const std::string& name) { return NodeBuilder(rg, name, false); }
};

// components/camera.rs:23-59; fov in radians (main.rs:72: 65.0_f32.to_radians())
struct Camera {
    float position[3], direction[3], fov, aspect_ratio, z_near = 0.1f, z_far = 1000.0f;
    // view_matrix / projection_matrix (:52-58) + their inverses + window size, as renderer::commands fills GConst (renderer/mod.rs:72-78)
    rt3_gconst gconst(uint32_t width, uint32_t height) const {
        rt3_gconst g;
        rt3_camera_gconst(position, direction, fov, aspect_ratio, z_near, z_far, (float)width, (float)height, &g);
        return g;
    }
};

}  // namespace rt3
