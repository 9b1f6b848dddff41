"""Host-side mirror of the reference's per-frame pass graph, driving librt3.so through its C ABI.

Reference (src/renderer/render_graph/*):
  * `RenderGraph::image / buffer / import` name-keyed transient resources        mod.rs:422-483
  * `RayTracingPass::new(rg, name).shader(..).entry(..).constants(&c).read(origin, h).write(origin, h)
        .read_write(origin, h).launch(WorkSize2D)`                                executions.rs:80-101, build.rs:66-209,371-399
  * `ComputePass::new(rg, name)....dispatch(DispatchSize)`                        executions.rs:15-36
  * execution order = reverse DFS over `origin` edges from the node touching the output image, de-duplicated
                                                                                  bake.rs:29-49
  * bindings = one u32 handle per non-attachment edge in builder order           bake.rs:51-83
Barriers / image layouts / descriptor heaps are Vulkan mechanics and have no HIP counterpart (single in-order stream).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib as L

IMPORTED = 0xFFFFFFFFFFFFFFFF  # render_graph/mod.rs:38  (`!0`)


class ImageSize:
    """build.rs:211-229"""

    def __init__(self, kind, a=0, b=0):
        self.kind, self.a, self.b = kind, a, b

    FullScreen = None  # filled below

    @staticmethod
    def FractionalFullScreen(dx, dy):
        return ImageSize("frac", dx, dy)

    @staticmethod
    def XY(x, y):
        return ImageSize("xy", x, y)

    def size(self, window):
        w, h = window
        if self.kind == "full":
            return w, h
        if self.kind == "frac":
            return -(-w // self.a), -(-h // self.b)
        return self.a, self.b


ImageSize.FullScreen = ImageSize("full")


class WorkSize2D:
    """executions.rs:57-78 : FullScreen = WINDOW_SIZE exactly (no group rounding)."""

    def __init__(self, kind, a=0, b=0):
        self.kind, self.a, self.b = kind, a, b

    FullScreen = None

    @staticmethod
    def FractionalFullScreen(x, y):
        return WorkSize2D("frac", x, y)

    @staticmethod
    def X(x):
        return WorkSize2D("x", x)

    @staticmethod
    def XY(x, y):
        return WorkSize2D("xy", x, y)

    def size(self, window):
        w, h = window
        return {"full": (w, h), "frac": (-(-w // max(self.a, 1)), -(-h // max(self.b, 1))), "x": (self.a, 1), "xy": (self.a, self.b)}[self.kind]


WorkSize2D.FullScreen = WorkSize2D("full")


class DispatchSize:
    """build.rs:231-263 : FullScreen = ceil(W/8) x ceil(H/8) x 1 groups."""

    def __init__(self, kind, a=0, b=0, c=0):
        self.kind, self.a, self.b, self.c = kind, a, b, c

    FullScreen = None

    @staticmethod
    def X(x):
        return DispatchSize("xyz", x, 1, 1)

    @staticmethod
    def XY(x, y):
        return DispatchSize("xyz", x, y, 1)

    @staticmethod
    def XYZ(x, y, z):
        return DispatchSize("xyz", x, y, z)

    def size(self, window):
        w, h = window
        if self.kind == "full":
            return -(-w // 8), -(-h // 8), 1
        return self.a, self.b, self.c


DispatchSize.FullScreen = DispatchSize("full")


@dataclass
class NodeEdge:  # render_graph/mod.rs:109-126
    origin: int | None
    edge_type: str  # ShaderRead | ShaderWrite | ShaderReadWrite
    resource: int


@dataclass
class Node:  # render_graph/mod.rs:101-107
    name: str
    kind: str  # "raytracing" | "compute"
    path: str
    entry: str
    size: object
    constants: object
    edges: list = field(default_factory=list)


class Context:
    """Owns the rt3_ctx (Context::new + RayTracingContext::new, renderer/mod.rs:32-45)."""

    def __init__(self, device: int = 0):
        self.lib = L.load()
        h = C.c_void_p()
        rc = self.lib.rt3_create(device, C.byref(h))
        if rc != 0:
            raise L.Rt3Error(rc, self.lib.rt3_last_error(None).decode())
        self.h = h

    def check(self, rc):
        if rc != 0:
            raise L.Rt3Error(rc, self.lib.rt3_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.rt3_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def device_name(self):
        buf = C.create_string_buffer(256)
        self.check(self.lib.rt3_device_name(self.h, buf, 256))
        return buf.value.decode()

    def set_option(self, opt, value):
        self.check(self.lib.rt3_set_option(self.h, opt, int(value)))

    # ---- world buffers (world/mod.rs:83-125)
    def upload_mesh(self, mesh):
        v = np.ascontiguousarray(mesh.vertices, np.float32)
        i = np.ascontiguousarray(mesh.indices, np.uint32)
        g = np.ascontiguousarray(mesh.geometries)
        pc = np.ascontiguousarray(mesh.prim_counts, np.uint32)
        self.check(self.lib.rt3_scene_set_vertices(self.h, v.ctypes.data, len(v)))
        self.check(self.lib.rt3_scene_set_indices(self.h, i.ctypes.data, len(i)))
        self.check(self.lib.rt3_scene_set_geometry(self.h, g.ctypes.data, pc.ctypes.data, len(g)))
        for i, t in enumerate(getattr(mesh, "textures", None) or []):  # base-colour textures, RGBA8 sRGB
            t = np.ascontiguousarray(t, np.uint8)
            self.check(self.lib.rt3_scene_set_texture(self.h, i, t.ctypes.data, t.shape[1], t.shape[0]))

    def set_instances(self, instances):
        """instances: [(geometry_first, geometry_count, 4x4 matrix as numpy, object -> world)] -- Instance + Transform of
        src/renderer/world/mod.rs:46-60; [] = every geometry once under the identity.  Takes effect at the next build_accel()."""
        arr = (L.Instance * max(1, len(instances)))()
        for k, (first, count, m) in enumerate(instances):
            arr[k].geometry_first, arr[k].geometry_count = int(first), int(count)
            arr[k].transform[:] = [float(x) for x in np.asarray(m, np.float32).T.ravel()]  # column-major, like glam's Mat4
        self.check(self.lib.rt3_scene_set_instances(self.h, C.byref(arr), len(instances)))

    def set_sky(self, rgb):
        s = np.ascontiguousarray(rgb, np.float32)
        self.check(self.lib.rt3_scene_set_sky(self.h, s.ctypes.data, s.shape[1], s.shape[0]))

    def set_bluenoise(self, rgba):
        b = np.ascontiguousarray(rgba, np.uint8)
        self.check(self.lib.rt3_scene_set_bluenoise(self.h, b.ctypes.data, b.shape[1], b.shape[0]))

    def build_accel(self) -> int:
        out = C.c_uint32()
        self.check(self.lib.rt3_accel_build(self.h, C.byref(out)))
        return out.value

    def accel_import(self, nodes, tris):
        """install a tree built elsewhere over the same triangles (rt3_accel_download's format)"""
        n = np.ascontiguousarray(nodes)
        t = np.ascontiguousarray(tris)
        self.check(self.lib.rt3_accel_import(self.h, n.ctypes.data, n.nbytes, t.ctypes.data, t.nbytes))

    def accel_info(self):
        """(n_nodes, n_tris, levels, node_bytes)"""
        a, b, c, d = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        self.check(self.lib.rt3_accel_info(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return a.value, b.value, c.value, d.value

    def accel_download(self):
        nn, nt, _, nb = self.accel_info()
        nodes = np.empty((nn, nb // 4), np.uint32)
        tris = np.empty((nt, 12), np.uint32)
        self.check(self.lib.rt3_accel_download(self.h, nodes.ctypes.data, nodes.nbytes, tris.ctypes.data, tris.nbytes))
        return nodes, tris

    def sky_download(self, w, h):
        """(alias words, RGB9E5 texels, marginal CDF, realised (u,v) density)"""
        al, tx, cm, pu = np.empty((h, w), np.uint32), np.empty((h, w), np.uint32), np.empty(h, np.float32), np.empty((h, w), np.float32)
        self.check(self.lib.rt3_sky_download(self.h, al.ctypes.data, tx.ctypes.data, cm.ctypes.data, pu.ctypes.data))
        return al, tx, cm, pu

    def trace_rays(self, rays, any_hit=False, counts=False, repeat=1):
        """rays (8, n) float32 host SoA -> (t,u,v,prim[,n_nodes,n_tris], kernel_ms)"""
        rays = np.ascontiguousarray(rays, np.float32)
        n = rays.shape[1]
        t, u, v = (np.zeros(n, np.float32) for _ in range(3))
        prim = np.zeros(n, np.uint32)
        cn = np.zeros(n, np.uint32) if counts else None
        ct = np.zeros(n, np.uint32) if counts else None
        ms = C.c_double()
        self.check(self.lib.rt3_trace_rays(self.h, rays.ctypes.data, n, int(any_hit), t.ctypes.data, u.ctypes.data, v.ctypes.data, prim.ctypes.data,
                                           cn.ctypes.data if counts else None, ct.ctypes.data if counts else None, repeat, C.byref(ms)))
        return (t, u, v, prim, cn, ct, ms.value) if counts else (t, u, v, prim, ms.value)

    def selftest(self, op, inp, out_words, out_dtype=np.uint32):
        """Evaluate device function `op` (include/rt3.h: rt3_selftest_eval) on the rows of `inp` (32-bit words)."""
        inp = np.ascontiguousarray(inp)
        assert inp.dtype.itemsize == 4
        n = inp.shape[0]
        out = np.zeros((n, out_words), out_dtype)
        self.check(self.lib.rt3_selftest_eval(self.h, op, inp.ctypes.data, n, out.ctypes.data))
        return out

    def stats_reset(self):
        self.check(self.lib.rt3_stats_reset(self.h))

    def stats(self) -> L.Stats:
        s = L.Stats()
        self.check(self.lib.rt3_stats_get(self.h, C.byref(s)))
        return s

    def wait(self):
        self.check(self.lib.rt3_frame_wait(self.h))

    def set_tile_partition(self, w, h, rank, n_ranks):
        self.check(self.lib.rt3_set_tile_partition(self.h, w, h, rank, n_ranks))

    def tile_pixel_count(self, rank, n_ranks):
        out = C.c_uint32()
        self.check(self.lib.rt3_tile_pixel_count(self.h, rank, n_ranks, C.byref(out)))
        return out.value

    # ---- frame-end gather (include/rt3.h: rt3_comm_* / rt3_gather_*)
    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(L.COMM_ID_BYTES)
        rc = self.lib.rt3_comm_unique_id(buf)
        if rc != 0:
            raise L.Rt3Error(rc, self.lib.rt3_last_error(None).decode())
        return buf.raw

    def comm_init(self, uid: bytes, rank: int, n_ranks: int):
        assert len(uid) == L.COMM_ID_BYTES
        self.check(self.lib.rt3_comm_init(self.h, C.c_char_p(uid), rank, n_ranks))

    def comm_destroy(self):
        self.check(self.lib.rt3_comm_destroy(self.h))

    def gather_tiles(self, image, root=0):
        self.check(self.lib.rt3_gather_tiles(self.h, image, root))

    def gather_layout(self, image, root, n_ranks):
        off = (C.c_uint64 * (n_ranks + 1))()
        self.check(self.lib.rt3_gather_layout(self.h, image, root, n_ranks, off))
        return [int(x) for x in off]

    def gather_unpack(self, image, root, n_ranks, recv_device_ptr):
        self.check(self.lib.rt3_gather_unpack(self.h, image, root, n_ranks, C.c_void_p(recv_device_ptr)))


class NodeBuilder:
    """build.rs:32-209"""

    def __init__(self, rg, name, kind):
        if any(n.name == name for n in rg.nodes):
            raise ValueError(f"Node name {name} allready used")  # build.rs:57-59 (panic there)
        self.rg = rg
        self.node = Node(name, kind, "", "main", WorkSize2D.FullScreen if kind == "raytracing" else DispatchSize.FullScreen, None)

    def shader(self, path):
        self.node.path = path
        return self

    def entry(self, entry):
        self.node.entry = entry
        return self

    def constants(self, c):
        self.node.constants = c
        return self

    def _edge(self, origin, handle, typ):
        if typ == "ShaderRead" and origin != IMPORTED:  # build.rs:96-107
            prev = [e for e in self.rg.nodes[origin].edges if e.resource == handle]
            if not prev:
                raise ValueError("Origin doesnt write to handle")
            if prev[0].edge_type == "ShaderRead":
                raise ValueError("Origin contains handle but does not write to it")
        self.node.edges.append(NodeEdge(None if origin == IMPORTED else origin, typ, handle))
        return self

    def read(self, origin, handle):
        return self._edge(origin, handle, "ShaderRead")

    def write(self, last_read, handle):
        return self._edge(last_read, handle, "ShaderWrite")

    def read_write(self, origin, handle):
        return self._edge(origin, handle, "ShaderReadWrite")

    def _build(self):
        seen = set()
        for e in self.node.edges:  # build.rs:195-198
            if e.resource in seen:
                raise ValueError(f"resource: {e.resource} is duplicate")
            seen.add(e.resource)
        self.rg.nodes.append(self.node)
        return len(self.rg.nodes) - 1

    def launch(self, size=None):  # RayTracingPass, build.rs:380-383
        assert self.node.kind == "raytracing"
        self.node.size = size or WorkSize2D.FullScreen
        return self._build()

    def dispatch(self, size=None):  # ComputePass, build.rs:395-398
        assert self.node.kind == "compute"
        self.node.size = size or DispatchSize.FullScreen
        return self._build()


class RayTracingPass:
    @staticmethod
    def new(rg, name):  # executions.rs:86-101
        return NodeBuilder(rg, name, "raytracing")


class ComputePass:
    @staticmethod
    def new(rg, name):  # executions.rs:21-36
        return NodeBuilder(rg, name, "compute")


class RenderGraph:
    """render_graph/mod.rs:291-532 reduced to what a HIP stream needs: named resources + per-frame node list."""

    def __init__(self, ctx: Context, window):
        self.ctx, self.window = ctx, (int(window[0]), int(window[1]))
        self.nodes: list[Node] = []
        self.named: dict[str, int] = {}
        self.frame_number = 0

    # mod.rs:440-483 : created on first use, looked up by name afterwards
    def image(self, size: ImageSize, fmt: int, name: str) -> int:
        if name in self.named:
            return self.named[name]
        w, h = size.size(self.window)
        out = C.c_uint32()
        self.ctx.check(self.ctx.lib.rt3_image_create(self.ctx.h, w, h, fmt, C.byref(out)))
        self.named[name] = out.value
        return out.value

    def buffer(self, size: int, name: str) -> int:
        if name in self.named:
            return self.named[name]
        out = C.c_uint32()
        self.ctx.check(self.ctx.lib.rt3_buffer_create(self.ctx.h, size, C.byref(out)))
        self.named[name] = out.value
        return out.value

    def import_image(self, device_ptr: int, w: int, h: int, fmt: int, name: str) -> int:  # mod.rs:426-438
        out = C.c_uint32()
        self.ctx.check(self.ctx.lib.rt3_image_import(self.ctx.h, C.c_void_p(device_ptr), w, h, fmt, C.byref(out)))
        self.named[name] = out.value
        return out.value

    def begin_frame(self):  # mod.rs:656-686
        self.nodes = []

    def bake(self, root: int):  # bake.rs:29-49
        order = []

        def flatten(n):
            order.append(n)
            for e in self.nodes[n].edges:
                if e.origin is not None:
                    flatten(e.origin)

        flatten(root)
        out = []
        for n in reversed(order):
            if n not in out:
                out.append(n)
        return out

    def draw_frame(self, output: int, wait: bool = False):
        """mod.rs:534-655 : find the root (the node touching `output`, :553-562), bake, execute in order."""
        roots = [i for i, n in enumerate(self.nodes) if any(e.resource == output for e in n.edges)]
        if not roots:
            raise ValueError("no node touches the output resource")
        for ni in self.bake(roots[-1]):
            n = self.nodes[ni]
            x, y, z = (*n.size.size(self.window), 1) if n.kind == "raytracing" else n.size.size(self.window)
            b = (C.c_uint32 * len(n.edges))(*[e.resource for e in n.edges])  # bake.rs:51-83
            cst = n.constants
            self.ctx.check(self.ctx.lib.rt3_pass_launch(self.ctx.h, n.path.encode(), n.entry.encode(), x, y, z, C.byref(cst), C.sizeof(cst),
                                                        b, len(n.edges)))
        self.frame_number += 1
        if wait:
            self.ctx.wait()

    def download(self, handle: int, shape, dtype) -> np.ndarray:
        out = np.empty(shape, dtype)
        self.ctx.check(self.ctx.lib.rt3_resource_download(self.ctx.h, handle, out.ctypes.data, out.nbytes))
        return out

    def upload(self, handle: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        self.ctx.check(self.ctx.lib.rt3_resource_upload(self.ctx.h, handle, arr.ctypes.data, arr.nbytes))

    def device_ptr(self, handle: int):
        p, n = C.c_void_p(), C.c_size_t()
        self.ctx.check(self.ctx.lib.rt3_resource_device_ptr(self.ctx.h, handle, C.byref(p), C.byref(n)))
        return p.value, n.value
