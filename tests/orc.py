"""ctypes binding of the CPU oracle (oracle/librt3_oracle.so).  TEST INFRASTRUCTURE ONLY -- the product never imports this."""
import ctypes as C
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
_lib = None

MISS = 0xFFFFFFFF
BACKGROUND_DEPTH = 100000.0
F_NEE_SKY, F_BLUENOISE, F_SPECULAR, F_FACEFORWARD, F_PROBE_RADIANCE = 1, 2, 4, 8, 16


class GConst(C.Structure):
    _fields_ = [("proj", C.c_float * 16), ("view", C.c_float * 16), ("proj_inverse", C.c_float * 16), ("view_inverse", C.c_float * 16),
                ("window_size", C.c_float * 2), ("frame", C.c_uint32), ("blendfactor", C.c_float), ("bounces", C.c_uint32),
                ("samples", C.c_uint32), ("proberng", C.c_uint32), ("cell_size", C.c_float), ("mouse", C.c_uint32 * 2), ("pad", C.c_uint32 * 2)]


assert C.sizeof(GConst) == 304


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(str(ROOT / "oracle" / "librt3_oracle.so"))
        L = _lib
        u32, f32, vp = C.c_uint32, C.c_float, C.c_void_p
        L.orc_hash.restype = u32; L.orc_hash.argtypes = [u32]
        L.orc_zcurve.restype = u32; L.orc_zcurve.argtypes = [u32, u32]
        L.orc_rng_seed.restype = u32; L.orc_rng_seed.argtypes = [u32, u32, u32]
        L.orc_murmur3.restype = u32; L.orc_murmur3.argtypes = [u32, u32]
        L.orc_uniform_float.restype = f32; L.orc_uniform_float.argtypes = [u32, u32]
        L.orc_radical_inverse_bits.restype = u32; L.orc_radical_inverse_bits.argtypes = [u32]
        L.orc_pack_color_888.restype = u32; L.orc_pack_normal_11_10_11.restype = u32
        L.orc_pack_2x16f.restype = u32; L.orc_pack_2x16f.argtypes = [f32, f32]
        L.orc_float3_to_rgb9e5.restype = u32
        L.orc_atan2.restype = f32; L.orc_atan2.argtypes = [f32, f32]
        L.orc_sincos_2pi.argtypes = [f32, vp, vp]
        L.orc_diffuse_sample.argtypes = [f32, f32, vp]
        L.orc_camera_gconst.argtypes = [vp, vp, f32, f32, f32, f32, f32, f32, vp]
        L.orc_primary_ray.argtypes = [vp, u32, u32, vp, vp]
        L.orc_primary_rays.argtypes = [vp, vp, vp, u32, f32, f32, vp]
        L.orc_scene_create.restype = vp
        for n in ("orc_scene_destroy", "orc_accel_build"):
            getattr(L, n).argtypes = [vp]
        L.orc_accel_set_layout.argtypes = [vp, u32, u32, u32]
        L.orc_accel_set_collapse.argtypes = [vp, u32]
        L.orc_accel_set_sah_top.argtypes = [vp, u32]
        L.orc_accel_set_tree_order.argtypes = [vp, u32]
        L.orc_accel_set_dp_costs.argtypes = [vp, f32, f32]
        L.orc_accel_set_top_opt.argtypes = [vp, u32, u32]
        L.orc_accel_set_recull.argtypes = [vp, u32]
        L.orc_scene_set_instances.argtypes = [vp, vp, u32]
        L.orc_scene_set_instances.restype = C.c_int
        L.orc_octa_encode16.argtypes = [vp]
        L.orc_octa_encode16.restype = u32
        L.orc_octa_decode16.argtypes = [u32, vp]
        L.orc_accel_node_words.restype = u32; L.orc_accel_node_words.argtypes = [vp]
        L.orc_scene_set_vertices.argtypes = [vp, vp, u32]
        L.orc_scene_set_indices.argtypes = [vp, vp, u32]
        L.orc_scene_set_geometry.argtypes = [vp, vp, vp, u32]
        L.orc_scene_set_sky.argtypes = [vp, vp, u32, u32]
        L.orc_scene_set_bluenoise.argtypes = [vp, vp, u32, u32]
        L.orc_scene_set_texture.argtypes = [vp, u32, vp, u32, u32]
        for n in ("orc_accel_num_tris", "orc_accel_num_nodes", "orc_accel_max_depth"):
            getattr(L, n).restype = u32; getattr(L, n).argtypes = [vp]
        for n in ("orc_accel_nodes", "orc_accel_tris", "orc_accel_codes", "orc_sky_alias", "orc_sky_texels", "orc_sky_cdf_marg", "orc_sky_pdf_uv"):
            getattr(L, n).restype = vp; getattr(L, n).argtypes = [vp]
        L.orc_trace_closest.argtypes = [vp, vp, u32, vp, vp, vp, vp, vp, vp, C.c_int]
        L.orc_trace_any.argtypes = [vp, vp, u32, vp, vp, vp, C.c_int]
        L.orc_trace_brute.argtypes = [vp, vp, u32, vp, vp, vp, vp, C.c_int, C.c_int]
        L.orc_hit_info.argtypes = [vp, u32, f32, f32, vp]
        L.orc_pass_gbuffer.argtypes = [vp, vp, u32, u32, u32, u32, vp, vp, C.c_int]
        L.orc_pass_reference_mode.argtypes = [vp, vp, u32, u32, u32, u32, vp, vp, vp, vp, vp, C.c_int]
        L.orc_pass_postprocess.argtypes = [vp, vp, u32, u32, u32, u32, vp, vp, vp, C.c_int]
        L.orc_bounce1_rays.restype = u32; L.orc_bounce1_rays.argtypes = [vp, vp, vp, vp, vp, u32]
        L.orc_tile_pixels.restype = u32; L.orc_tile_pixels.argtypes = [u32, u32, u32, u32, vp]
        L.orc_octa_decode.argtypes = [f32, f32, vp]
        L.orc_sh3_evaluate.argtypes = [vp, vp]
        L.orc_wave_sort64.argtypes = [vp, vp]
        L.orc_wave_sum64.restype = f32; L.orc_wave_sum64.argtypes = [vp]
        L.orc_pass_structured_importance_sampling.argtypes = [vp, u32, u32, vp, vp, vp]
        L.orc_pass_trace_probes.argtypes = [vp, vp, u32, u32, vp, vp, vp, vp, vp, C.c_int]
        L.orc_pass_sh_conversion.argtypes = [u32, u32, vp, vp]
        L.orc_pass_interpolate_probes.argtypes = [vp, vp, vp, vp, vp]
    return _lib


def ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def f3(x):
    return (C.c_float * 3)(*[float(v) for v in x])


def camera_gconst(position, direction, fov_deg, width, height, z_near=0.1, z_far=1000.0, aspect=None):
    g = GConst()
    asp = (width / height) if aspect is None else aspect
    lib().orc_camera_gconst(f3(position), f3(direction), float(np.float32(np.deg2rad(np.float32(fov_deg)))), float(np.float32(asp)),
                            z_near, z_far, float(width), float(height), C.byref(g))
    return g


INSTANCE_DTYPE = np.dtype([("geometry_first", np.uint32), ("geometry_count", np.uint32), ("transform", np.float32, 16)])


class Scene:
    def __init__(self, mesh, sky=None, bluenoise=None, build=True, leaf_size=2, node_width=4, quantized=1, collapse=2, sah_top=1, tree_order=0, instances=None, dp_costs=None, top_opt=None, recull=0):
        L = lib()
        self.h = L.orc_scene_create()
        L.orc_accel_set_layout(self.h, leaf_size, node_width, quantized)
        L.orc_accel_set_collapse(self.h, collapse)
        L.orc_accel_set_sah_top(self.h, sah_top)
        L.orc_accel_set_tree_order(self.h, tree_order)
        if dp_costs:
            L.orc_accel_set_dp_costs(self.h, *dp_costs)
        if top_opt:
            L.orc_accel_set_top_opt(self.h, *top_opt)
        L.orc_accel_set_recull(self.h, recull)
        self.mesh = mesh
        v = np.ascontiguousarray(mesh.vertices, np.float32); i = np.ascontiguousarray(mesh.indices, np.uint32)
        L.orc_scene_set_vertices(self.h, ptr(v), len(v))
        L.orc_scene_set_indices(self.h, ptr(i), len(i))
        g = np.ascontiguousarray(mesh.geometries); pc = np.ascontiguousarray(mesh.prim_counts, np.uint32)
        L.orc_scene_set_geometry(self.h, ptr(g), ptr(pc), len(g))
        if instances:
            self.set_instances(instances, build=False)
        if sky is not None:
            s = np.ascontiguousarray(sky, np.float32)
            L.orc_scene_set_sky(self.h, ptr(s), s.shape[1], s.shape[0])
        if bluenoise is not None:
            b = np.ascontiguousarray(bluenoise, np.uint8)
            L.orc_scene_set_bluenoise(self.h, ptr(b), b.shape[1], b.shape[0])
        for i, t in enumerate(getattr(mesh, "textures", None) or []):
            t = np.ascontiguousarray(t, np.uint8)
            assert L.orc_scene_set_texture(self.h, i, ptr(t), t.shape[1], t.shape[0]) == 0
        if build:
            L.orc_accel_build(self.h)

    def set_instances(self, instances, build=True):
        """[(geometry_first, geometry_count, 4x4 object -> world)], as Context.set_instances of the product"""
        arr = np.zeros(max(1, len(instances)), INSTANCE_DTYPE)
        for k, (first, count, m) in enumerate(instances):
            arr[k]["geometry_first"], arr[k]["geometry_count"] = first, count
            arr[k]["transform"] = np.asarray(m, np.float32).T.ravel()  # column-major
        assert lib().orc_scene_set_instances(self.h, ptr(arr), len(instances)) == 0
        if build:
            lib().orc_accel_build(self.h)

    def __del__(self):
        try:
            lib().orc_scene_destroy(self.h)
        except Exception:
            pass

    @property
    def n_tris(self):
        return lib().orc_accel_num_tris(self.h)

    @property
    def n_nodes(self):
        return lib().orc_accel_num_nodes(self.h)

    @property
    def max_depth(self):
        return lib().orc_accel_max_depth(self.h)

    def nodes(self):
        n, w = self.n_nodes, lib().orc_accel_node_words(self.h)
        return np.ctypeslib.as_array(C.cast(lib().orc_accel_nodes(self.h), C.POINTER(C.c_uint32)), (n, w)).copy()

    def tris(self):
        n = self.n_tris
        return np.ctypeslib.as_array(C.cast(lib().orc_accel_tris(self.h), C.POINTER(C.c_uint32)), (n, 12)).copy()

    def codes(self):
        n = self.n_tris
        return np.ctypeslib.as_array(C.cast(lib().orc_accel_codes(self.h), C.POINTER(C.c_uint64)), (n,)).copy()

    def sky_tables(self, w, h):
        L = lib()
        al = np.ctypeslib.as_array(C.cast(L.orc_sky_alias(self.h), C.POINTER(C.c_uint32)), (h, w)).copy()
        tx = np.ctypeslib.as_array(C.cast(L.orc_sky_texels(self.h), C.POINTER(C.c_uint32)), (h, w)).copy()
        cm = np.ctypeslib.as_array(C.cast(L.orc_sky_cdf_marg(self.h), C.POINTER(C.c_float)), (h,)).copy()
        pu = np.ctypeslib.as_array(C.cast(L.orc_sky_pdf_uv(self.h), C.POINTER(C.c_float)), (h, w)).copy()
        return al, tx, cm, pu

    def trace_closest(self, rays, threads=8, counts=False):
        """rays: (8, n) float32 SoA ox,oy,oz,dx,dy,dz,tmin,tmax"""
        rays = np.ascontiguousarray(rays, np.float32); n = rays.shape[1]
        t = np.empty(n, np.float32); u = np.empty(n, np.float32); v = np.empty(n, np.float32); p = np.empty(n, np.uint32)
        nn = np.empty(n, np.uint32) if counts else None; nt = np.empty(n, np.uint32) if counts else None
        lib().orc_trace_closest(self.h, ptr(rays), n, ptr(t), ptr(u), ptr(v), ptr(p), ptr(nn), ptr(nt), threads)
        return (t, u, v, p, nn, nt) if counts else (t, u, v, p)

    def trace_any(self, rays, threads=8, counts=False):
        rays = np.ascontiguousarray(rays, np.float32); n = rays.shape[1]
        occ = np.empty(n, np.uint32)
        nn = np.empty(n, np.uint32) if counts else None; nt = np.empty(n, np.uint32) if counts else None
        lib().orc_trace_any(self.h, ptr(rays), n, ptr(occ), ptr(nn), ptr(nt), threads)
        return (occ, nn, nt) if counts else occ

    def trace_brute(self, rays, mode=0, threads=8):
        rays = np.ascontiguousarray(rays, np.float32); n = rays.shape[1]
        t = np.empty(n, np.float32); u = np.empty(n, np.float32); v = np.empty(n, np.float32); p = np.empty(n, np.uint32)
        lib().orc_trace_brute(self.h, ptr(rays), n, ptr(t), ptr(u), ptr(v), ptr(p), mode, threads)
        return t, u, v, p

    def gbuffer(self, g, rect=None, threads=8):
        W, H = int(g.window_size[0]), int(g.window_size[1])
        x0, y0, x1, y1 = rect or (0, 0, W, H)
        gb = np.zeros((H, W, 4), np.uint32); depth = np.zeros((H, W), np.float32)
        lib().orc_pass_gbuffer(self.h, C.byref(g), x0, y0, x1, y1, ptr(gb), ptr(depth), threads)
        return gb, depth

    def reference_mode(self, g, gb, depth, prev=None, rect=None, threads=8):
        W, H = int(g.window_size[0]), int(g.window_size[1])
        x0, y0, x1, y1 = rect or (0, 0, W, H)
        light = np.zeros((H, W, 4), np.float32) if prev is None else prev.copy()
        prev = np.zeros((H, W, 4), np.float32) if prev is None else np.ascontiguousarray(prev, np.float32)
        counts = np.zeros(6, np.uint64)
        lib().orc_pass_reference_mode(self.h, C.byref(g), x0, y0, x1, y1, ptr(gb), ptr(depth), ptr(prev), ptr(light), ptr(counts), threads)
        return light, counts

    def bounce1_rays(self, g, gb, depth):
        """(8, n) float32: the extension rays the path tracer traces after the first shade of sample 0 (SURVEY 8d's bounce-1 batch)"""
        cap = int(depth.size)
        rays = np.zeros((8, cap), np.float32)
        n = lib().orc_bounce1_rays(self.h, C.byref(g), ptr(gb), ptr(depth), ptr(rays), cap)
        return np.ascontiguousarray(rays[:, :n])

    def postprocess(self, g, depth, img, rect=None, threads=8):
        W, H = int(g.window_size[0]), int(g.window_size[1])
        x0, y0, x1, y1 = rect or (0, 0, W, H)
        out = np.zeros((H, W, 4), np.float32)
        lib().orc_pass_postprocess(self.h, C.byref(g), x0, y0, x1, y1, ptr(depth), ptr(np.ascontiguousarray(img, np.float32)), ptr(out), threads)
        return out


# ---- probe-GI passes (oracle/rt3_oracle_probes.c)
def sh_buffer_floats(probes_x, probes_y):
    """floats in a float3x3 buffer indexed by zcurve(3 * gx + c, gy) (12 floats per element)"""
    return 12 * (lib().orc_zcurve(probes_x * 3 - 1, probes_y - 1) + 1)


def structured_importance_sampling(g, gb, probes_x, probes_y):
    out = np.zeros((probes_y * 8, probes_x * 8), np.uint16); dbg = np.zeros((probes_y * 8, probes_x * 8), np.float32)
    lib().orc_pass_structured_importance_sampling(C.byref(g), probes_x, probes_y, ptr(np.ascontiguousarray(gb, np.uint32)), ptr(out), ptr(dbg))
    return out, dbg


def trace_probes(scene, g, gb, depth, directions, prev_atlas, threads=8):
    ah, aw = directions.shape
    atlas = np.zeros((ah, aw, 4), np.float32)
    lib().orc_pass_trace_probes(scene.h, C.byref(g), aw // 8, ah // 8, ptr(np.ascontiguousarray(gb, np.uint32)), ptr(np.ascontiguousarray(depth, np.float32)),
                                ptr(np.ascontiguousarray(directions, np.uint16)), ptr(np.ascontiguousarray(prev_atlas, np.float32)), ptr(atlas), threads)
    return atlas


def sh_conversion(atlas):
    ah, aw = atlas.shape[:2]
    out = np.zeros(sh_buffer_floats(aw // 8, ah // 8), np.float32)
    lib().orc_pass_sh_conversion(aw // 8, ah // 8, ptr(np.ascontiguousarray(atlas, np.float32)), ptr(out))
    return out


def interpolate_probes(g, gb, depth, sh, light_in=None):
    W, H = int(g.window_size[0]), int(g.window_size[1])
    light = np.zeros((H, W, 4), np.float32) if light_in is None else np.ascontiguousarray(light_in, np.float32).copy()
    lib().orc_pass_interpolate_probes(C.byref(g), ptr(np.ascontiguousarray(gb, np.uint32)), ptr(np.ascontiguousarray(depth, np.float32)),
                                      ptr(np.ascontiguousarray(sh, np.float32)), ptr(light))
    return light


def primary_rays(g, xs, ys, tmin=0.0, tmax=BACKGROUND_DEPTH):
    xs = np.ascontiguousarray(xs, np.uint32).ravel(); ys = np.ascontiguousarray(ys, np.uint32).ravel()
    n = len(xs)
    rays = np.empty((8, n), np.float32)
    lib().orc_primary_rays(C.byref(g), ptr(xs), ptr(ys), n, tmin, tmax, ptr(rays))
    return rays


def tile_pixels(w, h, rank, n_ranks):
    n = lib().orc_tile_pixels(w, h, rank, n_ranks, None)
    out = np.empty((n, 2), np.uint32)
    lib().orc_tile_pixels(w, h, rank, n_ranks, ptr(out))
    return out


def shared_edge_rays(mesh, h=0.02, seed=0, limit=None):
    """Watertightness sweep: for every edge that two triangles share (bit-identical end points) and every vertex whose edges are all
    shared, rays from both sides of the surface (offset `h` along the triangle's geometric normal) aimed exactly at the edge midpoint /
    the vertex.  Returns (rays (8, m) float32, h): a ray that does not hit within h + 1e-3 has slipped between the two triangles."""
    tri = mesh.triangle_positions().astype(np.float32)
    n = np.cross((tri[:, 1] - tri[:, 0]).astype(np.float64), (tri[:, 2] - tri[:, 0]).astype(np.float64))
    ln = np.linalg.norm(n, axis=1)
    ok = ln > 1e-12
    tri, n = tri[ok], (n[ok] / ln[ok, None])
    keys = tri.view(np.uint32).reshape(len(tri), 3, 3)
    _, vid = np.unique(keys.reshape(-1, 3), axis=0, return_inverse=True)  # vertex identity = its exact position
    vid = vid.reshape(-1, 3)
    e = np.stack([vid[:, [0, 1]], vid[:, [1, 2]], vid[:, [2, 0]]], 1)  # (t, 3, 2)
    ek = np.sort(e.reshape(-1, 2), axis=1)
    ekey = ek[:, 0].astype(np.int64) * (vid.max() + 1) + ek[:, 1]
    uniq, inv, cnt = np.unique(ekey, return_inverse=True, return_counts=True)
    shared = (cnt[inv] == 2).reshape(-1, 3)  # per triangle edge
    mids = np.stack([(tri[:, 0].astype(np.float64) + tri[:, 1]) / 2, (tri[:, 1].astype(np.float64) + tri[:, 2]) / 2, (tri[:, 2].astype(np.float64) + tri[:, 0]) / 2], 1)
    targets = [mids[shared], ]
    normals = [np.repeat(n[:, None, :], 3, 1)[shared], ]
    # vertices all of whose incident edges are shared
    bad_v = np.zeros(vid.max() + 1, bool)
    np.logical_or.at(bad_v, e[~shared].reshape(-1), True)
    vgood = ~bad_v[vid]
    targets.append(tri.astype(np.float64)[vgood])
    normals.append(np.repeat(n[:, None, :], 3, 1)[vgood])
    T, N = np.concatenate(targets), np.concatenate(normals)
    if limit and len(T) > limit:
        pick = np.random.default_rng(seed).choice(len(T), limit, replace=False)
        T, N = T[pick], N[pick]
    rays = []
    for sgn in (1.0, -1.0):
        o = (T + sgn * h * N).astype(np.float32)
        d = T.astype(np.float32) - o
        d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        rays.append(np.concatenate([o.T, d.T, np.full((1, len(o)), 1e-4, np.float32), np.full((1, len(o)), 1e5, np.float32)]))
    return np.ascontiguousarray(np.concatenate(rays, 1), np.float32), h
