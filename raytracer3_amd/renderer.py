"""Frame description: the path-tracing counterpart of `renderer::commands` (src/renderer/mod.rs:65-106) and of the
`Camera` component (src/components/camera.rs:23-59), plus the tile-partitioned multi-GPU frame (north_star).

Per frame, three passes exactly as the old shaders were wired:
  RayTracingPass("gbuffer")        -> packed G-buffer + depth                 shaders/old/gbuffer.slang
  RayTracingPass("refrence_mode")  -> Light (RGBA32F linear radiance)          shaders/old/refrence_mode.slang
  ComputePass("postprocess")       -> display image (AgX)                      shaders/old/postprocess.slang
and, as a second frame description, the probe-GI chain of the old shaders (`probe_commands`):
  gbuffer -> structured_importance_sampling -> trace_probes -> spherical_harmonic_conversion -> interpolate_probes
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _lib as L
from .render_graph import IMPORTED, ComputePass, Context, DispatchSize, ImageSize, RayTracingPass, RenderGraph, WorkSize2D

DEFAULT_FLAGS = L.F_NEE_SKY | L.F_BLUENOISE | L.F_FACEFORWARD | L.F_SPECULAR  # the full north_star estimator


class Camera:
    """components/camera.rs:23-59.  `fov` in radians, like the reference (`65.0_f32.to_radians()`, main.rs:72)."""

    def __init__(self, position, direction, fov, aspect_ratio, z_near=0.1, z_far=1000.0):
        d = np.asarray(direction, np.float32)
        self.position = np.asarray(position, np.float32)
        self.direction = d / np.float32(np.linalg.norm(d))  # Camera::new normalises, camera.rs:43
        self.fov, self.aspect_ratio, self.z_near, self.z_far = float(np.float32(fov)), float(np.float32(aspect_ratio)), z_near, z_far

    def gconst(self, window) -> L.GConst:
        """view_matrix / projection_matrix (camera.rs:52-58) + the GConst fill of renderer/mod.rs:72-78."""
        g = L.GConst()
        p = (C.c_float * 3)(*[float(x) for x in self.position])
        d = (C.c_float * 3)(*[float(x) for x in self.direction])
        L.load().rt3_camera_gconst(p, d, self.fov, self.aspect_ratio, self.z_near, self.z_far, float(window[0]), float(window[1]), C.byref(g))
        return g

    def view_matrix(self, window=(1, 1)):
        return np.array(self.gconst(window).view[:], np.float32).reshape(4, 4).T

    def projection_matrix(self, window=(1, 1)):
        return np.array(self.gconst(window).proj[:], np.float32).reshape(4, 4).T


class PathTracer:
    """One GPU's share of the frame.  `rank` / `n_ranks` select the interleaved 64x64 tiles this process renders."""

    def __init__(self, window, device=0, rank=0, n_ranks=1):
        self.window = (int(window[0]), int(window[1]))
        self.ctx = Context(device)
        self.rank, self.n_ranks = rank, n_ranks
        self.ctx.set_tile_partition(self.window[0], self.window[1], rank, n_ranks)
        self.rg = RenderGraph(self.ctx, self.window)
        self._accel = None
        self.comm_ready = False  # True once init_comm() has joined librt3's RCCL communicator
        self._stage = None       # rehearsal path only: staging buffer of the host-moved gather
        self.host_group = None   # process group of the host-moved gather (None = the default group)

    def close(self):
        self.ctx.close()

    def set_scene(self, mesh, sky=None, bluenoise=None):
        self.ctx.upload_mesh(mesh)
        if sky is not None:
            self.ctx.set_sky(sky)
        if bluenoise is not None:
            self.ctx.set_bluenoise(bluenoise)
        self._accel = self.ctx.build_accel()

    def make_gconst(self, camera: Camera, samples, bounces=4, frame=0, blendfactor=1.0, flags=DEFAULT_FLAGS) -> L.GConst:
        g = camera.gconst(self.window)
        g.frame, g.samples, g.bounces, g.blendfactor = frame, samples, bounces, blendfactor
        g.pad[0] = flags
        return g

    def commands(self, gconst: L.GConst, postprocess=True):
        """Build this frame's nodes (the analogue of renderer::commands, renderer/mod.rs:65-106)."""
        rg = self.rg
        rg.begin_frame()
        gbuffer = rg.image(ImageSize.FullScreen, L.FORMAT_R32G32B32A32_UINT, "gbuffer")
        depth = rg.image(ImageSize.FullScreen, L.FORMAT_R32_SFLOAT, "gbuffer_depth")
        light = rg.image(ImageSize.FullScreen, L.FORMAT_R32G32B32A32_SFLOAT, "Light")
        prev = rg.image(ImageSize.FullScreen, L.FORMAT_R32G32B32A32_SFLOAT, "PrevLight")
        out = rg.image(ImageSize.FullScreen, L.FORMAT_R32G32B32A32_SFLOAT, "color")
        gb = (RayTracingPass.new(rg, "gbuffer").shader("gbuffer").constants(gconst)
              .write(IMPORTED, gbuffer).write(IMPORTED, depth).launch(WorkSize2D.FullScreen))
        pt = (RayTracingPass.new(rg, "refrence_mode").shader("refrence_mode").constants(gconst)
              .read(gb, gbuffer).read(gb, depth).write(IMPORTED, light).read(IMPORTED, prev).launch(WorkSize2D.FullScreen))
        if postprocess:
            (ComputePass.new(rg, "postprocess").shader("postprocess").constants(gconst)
             .read(gb, depth).write(IMPORTED, out).read(pt, light).dispatch(DispatchSize.FullScreen))
        self.handles = dict(gbuffer=gbuffer, depth=depth, light=light, prev=prev, color=out)
        return self.handles

    def probe_commands(self, gconst: L.GConst):
        """The probe-GI frame: one probe per 16x16 pixel block, 8x8 rays per probe in the probe atlas.  Returns the handles; `Light`
        receives the interpolated image.  Under a tile partition (n_ranks > 1) the chain runs REPLICATED: it reads the whole G-buffer
        (jittered neighbours, probes up to two cells away, red marks scattered to other pixels) and is launch-bound at well under a
        millisecond, so every rank renders it for the full window (render_probes switches the partition off around it) and the
        frame-end gather of each rank's own tiles assembles the same image on the root as a single rank would produce."""
        rg = self.rg
        rg.begin_frame()
        W, H = self.window
        px, py = W // 16, H // 16
        asize = ImageSize.XY(px * 8, py * 8)
        gbuffer = rg.image(ImageSize.FullScreen, L.FORMAT_R32G32B32A32_UINT, "gbuffer")
        depth = rg.image(ImageSize.FullScreen, L.FORMAT_R32_SFLOAT, "gbuffer_depth")
        light = rg.image(ImageSize.FullScreen, L.FORMAT_R32G32B32A32_SFLOAT, "Light")
        directions = rg.image(asize, L.FORMAT_R16_UINT, "probe_directions")
        debug = rg.image(asize, L.FORMAT_R32_SFLOAT, "probe_debug")
        atlas = rg.image(asize, L.FORMAT_R32G32B32A32_SFLOAT, "probe_atlas")
        prev_atlas = rg.image(asize, L.FORMAT_R32G32B32A32_SFLOAT, "prev_probe_atlas")
        sh = rg.buffer(sh_buffer_bytes(px, py), "sh_coeficents")
        gb = (RayTracingPass.new(rg, "gbuffer").shader("gbuffer").constants(gconst)
              .write(IMPORTED, gbuffer).write(IMPORTED, depth).launch(WorkSize2D.FullScreen))
        sis = (ComputePass.new(rg, "structured_importance_sampling").shader("structured_importance_sampling").constants(gconst)
               .read(gb, gbuffer).read(gb, depth).write(IMPORTED, directions).write(IMPORTED, debug).read(IMPORTED, atlas)
               .dispatch(DispatchSize.XY(px, py)))
        tp = (RayTracingPass.new(rg, "trace_probes").shader("trace_probes").constants(gconst)
              .read(gb, gbuffer).read(gb, depth).read(sis, directions).write(IMPORTED, atlas).read(IMPORTED, prev_atlas)
              .launch(WorkSize2D.XY(px * 8, py * 8)))
        shc = (ComputePass.new(rg, "spherical_harmonic_conversion").shader("spherical_harmonic_conversion").constants(gconst)
               .write(IMPORTED, sh).read(tp, atlas).dispatch(DispatchSize.XY(px, py)))
        (ComputePass.new(rg, "interpolate_probes").shader("interpolate_probes").constants(gconst)
         .read(gb, gbuffer).read(gb, depth).read(shc, sh).write(IMPORTED, light).dispatch(DispatchSize.FullScreen))
        self.handles = dict(gbuffer=gbuffer, depth=depth, light=light, directions=directions, debug=debug, atlas=atlas, prev_atlas=prev_atlas, sh=sh)
        return self.handles

    def render_probes(self, gconst, wait=True):
        h = self.probe_commands(gconst)
        W, H = self.window
        if self.n_ranks > 1:  # replicas: the whole window on every rank (see probe_commands)
            self.ctx.set_tile_partition(W, H, 0, 1)
        try:
            self.rg.draw_frame(h["light"], wait=wait)
        finally:
            if self.n_ranks > 1:  # launches read the partition when they are enqueued: safe to restore behind them
                self.ctx.set_tile_partition(W, H, self.rank, self.n_ranks)
        return h

    def copy_atlas_to_prev(self):
        """Temporal blend input of trace_probes (prev_probe_atlas, trace_probes.slang:12,74)."""
        W, H = self.window
        self.rg.upload(self.handles["prev_atlas"], self.rg.download(self.handles["atlas"], (H // 16 * 8, W // 16 * 8, 4), np.float32))

    def render(self, gconst, postprocess=False, wait=True):
        h = self.commands(gconst, postprocess)
        self.rg.draw_frame(h["color"] if postprocess else h["light"], wait=wait)
        return h

    # ---- results
    def light(self):
        W, H = self.window
        return self.rg.download(self.handles["light"], (H, W, 4), np.float32)

    def color(self):
        W, H = self.window
        return self.rg.download(self.handles["color"], (H, W, 4), np.float32)

    def gbuffer(self):
        W, H = self.window
        return self.rg.download(self.handles["gbuffer"], (H, W, 4), np.uint32), self.rg.download(self.handles["depth"], (H, W), np.float32)

    def copy_light_to_prev(self):
        """Progressive accumulation: PrevLight <- Light (refrence_mode.slang:11,61-65), through the host (tests)."""
        self.rg.upload(self.handles["prev"], self.light())

    def swap_light_prev(self):
        """Progressive accumulation without a copy: the two images trade names, so the next frame's `PrevLight` is this frame's
        `Light` (refrence_mode.slang:10-11,61-65; the reference's two images are distinct resources as well)."""
        n = self.rg.named
        n["Light"], n["PrevLight"] = n["PrevLight"], n["Light"]

    def load_prev(self, image):
        """Resume a progressive render: `image` (H, W, 4) float32 becomes the next frame's `PrevLight` (checkpoint row)."""
        W, H = self.window
        h = self.rg.image(ImageSize.FullScreen, L.FORMAT_R32G32B32A32_SFLOAT, "PrevLight")
        self.rg.upload(h, np.ascontiguousarray(image, np.float32).reshape(H, W, 4))

    # ---- multi-GPU: ONE gather of the per-rank tile buffers at frame end (include/rt3.h: rt3_gather_tiles, RCCL inside librt3)
    def tile_pixel_count(self, rank):
        return self.ctx.tile_pixel_count(rank, self.n_ranks)

    def init_comm(self, uid: bytes):
        """Collective over all ranks: join librt3's own RCCL communicator.  `uid` is rank 0's `ctx.comm_unique_id()`, carried to the
        other ranks by the host (bench.py: through the torch.distributed key-value store)."""
        self.ctx.comm_init(uid, self.rank, self.n_ranks)
        self.comm_ready = True

    def gather_light(self, dist=None, torch=None, dst=0, download=True):
        """Assemble the frame on rank `dst` from every rank's tiles of `Light` with the frame's one collective.
        With a communicator (`init_comm`): `rt3_gather_tiles`, enqueued on librt3's stream, no host synchronisation.
        Without one (REHEARSAL on a one-GPU box, `RT3_DIST_BACKEND=gloo`): the same layout and the same single untile launch
        (`rt3_gather_layout` / `rt3_gather_unpack`), the bytes moved through host tensors by `exchange_tiles_host`.
        Returns the full (H, W, 4) image on `dst` (None elsewhere); with download=False the assembled frame stays in `dst`'s HBM
        (the `Light` image) and True is returned on `dst` instead.  With n_ranks == 1 no collective is issued."""
        img = self.handles["light"]
        if self.n_ranks > 1:
            if self.comm_ready:
                self.ctx.gather_tiles(img, dst)
            else:
                off = self.ctx.gather_layout(img, dst, self.n_ranks)
                mine = None
                if self.rank != dst:
                    n = self.tile_pixel_count(self.rank)
                    if self._stage is None:
                        self._stage = self.rg.buffer(max(n, 1) * 16, "gather_stage")
                    ptr, _ = self.rg.device_ptr(self._stage)
                    self.ctx.check(self.ctx.lib.rt3_image_pack_tiles(self.ctx.h, img, self.rank, self.n_ranks, C.c_void_p(ptr)))
                    mine = self.rg.download(self._stage, (max(n, 1), 4), np.float32)[:n]
                recv = exchange_tiles_host(dist, torch, self.rank, self.n_ranks, off, mine, dst, group=self.host_group)
                if self.rank == dst and off[-1]:
                    if self._stage is None:
                        self._stage = self.rg.buffer(off[-1] * 16, "gather_stage")
                    self.rg.upload(self._stage, recv)
                    self.ctx.gather_unpack(img, dst, self.n_ranks, self.rg.device_ptr(self._stage)[0])
            if self.rank != dst:
                return None
        return self.light() if download else True


def gather_offsets(counts, root):
    """Pixel offsets of the frame-end gather's receive buffer (what `rt3_gather_layout` returns): ranks in ascending order, exact
    counts, nothing from `root` itself (its tiles are already in its image).  n_ranks + 1 entries."""
    off = [0]
    for r, c in enumerate(counts):
        off.append(off[-1] + (0 if r == root else int(c)))
    return off


def exchange_tiles_host(dist, torch, rank, n_ranks, offsets, mine, dst=0, group=None):
    """Host-memory stand-in for the RCCL exchange inside `rt3_gather_tiles` (gloo: CPU tests, one-GPU rehearsal): every rank
    but `dst` sends its packed tiles (n x 4 float32) point to point; `dst` receives rank r's at offsets[r]..offsets[r+1] of ONE
    contiguous buffer, all receives posted together.  Returns that buffer on `dst`, None elsewhere."""
    if rank != dst:
        if mine is not None and len(mine):
            dist.send(torch.from_numpy(np.ascontiguousarray(mine, np.float32)), dst, group=group)
        return None
    recv = torch.empty((offsets[-1], 4), dtype=torch.float32)
    reqs = [dist.irecv(recv[offsets[r]:offsets[r + 1]], src=r, group=group) for r in range(n_ranks) if r != dst and offsets[r + 1] > offsets[r]]
    for q in reqs:
        q.wait()
    return recv.numpy()


def zcurve(x, y):
    """ZCurveToLinearIndex (shaders/include/math.slang:105-117)"""
    def explode(v):
        v = (v | (v << 8)) & 0x00FF00FF
        v = (v | (v << 4)) & 0x0F0F0F0F
        v = (v | (v << 2)) & 0x33333333
        return (v | (v << 1)) & 0x55555555
    return explode(x) | (explode(y) << 1)


def sh_buffer_bytes(probes_x, probes_y):
    """Bytes of the float3x3 buffer spherical_harmonic_conversion writes at zcurve(3 * gx + c, gy) (48 B per element)."""
    return 48 * (zcurve(probes_x * 3 - 1, probes_y - 1) + 1)


def default_camera(window, position, direction, fov_deg):
    return Camera(position, direction, math.radians(fov_deg), window[0] / window[1])
