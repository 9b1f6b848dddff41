#!/usr/bin/env python3
"""Experiment: can the GPU overlap two wavefront batches?  One context renders 64 spp; two contexts (own streams) render
32 spp each concurrently.  Same total work; the wall-time ratio bounds what batch interleaving inside one context could buy."""
import math
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: F401  (first, see _lib.load)

from raytracer3_amd import assets, scenes
from raytracer3_amd.renderer import DEFAULT_FLAGS, Camera, PathTracer

W, H = 1920, 1080
mesh, sky, bn = scenes.atrium(1.0), scenes.sky(2048, 1024), assets.load_bluenoise()
cam = Camera(scenes.ATRIUM_CAMERA["position"], scenes.ATRIUM_CAMERA["direction"], math.radians(scenes.ATRIUM_CAMERA["fov_deg"]), W / H)


def make():
    pt = PathTracer((W, H), device=0)
    pt.set_scene(mesh, sky, bn)
    return pt


from raytracer3_amd import _lib as L  # noqa: E402

a, b = make(), make()
for blocks in (2048, 1536, 1024):  # persistent traversal workgroups per launch (process-wide knob): fewer leave wave slots to the other stream
    a.ctx.set_option(L.OPT_TRACE_BLOCKS, blocks)
    for spp_each, label in ((64, "one context, 64 spp"), (32, "two contexts, 32 spp each, concurrent"), (-32, "two contexts, 32 spp each, second one delayed")):
        for rep in range(3):
            ga = a.make_gconst(cam, abs(spp_each), 4, frame=rep, flags=DEFAULT_FLAGS)
            gb = b.make_gconst(cam, abs(spp_each), 4, frame=rep + 100, flags=DEFAULT_FLAGS)
            a.ctx.wait(); b.ctx.wait()
            t0 = time.perf_counter()
            a.render(ga, postprocess=False, wait=False)
            if spp_each < 0:
                time.sleep(0.008)  # roughly half a bounce: B's traversal then runs beside A's shading
            if spp_each != 64:
                b.render(gb, postprocess=False, wait=False)
            a.ctx.wait(); b.ctx.wait()
            dt = time.perf_counter() - t0
        print(f"blocks {blocks}: {label}: {dt * 1e3:.1f} ms", flush=True)
