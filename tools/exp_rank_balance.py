#!/usr/bin/env python3
"""What the 1 -> N strong-scaling run will see, measured on ONE GPU: render only rank r's tiles of the C3 frame for every
r of an N-rank partition and compare with 1/N of the full-frame time (load balance of the Z-order tile interleave and the
fixed per-frame cost of the launch sequence)."""
import math
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: F401

from raytracer3_amd import assets, scenes
from raytracer3_amd.renderer import DEFAULT_FLAGS, Camera, PathTracer

W, H, SPP = 1920, 1080, 64
mesh, sky, bn = scenes.atrium(1.0), scenes.sky(2048, 1024), assets.load_bluenoise()
cam = Camera(scenes.ATRIUM_CAMERA["position"], scenes.ATRIUM_CAMERA["direction"], math.radians(scenes.ATRIUM_CAMERA["fov_deg"]), W / H)


import os
from raytracer3_amd import _lib as L

FUSED = int(os.environ.get("RT3_EXP_FUSED", "-1"))  # -1: library default
CHUNK = int(os.environ.get("RT3_EXP_CHUNK", "0"))
BLOCKS = int(os.environ.get("RT3_EXP_BLOCKS", "0"))
NS = [int(x) for x in os.environ.get("RT3_EXP_N", "2,4,8").split(",")]


def timed(rank, n):
    pt = PathTracer((W, H), device=0, rank=rank, n_ranks=n)
    if FUSED >= 0:
        pt.ctx.set_option(L.OPT_FUSED_TRACE, FUSED)
    if CHUNK:
        pt.ctx.set_option(L.OPT_POOL_CHUNK, CHUNK)
    if BLOCKS:
        pt.ctx.set_option(L.OPT_TRACE_BLOCKS, BLOCKS)
    pt.set_scene(mesh, sky, bn)
    best = 1e9
    for rep in range(3):
        g = pt.make_gconst(cam, SPP, 4, frame=rep, flags=DEFAULT_FLAGS)
        pt.ctx.wait()
        t0 = time.perf_counter()
        pt.render(g, postprocess=False, wait=True)
        best = min(best, time.perf_counter() - t0)
    st = pt.ctx.stats()
    pt.close()
    return best * 1e3, (st.extension_rays + st.shadow_rays) / 3


full, rays = timed(0, 1)
print(f"N=1: {full:.2f} ms, {rays / 1e6:.0f} Mrays")
for n in NS:
    ts = [timed(r, n) for r in range(n)]
    worst = max(t for t, _ in ts)
    print(f"N={n}: per-rank ms " + " ".join(f"{t:.2f}" for t, _ in ts) + f" | slowest {worst:.2f} vs ideal {full / n:.2f} -> efficiency bound {full / n / worst:.3f}"
          f" | rays max/mean {max(r for _, r in ts) / (sum(r for _, r in ts) / n):.3f}")
