#!/usr/bin/env python3
"""Where does a rank's time go as the frame is split?  Per-kernel-category HIP-event times of rank 0 of N for N = 1, 2, 4, 8 on ONE
GPU (C3 frame), times N, next to the single-rank frame: what does not shrink with 1/N is the strong-scaling loss."""
import math
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: F401

from raytracer3_amd import _lib as L
from raytracer3_amd import assets, scenes
from raytracer3_amd.renderer import DEFAULT_FLAGS, Camera, PathTracer

W, H, SPP = 1920, 1080, 64
mesh, sky, bn = scenes.atrium(1.0), scenes.sky(2048, 1024), assets.load_bluenoise()
cam = Camera(scenes.ATRIUM_CAMERA["position"], scenes.ATRIUM_CAMERA["direction"], math.radians(scenes.ATRIUM_CAMERA["fov_deg"]), W / H)
for n in (1, 2, 4, 8):
    pt = PathTracer((W, H), device=0, rank=0, n_ranks=n)
    pt.set_scene(mesh, sky, bn)
    g = pt.make_gconst(cam, SPP, 4, frame=0, flags=DEFAULT_FLAGS)
    pt.render(g, postprocess=False, wait=True)
    pt.ctx.set_option(L.OPT_PROFILE, 1)
    pt.ctx.stats_reset()
    for f in range(3):
        pt.render(pt.make_gconst(cam, SPP, 4, frame=1 + f, flags=DEFAULT_FLAGS), postprocess=False, wait=True)
    st = pt.ctx.stats()
    print(f"N={n}: per frame of rank 0  extend {st.extend_ms / 3:.3f} ms ({st.extend_launches // 3} launches)  shadow {st.shadow_ms / 3:.3f}  shade {st.shade_ms / 3:.3f}  "
          f"other {st.other_ms / 3:.3f}   x N = {n * (st.extend_ms + st.shadow_ms + st.shade_ms + st.other_ms) / 3:.2f} ms", flush=True)
    pt.close()
