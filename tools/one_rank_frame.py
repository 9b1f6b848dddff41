#!/usr/bin/env python3
"""One rank's share of the C3 frame on one GPU, for a kernel trace of what a rank of an N-GPU run executes:
    rocprofv3 --kernel-trace --stats --output-format csv -d out -- python3 tools/one_rank_frame.py --ranks 8 --rank 0 --frames 3"""
import argparse
import math
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: F401  (librt3 binds to torch's HIP runtime)

from raytracer3_amd import assets, scenes
from raytracer3_amd.renderer import DEFAULT_FLAGS, Camera, PathTracer

ap = argparse.ArgumentParser()
ap.add_argument("--ranks", type=int, default=8)
ap.add_argument("--rank", type=int, default=0)
ap.add_argument("--frames", type=int, default=3)
args = ap.parse_args()
W, H, SPP = 1920, 1080, 64
mesh, sky, bn = scenes.atrium(1.0), scenes.sky(2048, 1024), assets.load_bluenoise()
cam = Camera(scenes.ATRIUM_CAMERA["position"], scenes.ATRIUM_CAMERA["direction"], math.radians(scenes.ATRIUM_CAMERA["fov_deg"]), W / H)
pt = PathTracer((W, H), device=0, rank=args.rank, n_ranks=args.ranks)
pt.set_scene(mesh, sky, bn)
for f in range(args.frames + 1):  # the first frame is the warm-up
    pt.render(pt.make_gconst(cam, SPP, 4, frame=f, flags=DEFAULT_FLAGS), postprocess=False, wait=True)
st = pt.ctx.stats()
print(f"rank {args.rank} of {args.ranks}: {args.frames + 1} frames, {(st.extension_rays + st.shadow_rays) / (args.frames + 1) / 1e6:.1f} Mrays per frame")
pt.close()
