#!/usr/bin/env python3
"""Render the probe-GI frame of the old shaders (gbuffer -> structured_importance_sampling -> trace_probes ->
spherical_harmonic_conversion -> interpolate_probes) and time it.

  python tools/probe_frame.py --scene cornell --size 1920x1080 --frames 32 --out gpurun_out/probes.png

Frames accumulate through trace_probes' temporal blend (prev_probe_atlas, blendfactor); the radiance store is the one
trace_probes.slang:74 keeps in a comment (RT3_F_PROBE_RADIANCE) unless --as-written is given, which shows the debug state.
"""
import argparse
import math
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="cornell", choices=["atrium", "cornell"])
    ap.add_argument("--size", default="1920x1080")
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--blend", type=float, default=0.1)
    ap.add_argument("--as-written", action="store_true")
    ap.add_argument("--out", default="gpurun_out/probes.png")
    args = ap.parse_args()

    from PIL import Image

    from raytracer3_amd import _lib as L
    from raytracer3_amd import scenes
    from raytracer3_amd.renderer import Camera, PathTracer

    W, H = (int(x) for x in args.size.split("x"))
    mesh, cam_kw = (scenes.cornell(), scenes.CORNELL_CAMERA) if args.scene == "cornell" else (scenes.atrium(1.0), scenes.ATRIUM_CAMERA)
    pt = PathTracer((W, H))
    pt.set_scene(mesh, scenes.sky(512, 256) if args.scene == "atrium" else None)
    pt.ctx.set_option(L.OPT_PROFILE, 1)
    cam = Camera(cam_kw["position"], cam_kw["direction"], math.radians(cam_kw["fov_deg"]), W / H)
    flags = 0 if args.as_written else L.F_PROBE_RADIANCE
    wall = []
    for f in range(args.frames):
        g = pt.make_gconst(cam, 1, 1, frame=f, blendfactor=1.0 if f == 0 else args.blend, flags=flags)
        if f == 1:
            pt.ctx.stats_reset()
        t0 = time.perf_counter()
        pt.render_probes(g, wait=True)
        wall.append(time.perf_counter() - t0)
        pt.copy_atlas_to_prev()
    st = pt.ctx.stats()
    n = max(args.frames - 1, 1)
    print(f"{W}x{H}: {W // 16}x{H // 16} probes, {(W // 16) * (H // 16) * 64} probe rays + {W * H} primary rays per frame")
    print(f"per frame (HIP events, frames 1..): traversal {st.extend_ms / n:.3f} ms, other kernels {st.other_ms / n:.3f} ms; "
          f"host wall of render_probes {1e3 * float(np.median(wall)):.2f} ms")
    light = pt.light()
    rgb = np.clip(light[..., :3] / (1.0 + light[..., :3]), 0, 1) ** (1 / 2.2)  # Reinhard + gamma, for a quick look only
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Image.fromarray((rgb * 255 + 0.5).astype(np.uint8)).save(args.out)
    failed = (light[..., 0] == 1) & (light[..., 1] == 0) & (light[..., 2] == 0)
    print(f"wrote {args.out}; 'interpolation failed' marks: {100 * failed.mean():.1f} % of the pixels")
    pt.close()


if __name__ == "__main__":
    main()
