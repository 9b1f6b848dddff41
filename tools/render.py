#!/usr/bin/env python3
"""Render a scene to PNG through the three passes (gbuffer -> refrence_mode -> postprocess), optionally progressively
(BASELINE.json configs[4]: accumulation over passes with blendfactor = 1/(pass+1), RMSE-vs-spp curve against the final image).

  python tools/render.py --scene atrium --size 960x540 --spp 64 --passes 16 --out gpurun_out/atrium.png --curve gpurun_out/curve.json
  python tools/render.py --scene cornell_ref --size 1920x1080 --spp 64 --passes 64 --bounces 2 --flags 0 \
                         --compare resources/refrence_480x270.png --side-by-side profiles/r02_cornell_ref_side_by_side.png
                          (informational only: a 480x270 copy of the reference tree's resources/refrence.png, a Blender-Cycles render of
                          the box scenes.cornell_ref() rebuilds from the reference's processed asset; Cycles is not this estimator)
  python tools/render.py --glb resources/sponza_scene.glb --exr resources/skybox2.exr ...               (if the real assets are dropped in)
"""
import argparse
import json
import math
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="atrium", choices=["atrium", "cornell", "cornell_ref"])
    ap.add_argument("--glb", default=None)
    ap.add_argument("--exr", default=None)
    ap.add_argument("--detail", type=float, default=1.0)
    ap.add_argument("--size", default="960x540")
    ap.add_argument("--spp", type=int, default=64, help="samples per pass")
    ap.add_argument("--passes", type=int, default=1)
    ap.add_argument("--bounces", type=int, default=4)
    ap.add_argument("--flags", type=int, default=-1)
    ap.add_argument("--out", default="gpurun_out/render.png")
    ap.add_argument("--curve", default=None, help="write the RMSE-vs-spp convergence curve (JSON) here")
    ap.add_argument("--compare", default=None, help="PNG to compare the tone-mapped result with (RMSE of 8-bit values / 255)")
    ap.add_argument("--side-by-side", default=None, help="with --compare: write [render | reference] at the reference's size here")
    ap.add_argument("--report", default=None, help="with --compare: write the numbers and the scene parameters as JSON here")
    args = ap.parse_args()

    from PIL import Image

    from raytracer3_amd import _lib as L
    from raytracer3_amd import assets, scenes
    from raytracer3_amd.renderer import DEFAULT_FLAGS, Camera, PathTracer

    W, H = (int(x) for x in args.size.split("x"))
    if args.glb:
        mesh, cam_kw = assets.GltfMeshLoader.load(args.glb), scenes.ATRIUM_CAMERA
    elif args.scene == "cornell":
        mesh, cam_kw = scenes.cornell(), scenes.CORNELL_CAMERA
    elif args.scene == "cornell_ref":
        mesh, cam_kw = scenes.cornell_ref(), scenes.CORNELL_REF_CAMERA
    else:
        mesh, cam_kw = scenes.atrium(args.detail), scenes.ATRIUM_CAMERA
    sky = assets.read_exr(args.exr) if args.exr else (scenes.sky(2048, 1024) if args.scene == "atrium" or args.glb else None)
    if sky is not None:  # rt3_scene_set_sky wants finite, non-negative radiance; HDR files carry inf / NaN / tiny negatives
        sky = np.clip(np.nan_to_num(sky, nan=0.0, posinf=65504.0, neginf=0.0), 0.0, 65504.0).astype(np.float32)
    flags = args.flags if args.flags >= 0 else (DEFAULT_FLAGS if sky is not None else L.F_FACEFORWARD | L.F_SPECULAR)
    pt = PathTracer((W, H))
    pt.set_scene(mesh, sky, assets.load_bluenoise())
    cam = Camera(cam_kw["position"], cam_kw["direction"], math.radians(cam_kw["fov_deg"]), W / H)
    history, t0 = [], time.perf_counter()
    for p in range(args.passes):
        g = pt.make_gconst(cam, args.spp, args.bounces, frame=p, blendfactor=1.0 / (p + 1), flags=flags)
        pt.render(g, postprocess=(p == args.passes - 1))
        if args.curve:
            history.append(pt.light()[..., :3].copy())
        if p != args.passes - 1:
            pt.copy_light_to_prev()
    dt = time.perf_counter() - t0
    st = pt.ctx.stats()
    color = pt.color()
    light = pt.light()
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    img8 = (np.clip(color[..., :3], 0, 1) * 255 + 0.5).astype(np.uint8)
    Image.fromarray(img8).save(args.out)
    rays = st.extension_rays + st.shadow_rays
    print(f"{args.out}: {W}x{H} {args.passes} x {args.spp} spp, {rays / 1e6:.0f} Mrays in {dt:.2f} s ({rays / dt / 1e6:.0f} Mrays/s incl. host), mean radiance {light[..., :3].mean():.4f}")
    if args.curve:
        ref = history[-1].astype(np.float64)
        curve = [{"spp": (i + 1) * args.spp, "rmse_vs_final": float(np.sqrt(np.mean((h - ref) ** 2)))} for i, h in enumerate(history[:-1])]
        Path(args.curve).write_text(json.dumps({"size": [W, H], "spp_per_pass": args.spp, "passes": args.passes, "curve": curve}, indent=1))
        print("convergence:", ", ".join(f"{c['spp']}spp {c['rmse_vs_final']:.4f}" for c in curve[:: max(1, len(curve) // 8)]))
    if args.compare:
        refimg = Image.open(args.compare).convert("RGB")
        rw, rh = refimg.size
        if W % rw == 0 and H % rh == 0 and W // rw == H // rh:  # box-filter the render down to the reference's size
            k = W // rw
            mine = img8.reshape(rh, k, rw, k, 3).astype(np.float64).mean((1, 3)) / 255.0
            ref = np.array(refimg, np.float64) / 255
        else:
            mine, ref = img8 / 255.0, np.array(refimg.resize((W, H)), np.float64) / 255
        rmse = float(np.sqrt(np.mean((mine - ref) ** 2)))
        print(f"informational RMSE vs {args.compare} (8-bit, display referred, another renderer): {rmse:.4f}")
        if args.side_by_side:
            Image.fromarray(np.concatenate([(mine * 255 + 0.5).astype(np.uint8), (ref * 255 + 0.5).astype(np.uint8)], 1)).save(args.side_by_side)
        if args.report:
            Path(args.report).write_text(json.dumps({
                "scene": args.scene, "render": {"size": [W, H], "spp": args.spp * args.passes, "bounces": args.bounces, "flags": flags},
                "camera": {k: (list(v) if isinstance(v, tuple) else v) for k, v in cam_kw.items()},
                "light_emission": scenes.CORNELL_REF_EMISSION if args.scene == "cornell_ref" else None,
                "compare": args.compare, "compared_at": [int(mine.shape[1]), int(mine.shape[0])], "rmse_8bit": rmse,
                "note": "informational: the reference image is a Blender-Cycles render (100 spp, its own view transform); scene rebuilt from the "
                        "eight cubes / materials of the reference's processed box.glb, placements and emission fitted to the image"}, indent=1))
    pt.close()


if __name__ == "__main__":
    main()
