#!/usr/bin/env python3
"""Choose the one free photometric parameter of scenes.cornell_ref() -- the emission of its light cube, which the reference's
processed asset does not carry -- against resources/refrence_480x270.png (a 480x270 copy of the reference tree's only image,
resources/refrence.png), on the CPU oracle (test infrastructure, hence this script lives under tests/; the GPU side-by-side is
tools/render.py --scene cornell_ref).
Radiance is linear in the emission, so ONE render with emission 1 is scaled before the display transform.
Also reports which path depth resembles the image most (the Cycles render looks like direct light only)."""
import sys
from pathlib import Path

import numpy as np
from PIL import Image

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import orc  # noqa: E402
from raytracer3_amd import scenes  # noqa: E402

W, H, SPP = 480, 270, 512
ref = np.array(Image.open(ROOT / "resources" / "refrence_480x270.png").convert("RGB"), np.float64) / 255.0
osc = orc.Scene(scenes.cornell_ref(emission=1.0))
for B in (2, 3, 5):
    g = orc.camera_gconst(width=W, height=H, **scenes.CORNELL_REF_CAMERA)
    g.bounces, g.samples, g.blendfactor, g.frame = B, SPP, 1.0, 0
    g.pad[0] = 0  # reference semantics: diffuse BSDF, emissive-only transport
    gb, depth = osc.gbuffer(g, threads=8)
    light, _ = osc.reference_mode(g, gb, depth, threads=8)
    best = None
    for e in np.geomspace(0.02, 4.0, 61):
        img = osc.postprocess(g, depth, (light * np.float32(e)).astype(np.float32), threads=8)
        i8 = np.floor(np.clip(img[..., :3], 0, 1) * 255 + 0.5) / 255.0
        rmse = float(np.sqrt(np.mean((i8 - ref) ** 2)))
        if best is None or rmse < best[0]:
            best = (rmse, float(e), i8)
    print(f"B={B}: best emission {best[1]:.3f} -> 8-bit RMSE {best[0]:.4f}")
    if len(sys.argv) > 1:
        Image.fromarray(np.concatenate([(best[2] * 255 + 0.5).astype(np.uint8), (ref * 255 + 0.5).astype(np.uint8)], 1)).save(f"{sys.argv[1]}_B{B}.png")
