#!/usr/bin/env python3
"""Soak: random scenes / cameras / sizes / path depths / feature flags / tile partitions, GPU frame against the oracle bit for bit,
for a given number of seconds (default 240).  Not a pytest test (its running time is the point); run on the GPU box:
    python tests/experiments/soak_parity.py [seconds] [seed]
Prints one line per case and a summary; exits non-zero at the first mismatch with the parameters that produced it."""
import math
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import orc  # noqa: E402
import torch  # noqa: E402,F401

from raytracer3_amd import _lib as L  # noqa: E402
from raytracer3_amd import assets, scenes  # noqa: E402
from raytracer3_amd.renderer import Camera, PathTracer  # noqa: E402
from test_gpu_parity import as_orc  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
bn = assets.load_bluenoise()
skies = [scenes.sky(256, 128), scenes.sky(512, 256)]
pool = [("atrium0.2", scenes.atrium(0.2), (-12, 0.5, -6), (12, 10, 6)), ("atrium0.35", scenes.atrium(0.35), (-12, 0.5, -6), (12, 10, 6)),
        ("cornell", scenes.cornell(), (-0.9, 0.1, -0.9), (0.9, 1.9, 0.9)), ("cornell_ref", scenes.cornell_ref(), (-0.9, -0.9, -0.9), (0.9, 0.9, 6.0))]
t_end = time.time() + seconds
n_cases = 0
while time.time() < t_end:
    name, mesh, lo, hi = pool[int(rng.integers(len(pool)))]
    sky = skies[int(rng.integers(len(skies)))] if rng.random() < 0.85 else None
    W, H = int(rng.integers(8, 40)) * 8, int(rng.integers(6, 30)) * 8
    spp, bounces = int(rng.integers(1, 7)), int(rng.integers(1, 6))
    flags = int(rng.integers(0, 16))
    if sky is None:
        flags &= ~L.F_NEE_SKY
    frame = int(rng.integers(0, 1000))
    n_ranks = int(rng.choice([1, 1, 2, 3, 5, 8]))
    rank = int(rng.integers(n_ranks))
    batch = int(rng.choice([0, 0, 1, 2]))
    leaf, T = int(rng.choice([1, 2, 2, 3, 4])), int(rng.choice([0, 1, 1, 2, 4, 16]))
    collapse = int(rng.choice([2, 2, 2, 1, 0]))  # cost-driven (default), surface area, even depth
    # instances (round 3): the whole scene once under the identity plus, sometimes, one or two of its geometries placed again under random
    # rigid / scaled matrices inside the scene's box
    inst = []
    if rng.random() < 0.35:
        ng = len(mesh.geometries)
        inst.append((0, ng, np.eye(4, dtype=np.float32)))
        for _ in range(int(rng.integers(1, 3))):
            gsel = int(rng.integers(ng))
            a, sc, tr = float(rng.uniform(-3.1, 3.1)), rng.uniform(0.3, 1.5, 3), rng.uniform(lo, hi) * 0.3
            m = np.array([[math.cos(a) * sc[0], 0, math.sin(a) * sc[2], tr[0]], [0, sc[1], 0, tr[1]], [-math.sin(a) * sc[0], 0, math.cos(a) * sc[2], tr[2]], [0, 0, 0, 1]], np.float32)
            inst.append((gsel, 1, m))
    pos = rng.uniform(lo, hi)
    d = rng.normal(size=3)
    d /= np.linalg.norm(d)
    fov = float(rng.uniform(20, 100))
    params = dict(scene=name, sky=None if sky is None else sky.shape[:2], size=(W, H), spp=spp, bounces=bounces, flags=flags, frame=frame, rank=(rank, n_ranks), batch=batch,
                  leaf=leaf, T=T, collapse=collapse, instances=len(inst), pos=[round(float(x), 4) for x in pos], dir=[round(float(x), 4) for x in d], fov=round(fov, 2))
    pt = PathTracer((W, H), rank=rank, n_ranks=n_ranks)
    pt.ctx.set_option(L.OPT_LEAF_SIZE, leaf)
    pt.ctx.set_option(L.OPT_SAH_TOP, T)
    pt.ctx.set_option(L.OPT_WIDE_COLLAPSE, collapse)
    if inst:
        pt.ctx.upload_mesh(mesh)
        pt.ctx.set_instances(inst)
    pt.set_scene(mesh, sky, bn)
    if batch:
        pt.ctx.set_option(L.OPT_BATCH_SPP, batch)
    cam = Camera(tuple(float(x) for x in pos), tuple(float(x) for x in d), math.radians(fov), W / H)
    g = pt.make_gconst(cam, spp, bounces, frame=frame, flags=flags)
    pt.render(g, postprocess=False)
    light = pt.light()
    _, depth = pt.gbuffer()
    st = pt.ctx.stats()
    pt.close()
    osc = orc.Scene(mesh, sky, bn, leaf_size=leaf, sah_top=T, collapse=collapse, instances=inst or None)
    og = as_orc(g)
    ogb, odepth = osc.gbuffer(og)
    olight, counts = osc.reference_mode(og, ogb, odepth)
    xy = orc.tile_pixels(W, H, rank, n_ranks)  # this rank's pixels; it must not have touched any other
    own = np.zeros((H, W), bool)
    own[xy[:, 1], xy[:, 0]] = True
    ok = (np.array_equal(light.view(np.uint32)[own], olight.view(np.uint32)[own]) and np.array_equal(depth.view(np.uint32)[own], odepth.view(np.uint32)[own])
          and (light[~own] == 0).all())
    n_cases += 1
    print(("ok  " if ok else "FAIL"), params, "rays", st.extension_rays + st.shadow_rays, flush=True)
    if not ok:
        diff = np.abs(light[..., :3].astype(np.float64) - olight[..., :3])
        print("max abs diff", diff.max(), "pixels differing", int((light.view(np.uint32) != olight.view(np.uint32)).any(-1).sum()))
        sys.exit(1)
print(f"soak: {n_cases} cases bit-exact in {seconds:.0f} s")
