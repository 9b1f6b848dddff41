#!/usr/bin/env python3
"""What would a better tree buy ON THE GPU?  The oracle builds tree variants of the bench scene (tests/experiments/tree_quality.py has
their visit counts), rt3_accel_import installs each in the product, and the bench frame (1080p @ 64 spp, B = 4, all flags) is timed per
kernel with HIP events -- before any device builder is written for a variant.  Radiance must not depend on the tree: the light image of
every variant is compared with the default tree's bit for bit.  Test infrastructure (uses the oracle): lives under tests/.
    python tests/experiments/tree_quality_gpu.py [--spp 64]"""
import argparse, math, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
import orc
from raytracer3_amd import _lib as L, assets, scenes
from raytracer3_amd.renderer import DEFAULT_FLAGS, Camera, PathTracer

ap = argparse.ArgumentParser(); ap.add_argument('--spp', type=int, default=64); ap.add_argument('--steps', type=int, default=3)
args = ap.parse_args()
W, H = 1920, 1080
mesh = scenes.atrium(1.0); sky = scenes.sky(2048, 1024); bn = assets.load_bluenoise()
pt = PathTracer((W, H)); pt.set_scene(mesh, sky, bn)
cam = Camera(scenes.ATRIUM_CAMERA['position'], scenes.ATRIUM_CAMERA['direction'], math.radians(scenes.ATRIUM_CAMERA['fov_deg']), W / H)

def run(tag):
    g = pt.make_gconst(cam, args.spp, 4, frame=0, flags=DEFAULT_FLAGS)
    pt.render(g, postprocess=False)  # warm-up
    pt.ctx.set_option(L.OPT_COUNT_TRAVERSAL, 1); pt.ctx.stats_reset(); pt.render(g, postprocess=False); c = pt.ctx.stats(); pt.ctx.set_option(L.OPT_COUNT_TRAVERSAL, 0)
    pt.ctx.set_option(L.OPT_PROFILE, 1); pt.ctx.stats_reset()
    t0 = time.perf_counter()
    for i in range(args.steps): pt.render(pt.make_gconst(cam, args.spp, 4, frame=i, flags=DEFAULT_FLAGS), postprocess=False, wait=False)
    pt.ctx.wait(); dt = (time.perf_counter() - t0) / args.steps * 1e3
    st = pt.ctx.stats(); pt.ctx.set_option(L.OPT_PROFILE, 0)
    n = args.steps
    print(f"{tag:28s} frame {dt:7.2f} ms  k_extend {st.extend_ms / n:6.2f}  k_shadow {st.shadow_ms / n:6.2f}  k_shade {st.shade_ms / n:6.2f} | "
          f"ext {c.nodes_visited / c.extension_rays:.2f}+{c.tris_tested / c.extension_rays:.2f}  shadow {c.shadow_nodes_visited / c.shadow_rays:.2f}+{c.shadow_tris_tested / c.shadow_rays:.2f}  nodes {pt.ctx.accel_info()[0]}", flush=True)
    pt.render(pt.make_gconst(cam, 4, 4, frame=0, flags=DEFAULT_FLAGS), postprocess=False)
    return pt.light().copy()

ref = run('device build (default)')
variants = [('oracle default (same tree)', dict()), ('round 2: T=2, area collapse', dict(sah_top=2, collapse=1)), ('T=2 DP collapse', dict(sah_top=2, collapse=2)),
            ('T=1 DP leaf<=3', dict(sah_top=1, collapse=2, leaf_size=3)), ('T=1 DP leaf<=4', dict(sah_top=1, collapse=2, leaf_size=4)),
            ('T=1 area, tree order', dict(sah_top=1, collapse=1, tree_order=1)),
            ('DP c_node 1 c_tri 0.5', dict(dp_costs=(1.0, 0.5))), ('DP c_node 1 c_tri 1.5', dict(dp_costs=(1.0, 1.5))), ('DP c_node 1 c_tri 2', dict(dp_costs=(1.0, 2.0))),
            ('DP c_node 2 c_tri 1', dict(dp_costs=(2.0, 1.0))), ('DP c_node 0.5 c_tri 1, leaf<=4', dict(dp_costs=(0.5, 1.0), leaf_size=4))]
for name, kw in variants:
    osc = orc.Scene(mesh, **kw)
    pt.ctx.accel_import(osc.nodes(), osc.tris())
    img = run(name)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), name + ': radiance depends on the tree'
pt.close()
