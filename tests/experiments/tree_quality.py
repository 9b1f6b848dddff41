#!/usr/bin/env python3
"""Tree-quality prototype: node / triangle steps per ray of the bench frame (atrium, full estimator, B = 4) on the
CPU oracle, for several builder settings, BEFORE touching the device builder.  A reduced window of the bench camera is enough: the per-ray
averages at 480x270 @ 2 spp agree with the full frame's to ~1 %.  Test infrastructure (uses the oracle): lives under tests/."""
import sys, time, ctypes as C, numpy as np
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
import orc
from raytracer3_amd import scenes, assets

def measure(osc, W=480, H=270, spp=2, flags=15):
    g = orc.camera_gconst(width=W, height=H, **scenes.ATRIUM_CAMERA)
    g.samples = spp; g.bounces = 4; g.frame = 0; g.blendfactor = 1.0; g.pad[0] = flags
    gb, depth = osc.gbuffer(g)
    _, c = osc.reference_mode(g, gb, depth)
    c = [float(x) for x in c]
    n_ext, n_sh = c[0], c[1]
    en, et = (c[2] - c[4]) / n_ext, (c[3] - c[5]) / n_ext
    sn, st = c[4] / max(n_sh, 1), c[5] / max(n_sh, 1)
    # primary rays separately
    ys, xs = np.mgrid[0:H, 0:W]
    pr = orc.primary_rays(g, xs.ravel(), ys.ravel())
    _, _, _, _, pn, pt = osc.trace_closest(pr, counts=True)
    return dict(ext_nodes=en, ext_tris=et, sh_nodes=sn, sh_tris=st, prim_nodes=float(pn.mean()), prim_tris=float(pt.mean()), n_ext=n_ext, n_sh=n_sh)

if __name__ == '__main__':
    mesh = scenes.atrium(1.0); sky = scenes.sky(512, 256); bn = assets.load_bluenoise()
    variants = [
        ('default (T=1 DP)', dict()),
        ('recull (12-bit distances)', dict(recull=1)),
        ('top-opt k=64 p=2', dict(top_opt=(64, 2))),
        ('top-opt k=256 p=2', dict(top_opt=(256, 2))),
        ('top-opt k=16 p=3', dict(top_opt=(16, 3))),
        ('top-opt k=1024 p=3', dict(top_opt=(1024, 3))),
        ('baseline T=2 area', dict(sah_top=2, collapse=1)),
        ('T=2 area tree-order', dict(sah_top=2, collapse=1, tree_order=1)),
        ('T=1 area tree-order', dict(sah_top=1, collapse=1, tree_order=1)),
        ('T=2 DP', dict(sah_top=2, collapse=2)),
        ('T=1 DP', dict(sah_top=1, collapse=2)),
        ('T=1 DP leaf3', dict(sah_top=1, collapse=2, leaf_size=3)),
        ('T=1 DP leaf4', dict(sah_top=1, collapse=2, leaf_size=4)),
    ]
    if len(sys.argv) > 1: variants = [v for v in variants if any(a in v[0] for a in sys.argv[1:])]
    for name, kw in variants:
        t0 = time.time(); osc = orc.Scene(mesh, sky, bn, **kw); tb = time.time() - t0
        r = measure(osc)
        # bench mix: per frame 1 primary launch + 3 bounce launches of ~equal ray counts -> ext mean over all four
        print(f"{name:24s} nodes {osc.n_nodes:6d} build {tb:5.1f}s | bounce ext {r['ext_nodes']:.2f}+{r['ext_tris']:.2f}  shadow {r['sh_nodes']:.2f}+{r['sh_tris']:.2f}  primary {r['prim_nodes']:.2f}+{r['prim_tris']:.2f}", flush=True)
