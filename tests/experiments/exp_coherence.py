"""Experiment: the same ray set traced in different orders (pixel/tile order, shuffled, sorted by origin-Morton + octant)."""
import sys, math, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from raytracer3_amd import _lib as L, scenes
from raytracer3_amd.render_graph import Context
from raytracer3_amd.renderer import Camera

def morton3(q):
    def part(x):
        x = x.astype(np.uint64) & 0x3FF
        x = (x | (x << 16)) & 0x30000FF
        x = (x | (x << 8)) & 0x300F00F
        x = (x | (x << 4)) & 0x30C30C3
        x = (x | (x << 2)) & 0x9249249
        return x
    return (part(q[0]) << 2) | (part(q[1]) << 1) | part(q[2])

mesh = scenes.atrium(1.0)
for leaf, width in ((1, 2), (2, 4)):
    ctx = Context(0)
    ctx.set_option(L.OPT_LEAF_SIZE, leaf); ctx.set_option(L.OPT_NODE_WIDTH, width)
    ctx.upload_mesh(mesh); ctx.build_accel()
    W, H = 1920, 1080
    cam = Camera((-10, 2, 0), (1, 0.1, 0), math.radians(65.0), W / H)
    g = cam.gconst((W, H))
    # primary rays in tile / Z-curve order, computed on the host in float32 like the kernel (approximately)
    import orc, ctypes as C
    og = orc.GConst(); C.memmove(C.byref(og), C.byref(g), 304)
    tiles = orc.tile_pixels(W, H, 0, 1)[: 1 << 20]
    pinv = np.array(og.proj_inverse[:], np.float32).reshape(4, 4).T; vinv = np.array(og.view_inverse[:], np.float32).reshape(4, 4).T
    cx = (tiles[:, 0] + 0.5) / W * 2 - 1; cy = -((tiles[:, 1] + 0.5) / H * 2 - 1)
    tgt = (pinv @ np.stack([cx, cy, np.ones_like(cx), np.ones_like(cx)])).astype(np.float32)[:3]
    tgt /= np.linalg.norm(tgt, axis=0)
    d = (vinv[:3, :3] @ tgt).astype(np.float32)
    n = d.shape[1]
    o = np.repeat(vinv[:3, 3:4], n, axis=1).astype(np.float32)
    prim = np.concatenate([o, d, np.zeros((1, n), np.float32), np.full((1, n), 1e5, np.float32)]).astype(np.float32)
    t, u, v, p, ms = ctx.trace_rays(prim, repeat=5)
    print(f"layout leaf {leaf} width {width}: primary tile order   {n / ms / 1e3:8.1f} Mrays/s")
    perm = np.random.default_rng(0).permutation(n)
    ms2 = ctx.trace_rays(np.ascontiguousarray(prim[:, perm]), repeat=5)[4]
    print(f"                          primary shuffled      {n / ms2 / 1e3:8.1f} Mrays/s")
    # diffuse bounce rays from the primary hits
    hit = p != L.MISS
    hp = (o + d * t)[:, hit]
    rng = np.random.default_rng(1)
    dd = rng.normal(size=hp.shape).astype(np.float32); dd /= np.linalg.norm(dd, axis=0)
    m = hp.shape[1]
    b = np.concatenate([hp, dd, np.full((1, m), 1e-3, np.float32), np.full((1, m), 1e5, np.float32)]).astype(np.float32)
    ms3 = ctx.trace_rays(b, repeat=5)[4]
    print(f"                          bounce, pixel order   {m / ms3 / 1e3:8.1f} Mrays/s")
    ms4 = ctx.trace_rays(np.ascontiguousarray(b[:, rng.permutation(m)]), repeat=5)[4]
    print(f"                          bounce, shuffled      {m / ms4 / 1e3:8.1f} Mrays/s")
    lo, hi = hp.min(axis=1, keepdims=True), hp.max(axis=1, keepdims=True)
    q = np.clip((hp - lo) / (hi - lo) * 1023, 0, 1023).astype(np.uint32)
    octant = ((dd[0] > 0).astype(np.uint64) << 2) | ((dd[1] > 0).astype(np.uint64) << 1) | (dd[2] > 0).astype(np.uint64)
    for name, key in (("origin morton", morton3(q)), ("octant|morton", (octant << 30) | morton3(q)), ("morton|octant", (morton3(q) << 3) | octant), ("morton>>9|oct|morton", ((morton3(q) >> 9) << 12) | (octant << 9) | (morton3(q) & 511))):
        order = np.argsort(key, kind="stable")
        ms5 = ctx.trace_rays(np.ascontiguousarray(b[:, order]), repeat=5)[4]
        print(f"                          bounce, sorted {name:22s} {m / ms5 / 1e3:8.1f} Mrays/s")
    ctx.close()
