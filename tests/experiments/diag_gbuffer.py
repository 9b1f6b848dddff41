import sys, math, ctypes as C
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, orc
from raytracer3_amd import _lib as L, assets, scenes
from raytracer3_amd.renderer import Camera, PathTracer
mesh = scenes.atrium(0.3); osc = orc.Scene(mesh)
W, H = 160, 90
pt = PathTracer((W, H)); pt.set_scene(mesh)
cam = Camera((-10, 2, 0), (1, 0.1, 0), math.radians(65.0), W / H)
g = pt.make_gconst(cam, 1, 1, flags=0)
pt.render(g)
gb, depth = pt.gbuffer()
og = orc.GConst(); C.memmove(C.byref(og), C.byref(g), 304)
ogb, odepth = osc.gbuffer(og)
bad = np.argwhere(depth != odepth)
print('differing depth pixels', len(bad), 'of', W * H)
ys, xs = np.mgrid[0:H, 0:W]
rays = orc.primary_rays(og, xs.ravel(), ys.ravel())
t, u, v, p, _ = pt.ctx.trace_rays(rays)
t = np.where(p == L.MISS, 1e5, t).reshape(H, W)
print('gpu trace of oracle rays vs oracle depth', (t != odepth).sum(), ' vs gpu depth', (t != depth).sum())
for y, x in bad[:10]:
    print(x, y, depth[y, x], odepth[y, x], depth[y, x] - odepth[y, x])
print('gbuffer word mismatches where depth equal:', [(gb[..., k] != ogb[..., k])[(depth == odepth) & (depth != 1e5)].sum() for k in range(4)])
