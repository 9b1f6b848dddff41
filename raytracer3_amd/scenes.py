"""Deterministic procedural stand-ins for the assets the reference tree does not ship
(/root/reference/.MISSING_LARGE_BLOBS: resources/sponza_scene.glb, resources/skybox2.exr) -- SURVEY.md section 8d.

* `atrium(detail)`  Sponza stand-in: two-storey colonnade around an open-roofed court, floor tiles, draped curtains,
                    emissive ceiling panels.  detail=1.0 -> 262 144 +- 2 % triangles, AABB 29.8 x 12.4 x 18.3 m.
* `cornell()`       closed Cornell box with an emissive ceiling panel (reference semantics need no sky).
* `cornell_ref()`   the box of the ONE image the reference tree holds (resources/refrence.png, a Blender-Cycles render): the eight
                    unit cubes and eight materials of the reference's processed asset `imported_assets/Default/box.glb`, placed
                    (the asset lost its node transforms and its emission) so that a render from CORNELL_REF_CAMERA lines up with
                    that image.  Informational: Cycles is not this estimator and the display transforms differ.
* `sky(w, h)`       equirect RGB32F gradient sky + 0.5 degree sun disc (peak radiance 5e4).
All surfaces carry normals facing the side they are meant to be seen from.
"""
from __future__ import annotations

import numpy as np

from .assets import Material, Mesh, MeshBuilder

ATRIUM_SEED = 0x5F0A2A
SKY_SEED = 7
ATRIUM_CAMERA = dict(position=(-10.0, 2.0, 0.0), direction=(1.0, 0.1, 0.0), fov_deg=65.0)
# slightly off-axis: an exactly symmetric view sends rays through the shared diagonals of the wall quads, where the fp32
# Moeller-Trumbore test is not watertight (known limitations)
CORNELL_CAMERA = dict(position=(0.0137, 1.0071, 3.4), direction=(0.0041, -0.0033, -1.0), fov_deg=40.0)


# Fitted to /root/reference/resources/refrence.png (1920x1080): least squares over the eight interior corners of the box as they
# appear in the image (box interior = the asset's unit cube [-1, 1]^3, open towards +z), no camera roll: 7 px RMS.
CORNELL_REF_CAMERA = dict(position=(-0.18653, 0.33087, 7.38676), direction=(0.018591, -0.058248, -0.998129), fov_deg=22.8837)
CORNELL_REF_EMISSION = 0.48  # GeometryInfo.emission of the light cube (x 12 in hit_info); chosen by tests/experiments/fit_cornell_ref.py


def _grid(origin, du, dv, nu, nv, normal=None, disp=None):
    """(nu x nv) quad grid spanning origin + s*du + t*dv, s,t in [0,1]; CCW seen from cross(du,dv)."""
    o, du, dv = (np.asarray(a, np.float64) for a in (origin, du, dv))
    s, t = np.meshgrid(np.linspace(0, 1, nu + 1), np.linspace(0, 1, nv + 1), indexing="xy")
    pos = o + s[..., None] * du + t[..., None] * dv
    n = np.cross(du, dv) if normal is None else np.asarray(normal, np.float64)
    n = n / np.linalg.norm(n)
    nrm = np.broadcast_to(n, pos.shape).copy()
    if disp is not None:
        pos, nrm = disp(pos, nrm, s, t)
    uv = np.stack([s, t], -1)
    i = np.arange(nu)[None, :] + (nu + 1) * np.arange(nv)[:, None]
    a, b, c, d = i, i + 1, i + nu + 2, i + nu + 1
    tris = np.stack([np.stack([a, b, c], -1), np.stack([a, c, d], -1)], -2).reshape(-1, 3)
    return pos.reshape(-1, 3), nrm.reshape(-1, 3), uv.reshape(-1, 2), tris


def _box(mb, name, lo, hi, mat, n=1, inward=False):
    lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
    e = hi - lo
    faces = [  # origin, du, dv with cross(du,dv) pointing outward
        (lo + [0, 0, e[2]], [e[0], 0, 0], [0, e[1], 0]),  # +z
        (lo + [e[0], 0, 0], [-e[0], 0, 0], [0, e[1], 0]),  # -z
        (lo + [e[0], 0, e[2]], [0, 0, -e[2]], [0, e[1], 0]),  # +x
        (lo, [0, 0, e[2]], [0, e[1], 0]),  # -x
        (lo + [0, e[1], e[2]], [e[0], 0, 0], [0, 0, -e[2]]),  # +y
        (lo, [e[0], 0, 0], [0, 0, e[2]]),  # -y
    ]
    P, N, U, T, off = [], [], [], [], 0
    for o, du, dv in faces:
        if inward:
            du, dv = dv, du
        p, nn, uv, t = _grid(o, du, dv, n, n)
        P.append(p); N.append(nn); U.append(uv); T.append(t + off); off += len(p)
    mb.add(name, np.concatenate(P), np.concatenate(N), np.concatenate(U), np.concatenate(T), mat)


def _cylinder(mb, name, base, radius, height, nseg, nring, mat, flute=0.0):
    th = np.linspace(0, 2 * np.pi, nseg + 1)
    y = np.linspace(0, 1, nring + 1)
    T, Y = np.meshgrid(th, y, indexing="xy")
    r = radius * (1.0 + flute * np.cos(12 * T)) * (1.0 - 0.12 * Y + 0.10 * np.exp(-((Y - 0.02) / 0.04) ** 2) + 0.10 * np.exp(-((Y - 0.98) / 0.04) ** 2))
    pos = np.stack([base[0] + r * np.cos(T), base[1] + Y * height, base[2] + r * np.sin(T)], -1)
    nrm = np.stack([np.cos(T), np.zeros_like(T), np.sin(T)], -1)
    uv = np.stack([T / (2 * np.pi), Y], -1)
    i = np.arange(nseg)[None, :] + (nseg + 1) * np.arange(nring)[:, None]
    a, b, c, d = i, i + 1, i + nseg + 2, i + nseg + 1
    tris = np.stack([np.stack([a, c, b], -1), np.stack([a, d, c], -1)], -2).reshape(-1, 3)  # outward
    mb.add(name, pos.reshape(-1, 3), nrm.reshape(-1, 3), uv.reshape(-1, 2), tris, mat)


def _arch(mb, name, p0, p1, rise, tube, nseg, nring, mat):
    """Half-ellipse tube from p0 to p1 (same height), bulging up by `rise`."""
    p0, p1 = np.asarray(p0, np.float64), np.asarray(p1, np.float64)
    mid, half = 0.5 * (p0 + p1), 0.5 * (p1 - p0)
    L = np.linalg.norm(half)
    ax = half / L
    a = np.linspace(np.pi, 0, nseg + 1)
    centre = mid + np.cos(a)[:, None] * half + np.sin(a)[:, None] * np.array([0, rise, 0])
    tang = -np.sin(a)[:, None] * half + np.cos(a)[:, None] * np.array([0, rise, 0])
    tang /= np.linalg.norm(tang, axis=1, keepdims=True)
    side = np.cross(ax, [0, 1, 0]); side /= np.linalg.norm(side)
    up = np.cross(side[None, :], tang)
    ph = np.linspace(0, 2 * np.pi, nring + 1)
    nrm = np.cos(ph)[None, :, None] * up[:, None, :] + np.sin(ph)[None, :, None] * side[None, None, :]
    pos = centre[:, None, :] + tube * nrm
    uv = np.stack(np.meshgrid(ph / (2 * np.pi), np.linspace(0, 1, nseg + 1), indexing="xy"), -1)
    i = np.arange(nring)[None, :] + (nring + 1) * np.arange(nseg)[:, None]
    a_, b_, c_, d_ = i, i + 1, i + nring + 2, i + nring + 1
    tris = np.stack([np.stack([a_, b_, c_], -1), np.stack([a_, c_, d_], -1)], -2).reshape(-1, 3)
    # orientation check: make the geometric normal agree with the shading normal
    P, N = pos.reshape(-1, 3), nrm.reshape(-1, 3)
    fn = np.cross(P[tris[:, 1]] - P[tris[:, 0]], P[tris[:, 2]] - P[tris[:, 0]])
    if np.sum(np.einsum("ij,ij->i", fn, N[tris[:, 0]])) < 0:
        tris = tris[:, ::-1]
    mb.add(name, P, N, uv.reshape(-1, 2), tris, mat)


def _sphere(mb, name, centre, radius, nseg, nring, mat):
    th = np.linspace(0, 2 * np.pi, nseg + 1)
    ph = np.linspace(0, np.pi, nring + 1)
    T, Pp = np.meshgrid(th, ph, indexing="xy")
    nrm = np.stack([np.sin(Pp) * np.cos(T), np.cos(Pp), np.sin(Pp) * np.sin(T)], -1)
    pos = np.asarray(centre, np.float64) + radius * nrm
    uv = np.stack([T / (2 * np.pi), Pp / np.pi], -1)
    i = np.arange(nseg)[None, :] + (nseg + 1) * np.arange(nring)[:, None]
    a, b, c, d = i, i + 1, i + nseg + 2, i + nseg + 1
    tris = np.stack([np.stack([a, b, c], -1), np.stack([a, c, d], -1)], -2).reshape(-1, 3)
    P, N = pos.reshape(-1, 3), nrm.reshape(-1, 3)
    fn = np.cross(P[tris[:, 1]] - P[tris[:, 0]], P[tris[:, 2]] - P[tris[:, 0]])
    area = np.linalg.norm(fn, axis=1)
    tris = tris[area > 1e-12]  # drop the degenerate pole triangles
    fn = fn[area > 1e-12]
    if np.sum(np.einsum("ij,ij->i", fn, N[tris[:, 0]])) < 0:
        tris = tris[:, ::-1]
    mb.add(name, P, N, uv.reshape(-1, 2), tris, mat)


def atrium(detail: float = 1.0, seed: int = ATRIUM_SEED) -> Mesh:
    """Sponza stand-in.  `detail` scales the tessellation (1.0 = full ~262k triangles; 0.1 ~ a few thousand)."""
    rng = np.random.default_rng(seed)
    alb = rng.uniform(0.2, 0.8, size=(24, 3))
    mats = []
    for k in range(24):
        rough = 0.5 if k % 5 == 0 else 1.0
        metal = 1.0 if k in (7, 19) else 0.0
        mats.append(Material(tuple(float(x) for x in alb[k]), metal, rough))
    emissive = Material((0.8, 0.8, 0.8), 0.0, 1.0, emission=(1.0, 0.95, 0.85))
    mb = MeshBuilder()
    X, Y, Z = 14.9, 12.4, 9.15  # half extents in x,z ; full height
    q = lambda n: max(1, int(round(n * detail)))  # noqa: E731

    # floor: 60 x 36 tiles in two alternating materials (normals up)
    nx, nz = q(60), q(36)
    for par in (0, 1):
        P, N, U, T, off = [], [], [], [], 0
        for ix in range(nx):
            for iz in range(nz):
                if (ix + iz) % 2 != par:
                    continue
                o = [-X + 2 * X * ix / nx, 0.0, -Z + 2 * Z * iz / nz]
                p, n, uv, t = _grid(o, [0, 0, 2 * Z / nz], [2 * X / nx, 0, 0], 1, 1)
                P.append(p); N.append(n); U.append(uv); T.append(t + off); off += 4
        if P:
            mb.add(f"floor{par}", np.concatenate(P), np.concatenate(N), np.concatenate(U), np.concatenate(T), mats[par])
    # outer walls (normals inward)
    wq = (q(40), q(16))
    p = _grid([-X, 0, -Z], [2 * X, 0, 0], [0, Y, 0], *wq); mb.add("wall_z-", *p, mats[2])
    p = _grid([X, 0, Z], [-2 * X, 0, 0], [0, Y, 0], *wq); mb.add("wall_z+", *p, mats[3])
    p = _grid([-X, 0, Z], [0, 0, -2 * Z], [0, Y, 0], *wq); mb.add("wall_x-", *p, mats[4])
    p = _grid([X, 0, -Z], [0, 0, 2 * Z], [0, Y, 0], *wq); mb.add("wall_x+", *p, mats[5])
    # ceiling ring around an open skylight x in [-8,8], z in [-4,4] (normals down)
    hx, hz = 8.0, 4.0
    cq = (q(24), q(8))
    p = _grid([-X, Y, -Z], [0, 0, Z - hz], [2 * X, 0, 0], *cq[::-1]); mb.add("ceil_z-", *p, mats[6])
    p = _grid([-X, Y, hz], [0, 0, Z - hz], [2 * X, 0, 0], *cq[::-1]); mb.add("ceil_z+", *p, mats[6])
    p = _grid([-X, Y, -hz], [0, 0, 2 * hz], [X - hx, 0, 0], *cq[::-1]); mb.add("ceil_x-", *p, mats[6])
    p = _grid([hx, Y, -hz], [0, 0, 2 * hz], [X - hx, 0, 0], *cq[::-1]); mb.add("ceil_x+", *p, mats[6])
    # gallery (upper floor ring at y = 6 around the court x in [-9.6,9.6], z in [-5.6,5.6]): slabs with thickness
    gx, gz, gy = 9.6, 5.6, 6.0
    _box(mb, "gallery_z-", [-X, gy - 0.3, -Z], [X, gy, -gz], mats[8], n=q(6))
    _box(mb, "gallery_z+", [-X, gy - 0.3, gz], [X, gy, Z], mats[8], n=q(6))
    _box(mb, "gallery_x-", [-X, gy - 0.3, -gz], [-gx, gy, gz], mats[9], n=q(6))
    _box(mb, "gallery_x+", [gx, gy - 0.3, -gz], [X, gy, gz], mats[9], n=q(6))
    # colonnade: columns on the court rectangle, both storeys, arches between neighbours
    cx, cz = 9.0, 5.0
    xs = np.linspace(-cx, cx, 7)
    zs = np.linspace(-cz, cz, 4)
    ring = [(x, -cz) for x in xs] + [(cx, z) for z in zs[1:]] + [(x, cz) for x in xs[::-1][1:]] + [(-cx, z) for z in zs[::-1][1:-1]]
    seg, rings = q(32), q(20)
    for storey, (y0, hgt) in enumerate(((0.0, 5.0), (gy, 5.2))):
        for k, (x, z) in enumerate(ring):
            _cylinder(mb, f"col{storey}_{k}", (x, y0, z), 0.33 if storey == 0 else 0.26, hgt, seg, rings, mats[10 + (k % 3)], flute=0.03)
        for k in range(len(ring)):
            (xa, za), (xb, zb) = ring[k], ring[(k + 1) % len(ring)]
            _arch(mb, f"arch{storey}_{k}", (xa, y0 + hgt, za), (xb, y0 + hgt, zb), 0.7, 0.16, q(28), q(12), mats[13 + storey])
    # curtains: 16 sin-displaced sheets hanging between upper columns (double sided look via two-sided shading)
    ncur = 16
    # remaining triangle budget goes to the curtains so that detail=1 lands on 262 144 +- 2 %
    so_far = sum(mb.c)
    extras = 8 * 2 * q(48) * q(24) + 4 * 8 + 12 * 6 * 2  # spheres + panels + plinths (estimate)
    target = int(262144 * detail * detail) if detail < 1.0 else 262144
    per = max(8, (target - so_far - extras) // ncur)
    cn = max(2, int(np.sqrt(per / 2.0)))
    cu, cv = cn, max(2, per // (2 * cn))
    for k in range(ncur):
        (xa, za), (xb, zb) = ring[(2 * k) % len(ring)], ring[(2 * k + 1) % len(ring)]
        ph = float(rng.uniform(0, 2 * np.pi))
        amp = float(rng.uniform(0.10, 0.22))

        def disp(pos, nrm, s, t, ph=ph, amp=amp):
            n0 = nrm[0, 0]
            w = amp * np.sin(6 * np.pi * s + ph) * (0.35 + 0.65 * t) + 0.05 * np.sin(2 * np.pi * t * 3 + ph)
            dws = amp * 6 * np.pi * np.cos(6 * np.pi * s + ph) * (0.35 + 0.65 * t)
            pos = pos + w[..., None] * n0
            du = pos[0, -1] - pos[0, 0]
            du = du / np.linalg.norm(du)
            length = np.linalg.norm(pos[0, -1] - pos[0, 0])
            nn = n0[None, None, :] - (dws / length)[..., None] * du[None, None, :]
            return pos, nn

        o = np.array([xa, gy + 4.6, za]); du = np.array([xb - xa, 0, zb - za]); dv = np.array([0, -3.6, 0])
        # face the court centre
        n = np.cross(du, dv)
        if np.dot(n, -o * [1, 0, 1]) < 0:
            o, du = o + du, -du
        p = _grid(o + np.cross(du, dv) / np.linalg.norm(np.cross(du, dv)) * 0.45, du, dv, cu, cv, disp=disp)
        mb.add(f"curtain{k}", *p, mats[15 + (k % 6)])
    # decorative spheres on plinths in the court
    for k in range(8):
        x = -7.0 + 2.0 * k; z = 2.5 if k % 2 else -2.5
        _box(mb, f"plinth{k}", [x - 0.35, 0.0, z - 0.35], [x + 0.35, 0.8, z + 0.35], mats[21], n=1)
        _sphere(mb, f"sphere{k}", (x, 1.25, z), 0.45, q(48), q(24), mats[7] if k % 4 == 0 else (mats[19] if k % 4 == 2 else mats[22]))
    # 4 emissive ceiling panels under the ceiling ring (emission 1.0 -> x12 in hit_info)
    for k, (x, z) in enumerate(((-11.5, -6.5), (11.5, -6.5), (-11.5, 6.5), (11.5, 6.5))):
        p = _grid([x - 1.2, Y - 0.05, z - 0.8], [0, 0, 1.6], [2.4, 0, 0], 2, 2)
        mb.add(f"panel{k}", *p, emissive)
    return mb.build()


def cornell() -> Mesh:
    """Closed Cornell box, 2 x 2 x 2 m centred at (0,1,0): red/green side walls, two blocks, emissive ceiling panel."""
    white = Material((0.73, 0.73, 0.73)); red = Material((0.65, 0.05, 0.05)); green = Material((0.12, 0.45, 0.15))
    light = Material((0.78, 0.78, 0.78), emission=(1.4, 1.2, 0.9))
    mb = MeshBuilder()
    n = 4
    mb.add("floor", *_grid([-1, 0, -1], [0, 0, 2], [2, 0, 0], n, n), white)
    mb.add("ceiling", *_grid([-1, 2, -1], [2, 0, 0], [0, 0, 2], n, n), white)
    mb.add("back", *_grid([-1, 0, -1], [2, 0, 0], [0, 2, 0], n, n), white)
    mb.add("front", *_grid([1, 0, 4], [-2, 0, 0], [0, 2, 0], n, n), white)  # behind the camera: closes the box
    mb.add("floor2", *_grid([-1, 0, 1], [0, 0, 3], [2, 0, 0], n, n), white)
    mb.add("ceiling2", *_grid([-1, 2, 1], [2, 0, 0], [0, 0, 3], n, n), white)
    mb.add("left", *_grid([-1, 0, 4], [0, 0, -5], [0, 2, 0], n, n), red)
    mb.add("right", *_grid([1, 0, -1], [0, 0, 5], [0, 2, 0], n, n), green)
    mb.add("panel", *_grid([-0.35, 1.995, -0.35], [0.7, 0, 0], [0, 0, 0.7], 2, 2), light)
    _box(mb, "tall", [-0.65, 0.0, -0.65], [-0.05, 1.2, -0.05], white, n=2)
    _box(mb, "short", [0.1, 0.0, 0.0], [0.7, 0.6, 0.6], white, n=2)
    return mb.build()


def cornell_ref(asset_path=None, emission: float = CORNELL_REF_EMISSION) -> Mesh:
    """The Cornell box of resources/refrence.png rebuilt from the reference's own processed asset.

    `imported_assets/Default/box.glb` (bincode, older `Mesh` layout; tests/golden/processed_box.glb.bin is a byte-identical copy
    so that this also works where /root/reference is absent) holds 8 unit cubes (24 vertices each: position, face normal, uv)
    and 8 materials -- 0.8 grey x3, blue, red, green, 0.5 grey, 0.8 grey with roughness 0.5 -- but neither node transforms nor
    emission.  Cube k takes material k; which cube is which wall is decided by its colour, the placements below
    are read off the image: open front, black outside, a flat emissive box just under the ceiling (its side faces are what
    lights the ceiling in the image: the render is direct light only -- the unlit front of the small box is pure black)."""
    from pathlib import Path

    from .assets import read_processed_mesh

    path = Path(asset_path) if asset_path else Path(__file__).resolve().parent.parent / "tests" / "golden" / "processed_box.glb.bin"
    pm = read_processed_mesh(path, "old")
    if len(pm.materials) != 8 or len(pm.vertices) != 8 * 24:
        raise ValueError("expected the reference's box asset: 8 cubes x 24 vertices, 8 materials")

    def role(m):
        r, g, b = m.color
        if b > 0.5 and r < 0.1:
            return "right"
        if r > 0.5 and g < 0.1:
            return "left"
        if g > 0.5 and r < 0.1:
            return "floor"
        if abs(r - 0.5) < 0.01:
            return "short"
        return "light" if m.roughness_factor < 0.9 else "grey"

    t = 0.05  # wall thickness (never seen: the outside is black)
    place = {  # role -> (half size, centre, rotation about +y in degrees)
        "floor": ((1 + 2 * t, t, 1.0), (0.0, -1 - t, 0.0), 0.0),
        "ceiling": ((1 + 2 * t, t, 1.0), (0.0, 1 + t, 0.0), 0.0),
        "back": ((1 + 2 * t, 1 + 2 * t, t), (0.0, 0.0, -1 - t), 0.0),
        "left": ((t, 1.0, 1.0), (-1 - t, 0.0, 0.0), 0.0),
        "right": ((t, 1.0, 1.0), (1 + t, 0.0, 0.0), 0.0),
        "tall": ((0.247, 0.455, 0.25), (-0.535, -0.545, -0.37), -10.0),
        "short": ((0.215, 0.19, 0.22), (0.28, -0.81, 0.27), 0.0),
        "light": ((0.455, 0.02, 0.625), (-0.015, 0.975, -0.025), 0.0),
    }
    greys = iter(("back", "ceiling", "tall"))
    mb = MeshBuilder()
    for k, m in enumerate(pm.materials):
        r = role(m)
        name = next(greys) if r == "grey" else r
        half, centre, rot = place[name]
        v = pm.vertices[24 * k : 24 * k + 24].astype(np.float64)
        c, sn = np.cos(np.radians(rot)), np.sin(np.radians(rot))
        R = np.array([[c, 0.0, sn], [0.0, 1.0, 0.0], [-sn, 0.0, c]])
        pos = (v[:, :3] * np.asarray(half)) @ R.T + np.asarray(centre)
        nrm = v[:, 3:6] @ R.T  # axis-aligned face normals: unchanged by the axis-aligned scale
        tris = []
        for f in np.unique(np.round(v[:, 3:6]), axis=0):  # the four vertices of each face, fanned counter-clockwise about the outward normal
            idx = np.nonzero((np.round(v[:, 3:6]) == f).all(1))[0]
            assert len(idx) == 4
            ctr = v[idx, :3].mean(0)
            a = np.cross(f, [1.0, 0.0, 0.0]) if abs(f[0]) < 0.5 else np.cross(f, [0.0, 1.0, 0.0])
            bq = np.cross(f, a)
            ang = np.arctan2((v[idx, :3] - ctr) @ bq, (v[idx, :3] - ctr) @ a)
            q = idx[np.argsort(ang)]
            tris += [[q[0], q[1], q[2]], [q[0], q[2], q[3]]]
        mat = Material(m.color, m.metalic_factor, m.roughness_factor, (emission,) * 3 if name == "light" else (0.0, 0.0, 0.0))
        mb.add(name, pos, nrm, v[:, 6:8], np.asarray(tris), mat)
    return mb.build()


def textured_cornell() -> Mesh:
    """The Cornell box with base-colour textures (hit_logic.slang:31-33): checker floor, gradient back wall."""
    mesh = cornell()
    yy, xx = np.mgrid[0:64, 0:64]
    checker = np.where(((xx // 8) + (yy // 8)) % 2 == 0, 230, 40).astype(np.uint8)
    t0 = np.stack([checker, checker, (checker // 2 + 60).astype(np.uint8), np.full_like(checker, 255)], -1)
    yy, xx = np.mgrid[0:32, 0:48]
    t1 = np.stack([(xx * 5).astype(np.uint8), (yy * 7).astype(np.uint8), np.full(xx.shape, 128, np.uint8), np.full(xx.shape, 255, np.uint8)], -1)
    mesh.textures = [np.ascontiguousarray(t0), np.ascontiguousarray(t1)]
    for name, tex in (("floor", 0), ("floor2", 0), ("back", 1)):
        g = mesh.names.index(name)
        mesh.geometries["base_color_texture_index"][g] = tex
        mesh.geometries["base_color"][g] = (1.0, 1.0, 1.0, 1.0)
        vo = int(mesh.geometries["vertex_offset"][g])
        later = [int(x) for x in mesh.geometries["vertex_offset"] if int(x) > vo]
        mesh.vertices[vo : (min(later) if later else len(mesh.vertices)), 6:8] *= 3.0  # repeat addressing: uv up to 3
    return mesh


def sky(width: int = 2048, height: int = 1024, seed: int = SKY_SEED) -> np.ndarray:
    """Equirect RGB32F sky: horizon-to-zenith gradient, ground bounce, soft cloud noise, 0.5 degree sun (5e4)."""
    rng = np.random.default_rng(seed)
    v = (np.arange(height) + 0.5) / height
    u = (np.arange(width) + 0.5) / width
    U, V = np.meshgrid(u, v, indexing="xy")
    theta = np.pi * V
    phi = 2 * np.pi * (U - 0.5)
    d = np.stack([np.cos(phi) * np.sin(theta), np.cos(theta), np.sin(phi) * np.sin(theta)], -1)
    up = np.clip(d[..., 1], 0, 1)
    zen, hor, gnd = np.array([0.18, 0.36, 0.9]), np.array([0.9, 0.95, 1.0]), np.array([0.18, 0.16, 0.14])
    t = up[..., None] ** 0.45
    col = hor * (1 - t) + zen * t
    col = np.where(d[..., 1:2] < 0, gnd * (1.0 + 0.5 * d[..., 1:2]), col)
    # low-frequency cloud modulation from a few random sinusoids (deterministic)
    cl = np.zeros_like(U)
    for _ in range(6):
        fx, fy = rng.integers(1, 6), rng.integers(1, 5)
        cl += rng.uniform(0.05, 0.15) * np.sin(2 * np.pi * (fx * U + rng.uniform()) ) * np.sin(np.pi * fy * V + rng.uniform(0, 6.28))
    col = col * (1.0 + np.where(d[..., 1] > 0, cl, 0)[..., None])
    sun_dir = np.array([0.30, 0.88, 0.25]); sun_dir /= np.linalg.norm(sun_dir)
    cosang = d @ sun_dir
    ang = np.degrees(np.arccos(np.clip(cosang, -1, 1)))
    sun = np.where(ang < 0.25, 5.0e4, 0.0) + 40.0 * np.exp(-(ang / 2.5) ** 2)
    col = col + sun[..., None] * np.array([1.0, 0.93, 0.82])
    return np.ascontiguousarray(np.maximum(col, 0.0), np.float32)
