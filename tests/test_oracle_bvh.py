"""The oracle's LBVH + fp32 Moeller-Trumbore pinned against brute force (fp32: exact; fp64: up to grazing cases)."""
import numpy as np

import orc
from raytracer3_amd import scenes


def random_rays(n, seed, lo, hi):
    rng = np.random.default_rng(seed)
    o = rng.uniform(lo, hi, (n, 3)).T
    d = rng.normal(size=(3, n))
    d /= np.linalg.norm(d, axis=0)
    return np.concatenate([o, d, np.full((1, n), 0.001), np.full((1, n), 1e5)]).astype(np.float32)


def test_bvh_equals_bruteforce_fp32_and_tracks_fp64():
    mesh = scenes.atrium(0.2)
    sc = orc.Scene(mesh)
    assert sc.n_nodes < sc.n_tris and sc.max_depth < 22
    base = orc.Scene(mesh, leaf_size=1, node_width=2, quantized=0, sah_top=0)  # plain Karras tree: n - 1 binary nodes
    assert base.n_nodes == base.n_tris - 1
    codes = sc.codes()
    assert (np.diff(codes.astype(np.float64)) >= 0).all()  # Morton order
    rays = random_rays(6000, 3, [-14, 0.2, -8], [14, 12, 8])
    t, u, v, p, nn, nt = sc.trace_closest(rays, counts=True)
    bt, bu, bv, bp = sc.trace_brute(rays, 0)
    assert np.array_equal(p, bp) and np.array_equal(t, bt) and np.array_equal(u, bu) and np.array_equal(v, bv)
    dt, du, dv, dp = sc.trace_brute(rays, 1)
    # fp64 may pick another primitive where surfaces coincide (coplanar floor / plinth / slab faces: ties in t), so compare
    # the hit distance; hit-vs-miss may flip only for grazing rays at silhouettes
    both = (p != orc.MISS) & (dp != orc.MISS)
    assert ((p != orc.MISS) == (dp != orc.MISS)).mean() > 0.999
    assert np.quantile(np.abs(t[both] - dt[both]), 0.999) < 1e-3
    assert (dp == p).mean() > 0.95
    occ = sc.trace_any(rays)
    assert np.array_equal(occ != 0, p != orc.MISS)
    # the layout (leaf size, node width) changes the walk, never the answer
    for leaf, width, quant in ((1, 2, 0), (4, 2, 0), (1, 4, 0), (8, 4, 0), (2, 4, 0), (4, 4, 1), (2, 4, 2), (1, 4, 2), (8, 4, 2)):
        alt = orc.Scene(mesh, leaf_size=leaf, node_width=width, quantized=quant)
        at, au, av, ap = alt.trace_closest(rays)
        assert np.array_equal(ap, p) and np.array_equal(at, t) and np.array_equal(au, u) and np.array_equal(av, v)
        assert np.array_equal(alt.trace_any(rays) != 0, occ != 0)
    # the surface-area collapse only regroups the same binary tree into better four-wide nodes: same hits, fewer visits
    lb = orc.Scene(mesh, sah_top=0)
    lt, lu, lv, lp, lnn, lnt = lb.trace_closest(rays, counts=True)
    assert np.array_equal(lp, p) and np.array_equal(lt, t) and nn.mean() < lnn.mean()  # SAH top: same hits, fewer visits than the plain LBVH
    print("nodes/ray: plain LBVH %.2f, SAH top %.2f" % (lnn.mean(), nn.mean()))
    par = orc.Scene(mesh, collapse=0)
    pt_, pu_, pv_, pp_, pnn, pnt = par.trace_closest(rays, counts=True)
    assert np.array_equal(pp_, p) and np.array_equal(pt_, t) and nn.mean() < pnn.mean()
    print("nodes/ray: parity collapse %.2f, surface-area collapse %.2f; nodes %d -> %d" % (pnn.mean(), nn.mean(), par.n_nodes, sc.n_nodes))
    assert 5 < nn.mean() < 200 and nt.mean() < 20
    hit = p != orc.MISS
    assert (u[hit] >= 0).all() and (v[hit] >= 0).all() and (u[hit] + v[hit] <= 1).all() and (t[hit] > 0.001).all()


def test_tmin_tmax_interval_and_ties():
    from raytracer3_amd import assets

    mb = assets.MeshBuilder()
    tri = np.array([[-1, -1, 0], [1, -1, 0], [0, 1, 0]], np.float32)
    for k in range(3):  # three coincident copies at z = 0 and one at z = 1
        mb.add(f"c{k}", tri, np.tile([0, 0, 1], (3, 1)), None, [[0, 1, 2]], assets.Material())
    mb.add("far", tri + [0, 0, 1], np.tile([0, 0, 1], (3, 1)), None, [[0, 1, 2]], assets.Material())
    sc = orc.Scene(mb.build())
    def ray(oz, dz, tmin, tmax):
        return np.array([[0, 0, oz, 0, 0, dz, tmin, tmax]], np.float32).T.copy()
    t, u, v, p = sc.trace_closest(ray(-1, 1, 0, 1e5))
    assert p[0] == 0 and t[0] == 1.0  # tie at t = 1 -> lowest primitive id
    t, u, v, p = sc.trace_closest(ray(-1, 1, 1.0, 1e5))
    assert p[0] == 3 and t[0] == 2.0  # t must be > tmin: the coincident sheet at t == tmin is skipped
    t, u, v, p = sc.trace_closest(ray(-1, 1, 0, 0.5))
    assert p[0] == orc.MISS  # nothing inside (tmin, tmax)
    t, u, v, p = sc.trace_closest(ray(2, -1, 0, 1e5))
    assert p[0] == 3 and t[0] == 1.0  # back faces are hit too (no culling)


def edge_rays(mesh, origin):
    """rays from `origin` aimed exactly at every edge midpoint and vertex of the mesh"""
    tri = mesh.triangle_positions()
    targets = np.concatenate([(tri[:, 0] + tri[:, 1]) / 2, (tri[:, 1] + tri[:, 2]) / 2, (tri[:, 0] + tri[:, 2]) / 2, tri.reshape(-1, 3)])
    org = np.asarray(origin, np.float32)
    d = targets - org
    ln = np.linalg.norm(d, axis=1)
    d = (d[ln > 1e-3] / ln[ln > 1e-3, None]).astype(np.float32)
    m = len(d)
    return np.concatenate([np.tile(org[:, None], (1, m)), d.T, np.full((1, m), 0.001), np.full((1, m), 1e5)]).astype(np.float32)


def test_shared_edge_sweep_is_watertight():
    """The driver traversal of the reference is watertight by specification (raytracing.rs:88-148).  The edge-function triangle test
    evaluates a shared edge to the same number for both of its triangles, so no ray can pass between them: every edge two triangles of
    the atrium share, and every vertex surrounded by shared edges, is hit from both sides of the surface by a ray aimed exactly at it."""
    mesh = scenes.atrium(0.3)
    sc = orc.Scene(mesh)
    rays, h = orc.shared_edge_rays(mesh, limit=60000)
    assert rays.shape[1] > 100000
    t, u, v, p = sc.trace_closest(rays, threads=8)
    odd = np.nonzero((p == orc.MISS) | (t > h + 1e-3))[0]
    # a handful of targets sit on folds of the arch tubes where even an exact intersector finds nothing at distance h: a ray only
    # counts as leaked if the double-precision brute force over all triangles DOES hit there
    assert len(odd) < 1e-3 * rays.shape[1]
    if len(odd):
        td, _, _, pd = sc.trace_brute(rays[:, odd], mode=1, threads=8)
        assert ((pd == orc.MISS) | (td > h + 1e-3)).all()
    assert sc.trace_any(rays, threads=8).all()
    # and the fp32 brute force over all triangles agrees with the tree on a sample (the test itself, not the traversal, is watertight)
    tb, ub, vb, pb = sc.trace_brute(rays[:, ::97], mode=0, threads=8)
    assert np.array_equal(pb, p[::97]) and np.array_equal(tb, t[::97])


def test_closed_box_does_not_leak_through_shared_edges():
    """Rays aimed exactly at every edge midpoint and vertex of the Cornell box, T-junctions between separately tessellated walls
    included (no watertight test covers those: 42 of these 2160 rays slip through the exact sign test, 146 slipped through plain fp32
    Moeller-Trumbore): with barycentrics accepted down to -2^-20 none does."""
    mesh = scenes.cornell()
    sc = orc.Scene(mesh)
    rays = edge_rays(mesh, [0.0137, 1.0071, 0.3])
    t, u, v, p = sc.trace_closest(rays)
    assert len(p) > 2000 and (p != orc.MISS).all()
    assert sc.trace_any(rays).all()


def test_non_finite_rays_miss_without_walking_the_tree():
    """[rule] NaN / Inf in a ray's origin or direction (a NaN camera, a zero-length normal upstream) = miss, zero visits: with NaNs
    every min/max slab test passes and such a ray would otherwise visit every node and triangle."""
    import orc
    from raytracer3_amd import scenes
    s = orc.Scene(scenes.cornell())
    rays = np.tile(np.array([[0.0], [1.0], [3.0], [0.0], [0.0], [-1.0], [0.0], [1e5]], np.float32), (1, 8))
    bad = [np.nan, np.inf, -np.inf]
    for k in range(6):
        rays[k, k + 1] = bad[k % 3]
    t, u, v, p, nn, nt = s.trace_closest(rays, counts=True)
    assert p[0] != orc.MISS and nn[0] > 0  # the untouched ray hits the back wall
    assert (p[1:7] == orc.MISS).all() and (nn[1:7] == 0).all() and (nt[1:7] == 0).all() and p[7] == p[0]
    occ, on, ot = s.trace_any(rays, counts=True)
    assert occ[0] == 1 and (occ[1:7] == 0).all() and (on[1:7] == 0).all()


def test_edge_functions_are_exactly_antisymmetric():
    """What the watertightness of the triangle test rests on: the edge function of (P1, P2) seen from a triangle (P0, P1, P2) and the
    one of the same edge seen from its neighbour (P3, P2, P1) are the same number with opposite sign, bit for bit, for any ray --
    and with equal sign when the neighbour is wound the other way round (P3, P1, P2)."""
    import ctypes as C

    L = orc.lib()
    L.orc_tri_edge_functions.argtypes = [C.c_void_p] * 6
    L.orc_tri_edge_functions.restype = None
    rng = np.random.default_rng(5)
    out1, out2, out3 = (np.zeros(3, np.float32) for _ in range(3))
    for _ in range(2000):
        scale = np.float32(10.0 ** rng.uniform(-3, 3))
        P = (rng.normal(size=(4, 3)) * scale + rng.normal(size=3) * scale * 10).astype(np.float32)
        o = (rng.normal(size=3) * scale * 10).astype(np.float32)
        d = rng.normal(size=3).astype(np.float32)
        L.orc_tri_edge_functions(orc.ptr(P[0]), orc.ptr(P[1]), orc.ptr(P[2]), orc.ptr(o), orc.ptr(d), orc.ptr(out1))  # U = edge (P1, P2)
        L.orc_tri_edge_functions(orc.ptr(P[3]), orc.ptr(P[2]), orc.ptr(P[1]), orc.ptr(o), orc.ptr(d), orc.ptr(out2))  # U = edge (P2, P1)
        L.orc_tri_edge_functions(orc.ptr(P[3]), orc.ptr(P[1]), orc.ptr(P[2]), orc.ptr(o), orc.ptr(d), orc.ptr(out3))  # U = edge (P1, P2) again
        assert out1[0] == -out2[0] and out1[0] == out3[0]
        assert (out1.view(np.uint32)[0] ^ out2.view(np.uint32)[0]) in (0x80000000, 0) or out1[0] == 0.0


def test_instances_flatten_like_a_host_baked_mesh():
    """rt3_scene_set_instances / orc_scene_set_instances (world/mod.rs:46-60): a mesh placed twice under two matrices is traversed exactly
    like the mesh a host would get by transforming the positions itself with the same fp32 expression -- same triangle records, same
    nodes, same hits -- and hit_info turns the normal by the matrix (hit_logic.slang:23)."""
    import math

    from raytracer3_amd import assets

    base = scenes.cornell()
    k = list(base.names).index("tall")

    def trs(t, ry, s):
        c, sn = math.cos(ry), math.sin(ry)
        return np.array([[c * s[0], 0, sn * s[2], t[0]], [0, s[1], 0, t[1]], [-sn * s[0], 0, c * s[2], t[2]], [0, 0, 0, 1]], np.float32)

    mats = [np.eye(4, dtype=np.float32), trs((0.5, 0.1, 1.5), 0.7, (0.5, 1.2, 0.8)), trs((-0.3, 0.0, 2.2), -1.1, (1.0, 0.5, 1.0))]
    n_g = len(base.geometries)
    inst = [(0, n_g, mats[0]), (k, 1, mats[1]), (k, 1, mats[2])]
    osc = orc.Scene(base, instances=inst)
    # the same world baked on the host: geometry k appended twice with transformed positions (normals left alone: compared separately)
    g = base.geometries[k]
    first = int(np.sum(base.prim_counts[:k]))
    idx = base.indices[g["index_offset"]: g["index_offset"] + 3 * int(base.prim_counts[k])] + g["vertex_offset"]
    verts = [base.vertices]
    geoms = [base.geometries.copy()]
    indices = [base.indices]
    pcs = list(base.prim_counts)
    for m in mats[1:]:
        used = np.unique(idx)
        v = base.vertices[used].copy()
        x, y, z = v[:, 0].copy(), v[:, 1].copy(), v[:, 2].copy()
        for r in range(3):  # glam transform_point3: w_axis + (z_axis z + (y_axis y + x_axis x)), fp32
            v[:, r] = m[r, 3] + (m[r, 2] * z + (m[r, 1] * y + m[r, 0] * x))
        remap = {int(u): j for j, u in enumerate(used)}
        gi = base.geometries[k:k + 1].copy()
        gi["vertex_offset"] = sum(len(a) for a in verts)
        gi["index_offset"] = sum(len(a) for a in indices)
        verts.append(v)
        indices.append(np.array([remap[int(u)] for u in idx], np.uint32))
        geoms.append(gi)
        pcs.append(base.prim_counts[k])
    baked = assets.Mesh(np.concatenate(verts), np.concatenate(indices), np.concatenate(geoms), np.array(pcs, np.uint32), list(base.names) + ["t1", "t2"])
    ob = orc.Scene(baked)
    assert osc.n_tris == ob.n_tris == base.n_triangles + 2 * int(base.prim_counts[k])
    assert np.array_equal(osc.tris(), ob.tris()) and np.array_equal(osc.nodes(), ob.nodes())
    rays = random_rays(20000, 4, [-0.9, 0.1, -0.9], [0.9, 1.9, 3.5])
    a, b = osc.trace_closest(rays, counts=True), ob.trace_closest(rays, counts=True)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    # normals: hit an instanced triangle and compare with M3 applied to the (quantised) object-space normal
    t, u, v, p = a[:4]
    sel = np.nonzero((p != 0xFFFFFFFF) & (p >= base.n_triangles))[0][:200]
    assert len(sel) > 20
    L = orc.lib()
    for i in sel:
        surf = np.zeros(11, np.float32)
        L.orc_hit_info(osc.h, int(p[i]), float(u[i]), float(v[i]), orc.ptr(surf))
        which = 1 if p[i] < base.n_triangles + int(base.prim_counts[k]) else 2
        local = int(p[i]) - base.n_triangles - (which - 1) * int(base.prim_counts[k])
        plain = np.zeros(11, np.float32)
        L.orc_hit_info(osc.h, first + local, float(u[i]), float(v[i]), orc.ptr(plain))  # the identity placement of the same triangle (instance 0)
        want = mats[which][:3, :3].astype(np.float64) @ plain[6:9].astype(np.float64)
        want /= np.linalg.norm(want)
        assert np.allclose(surf[6:9], want, atol=2e-6)
        assert np.array_equal(surf[:6], plain[:6]) and np.array_equal(surf[9:], plain[9:])  # the material travels with the geometry
