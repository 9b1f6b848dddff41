"""Committed golden fixtures (tests/golden/*.npz, produced by tests/golden/gen_golden.py from the CPU oracle; the reference
ships no golden vectors).  CPU suite: the oracle still reproduces them.  GPU suite: the HIP path does."""
import ctypes as C
from pathlib import Path

import numpy as np
import pytest

import orc
from raytracer3_amd import _lib as L
from raytracer3_amd.assets import Mesh

GOLD = Path(__file__).resolve().parent / "golden"
SCENES = ["cornell_ref", "atrium_full", "atrium_spec"]


def load(name):
    z = np.load(GOLD / f"{name}.npz")
    mesh = Mesh(z["vertices"], z["indices"], z["geometries"], z["prim_counts"])
    sky = z["sky"] if z["sky"].size else None
    bn = None
    if name.startswith("atrium"):
        from raytracer3_amd import assets

        bn = assets.load_bluenoise()
    return z, mesh, sky, bn


def gconst_from(z, cls):
    g = cls()
    C.memmove(C.byref(g), z["gconst"].tobytes(), 304)
    return g


@pytest.mark.parametrize("name", SCENES)
def test_oracle_reproduces_golden(name):
    z, mesh, sky, bn = load(name)
    osc = orc.Scene(mesh, sky, bn)
    assert np.array_equal(osc.nodes(), z["bvh_nodes"]) and np.array_equal(osc.tris(), z["bvh_tris"])
    t, u, v, p, nn, nt = osc.trace_closest(z["rays"], counts=True)
    assert np.array_equal(p, z["hit_prim"]) and np.array_equal(t, z["hit_t"]) and np.array_equal(u, z["hit_u"]) and np.array_equal(v, z["hit_v"])
    assert np.array_equal(nn, z["n_nodes"]) and np.array_equal(nt, z["n_tris"])
    assert np.array_equal(osc.trace_any(z["rays"]), z["occluded"])
    g = gconst_from(z, orc.GConst)
    gb, depth = osc.gbuffer(g)
    assert np.array_equal(gb, z["gbuffer"]) and np.array_equal(depth, z["depth"])
    light, counts = osc.reference_mode(g, gb, depth, threads=3)  # thread count must not matter
    assert np.array_equal(light.view(np.uint32), z["light"].view(np.uint32)) and np.array_equal(counts[:4], z["counts"][:4])


def test_oracle_reproduces_function_golden():
    z = np.load(GOLD / "functions.npz")
    Lo = orc.lib()
    for i in range(64):
        out = np.zeros(4, np.uint32)
        Lo.orc_gbuffer_pack(orc.ptr(np.ascontiguousarray(z["surf"][i])), orc.ptr(out))
        assert np.array_equal(out, z["packed"][i])
        un = np.zeros(11, np.float32)
        Lo.orc_gbuffer_unpack(orc.ptr(np.ascontiguousarray(z["packed"][i])), orc.ptr(un))
        assert np.array_equal(un, z["unpacked"][i])
    # lossy G-buffer: what survives the 128-bit texel (gbuffer_helpers.slang:22-34)
    assert np.abs(np.sqrt(z["unpacked"][:, 0:3]) - np.sqrt(z["surf"][:, 0:3])).max() <= 0.5 / 255 + 1e-6
    assert (np.einsum("ij,ij->i", z["unpacked"][:, 6:9], z["surf"][:, 6:9]) > 0.9999).all()
    assert np.abs(np.sqrt(z["unpacked"][:, 9]) - np.sqrt(z["surf"][:, 9])).max() < 1e-3
    # DiffuseBrdf.sample: unit vectors in the upper hemisphere, cos(theta) = sqrt(1 - u1) (brdf.slang:56-65)
    wi, ur = z["diffuse_wi"], z["urand"]
    assert np.allclose(np.linalg.norm(wi, axis=1), 1, atol=1e-6) and np.allclose(wi[:, 2], np.sqrt(1 - ur[:, 1]), atol=1e-6)
    assert np.allclose(np.arctan2(wi[:, 1], wi[:, 0]) % (2 * np.pi), 2 * np.pi * ur[:, 0], atol=1e-5)
    # build_orthonormal_basis: right-handed orthonormal frame (math.slang:29-50)
    n, ob = z["normals"], z["onb"]
    b1, b2 = ob[:, :3], ob[:, 3:]
    assert np.allclose(np.einsum("ij,ij->i", b1, b2), 0, atol=1e-6) and np.allclose(np.einsum("ij,ij->i", b1, n), 0, atol=1e-6)
    assert np.allclose(np.cross(b1, b2), n, atol=1e-6)


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", SCENES)
def test_gpu_reproduces_golden(name):
    from raytracer3_amd.renderer import PathTracer

    z, mesh, sky, bn = load(name)
    g = gconst_from(z, L.GConst)
    W, H = int(g.window_size[0]), int(g.window_size[1])
    pt = PathTracer((W, H))
    pt.set_scene(mesh, sky, bn)
    nodes, tris = pt.ctx.accel_download()
    assert np.array_equal(nodes, z["bvh_nodes"]) and np.array_equal(tris, z["bvh_tris"])
    t, u, v, p, cn, ct, _ = pt.ctx.trace_rays(z["rays"], counts=True)
    hit = p != L.MISS
    assert np.array_equal(p, z["hit_prim"]) and np.array_equal(t[hit], z["hit_t"][hit]) and np.array_equal(u[hit], z["hit_u"][hit])
    assert np.array_equal(v[hit], z["hit_v"][hit]) and np.array_equal(cn, z["n_nodes"]) and np.array_equal(ct, z["n_tris"])
    assert np.array_equal(pt.ctx.trace_rays(z["rays"], any_hit=True)[3], z["occluded"])
    pt.render(g)
    gb, depth = pt.gbuffer()
    hitpx = z["depth"] != L.BACKGROUND_DEPTH
    assert np.array_equal(depth, z["depth"]) and np.array_equal(gb[hitpx], z["gbuffer"][hitpx])
    light = pt.light()
    assert np.array_equal(light.view(np.uint32), z["light"].view(np.uint32))
    st = pt.ctx.stats()
    assert st.extension_rays == W * H + int(z["counts"][0]) and st.shadow_rays == int(z["counts"][1])
    pt.close()


@pytest.mark.gpu
def test_gpu_device_functions_known_answers():
    """SURVEY 8c integer KATs and the function fixtures, evaluated ON the GPU (rt3_selftest_eval)."""
    from raytracer3_amd.render_graph import Context

    ctx = Context(0)
    u32 = lambda a: np.array(a, np.uint32)  # noqa: E731
    assert ctx.selftest(0, u32([[0], [1], [0xDEADBEEF]]), 1).ravel().tolist() == [0x6B4ED927, 0xB48681B6, 0x7FF0EADA]
    assert ctx.selftest(1, u32([[3, 5], [1919, 1079], [65535, 65535]]), 1).ravel().tolist() == [39, 3481471, 0xFFFFFFFF]
    assert ctx.selftest(11, u32([[0, 0, 0], [1, 0, 0], [960, 540, 7]]), 1).ravel().tolist() == [0x6B4ED927, 0xB48681B6, 0xEE26C3FF]
    assert ctx.selftest(2, u32([[0x6B4ED927, 0], [0x6B4ED927, 1], [0x6B4ED927, 2], [0xEE26C3FF, 0]]), 1).ravel().tolist() == [
        0x312DDF77, 0xE5DEE3C0, 0x2276F4DC, 0x5006FA9B]
    fl = ctx.selftest(3, u32([[0x6B4ED927, 0], [0x6B4ED927, 1], [0x6B4ED927, 2]]), 1, np.float32).ravel().tolist()
    assert fl == [0.35838210582733154, 0.7413253784179688, 0.9293475151062012]
    z = np.load(GOLD / "functions.npz")
    seeds = np.array([[orc.lib().orc_rng_seed(int(a), int(b), int(c)), k] for a, b, c in z["rng_seeds"] for k in range(8)], np.uint32)
    assert np.array_equal(ctx.selftest(2, seeds, 1).reshape(-1, 8), z["rng_out"])
    assert np.array_equal(ctx.selftest(4, z["surf"].view(np.uint32), 4), z["packed"])
    assert np.array_equal(ctx.selftest(5, z["packed"], 11), z["unpacked"].view(np.uint32))
    # division-free integer helpers of k_shade (magic-number divide, conditional wrap) against Python's exact integers
    rng = np.random.default_rng(11)
    nd = np.stack([rng.integers(0, 2**32, 4096, dtype=np.uint64), rng.integers(1, 2**32, 4096, dtype=np.uint64)], 1)
    nd[:64, 1] = np.arange(1, 65)
    nd[64:128, 1] = 2 ** rng.integers(0, 32, 64)
    nd[128:192, 0] = 0xFFFFFFFF - np.arange(64)
    nd[192:256, 1] = 0xFFFFFFFF - np.arange(64)
    nd[256:320] = [[1920 * 1080 * 16 - 1 - k, 1920 * 1080] for k in range(64)]
    got = ctx.selftest(12, nd.astype(np.uint32), 3)
    assert np.array_equal(got[:, 0], (nd[:, 0] // nd[:, 1]).astype(np.uint32)) and np.array_equal(got[:, 1], (nd[:, 0] % nd[:, 1]).astype(np.uint32))
    xs = nd[:, 0].astype(np.uint32).view(np.int32).astype(np.int64)
    xs[:2048] = rng.integers(-70000, 140000, 2048)
    ws = (nd[:, 1] & 0xFFFF).astype(np.int64) + 1
    nd2 = np.stack([xs.astype(np.int32).view(np.uint32), nd[:, 1].astype(np.uint32)], 1)
    assert np.array_equal(ctx.selftest(12, nd2, 3)[:, 2].astype(np.int64), np.mod(xs, ws))
    assert np.array_equal(ctx.selftest(6, z["urand"].view(np.uint32), 3), z["diffuse_wi"].view(np.uint32))
    assert np.array_equal(ctx.selftest(7, z["normals"].view(np.uint32), 6), z["onb"].view(np.uint32))
    agx_in = np.stack([z["agx_in"], z["agx_in"] * np.float32(0.5), z["agx_in"] * np.float32(0.25)], 1).astype(np.float32)
    agx = ctx.selftest(8, agx_in.view(np.uint32), 3, np.float32)
    assert np.allclose(agx, z["agx_out"], atol=2e-5, rtol=1e-4)  # log2f / powf: libm vs device, tolerance
    us = np.linspace(0, 1, 4097, dtype=np.float32)[:-1].reshape(-1, 1)
    sc = ctx.selftest(9, us.view(np.uint32), 2, np.float32)
    s, c = C.c_float(), C.c_float()
    for i in range(0, len(us), 37):
        orc.lib().orc_sincos_2pi(float(us[i, 0]), C.byref(s), C.byref(c))
        assert sc[i, 0] == s.value and sc[i, 1] == c.value
    assert np.abs(sc[:, 0] - np.sin(2 * np.pi * us[:, 0].astype(np.float64))).max() < 3e-7
    yx = np.random.default_rng(0).normal(size=(512, 2)).astype(np.float32)
    at = ctx.selftest(10, yx.view(np.uint32), 1, np.float32).ravel()
    assert np.array_equal(at, np.array([orc.lib().orc_atan2(float(y), float(x)) for y, x in yx], np.float32))
    ctx.close()
