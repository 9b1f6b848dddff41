"""Probe-GI passes: structured_importance_sampling -> trace_probes -> spherical_harmonic_conversion ->
interpolate_probes.  Parity unpinned upstream (no host wiring, no outputs: see oracle/rt3_oracle_probes.c), so the CPU half
pins the oracle by known answers and physical properties and the GPU half demands bit-exact agreement with it."""
import ctypes as C
import math
from pathlib import Path

import numpy as np
import pytest

import orc
from raytracer3_amd import _lib as L
from raytracer3_amd import scenes

GOLDEN = Path(__file__).resolve().parent / "golden"
FW, FH, PX, PY = 96, 64, 6, 4  # the probes.npz fixture: window and probe grid


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def octa(fx, fy):
    n = (C.c_float * 3)()
    orc.lib().orc_octa_decode(float(fx), float(fy), n)
    return np.array(n[:], np.float32)


def sh9(d):
    out = np.zeros(9, np.float32)
    orc.lib().orc_sh3_evaluate(orc.ptr(np.ascontiguousarray(d, np.float32)), orc.ptr(out))
    return out


# ------------------------------------------------------------------------------------------------ CPU: oracle pinned
def test_octa_decode_known_answers():
    assert octa(0.5, 0.5).tolist() == [0.0, 0.0, 1.0]      # centre of the map = +z (packing.slang:77-86)
    assert octa(1.0, 0.5).tolist() == [1.0, 0.0, 0.0]
    assert octa(0.5, 0.0).tolist() == [0.0, -1.0, 0.0]
    assert octa(0.0, 0.0).tolist() == [0.0, 0.0, -1.0]     # the four corners fold onto -z
    g = (np.arange(8) + 0.5) / 8
    d = np.array([octa(x, y) for y in g for x in g], np.float64)
    assert np.allclose(np.linalg.norm(d, axis=1), 1.0, atol=1e-6)
    assert np.abs(d.mean(0)).max() < 1e-6                   # the 64 texel centres are point-symmetric
    assert (d[:, 2] > 0).sum() == 24 and (d[:, 2] == 0).sum() == 16  # 16 of the 8x8 texel centres lie on the fold |x| + |y| = 1


def test_sh3_basis_known_answers():
    z = sh9([0, 0, 1])
    assert z[0] == np.float32(0.28209479177387814) and z[2] == np.float32(0.4886025119029199)
    assert z[6] == np.float32(0.31539156525252) * np.float32(2.0) and z[1] == 0 and z[3] == 0 and z[4] == 0
    # orthonormality of the basis under uniform sphere sampling (Monte Carlo, 200k directions)
    rng = np.random.default_rng(0)
    v = rng.normal(size=(200_000, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    x, y, zz = v.T
    B = np.stack([np.full_like(x, 0.28209479), -0.48860251 * y, 0.48860251 * zz, -0.48860251 * x, 1.09254843 * x * y, 1.09254843 * y * zz,
                  0.31539157 * (3 * zz * zz - 1), 1.09254843 * x * zz, 0.54627422 * (x * x - y * y)])
    assert np.allclose(sh9(v[0]), B[:, 0], atol=1e-6)
    gram = 4 * math.pi * (B @ B.T) / len(v)
    assert np.abs(gram - np.eye(9)).max() < 0.02


def test_wave_sort_and_sum_rules():
    rng = np.random.default_rng(1)
    for trial in range(8):
        k = rng.normal(size=64).astype(np.float32)
        if trial == 7:
            k[::3] = k[0]  # ties
        k0, idx = k.copy(), np.zeros(64, np.uint32)
        orc.lib().orc_wave_sort64(orc.ptr(k), orc.ptr(idx))
        assert np.all(np.diff(k) >= 0) and np.array_equal(k0[idx], k) and sorted(idx.tolist()) == list(range(64))
        s = orc.lib().orc_wave_sum64(orc.ptr(k0))
        assert abs(s - float(k0.astype(np.float64).sum())) < 1e-4
    ones = np.ones(64, np.float32)
    assert orc.lib().orc_wave_sum64(orc.ptr(ones)) == 64.0


def fixture_mesh(z):
    m = scenes.cornell()
    assert np.array_equal(m.vertices, z["vertices"]) and np.array_equal(m.indices, z["indices"])
    return m


def test_oracle_reproduces_probe_fixture():
    z = np.load(GOLDEN / "probes.npz")
    g = orc.GConst.from_buffer_copy(z["gconst"].tobytes())
    assert g.pad[0] == orc.F_PROBE_RADIANCE
    osc = orc.Scene(fixture_mesh(z))
    gb, depth = z["gbuffer"], z["depth"]
    dirs, dbg = orc.structured_importance_sampling(g, gb, PX, PY)
    assert np.array_equal(dirs, z["directions"]) and np.array_equal(dbg, z["debug"])
    assert np.array_equal(bits(orc.trace_probes(osc, g, gb, depth, dirs, z["prev_atlas"])), bits(z["atlas_rad"]))
    g0 = orc.GConst.from_buffer_copy(bytes(g)); g0.pad[0] = 0
    assert np.array_equal(bits(orc.trace_probes(osc, g0, gb, depth, dirs, z["prev_atlas"])), bits(z["atlas"]))
    assert np.array_equal(bits(orc.trace_probes(osc, g0, gb, depth, z["directions_mixed"], z["prev_atlas"])), bits(z["atlas_mixed"]))
    sh = orc.sh_conversion(z["atlas_rad"])
    assert np.array_equal(bits(sh), bits(z["sh"]))
    assert np.array_equal(bits(orc.interpolate_probes(g, gb, depth, sh)), bits(z["light"]))
    g1 = orc.GConst.from_buffer_copy(bytes(g)); g1.proberng = 1
    assert np.array_equal(bits(orc.interpolate_probes(g1, gb, depth, sh)), bits(z["light_rng"]))


def test_importance_sampling_as_written():
    """Every ray of a probe except the one(s) with the smallest BRDF pdf is promoted to mip 1 at index 4 * thread
    (structured_importance_sampling.slang:55-69); debug ends as the never-assigned cull index -1 (:70)."""
    z = np.load(GOLDEN / "probes.npz")
    d = z["directions"].reshape(PY, 8, PX, 8).transpose(0, 2, 1, 3).reshape(PX * PY, 64)
    ti = np.arange(64)
    for probe in d:
        hi = (probe >> 15) == 1
        assert np.array_equal(probe[hi] & 0x7FFF, 4 * ti[hi]) and np.array_equal(probe[~hi], ti[~hi])
        assert 1 <= (~hi).sum() <= 32
    assert np.all(z["debug"] == -1.0)


def test_trace_probes_as_written_and_radiance_variant():
    z = np.load(GOLDEN / "probes.npz")
    depth, atlas, rad, prev = z["depth"], z["atlas"], z["atlas_rad"], z["prev_atlas"]
    g = orc.GConst.from_buffer_copy(z["gconst"].tobytes())
    for py in range(PY):
        for px in range(PX):
            cell = atlas[py * 8:py * 8 + 8, px * 8:px * 8 + 8]
            if depth[py * 16, px * 16] == orc.BACKGROUND_DEPTH:  # trace_probes.slang:29-31
                assert np.all(cell[..., :3] == 0) and np.all(cell[..., 3] == orc.BACKGROUND_DEPTH)
                assert np.all(rad[py * 8:py * 8 + 8, px * 8:px * 8 + 8, 3] == orc.BACKGROUND_DEPTH)
            else:  # written texels carry (direction_2d / size, 0, depth) and sit where that coordinate points (:74)
                wrote = cell[..., 3] != 0
                ys, xs = np.nonzero(wrote)
                assert wrote.any() and np.array_equal((cell[wrote][:, 0] * 8).astype(int), xs) and np.array_equal((cell[wrote][:, 1] * 8).astype(int), ys)
                assert np.all(cell[~wrote] == 0) and np.all(cell[..., 2] == 0)
    hit = rad[..., 3] != orc.BACKGROUND_DEPTH
    assert g.blendfactor == 0.25 and np.all(rad[hit][:, :3] >= 0.75 * prev[hit][:, :3] - 1e-6)  # lerp(prev, emissive >= 0, 0.25)
    assert (rad[hit][:, :3] > 0.75 * prev[hit][:, :3] + 0.5).any()  # some probe ray sees the lamp


def test_constant_radiance_gives_albedo_plus_emissive():
    """SH projection of radiance 1 -> cosine-lobe irradiance pi -> Light = albedo * 1 + emissive (interpolate_probes.slang:81-102),
    up to the unequal solid angles of the 64 octahedral texels (< 1 %)."""
    z = np.load(GOLDEN / "probes.npz")
    g = orc.GConst.from_buffer_copy(z["gconst"].tobytes())
    sh = orc.sh_conversion(np.ones((PY * 8, PX * 8, 4), np.float32))
    e = sh.reshape(-1, 12)[orc.lib().orc_zcurve(3 * 2 + 1, 1)]
    assert abs(e[0] - 2 * math.sqrt(math.pi)) < 1e-5 and np.abs(e[[1, 2, 4, 5, 6, 8, 9, 10]]).max() < 0.08  # only the zonal L2 term leaks (-0.06)
    light = orc.interpolate_probes(g, z["gbuffer"], z["depth"], sh)
    surf = np.zeros(11, np.float32)
    n = 0
    for y in range(0, FH, 3):
        for x in range(0, FW, 5):
            if z["depth"][y, x] == orc.BACKGROUND_DEPTH:
                assert np.all(light[y, x] == 0)  # :19-22 untouched
                continue
            if light[y, x].tolist() == [1.0, 0.0, 0.0, 1.0]:
                continue  # "interpolation failed" mark
            orc.lib().orc_gbuffer_unpack(orc.ptr(np.ascontiguousarray(z["gbuffer"][y, x])), orc.ptr(surf))
            assert np.abs(light[y, x, :3] - (surf[0:3] + surf[3:6])).max() < 0.01 and light[y, x, 3] == 1.0
            n += 1
    assert n > 100


# ------------------------------------------------------------------------------------------------ GPU: bit-exact vs the oracle
gpu = pytest.mark.gpu


def launch(pt, name, x, y, z, g, bindings):
    b = (C.c_uint32 * len(bindings))(*bindings)
    return pt.ctx.lib.rt3_pass_launch(pt.ctx.h, name.encode(), b"main", x, y, z, C.byref(g), C.sizeof(g), b, len(bindings))


def as_lib(g: orc.GConst) -> L.GConst:
    o = L.GConst()
    C.memmove(C.byref(o), C.byref(g), 304)
    return o


@gpu
def test_gpu_probe_device_functions_bit_exact():
    from raytracer3_amd.render_graph import Context
    ctx = Context(0)
    rng = np.random.default_rng(5)
    uv = np.concatenate([rng.uniform(0, 1, (500, 2)), [[0.5, 0.5], [0, 0], [1, 1], [1, 0.5], [0.5, 0], [0.25, 0.75]]]).astype(np.float32)
    got = ctx.selftest(13, uv.view(np.uint32), 3)
    assert np.array_equal(got, np.array([bits(octa(a, b)) for a, b in uv]))
    d = rng.normal(size=(300, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    assert np.array_equal(ctx.selftest(14, d.view(np.uint32), 9), np.array([bits(sh9(v)) for v in d]))
    keys = rng.normal(size=(40, 64)).astype(np.float32)
    keys[3, ::2] = keys[3, 0]; keys[4] = 1.0; keys[5] = np.arange(64, 0, -1)
    got = ctx.selftest(15, keys.view(np.uint32), 128)
    for row, k in zip(got, keys):
        kk, idx = k.copy(), np.zeros(64, np.uint32)
        orc.lib().orc_wave_sort64(orc.ptr(kk), orc.ptr(idx))
        assert np.array_equal(row[:64], bits(kk)) and np.array_equal(row[64:], idx)
    sums = ctx.selftest(16, keys.view(np.uint32), 1).ravel()
    want = np.array([orc.lib().orc_wave_sum64(orc.ptr(np.ascontiguousarray(k))) for k in keys], np.float32)
    assert np.array_equal(sums, bits(want))
    ctx.close()


def make_tracer(mesh, W, H, sky=None):
    from raytracer3_amd.renderer import PathTracer
    pt = PathTracer((W, H))
    pt.set_scene(mesh, sky)
    return pt


@gpu
def test_gpu_probe_passes_match_fixture_stage_by_stage():
    """Each pass is fed the fixture's inputs, so a difference is attributed to exactly one kernel."""
    z = np.load(GOLDEN / "probes.npz")
    og = orc.GConst.from_buffer_copy(z["gconst"].tobytes())
    W, H, AW, AH = FW, FH, PX * 8, PY * 8
    pt = make_tracer(fixture_mesh(z), W, H)
    g = as_lib(og)
    h = pt.probe_commands(g)
    rg = pt.rg
    rg.upload(h["gbuffer"], z["gbuffer"]); rg.upload(h["depth"], z["depth"]); rg.upload(h["prev_atlas"], z["prev_atlas"])
    # structured_importance_sampling
    assert launch(pt, "structured_importance_sampling", PX, PY, 1, g, [h["gbuffer"], h["depth"], h["directions"], h["debug"], h["atlas"]]) == 0
    assert np.array_equal(rg.download(h["directions"], (AH, AW), np.uint16), z["directions"])
    assert np.array_equal(rg.download(h["debug"], (AH, AW), np.float32), z["debug"])
    # trace_probes: radiance variant, as written, as written with colliding / out-of-map direction words
    tp = [h["gbuffer"], h["depth"], h["directions"], h["atlas"], h["prev_atlas"]]
    assert launch(pt, "trace_probes", AW, AH, 1, g, tp) == 0
    assert np.array_equal(bits(rg.download(h["atlas"], (AH, AW, 4), np.float32)), bits(z["atlas_rad"]))
    g0 = as_lib(og); g0.pad[0] = 0
    assert launch(pt, "trace_probes", AW, AH, 1, g0, tp) == 0
    assert np.array_equal(bits(rg.download(h["atlas"], (AH, AW, 4), np.float32)), bits(z["atlas"]))
    rg.upload(h["directions"], z["directions_mixed"])
    assert launch(pt, "trace_probes", AW, AH, 1, g0, tp) == 0
    assert np.array_equal(bits(rg.download(h["atlas"], (AH, AW, 4), np.float32)), bits(z["atlas_mixed"]))
    # spherical_harmonic_conversion
    rg.upload(h["atlas"], z["atlas_rad"])
    assert launch(pt, "spherical_harmonic_conversion", PX, PY, 1, g, [h["sh"], h["atlas"]]) == 0
    sh = rg.download(h["sh"], z["sh"].shape, np.float32)
    used = np.zeros(len(sh) // 12, bool)
    for gy in range(PY):
        for gx in range(3 * PX):
            used[orc.lib().orc_zcurve(gx, gy)] = True
    assert np.array_equal(bits(sh.reshape(-1, 12)[used]), bits(z["sh"].reshape(-1, 12)[used]))
    # interpolate_probes, SH and normal-debug (proberng) modes
    rg.upload(h["sh"], z["sh"])
    ip = [h["gbuffer"], h["depth"], h["sh"], h["light"]]
    for gg, key in ((g, "light"), (None, "light_rng")):
        if gg is None:
            gg = as_lib(og); gg.proberng = 1
        rg.upload(h["light"], np.zeros((H, W, 4), np.float32))
        assert launch(pt, "interpolate_probes", W // 8, H // 8, 1, gg, ip) == 0
        assert np.array_equal(bits(rg.download(h["light"], (H, W, 4), np.float32)), bits(z[key]))
    pt.close()


@gpu
@pytest.mark.parametrize("flags", [0, L.F_PROBE_RADIANCE])
def test_gpu_probe_frame_matches_oracle_chain(flags):
    """The whole probe frame through the render graph on the atrium (background probes, many failed interpolations),
    two frames so that the temporal blend input is exercised."""
    from raytracer3_amd.renderer import Camera
    mesh, sky = scenes.atrium(0.3), scenes.sky(64, 32)
    W, H = 176, 100  # not a multiple of 16: 11 x 6 probes, the last rows / columns of pixels use clamped probes
    pt = make_tracer(mesh, W, H, sky)
    osc = orc.Scene(mesh, sky)
    cam = Camera(scenes.ATRIUM_CAMERA["position"], scenes.ATRIUM_CAMERA["direction"], math.radians(scenes.ATRIUM_CAMERA["fov_deg"]), W / H)
    oprev = np.zeros((48, 88, 4), np.float32)
    olight = np.zeros((H, W, 4), np.float32)
    for frame in (4, 5):
        g = pt.make_gconst(cam, 1, 1, frame=frame, blendfactor=0.3, flags=flags)
        h = pt.render_probes(g)
        og = orc.GConst.from_buffer_copy(bytes(g))
        gb, depth = pt.gbuffer()
        ogb, odepth = osc.gbuffer(og)
        hit = odepth != orc.BACKGROUND_DEPTH
        assert np.array_equal(bits(depth), bits(odepth)) and np.array_equal(gb[hit], ogb[hit])
        gbx = np.where(hit[..., None], ogb, gb)  # background texels of the G-buffer are never written: take the device's
        odirs, odbg = orc.structured_importance_sampling(og, gbx, 11, 6)
        assert np.array_equal(pt.rg.download(h["directions"], (48, 88), np.uint16), odirs)
        oatlas = orc.trace_probes(osc, og, gbx, odepth, odirs, oprev)
        atlas = pt.rg.download(h["atlas"], (48, 88, 4), np.float32)
        assert np.array_equal(bits(atlas), bits(oatlas))
        osh = orc.sh_conversion(oatlas)
        olight = orc.interpolate_probes(og, gbx, odepth, osh, olight)  # Light keeps what a frame does not overwrite
        light = pt.light()
        assert np.array_equal(bits(light), bits(olight)), (frame, np.argwhere((bits(light) != bits(olight)).any(-1))[:5])
        red = (light[..., 0] == 1) & (light[..., 1] == 0) & (light[..., 3] == 1)
        assert 0.02 < red.mean() < 0.9 and (light[..., 3] == 0).any()
        pt.copy_atlas_to_prev()
        oprev = oatlas
    if flags:
        assert oatlas[..., :3].max() > 0
    pt.close()


@gpu
def test_gpu_probe_frame_under_the_tile_partition():
    """VERDICT r2 item 9 (SURVEY rows f4 x e): the probe-GI frame with n_ranks > 1.  The chain reads the whole G-buffer (jittered
    neighbours, probes two cells away, red marks scattered to other pixels) and is launch-bound under a millisecond, so it runs
    REPLICATED: every rank renders it for the whole window; the frame-end gather of each rank's own tiles (here: packed at the offsets
    rt3_gather_layout reports, one rt3_gather_unpack on the root -- RCCL cannot put three ranks on one GPU) then assembles the very image
    a single rank produces, two frames with the temporal blend input, and each rank's own full image equals it as well."""
    from raytracer3_amd.renderer import Camera, PathTracer

    mesh, sky = scenes.atrium(0.3), scenes.sky(64, 32)
    W, H, n, root = 176, 100, 3, 0
    cam = Camera(scenes.ATRIUM_CAMERA["position"], scenes.ATRIUM_CAMERA["direction"], math.radians(scenes.ATRIUM_CAMERA["fov_deg"]), W / H)
    solo = make_tracer(mesh, W, H, sky)
    pts = [PathTracer((W, H), rank=r, n_ranks=n) for r in range(n)]
    for pt in pts:
        pt.set_scene(mesh, sky, None)
    for frame in (4, 5):
        g = solo.make_gconst(cam, 1, 1, frame=frame, blendfactor=0.3, flags=L.F_PROBE_RADIANCE)
        solo.render_probes(g)
        ref = solo.light()
        for pt in pts:
            pt.render_probes(g)
            assert np.array_equal(bits(pt.light()), bits(ref))  # a replica: the whole window on every rank
        rootpt = pts[root]
        img = rootpt.handles["light"]
        off = rootpt.ctx.gather_layout(img, root, n)
        recv = rootpt.rg.buffer(off[-1] * 16, "recv")
        ptr, _ = rootpt.rg.device_ptr(recv)
        rootpt.rg.upload(img, np.zeros((H, W, 4), np.float32))  # the gather must rebuild everything the root does not own...
        own = np.zeros((H, W), bool)
        xy = orc.tile_pixels(W, H, root, n)
        own[xy[:, 1], xy[:, 0]] = True
        keep = np.where(own[..., None], ref, 0).astype(np.float32)
        rootpt.rg.upload(img, keep)                              # ... from the other ranks' tiles alone
        for r, pt in enumerate(pts):
            if r != root:
                pt.ctx.check(pt.ctx.lib.rt3_image_pack_tiles(pt.ctx.h, pt.handles["light"], r, n, C.c_void_p(ptr + off[r] * 16)))
                pt.ctx.wait()
        rootpt.ctx.gather_unpack(img, root, n, ptr)
        assert np.array_equal(bits(rootpt.light()), bits(ref))
        solo.copy_atlas_to_prev()
        for pt in pts:
            pt.copy_atlas_to_prev()
    for pt in pts + [solo]:
        pt.close()


@gpu
def test_gpu_probe_pass_error_behaviour():
    mesh = scenes.cornell()
    pt = make_tracer(mesh, 64, 48)
    cam_g = as_lib(orc.camera_gconst(width=64, height=48, **scenes.CORNELL_CAMERA))
    h = pt.probe_commands(cam_g)
    E = L.E_INVALID
    assert launch(pt, "structured_importance_sampling", 5, 3, 1, cam_g, [h["gbuffer"], h["depth"], h["directions"], h["debug"], h["atlas"]]) == E  # > W/16 probes
    assert launch(pt, "structured_importance_sampling", 4, 3, 1, cam_g, [h["gbuffer"], h["depth"], h["debug"], h["directions"], h["atlas"]]) == E  # formats swapped
    assert launch(pt, "structured_importance_sampling", 4, 3, 1, cam_g, [h["gbuffer"], h["depth"], h["directions"], h["debug"]]) == E
    assert launch(pt, "trace_probes", 30, 24, 1, cam_g, [h["gbuffer"], h["depth"], h["directions"], h["atlas"], h["prev_atlas"]]) == E  # not 8 per probe
    assert launch(pt, "trace_probes", 32, 24, 1, cam_g, [h["gbuffer"], h["depth"], h["directions"], h["atlas"], h["atlas"]]) == E   # aliasing
    assert launch(pt, "spherical_harmonic_conversion", 4, 3, 1, cam_g, [h["atlas"], h["atlas"]]) == E  # `out` must be a buffer
    small = pt.rg.buffer(48, "too_small")
    assert launch(pt, "spherical_harmonic_conversion", 4, 3, 1, cam_g, [small, h["atlas"]]) == E
    assert b"at least" in pt.ctx.lib.rt3_last_error(pt.ctx.h)
    assert launch(pt, "interpolate_probes", 8, 6, 1, cam_g, [h["gbuffer"], h["depth"], small, h["light"]]) == E
    assert launch(pt, "interpolate_probes", 7, 6, 1, cam_g, [h["gbuffer"], h["depth"], h["sh"], h["light"]]) == E
    assert launch(pt, "probe_magic", 1, 1, 1, cam_g, []) == E
    # (a multi-rank tracer no longer refuses the probe frame: it renders it replicated, test_gpu_probe_frame_under_the_tile_partition)
    pt.close()


@gpu
def test_gpu_cpp_host_renders_the_probe_frame(tmp_path):
    """The compiled host (raytracer3_amd/host/example_frame.cpp, `probes` mode) describes the same five nodes with the C++
    builder chain; its Light and probe atlas must equal the oracle chain bit for bit."""
    import struct
    import subprocess
    from raytracer3_amd.renderer import Camera
    exe = Path(__file__).resolve().parent.parent / "raytracer3_amd" / "host" / "example_frame"
    if not exe.exists():
        subprocess.check_call(["make", "-C", str(exe.parent)])
    mesh = scenes.cornell()
    W, H, frame = 128, 80, 6
    cam = Camera(scenes.CORNELL_CAMERA["position"], scenes.CORNELL_CAMERA["direction"], math.radians(scenes.CORNELL_CAMERA["fov_deg"]), W / H)
    scene = tmp_path / "scene.bin"
    with open(scene, "wb") as f:
        f.write(struct.pack("<8I", len(mesh.vertices), len(mesh.indices), len(mesh.geometries), 0, 0, 0, 0, 0))
        for arr in (mesh.vertices.astype("<f4"), mesh.indices.astype("<u4"), mesh.geometries, mesh.prim_counts.astype("<u4")):
            f.write(np.ascontiguousarray(arr).tobytes())
        f.write(np.array([*cam.position, *cam.direction, cam.fov, cam.aspect_ratio], "<f4").tobytes())
    out = tmp_path / "out.bin"
    subprocess.check_call([str(exe), str(scene), str(W), str(H), "1", "1", str(L.F_PROBE_RADIANCE), str(frame), str(out), "probes"])
    data = np.fromfile(out, "<f4")
    light, atlas = data[:W * H * 4].reshape(H, W, 4), data[W * H * 4:].reshape(H // 16 * 8, W // 16 * 8, 4)
    g = cam.gconst((W, H))
    g.samples, g.bounces, g.frame, g.blendfactor = 1, 1, frame, 1.0
    g.pad[0] = L.F_PROBE_RADIANCE
    og = orc.GConst.from_buffer_copy(bytes(g))
    osc = orc.Scene(mesh)
    ogb, odepth = osc.gbuffer(og)
    assert (odepth != orc.BACKGROUND_DEPTH).all()  # closed box: every G-buffer texel is written
    odirs, _ = orc.structured_importance_sampling(og, ogb, W // 16, H // 16)
    oatlas = orc.trace_probes(osc, og, ogb, odepth, odirs, np.zeros_like(atlas))
    assert np.array_equal(bits(atlas), bits(oatlas))
    assert np.array_equal(bits(light), bits(orc.interpolate_probes(og, ogb, odepth, orc.sh_conversion(oatlas))))
    assert oatlas[..., :3].max() > 1.0  # the lamp is seen
