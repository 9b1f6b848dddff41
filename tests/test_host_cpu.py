"""CPU-side checks of the product's host layer: the C-ABI library loads and exports every symbol of include/rt3.h,
refuses to run without a GPU, and the Python mirror of the reference pass graph orders / binds passes like bake.rs."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

import orc
from raytracer3_amd import _lib as L
from raytracer3_amd import assets, scenes
from raytracer3_amd.render_graph import IMPORTED, ComputePass, DispatchSize, ImageSize, NodeBuilder, RayTracingPass, RenderGraph, WorkSize2D
from raytracer3_amd.renderer import Camera

ROOT = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol():
    lib = L.load()
    header = (ROOT / "include" / "rt3.h").read_text()
    declared = set(re.findall(r"\b(rt3_[a-z0-9_]+)\s*\(", header))
    assert declared == set(L.EXPORTS), declared ^ set(L.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name


def test_no_cpu_fallback(has_gpu):
    if has_gpu:
        pytest.skip("GPU present")
    lib = L.load()
    h = C.c_void_p()
    assert lib.rt3_create(0, C.byref(h)) == L.E_NO_DEVICE
    assert b"no CPU fallback" in lib.rt3_last_error(None)


def test_gconst_layout_matches_reference_offsets():
    # renderer/mod.rs:47-63 -> offsets 0,64,128,192,256,264,268,272,276,280,284,288,296
    exp = dict(proj=0, view=64, proj_inverse=128, view_inverse=192, window_size=256, frame=264, blendfactor=268, bounces=272, samples=276,
               proberng=280, cell_size=284, mouse=288, pad=296)
    for k, v in exp.items():
        assert getattr(L.GConst, k).offset == v
        assert getattr(orc.GConst, k).offset == v
    assert C.sizeof(L.GConst) == 304
    assert assets.GEOMETRY_DTYPE.itemsize == 64 and assets.GEOMETRY_DTYPE.fields["emission"][1] == 32 and assets.GEOMETRY_DTYPE.fields["roughness"][1] == 48


def test_camera_matches_oracle_and_reference_defaults():
    cam = Camera((0, 0, -1), (0, 0, 1), np.deg2rad(65.0), 1920 / 1088)  # main.rs:69-76
    g = cam.gconst((1920, 1088))
    o = orc.camera_gconst((0, 0, -1), (0, 0, 1), 65.0, 1920, 1088)
    for f in ("proj", "view", "proj_inverse", "view_inverse"):
        assert np.allclose(np.array(getattr(g, f)[:]), np.array(getattr(o, f)[:]), rtol=2e-6, atol=1e-7), f
    assert tuple(g.window_size) == (1920.0, 1088.0) and g.blendfactor == 1.0


class FakeCtx:
    """Records rt3_pass_launch calls instead of running them (no GPU in the build container)."""

    def __init__(self):
        self.calls, self.n = [], 0
        self.lib, self.h = self, None

    def check(self, rc):
        assert rc == 0

    def rt3_image_create(self, h, w, hh, fmt, out):
        self.n += 1
        out._obj.value = (1 << 30) | self.n
        return 0

    def rt3_pass_launch(self, h, path, entry, x, y, z, cst, size, b, nb):
        self.calls.append((path.decode(), entry.decode(), x, y, z, size, [b[i] for i in range(nb)]))
        return 0

    def wait(self):
        pass


def build_frame(rg, g):
    gbuffer = rg.image(ImageSize.FullScreen, L.FORMAT_R32G32B32A32_UINT, "gbuffer")
    depth = rg.image(ImageSize.FullScreen, L.FORMAT_R32_SFLOAT, "gbuffer_depth")
    light = rg.image(ImageSize.FullScreen, L.FORMAT_R32G32B32A32_SFLOAT, "Light")
    prev = rg.image(ImageSize.FullScreen, L.FORMAT_R32G32B32A32_SFLOAT, "PrevLight")
    out = rg.image(ImageSize.FullScreen, L.FORMAT_R32G32B32A32_SFLOAT, "color")
    # declared out of execution order on purpose: bake() must order by data dependencies (bake.rs:29-49)
    gb = RayTracingPass.new(rg, "gbuffer").shader("gbuffer").constants(g).write(IMPORTED, gbuffer).write(IMPORTED, depth).launch(WorkSize2D.FullScreen)
    unrelated = ComputePass.new(rg, "unused").shader("postprocess").constants(g).write(IMPORTED, prev).dispatch(DispatchSize.XY(1, 1))
    pt = (RayTracingPass.new(rg, "refrence_mode").shader("refrence_mode").constants(g).read(gb, gbuffer).read(gb, depth)
          .write(IMPORTED, light).read(IMPORTED, prev).launch(WorkSize2D.FullScreen))
    ComputePass.new(rg, "postprocess").shader("postprocess").constants(g).read(gb, depth).write(IMPORTED, out).read(pt, light).dispatch(DispatchSize.FullScreen)
    return dict(gbuffer=gbuffer, depth=depth, light=light, prev=prev, out=out, unrelated=unrelated)


def test_render_graph_orders_and_binds_like_the_reference():
    ctx = FakeCtx()
    rg = RenderGraph(ctx, (1920, 1080))
    g = L.GConst()
    h = build_frame(rg, g)
    rg.draw_frame(h["out"])
    names = [c[0] for c in ctx.calls]
    assert names == ["gbuffer", "refrence_mode", "postprocess"]  # the node that does not feed the output is culled
    gbc, ptc, ppc = ctx.calls
    assert gbc[2:5] == (1920, 1080, 1) and ptc[2:5] == (1920, 1080, 1)  # WorkSize2D::FullScreen = window exactly
    assert ppc[2:5] == (240, 135, 1)  # DispatchSize::FullScreen = ceil(W/8) x ceil(H/8) groups (build.rs:254-258)
    assert gbc[6] == [h["gbuffer"], h["depth"]]
    assert ptc[6] == [h["gbuffer"], h["depth"], h["light"], h["prev"]]  # builder call order (bake.rs:51-83)
    assert ppc[6] == [h["depth"], h["out"], h["light"]]
    assert all(c[5] == 304 and c[1] == "main" for c in ctx.calls)
    # named resources are created once and looked up afterwards (mod.rs:440-483)
    assert rg.image(ImageSize.FullScreen, L.FORMAT_R32_SFLOAT, "gbuffer_depth") == h["depth"]
    assert rg.frame_number == 1


def test_render_graph_builder_errors():
    ctx = FakeCtx()
    rg = RenderGraph(ctx, (64, 64))
    g = L.GConst()
    a = rg.image(ImageSize.FullScreen, L.FORMAT_R32_SFLOAT, "a")
    n0 = RayTracingPass.new(rg, "p0").shader("gbuffer").constants(g).read(IMPORTED, a).launch()
    with pytest.raises(ValueError, match="allready used"):  # build.rs:57-59
        RayTracingPass.new(rg, "p0")
    with pytest.raises(ValueError, match="does not write"):  # build.rs:104-106
        RayTracingPass.new(rg, "p1").read(n0, a)
    b = rg.image(ImageSize.XY(8, 8), L.FORMAT_R32_SFLOAT, "b")
    with pytest.raises(ValueError, match="doesnt write"):  # build.rs:97-103
        RayTracingPass.new(rg, "p2").read(n0, b)
    with pytest.raises(ValueError, match="duplicate"):  # build.rs:195-198
        RayTracingPass.new(rg, "p3").write(IMPORTED, a).read_write(IMPORTED, a).launch()
    assert ImageSize.FractionalFullScreen(8, 8).size((1920, 1080)) == (240, 135)  # build.rs:219-226 ceil-div
    assert WorkSize2D.X(7).size((1, 1)) == (7, 1)


def test_glb_and_exr_roundtrip(tmp_path):
    mesh = scenes.atrium(0.1)
    p = tmp_path / "a.glb"
    assets.write_glb(p, mesh)
    back = assets.GltfMeshLoader.load(p)
    assert np.array_equal(back.vertices, mesh.vertices) and np.array_equal(back.indices, mesh.indices)
    assert np.array_equal(back.prim_counts, mesh.prim_counts)
    for f in ("base_color", "metallic_factor", "roughness", "emission", "index_offset", "vertex_offset"):
        assert np.array_equal(back.geometries[f], mesh.geometries[f]), f
    sky = scenes.sky(64, 32)
    e = tmp_path / "s.exr"
    assets.write_exr(e, sky)
    assert np.array_equal(assets.read_exr(e), sky)
    bn = assets.load_bluenoise()
    assert bn.shape == (256, 256, 4) and abs(bn.mean() - 127.5) < 0.01  # 4 independent uniform channels


def test_glb_textures_roundtrip(tmp_path):
    mesh = scenes.textured_cornell()
    p = tmp_path / "t.glb"
    assets.write_glb(p, mesh)
    back = assets.GltfMeshLoader.load(p)
    assert len(back.textures) == 2 and all(np.array_equal(a, b) for a, b in zip(back.textures, mesh.textures))
    assert np.array_equal(back.geometries["base_color_texture_index"], mesh.geometries["base_color_texture_index"])
    assert np.array_equal(back.vertices, mesh.vertices)


def test_glb_node_transforms_are_baked(tmp_path):
    import json, struct

    tri = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], "<f4")
    idx = np.array([0, 1, 2], "<u2")
    blob = tri.tobytes() + idx.tobytes() + b"\0\0"
    doc = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}],
           "nodes": [{"children": [1], "translation": [10, 0, 0]}, {"mesh": 0, "scale": [2, 2, 2]}],
           "meshes": [{"primitives": [{"attributes": {"POSITION": 0}, "indices": 1}, {"attributes": {"POSITION": 0}, "indices": 1, "material": 0}]}],
           "materials": [{"pbrMetallicRoughness": {"baseColorFactor": [0.1, 0.2, 0.3, 1], "roughnessFactor": 0.5, "metallicFactor": 0.25},
                          "emissiveFactor": [1, 1, 1], "extensions": {"KHR_materials_emissive_strength": {"emissiveStrength": 3.0}}}],
           "accessors": [{"bufferView": 0, "componentType": 5126, "count": 3, "type": "VEC3"}, {"bufferView": 1, "componentType": 5123, "count": 3, "type": "SCALAR"}],
           "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 36}, {"buffer": 0, "byteOffset": 36, "byteLength": 6}],
           "buffers": [{"byteLength": len(blob)}]}
    js = json.dumps(doc).encode()
    js += b" " * ((-len(js)) % 4)
    p = tmp_path / "t.glb"
    p.write_bytes(struct.pack("<4sII", b"glTF", 2, 12 + 8 + len(js) + 8 + len(blob)) + struct.pack("<I4s", len(js), b"JSON") + js
                  + struct.pack("<I4s", len(blob), b"BIN\0") + blob)
    m = assets.GltfMeshLoader.load(p)
    assert m.n_triangles == 2 and len(m.geometries) == 2  # both primitives (the reference reads only the first, assets/mod.rs:221)
    assert np.allclose(m.vertices[:3, :3], tri * 2 + [10, 0, 0])
    assert np.allclose(m.vertices[:3, 3:6], [0, 0, 1])  # flat normal generated, transformed by the inverse transpose
    assert np.allclose(m.geometries["base_color"][1], [0.1, 0.2, 0.3, 1]) and np.isclose(m.geometries["roughness"][1], 0.5)
    assert np.allclose(m.geometries["emission"][1][:3], 3.0) and np.isclose(m.geometries["metallic_factor"][1], 0.25)


def test_exr_zip_roundtrip(tmp_path):
    sky = scenes.sky(96, 40)  # 40 rows: ZIP blocks of 16 scanlines with a ragged last block
    for comp in ("zips", "zip"):
        p = tmp_path / f"{comp}.exr"
        assets.write_exr(p, sky, compression=comp)
        assert p.stat().st_size < sky.nbytes  # it really compresses
        assert np.array_equal(assets.read_exr(p), sky)


def test_processed_asset_cache_decodes_the_reference_fixture(tmp_path):
    """tests/golden/processed_box.glb.bin is the reference tree's own imported_assets/Default/box.glb (6253 B, data):
    bincode 2, big-endian, varint, OLD field order: 8 materials + 192 vertices of 8 unit cubes."""
    pm = assets.read_processed_mesh(ROOT / "tests" / "golden" / "processed_box.glb.bin", layout="old")
    assert len(pm.meshlets) == 0 and len(pm.indices) == 0 and pm.uploaded is False
    assert len(pm.materials) == 8 and pm.vertices.shape == (192, 8)
    cols = np.array([m.color for m in pm.materials])
    assert np.allclose(cols[0], 0.8, atol=1e-3) and np.allclose(cols[5], 0.5, atol=1e-3)
    assert np.allclose(cols[2], (0, 0, 0.8), atol=1e-3) and np.allclose(cols[3], (0.8, 0, 0.006), atol=1e-3) and np.allclose(cols[4], (0.009, 0.8, 0), atol=1e-3)
    assert [round(m.roughness_factor, 3) for m in pm.materials] == [1.0] * 7 + [0.5]
    assert all(m.texture_offset == -1 and m.metalic_factor == 0.0 for m in pm.materials)
    assert np.all(np.abs(pm.vertices[:, :3]) == 1.0)  # unit cubes: positions +-1
    assert np.allclose(np.linalg.norm(pm.vertices[:, 3:6], axis=1), 1.0)  # per-face normals
    with pytest.raises(ValueError):
        assets.read_processed_mesh(ROOT / "tests" / "golden" / "processed_box.glb.bin", layout="current")
    # current layout round trip (MeshSaver / MeshLoader, assets/mod.rs:151-168,299-314)
    pm2 = assets.ProcessedMesh(np.array([[0, 0, 64, 124], [64, 372, 3, 1]], np.uint32), pm.materials, pm.vertices, np.arange(300, dtype=np.uint8), True)
    q = tmp_path / "m.bin"
    assets.write_processed_mesh(q, pm2)
    back = assets.read_processed_mesh(q)
    assert np.array_equal(back.meshlets, pm2.meshlets) and np.array_equal(back.vertices, pm2.vertices) and np.array_equal(back.indices, pm2.indices)
    assert back.uploaded is True and [m.color for m in back.materials] == [m.color for m in pm2.materials]


def test_cornell_ref_is_built_from_the_reference_asset():
    """scenes.cornell_ref(): the eight cubes / materials of the reference's processed box.glb (decoded, not hard-coded), placed to
    match resources/refrence.png.  The camera fit is checked on the eight interior corners measured in that image (7 px RMS)."""
    import math

    from raytracer3_amd import assets, scenes

    pm = assets.read_processed_mesh(ROOT / "tests" / "golden" / "processed_box.glb.bin", "old")
    mesh = scenes.cornell_ref()
    assert len(mesh.geometries) == 8 and mesh.n_triangles == 8 * 12 and len(mesh.vertices) == 8 * 24
    for k, m in enumerate(pm.materials):  # cube k carries material k of the asset
        assert np.allclose(mesh.geometries["base_color"][k][:3], m.color) and mesh.geometries["roughness"][k] == np.float32(m.roughness_factor)
    lit = [n for n, e in zip(mesh.names, mesh.geometries["emission"]) if e[0] > 0]
    assert lit == ["light"] and sorted(mesh.names) == sorted(["back", "ceiling", "right", "left", "floor", "short", "tall", "light"])
    # project the interior corners of the [-1, 1]^3 box with the fitted camera and compare with their pixel positions in refrence.png
    measured = {(-1, 1, -1): (644, 185), (-1, -1, -1): (657, 803), (-1, -1, 1): (580, 941), (-1, 1, 1): (560, 114),
                (1, -1, 1): (1406, 929), (1, -1, -1): (1286, 800), (1, 1, -1): (1287, 162), (1, 1, 1): (1404, 92)}
    g = orc.camera_gconst(width=1920, height=1080, **scenes.CORNELL_REF_CAMERA)
    view, proj = np.array(g.view[:], np.float64).reshape(4, 4).T, np.array(g.proj[:], np.float64).reshape(4, 4).T
    err = []
    for p, (x, y) in measured.items():
        c = proj @ view @ np.array([*p, 1.0])
        err += [(c[0] / c[3] + 1) / 2 * 1920 - x, (1 - c[1] / c[3]) / 2 * 1080 - y]
    assert math.sqrt(np.mean(np.square(err))) < 9.0
    osc = orc.Scene(mesh)
    g = orc.camera_gconst(width=96, height=54, **scenes.CORNELL_REF_CAMERA)
    g.bounces, g.samples, g.blendfactor = 2, 8, 1.0
    gb, depth = osc.gbuffer(g, threads=2)
    light, _ = osc.reference_mode(g, gb, depth, threads=2)
    assert 0.2 < (depth < 1e5).mean() < 0.5 and light[..., :3].max() > 1.0 and light[0, 0, :3].sum() == 0  # box in the middle, black outside


def test_bench_self_launch_refuses_without_devices():
    """`python bench.py --gpus N` as the driver invokes it starts its ranks itself -- and must fail loudly, before starting any, when
    the devices are not there (this container has none): no JSON line, a non-zero exit code, a message.  (With devices the same command
    is covered by the GPU tests: the 3-rank rehearsal and the refusal of more ranks than GPUs.)"""
    import os
    import subprocess
    import sys

    import torch

    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is visible: covered by the GPU tests")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "no GPU visible" in r.stderr
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())
