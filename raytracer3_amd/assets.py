"""Asset IO for the path tracer: binary glTF (.glb) meshes, scanline EXR skies, the blue-noise PNG.

Mirrors the *shape* of the reference loader API (`GltfMeshLoader` -> `MeshTransformer` -> `Mesh{vertices, indices,
materials}`, /root/reference/src/assets/mod.rs:118-133,179-286) without its limitations: every mesh x primitive x node
transform is loaded (the reference reads only the first primitive of the first mesh, assets/mod.rs:221), roughness is
read from the roughness factor (the reference decodes it from the metallic bytes, :88,110), and indices stay u32
(the reference's `indices: Vec<u8>` are meshlet-local and unusable for ray tracing).  Meshlet building and the bincode
cache are raster-/bevy-specific and out of scope.

`Vertex` layout == assets/mod.rs:127-133: p[3] n[3] t[2], 32 bytes, interleaved float32.
`GeometryInfo` layout == shaders/include/datatypes.slang:11-19 padded to 64 bytes (include/rt3.h: rt3_geometry_info).
"""
from __future__ import annotations

import json
import struct
from dataclasses import dataclass, field
from pathlib import Path

import numpy as np

GEOMETRY_DTYPE = np.dtype(
    [
        ("base_color", "<f4", (4,)),
        ("base_color_texture_index", "<i4"),
        ("metallic_factor", "<f4"),
        ("index_offset", "<u4"),
        ("vertex_offset", "<u4"),
        ("emission", "<f4", (4,)),
        ("roughness", "<f4"),
        ("_pad", "<u4", (3,)),
    ]
)
assert GEOMETRY_DTYPE.itemsize == 64


@dataclass
class Material:
    """assets/mod.rs:52-59 (f16 fields there; plain floats here) + emission (datatypes.slang:17)."""

    color: tuple = (0.8, 0.8, 0.8)
    metalic_factor: float = 0.0
    roughness_factor: float = 1.0
    emission: tuple = (0.0, 0.0, 0.0)
    texture_offset: int = -1


@dataclass
class Mesh:
    """Flattened scene: global vertex / index buffers + one GeometryInfo per primitive (world/mod.rs:103-125)."""

    vertices: np.ndarray  # (n, 8) float32  p n t
    indices: np.ndarray  # (m,) uint32, relative to each geometry's vertex_offset
    geometries: np.ndarray  # (g,) GEOMETRY_DTYPE
    prim_counts: np.ndarray  # (g,) uint32
    names: list = field(default_factory=list)
    textures: list = field(default_factory=list)  # base-colour textures: (h, w, 4) uint8, sRGB-encoded colour

    @property
    def n_triangles(self) -> int:
        return int(self.prim_counts.sum())

    def triangle_positions(self) -> np.ndarray:
        """(n_tris, 3, 3) world positions, in global primitive order."""
        out = []
        for g, cnt in zip(self.geometries, self.prim_counts):
            io, vo = int(g["index_offset"]), int(g["vertex_offset"])
            idx = self.indices[io : io + 3 * int(cnt)].astype(np.int64) + vo
            out.append(self.vertices[idx, :3].reshape(-1, 3, 3))
        return np.concatenate(out, axis=0) if out else np.zeros((0, 3, 3), np.float32)


class MeshBuilder:
    """Accumulates (positions, normals, uvs, triangles, material) parts into a `Mesh`."""

    def __init__(self):
        self.v, self.i, self.g, self.c, self.names = [], [], [], [], []
        self.nv = 0
        self.ni = 0

    def add(self, name, pos, nrm, uv, tris, mat: Material, normalize=True):
        pos = np.asarray(pos, np.float32).reshape(-1, 3)
        if normalize:
            nrm = np.asarray(nrm, np.float64).reshape(-1, 3)
            ln = np.linalg.norm(nrm, axis=1, keepdims=True)
            nrm = nrm / np.maximum(ln, 1e-20)
        nrm = np.asarray(nrm, np.float32).reshape(-1, 3)
        uv = np.zeros((len(pos), 2), np.float32) if uv is None else np.asarray(uv, np.float32).reshape(-1, 2)
        tris = np.asarray(tris, np.uint32).reshape(-1, 3)
        assert tris.size == 0 or int(tris.max()) < len(pos)
        g = np.zeros((), GEOMETRY_DTYPE)
        g["base_color"] = (*mat.color, 1.0)
        g["base_color_texture_index"] = mat.texture_offset
        g["metallic_factor"] = mat.metalic_factor
        g["roughness"] = mat.roughness_factor
        g["emission"] = (*mat.emission, 0.0)
        g["index_offset"] = self.ni
        g["vertex_offset"] = self.nv
        self.v.append(np.concatenate([pos, nrm, uv], axis=1))
        self.i.append(tris.reshape(-1))
        self.g.append(g)
        self.c.append(len(tris))
        self.names.append(name)
        self.nv += len(pos)
        self.ni += tris.size

    def build(self) -> Mesh:
        return Mesh(
            np.ascontiguousarray(np.concatenate(self.v, axis=0), np.float32),
            np.ascontiguousarray(np.concatenate(self.i), np.uint32),
            np.array(self.g, GEOMETRY_DTYPE),
            np.array(self.c, np.uint32),
            list(self.names),
        )


# ----------------------------------------------------------------------------------------------- glTF (.glb)
_COMP = {5120: ("i1", 1), 5121: ("u1", 1), 5122: ("<i2", 2), 5123: ("<u2", 2), 5125: ("<u4", 4), 5126: ("<f4", 4)}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


def write_glb(path, mesh: Mesh) -> None:
    """One glTF mesh+node per geometry, identity transforms, u32 indices, one material per geometry."""
    bin_parts, views, accessors, meshes, nodes, materials = [], [], [], [], [], []
    off = 0

    def push(arr, target=None):
        nonlocal off
        b = np.ascontiguousarray(arr).tobytes()
        pad = (-len(b)) % 4
        view = {"buffer": 0, "byteOffset": off, "byteLength": len(b)}
        if target:
            view["target"] = target
        views.append(view)
        bin_parts.append(b + b"\0" * pad)
        off += len(b) + pad
        return len(views) - 1

    for gi, (g, cnt) in enumerate(zip(mesh.geometries, mesh.prim_counts)):
        io, vo, cnt = int(g["index_offset"]), int(g["vertex_offset"]), int(cnt)
        idx = mesh.indices[io : io + 3 * cnt]
        later = [int(x) for x in mesh.geometries["vertex_offset"] if int(x) > vo]
        nv = (min(later) if later else len(mesh.vertices)) - vo  # keep unreferenced vertices: the round trip is exact
        v = mesh.vertices[vo : vo + nv]
        acc = []
        for col, typ in ((slice(0, 3), "VEC3"), (slice(3, 6), "VEC3"), (slice(6, 8), "VEC2")):
            data = np.ascontiguousarray(v[:, col], "<f4")
            a = {"bufferView": push(data, 34962), "componentType": 5126, "count": nv, "type": typ}
            if typ == "VEC3" and col.start == 0:
                a["min"] = [float(x) for x in data.min(axis=0)]
                a["max"] = [float(x) for x in data.max(axis=0)]
            accessors.append(a)
            acc.append(len(accessors) - 1)
        accessors.append({"bufferView": push(idx.astype("<u4"), 34963), "componentType": 5125, "count": 3 * cnt, "type": "SCALAR"})
        ia = len(accessors) - 1
        pbr = {
            "baseColorFactor": [float(x) for x in g["base_color"]],
            "metallicFactor": float(g["metallic_factor"]),
            "roughnessFactor": float(g["roughness"]),
        }
        if int(g["base_color_texture_index"]) > -1:
            pbr["baseColorTexture"] = {"index": int(g["base_color_texture_index"])}
        materials.append({"name": f"mat{gi}", "pbrMetallicRoughness": pbr, "emissiveFactor": [float(x) for x in g["emission"][:3]]})
        meshes.append({"name": mesh.names[gi] if gi < len(mesh.names) else f"g{gi}",
                       "primitives": [{"attributes": {"POSITION": acc[0], "NORMAL": acc[1], "TEXCOORD_0": acc[2]}, "indices": ia, "material": gi}]})
        nodes.append({"mesh": gi})
    images, textures = [], []
    for ti, tex in enumerate(mesh.textures):  # embedded PNG (lossless), one glTF texture per image
        import io

        from PIL import Image

        buf = io.BytesIO()
        Image.fromarray(np.ascontiguousarray(tex, np.uint8), "RGBA").save(buf, format="PNG")
        images.append({"bufferView": push(np.frombuffer(buf.getvalue(), np.uint8)), "mimeType": "image/png"})
        textures.append({"source": ti})
    doc = {
        "asset": {"version": "2.0", "generator": "raytracer3_amd.assets"},
        "scene": 0,
        "scenes": [{"nodes": list(range(len(nodes)))}],
        "nodes": nodes, "meshes": meshes, "materials": materials, "accessors": accessors, "bufferViews": views,
        "buffers": [{"byteLength": off}],
    }
    if images:
        doc["images"], doc["textures"] = images, textures
    js = json.dumps(doc, separators=(",", ":")).encode()
    js += b" " * ((-len(js)) % 4)
    blob = b"".join(bin_parts)
    total = 12 + 8 + len(js) + 8 + len(blob)
    with open(path, "wb") as f:
        f.write(struct.pack("<4sII", b"glTF", 2, total))
        f.write(struct.pack("<I4s", len(js), b"JSON"))
        f.write(js)
        f.write(struct.pack("<I4s", len(blob), b"BIN\0"))
        f.write(blob)


def _node_matrix(node) -> np.ndarray:
    if "matrix" in node:
        return np.array(node["matrix"], np.float64).reshape(4, 4).T  # glTF is column-major
    m = np.eye(4)
    if "scale" in node:
        m = np.diag([*node["scale"], 1.0]) @ m
    if "rotation" in node:
        x, y, z, w = node["rotation"]
        r = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 0],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w), 0],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y), 0],
                      [0, 0, 0, 1]], np.float64)
        m = r @ m
    if "translation" in node:
        t = np.eye(4)
        t[:3, 3] = node["translation"]
        m = t @ m
    return m


class GltfMeshLoader:
    """`.glb` -> `Mesh` (assets/mod.rs:179-200 + :207-252, all primitives, transforms baked at load)."""

    extensions = ("glb",)

    @staticmethod
    def load(path) -> Mesh:
        data = Path(path).read_bytes()
        magic, version, total = struct.unpack_from("<4sII", data, 0)
        if magic != b"glTF" or version != 2:
            raise ValueError(f"{path}: not a glTF 2 binary")
        p, doc, blob = 12, None, b""
        while p < total:
            ln, typ = struct.unpack_from("<I4s", data, p)
            chunk = data[p + 8 : p + 8 + ln]
            if typ == b"JSON":
                doc = json.loads(chunk.decode())
            elif typ == b"BIN\0":
                blob = chunk
            p += 8 + ln
        if doc is None:
            raise ValueError(f"{path}: no JSON chunk")

        def accessor(i):
            a = doc["accessors"][i]
            bv = doc["bufferViews"][a["bufferView"]]
            dt, sz = _COMP[a["componentType"]]
            nc = _NCOMP[a["type"]]
            start = bv.get("byteOffset", 0) + a.get("byteOffset", 0)
            stride = bv.get("byteStride", 0) or sz * nc
            cnt = a["count"]
            if stride == sz * nc:
                arr = np.frombuffer(blob, dt, cnt * nc, start).reshape(cnt, nc)
            else:
                arr = np.stack([np.frombuffer(blob, dt, nc, start + k * stride) for k in range(cnt)])
            if a.get("normalized") and a["componentType"] != 5126:
                arr = arr.astype(np.float32) / float(np.iinfo(np.dtype(dt)).max)
            return arr

        mb = MeshBuilder()
        mats = doc.get("materials", [])

        def visit(ni, parent):
            node = doc["nodes"][ni]
            m = parent @ _node_matrix(node)
            if "mesh" in node:
                gm = doc["meshes"][node["mesh"]]
                nm = np.linalg.inv(m[:3, :3]).T
                for pi, prim in enumerate(gm["primitives"]):
                    if prim.get("mode", 4) != 4:
                        continue
                    at = prim["attributes"]
                    identity = np.array_equal(m, np.eye(4))
                    pos = accessor(at["POSITION"]).astype(np.float64)
                    pos = (pos if identity else pos @ m[:3, :3].T + m[:3, 3]).astype(np.float32)
                    keep = False
                    if "NORMAL" in at:
                        nrm = accessor(at["NORMAL"])
                        keep = identity and nrm.dtype == np.float32  # stored unit normals pass through bit for bit
                        if not keep:
                            nrm = nrm.astype(np.float64) @ nm.T
                    else:
                        nrm = None
                    uv = accessor(at["TEXCOORD_0"]).astype(np.float32) if "TEXCOORD_0" in at else None
                    idx = accessor(prim["indices"]).reshape(-1).astype(np.uint32) if "indices" in prim else np.arange(len(pos), dtype=np.uint32)
                    tris = idx[: len(idx) // 3 * 3].reshape(-1, 3)
                    if nrm is None:  # flat normals from geometry
                        fn = np.cross(pos[tris[:, 1]] - pos[tris[:, 0]], pos[tris[:, 2]] - pos[tris[:, 0]])
                        nrm = np.zeros_like(pos, np.float64)
                        for k in range(3):
                            np.add.at(nrm, tris[:, k], fn)
                    mat = Material()
                    if "material" in prim:
                        gmtl = mats[prim["material"]]
                        pbr = gmtl.get("pbrMetallicRoughness", {})
                        bc = pbr.get("baseColorFactor", [1, 1, 1, 1])
                        em = np.array(gmtl.get("emissiveFactor", [0, 0, 0]), np.float64)
                        em = em * gmtl.get("extensions", {}).get("KHR_materials_emissive_strength", {}).get("emissiveStrength", 1.0)
                        tex = pbr.get("baseColorTexture", {}).get("index", -1)
                        mat = Material(tuple(bc[:3]), pbr.get("metallicFactor", 1.0), pbr.get("roughnessFactor", 1.0), tuple(em), tex)
                    mb.add(f"{gm.get('name', 'mesh')}.{pi}", pos, nrm, uv, tris, mat, normalize=not keep)
            for c in node.get("children", []):
                visit(c, m)

        scene = doc["scenes"][doc.get("scene", 0)]
        for root in scene["nodes"]:
            visit(root, np.eye(4))
        mesh = mb.build()
        # textures -> images (embedded PNG / JPEG, decoded by Pillow); Material.texture_offset indexes doc["textures"]
        for tex in doc.get("textures", []):
            img = doc["images"][tex["source"]]
            if "bufferView" in img:
                bv = doc["bufferViews"][img["bufferView"]]
                raw = blob[bv.get("byteOffset", 0) : bv.get("byteOffset", 0) + bv["byteLength"]]
            else:
                raw = (Path(path).parent / img["uri"]).read_bytes()
            import io

            from PIL import Image

            mesh.textures.append(np.ascontiguousarray(np.array(Image.open(io.BytesIO(raw)).convert("RGBA"), np.uint8)))
        return mesh


# ----------------------------------------------------------------------------------------------- EXR (scanline, no compression)
def _exr_zip_pack(raw: bytes) -> bytes:
    """OpenEXR ZIP/ZIPS block: de-interleave even/odd bytes, delta-predict, deflate."""
    import zlib

    a = np.frombuffer(raw, np.uint8)
    t = np.concatenate([a[0::2], a[1::2]]).astype(np.int16)
    d = t.copy()
    d[1:] = (t[1:] - t[:-1] + 128 + 256) % 256
    return zlib.compress(d.astype(np.uint8).tobytes(), 6)


def _exr_zip_unpack(comp: bytes, size: int) -> bytes:
    import zlib

    d = np.frombuffer(zlib.decompress(comp), np.uint8)
    if len(d) != size:
        raise ValueError("EXR ZIP block has the wrong size")
    return _exr_predictor_undo(d, size)

# ---- OpenEXR RLE and PIZ codecs, restated from the published OpenEXR algorithm (ImfRle / ImfPizCompressor / ImfWav / ImfHuf).
# Parity unpinned: the reference tree ships no EXR (its `image::open("./assets/skybox2.exr")` is commented out, main.rs:94) and
# this image has no other EXR implementation to compare with; the decoders are checked against the encoders below and against the
# C++ decoder of host/assets.hpp (tests/test_host_assets.py).
def _exr_predictor_undo(d: np.ndarray, size: int) -> bytes:
    """Shared tail of ZIP and RLE blocks: undo the byte delta predictor, then re-interleave the two halves."""
    d = d.astype(np.int64)
    d[1:] -= 128
    t = (np.cumsum(d) % 256).astype(np.uint8)
    half = (size + 1) // 2
    out = np.empty(size, np.uint8)
    out[0::2] = t[:half]
    out[1::2] = t[half:]
    return out.tobytes()


def _exr_predictor_apply(raw: bytes) -> np.ndarray:
    a = np.frombuffer(raw, np.uint8)
    t = np.concatenate([a[0::2], a[1::2]]).astype(np.int64)
    d = t.copy()
    d[1:] = (t[1:] - t[:-1] + 128) % 256
    return d.astype(np.uint8)


def _exr_rle_unpack(comp: bytes, size: int) -> bytes:
    out, p = bytearray(), 0
    while p < len(comp):
        c = comp[p] - 256 if comp[p] > 127 else comp[p]
        p += 1
        if c < 0:  # -c literal bytes
            out += comp[p : p - c]
            p += -c
        else:  # the next byte c + 1 times
            out += bytes([comp[p]]) * (c + 1)
            p += 1
    if len(out) != size:
        raise ValueError("EXR RLE block has the wrong size")
    return _exr_predictor_undo(np.frombuffer(bytes(out), np.uint8), size)


def _exr_rle_pack(raw: bytes) -> bytes:
    d = _exr_predictor_apply(raw).tobytes()
    out, p, n = bytearray(), 0, len(d)
    while p < n:
        r = 1
        while p + r < n and d[p + r] == d[p] and r < 128:
            r += 1
        if r >= 3:
            out += bytes([r - 1, d[p]])
            p += r
        else:
            q = p
            while q < n and q - p < 127 and not (q + 2 < n and d[q] == d[q + 1] == d[q + 2]):
                q += 1
            out += bytes([(256 - (q - p)) & 0xFF]) + d[p:q]
            p = q
    return bytes(out)


_HUF_ENCSIZE, _HUF_DECBITS = (1 << 16) + 1, 14
_SHORT_ZERO_RUN, _LONG_ZERO_RUN = 59, 63
_SHORTEST_LONG_RUN = 2 + _LONG_ZERO_RUN - _SHORT_ZERO_RUN  # 6
_LONGEST_LONG_RUN = 255 + _SHORTEST_LONG_RUN


def _huf_canonical(lengths):
    """ImfHuf hufCanonicalCodeTable: code lengths -> canonical codes (longer codes get the numerically smaller values)."""
    n = [0] * 59
    for l in lengths:
        n[l] += 1
    c = 0
    for i in range(58, 0, -1):
        nc = (c + n[i]) >> 1
        n[i] = c
        c = nc
    codes = [0] * len(lengths)
    for i, l in enumerate(lengths):
        if l:
            codes[i] = n[l]
            n[l] += 1
    return codes


class _Bits:
    def __init__(self, data=b"", p=0):
        self.d, self.p, self.c, self.lc, self.out = data, p, 0, 0, bytearray()

    def get(self, n):
        while self.lc < n:
            self.c = ((self.c << 8) | (self.d[self.p] if self.p < len(self.d) else 0)) & 0xFFFFFFFFFFFFFFFF
            self.p += 1
            self.lc += 8
        self.lc -= n
        return (self.c >> self.lc) & ((1 << n) - 1)

    def put(self, n, v):
        self.c = (self.c << n) | v
        self.lc += n
        while self.lc >= 8:
            self.lc -= 8
            self.out.append((self.c >> self.lc) & 0xFF)
        self.c &= (1 << self.lc) - 1

    def flush(self):
        if self.lc:
            self.out.append((self.c << (8 - self.lc)) & 0xFF)
        return bytes(self.out)


def _huf_uncompress(comp: bytes, n_raw: int) -> np.ndarray:
    if n_raw == 0:
        return np.zeros(0, np.uint16)
    im, i_max, _table_len, n_bits = struct.unpack_from("<IIII", comp, 0)
    if im >= _HUF_ENCSIZE or i_max >= _HUF_ENCSIZE:
        raise ValueError("EXR PIZ: bad Huffman header")
    br = _Bits(comp, 20)
    lengths = [0] * _HUF_ENCSIZE
    s = im
    while s <= i_max:  # hufUnpackEncTable
        l = br.get(6)
        if l == _LONG_ZERO_RUN:
            s += br.get(8) + _SHORTEST_LONG_RUN
        elif l >= _SHORT_ZERO_RUN:
            s += l - _SHORT_ZERO_RUN + 2
        else:
            lengths[s] = l
            s += 1
    codes = _huf_canonical(lengths)
    data_start = 20 + _table_len
    tab_len, tab_sym, long_codes = [0] * (1 << _HUF_DECBITS), [0] * (1 << _HUF_DECBITS), {}
    for sym in range(im, i_max + 1):  # hufBuildDecTable
        l = lengths[sym]
        if not l:
            continue
        if l > _HUF_DECBITS:
            long_codes.setdefault(codes[sym] >> (l - _HUF_DECBITS), []).append((l, codes[sym], sym))
        else:
            base = codes[sym] << (_HUF_DECBITS - l)
            for k in range(base, base + (1 << (_HUF_DECBITS - l))):
                tab_len[k], tab_sym[k] = l, sym
    out = np.zeros(n_raw, np.uint16)
    d, p, end = comp, data_start, min(len(comp), data_start + (n_bits + 7) // 8)
    c, lc, produced, rlc = 0, 0, 0, i_max
    while produced < n_raw:  # hufDecode; bits past the end read as zero
        while lc < 58 and p < end:
            c = (c << 8) | d[p]
            p += 1
            lc += 8
        while lc < _HUF_DECBITS:  # past the end of the data: zero bits, like hufDecode's final shift
            c <<= 8
            p += 1
            lc += 8
            if p > end + 16:
                raise ValueError("EXR PIZ: Huffman data ends early")
        idx = (c >> (lc - _HUF_DECBITS)) & 0x3FFF
        l = tab_len[idx]
        if l:
            sym = tab_sym[idx]
        else:
            for l, code, sym in long_codes.get(idx, ()):
                while lc < l:
                    c = (c << 8) | (d[p] if p < end else 0)
                    p += 1
                    lc += 8
                if (c >> (lc - l)) & ((1 << l) - 1) == code:
                    break
            else:
                raise ValueError("EXR PIZ: invalid Huffman code")
        lc -= l
        c &= (1 << lc) - 1
        if sym == rlc:  # run: repeat the previous value
            while lc < 8:
                c = (c << 8) | (d[p] if p < end else 0)
                p += 1
                lc += 8
            lc -= 8
            run = (c >> lc) & 0xFF
            c &= (1 << lc) - 1
            if produced == 0 or produced + run > n_raw:
                raise ValueError("EXR PIZ: bad run length")
            out[produced : produced + run] = out[produced - 1]
            produced += run
        else:
            out[produced] = sym
            produced += 1
    return out


def _huf_compress(raw: np.ndarray) -> bytes:
    """Test-side encoder (hufCompress): Huffman codes of at most 58 bits, run-length pseudo symbol iM."""
    import heapq

    if len(raw) == 0:
        return b""
    freq = np.bincount(raw, minlength=_HUF_ENCSIZE).tolist()
    used = [i for i, f in enumerate(freq) if f]
    im, i_max = used[0], used[-1] + 1
    freq[i_max] = 1  # the run-length symbol
    used.append(i_max)
    lengths = [0] * _HUF_ENCSIZE
    heap = [(freq[i], i, (i,)) for i in used]
    heapq.heapify(heap)
    if len(heap) == 1:
        lengths[heap[0][1]] = 1
    while len(heap) > 1:
        fa, ka, sa = heapq.heappop(heap)
        fb, kb, sb = heapq.heappop(heap)
        for i in sa + sb:
            lengths[i] += 1
        heapq.heappush(heap, (fa + fb, min(ka, kb), sa + sb))
    if max(lengths) > 58:
        raise ValueError("Huffman code longer than 58 bits")
    codes = _huf_canonical(lengths)
    bw = _Bits()
    s = im
    while s <= i_max:  # hufPackEncTable
        l = lengths[s]
        if l == 0:
            z = 1
            while s + z <= i_max and lengths[s + z] == 0 and z < _LONGEST_LONG_RUN:
                z += 1
            if z >= 2:
                if z >= _SHORTEST_LONG_RUN:
                    bw.put(6, _LONG_ZERO_RUN)
                    bw.put(8, z - _SHORTEST_LONG_RUN)
                else:
                    bw.put(6, _SHORT_ZERO_RUN + z - 2)
                s += z
                continue
        bw.put(6, l)
        s += 1
    table = bw.flush()
    bw = _Bits()
    n_bits = 0

    def send(sym, run):  # sendCode
        nonlocal n_bits
        ls, lr = lengths[sym], lengths[i_max]
        if ls + lr + 8 < ls * run:
            bw.put(ls, codes[sym]); bw.put(lr, codes[i_max]); bw.put(8, run)
            n_bits += ls + lr + 8
        else:
            for _ in range(run + 1):
                bw.put(ls, codes[sym])
            n_bits += ls * (run + 1)

    vals = raw.tolist()
    cur, run = vals[0], 0
    for v in vals[1:]:
        if v == cur and run < 255:
            run += 1
        else:
            send(cur, run)
            run = 0
        cur = v
    send(cur, run)
    data = bw.flush()
    return struct.pack("<IIIII", im, i_max, len(table), n_bits, 0) + table + data


def _wav_levels(nx, ny):
    n, p = min(nx, ny), 1
    while p <= n:
        p <<= 1
    p >>= 1
    return p  # largest power of two <= min(nx, ny) (0 if that is 0)


def _wav_pairs(plane, a_idx, b_idx, w14, decode):
    """One butterfly over the element pairs (a, b) given as index tuples into `plane` (uint16); returns (a', b')."""
    a, b = plane[a_idx].astype(np.int64), plane[b_idx].astype(np.int64)
    if decode:
        if w14:  # wdec14(l = a, h = b)
            ls, hs = ((a + 0x8000) & 0xFFFF) - 0x8000, ((b + 0x8000) & 0xFFFF) - 0x8000
            ai = ls + (hs & 1) + (hs >> 1)
            return ai & 0xFFFF, (ai - hs) & 0xFFFF
        bb = (a - (b >> 1)) & 0xFFFF  # wdec16
        return (b + bb - 0x8000) & 0xFFFF, bb
    if w14:  # wenc14
        sa, sb = ((a + 0x8000) & 0xFFFF) - 0x8000, ((b + 0x8000) & 0xFFFF) - 0x8000
        return ((sa + sb) >> 1) & 0xFFFF, (sa - sb) & 0xFFFF
    ao = (a + 0x8000) & 0xFFFF  # wenc16
    m, dd = (ao + b) >> 1, ao - b
    m = np.where(dd < 0, (m + 0x8000) & 0xFFFF, m)
    return m, dd & 0xFFFF


def _wav2(plane: np.ndarray, mx: int, decode: bool) -> None:
    """ImfWav wav2Encode / wav2Decode on one (ny, nx) uint16 plane, in place; every level is vectorised over its 2x2 cells."""
    ny, nx = plane.shape
    w14 = mx < (1 << 14)
    top = _wav_levels(nx, ny)
    levels = []
    p = 1
    while 2 * p <= top:
        levels.append(p)
        p *= 2
    for p in (reversed(levels) if decode else levels):
        p2 = 2 * p
        ys, xs = np.arange(0, ny - p2 + 1, p2), np.arange(0, nx - p2 + 1, p2)
        yl, xl = (ys[-1] + p2 if len(ys) else 0), (xs[-1] + p2 if len(xs) else 0)
        Y, X = np.meshgrid(ys, xs, indexing="ij")
        i00, i01, i10, i11 = (Y, X), (Y, X + p), (Y + p, X), (Y + p, X + p)
        if decode:
            a, c = _wav_pairs(plane, i00, i10, w14, True)  # wdec(*px, *p10 -> i00, i10)
            b, d = _wav_pairs(plane, i01, i11, w14, True)
            tmp = plane.copy()
            tmp[i00], tmp[i10], tmp[i01], tmp[i11] = a, c, b, d
            r00, r01 = _wav_pairs(tmp, i00, i01, w14, True)
            r10, r11 = _wav_pairs(tmp, i10, i11, w14, True)
            plane[i00], plane[i01], plane[i10], plane[i11] = r00, r01, r10, r11
        else:
            a, b = _wav_pairs(plane, i00, i01, w14, False)  # wenc(*px, *p01 -> i00, i01)
            c, d = _wav_pairs(plane, i10, i11, w14, False)
            tmp = plane.copy()
            tmp[i00], tmp[i01], tmp[i10], tmp[i11] = a, b, c, d
            r00, r10 = _wav_pairs(tmp, i00, i10, w14, False)
            r01, r11 = _wav_pairs(tmp, i01, i11, w14, False)
            plane[i00], plane[i10], plane[i01], plane[i11] = r00, r10, r01, r11
        if nx & p:  # odd column: 1-D step along y
            ia, ib = (ys, np.full_like(ys, xl)), (ys + p, np.full_like(ys, xl))
            ra, rb = _wav_pairs(plane, ia, ib, w14, decode)
            plane[ia], plane[ib] = ra, rb
        if ny & p:  # odd row: 1-D step along x
            ia, ib = (np.full_like(xs, yl), xs), (np.full_like(xs, yl), xs + p)
            ra, rb = _wav_pairs(plane, ia, ib, w14, decode)
            plane[ia], plane[ib] = ra, rb


def _piz_unpack(comp: bytes, chans, w: int, n_lines: int) -> bytes:
    """PIZ block -> the uncompressed scanline block (per line: each channel's row).  chans = [(name, pixel_type)]."""
    sizes = [1 if pt == 1 else 2 for _, pt in chans]
    total = sum(sizes) * w * n_lines
    lo, hi = struct.unpack_from("<HH", comp, 0)
    bitmap = np.zeros(8192, np.uint8)
    p = 4
    if lo <= hi:
        bitmap[lo : hi + 1] = np.frombuffer(comp, np.uint8, hi - lo + 1, p)
        p += hi - lo + 1
    (length,) = struct.unpack_from("<i", comp, p)
    p += 4
    present = np.unpackbits(bitmap, bitorder="little").astype(bool)
    present[0] = True
    lut = np.zeros(65536, np.uint16)
    vals = np.nonzero(present)[0]
    lut[: len(vals)] = vals
    mx = len(vals) - 1
    buf = _huf_uncompress(comp[p : p + length], total)
    o = 0
    planes = []
    for sz in sizes:
        region = buf[o : o + sz * w * n_lines].reshape(n_lines, w, sz)
        for j in range(sz):
            plane = np.ascontiguousarray(region[:, :, j])
            _wav2(plane, mx, True)
            region[:, :, j] = plane
        planes.append(region)
        o += sz * w * n_lines
    out = bytearray()
    for y in range(n_lines):
        for region in planes:
            out += lut[region[y].reshape(-1)].astype("<u2").tobytes()
    return bytes(out)


def _piz_pack(raw: bytes, chans, w: int, n_lines: int) -> bytes:
    sizes = [1 if pt == 1 else 2 for _, pt in chans]
    src = np.frombuffer(raw, "<u2")
    regions = [np.zeros((n_lines, w, sz), np.uint16) for sz in sizes]
    o = 0
    for y in range(n_lines):
        for region, sz in zip(regions, sizes):
            region[y] = src[o : o + sz * w].reshape(w, sz)
            o += sz * w
    allv = np.concatenate([r.reshape(-1) for r in regions])
    present = np.zeros(65536, bool)
    present[allv] = True
    present[0] = False  # zero is implicit
    bitmap = np.packbits(present, bitorder="little")
    nz = np.nonzero(bitmap)[0]
    lo, hi = (int(nz[0]), int(nz[-1])) if len(nz) else (8191, 0)
    present[0] = True
    fwd = np.zeros(65536, np.uint16)
    vals = np.nonzero(present)[0]
    fwd[vals] = np.arange(len(vals), dtype=np.uint16)
    mx = len(vals) - 1
    parts = []
    for region, sz in zip(regions, sizes):
        region[...] = fwd[region]
        for j in range(sz):
            plane = np.ascontiguousarray(region[:, :, j])
            _wav2(plane, mx, False)
            region[:, :, j] = plane
        parts.append(region.reshape(-1))
    huf = _huf_compress(np.concatenate(parts))
    return struct.pack("<HH", lo, hi) + (bitmap[lo : hi + 1].tobytes() if lo <= hi else b"") + struct.pack("<i", len(huf)) + huf



def write_exr(path, rgb: np.ndarray, compression: str = "none", half: bool = False) -> None:
    """RGB float32 image -> scanline OpenEXR (channels B,G,R as FLOAT, or HALF with half=True);
    compression "none" | "rle" | "zips" | "zip" | "piz"."""
    rgb = np.asarray(rgb, np.float32)
    h, w, _ = rgb.shape
    if compression != "none" or half:
        return _write_exr_blocks(path, rgb, {"none": 0, "rle": 1, "zips": 2, "zip": 3, "piz": 4}[compression], half)

    def attr(name, typ, payload):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<I", len(payload)) + payload

    chl = b"".join(n.encode() + b"\0" + struct.pack("<iBBBBii", 2, 0, 0, 0, 0, 1, 1) for n in ("B", "G", "R")) + b"\0"
    box = struct.pack("<iiii", 0, 0, w - 1, h - 1)
    hdr = (attr("channels", "chlist", chl) + attr("compression", "compression", b"\0") + attr("dataWindow", "box2i", box)
           + attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", b"\0") + attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
           + attr("screenWindowCenter", "v2f", struct.pack("<ff", 0.0, 0.0)) + attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0")
    head = struct.pack("<II", 20000630, 2) + hdr
    line = 8 + 12 * w
    table0 = len(head) + 8 * h
    with open(path, "wb") as f:
        f.write(head)
        f.write(np.arange(h, dtype="<u8").__mul__(line).__add__(table0).tobytes())
        for y in range(h):
            f.write(struct.pack("<iI", y, 12 * w))
            f.write(np.ascontiguousarray(rgb[y, :, ::-1].T, "<f4").tobytes())


def _exr_header(w, h, comp_code, pixel_type=2):
    def attr(name, typ, payload):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<I", len(payload)) + payload

    chl = b"".join(n.encode() + b"\0" + struct.pack("<iBBBBii", pixel_type, 0, 0, 0, 0, 1, 1) for n in ("B", "G", "R")) + b"\0"
    box = struct.pack("<iiii", 0, 0, w - 1, h - 1)
    hdr = (attr("channels", "chlist", chl) + attr("compression", "compression", bytes([comp_code])) + attr("dataWindow", "box2i", box)
           + attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", b"\0") + attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
           + attr("screenWindowCenter", "v2f", struct.pack("<ff", 0.0, 0.0)) + attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0")
    return struct.pack("<II", 20000630, 2) + hdr


def _write_exr_blocks(path, rgb, comp_code, half):
    h, w, _ = rgb.shape
    lines_per_block = {0: 1, 1: 1, 2: 1, 3: 16, 4: 32}[comp_code]
    pt, dt = (1, "<f2") if half else (2, "<f4")
    chans = [("B", pt), ("G", pt), ("R", pt)]
    head = _exr_header(w, h, comp_code, pt)
    blocks = []
    for y0 in range(0, h, lines_per_block):
        rows = rgb[y0 : y0 + lines_per_block]
        raw = b"".join(np.ascontiguousarray(r[:, ::-1].T.astype(dt)).tobytes() for r in rows)  # per scanline: B row, G row, R row
        if comp_code == 0:
            comp = raw
        elif comp_code == 1:
            comp = _exr_rle_pack(raw)
        elif comp_code == 4:
            comp = _piz_pack(raw, chans, w, len(rows))
        else:
            comp = _exr_zip_pack(raw)
        if len(comp) >= len(raw):
            comp = raw  # the format stores a block raw when compression does not help
        blocks.append(struct.pack("<iI", y0, len(comp)) + comp)
    table0 = len(head) + 8 * len(blocks)
    offs, o = [], table0
    for b in blocks:
        offs.append(o)
        o += len(b)
    with open(path, "wb") as f:
        f.write(head)
        f.write(np.array(offs, "<u8").tobytes())
        for b in blocks:
            f.write(b)


def read_exr(path) -> np.ndarray:
    """Scanline EXR (uncompressed, RLE, ZIPS, ZIP or PIZ) with HALF / FLOAT / UINT channels -> (h, w, 3) float32 RGB (missing
    channels = 0).  The `image` crate the reference would open its skybox with (main.rs:94) reads the same set plus PXR24 / B44."""
    data = Path(path).read_bytes()
    magic, ver = struct.unpack_from("<II", data, 0)
    if magic != 20000630:
        raise ValueError(f"{path}: not an OpenEXR file")
    if ver & 0x200:
        raise ValueError(f"{path}: tiled EXR is not supported")
    p, attrs = 8, {}
    while data[p] != 0:
        e = data.index(b"\0", p); name = data[p:e].decode(); p = e + 1
        e = data.index(b"\0", p); typ = data[p:e].decode(); p = e + 1
        (ln,) = struct.unpack_from("<I", data, p); p += 4
        attrs[name] = (typ, data[p : p + ln]); p += ln
    p += 1
    comp = attrs["compression"][1][0]
    if comp not in (0, 1, 2, 3, 4):
        raise ValueError(f"{path}: EXR compression {comp} is not supported (uncompressed, RLE, ZIPS, ZIP and PIZ are; PXR24 / B44 / DWA are not)")
    lines_per_block = {0: 1, 1: 1, 2: 1, 3: 16, 4: 32}[comp]
    chans, q, cl = [], 0, attrs["channels"][1]
    while cl[q] != 0:
        e = cl.index(b"\0", q); nm = cl[q:e].decode(); q = e + 1
        (pt,) = struct.unpack_from("<i", cl, q); q += 16
        chans.append((nm, pt))
    x0, y0, x1, y1 = struct.unpack("<iiii", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    n_blocks = (h + lines_per_block - 1) // lines_per_block
    offs = np.frombuffer(data, "<u8", n_blocks, p)
    out = np.zeros((h, w, 3), np.float32)
    line_bytes = sum({1: 2, 2: 4}.get(pt, 4) for _, pt in chans) * w
    for o in offs:
        y, size = struct.unpack_from("<iI", data, int(o))
        n_lines = min(lines_per_block, y1 + 1 - y)
        block = data[int(o) + 8 : int(o) + 8 + size]
        if comp and size < line_bytes * n_lines:  # a block that does not shrink is stored raw
            if comp == 1:
                block = _exr_rle_unpack(block, line_bytes * n_lines)
            elif comp == 4:
                block = _piz_unpack(block, chans, w, n_lines)
            else:
                block = _exr_zip_unpack(block, line_bytes * n_lines)
        q = 0
        for ly in range(n_lines):
            for nm, pt in chans:
                if pt == 2:
                    row = np.frombuffer(block, "<f4", w, q); q += 4 * w
                elif pt == 1:
                    row = np.frombuffer(block, "<f2", w, q).astype(np.float32); q += 2 * w
                else:
                    row = np.frombuffer(block, "<u4", w, q).astype(np.float32); q += 4 * w
                if nm in "RGB":
                    out[y + ly - y0, :, "RGB".index(nm)] = row
    return out


# ----------------------------------------------------------------------------------------------- bincode processed-asset cache
class _Bincode:
    """bincode 2 `standard().with_variable_int_encoding().with_big_endian()` (assets/mod.rs:135-137)."""

    def __init__(self, data: bytes):
        self.d, self.p = data, 0

    def take(self, n):
        b = self.d[self.p : self.p + n]
        if len(b) != n:
            raise ValueError("bincode: unexpected end of data")
        self.p += n
        return b

    def varint(self):
        b = self.take(1)[0]
        if b < 251:
            return b
        return int.from_bytes(self.take({251: 2, 252: 4, 253: 8, 254: 16}[b]), "big")

    def f16(self):
        return float(np.frombuffer(self.take(2), ">f2")[0])


@dataclass
class ProcessedMesh:
    """`Mesh` as serialised by `MeshSaver` (assets/mod.rs:118-125,299-314)."""

    meshlets: np.ndarray  # (n, 4) uint32: vertex_offset, triangle_offset, vertex_count, triangle_count
    materials: list  # Material
    vertices: np.ndarray  # (n, 8) float32
    indices: np.ndarray  # uint8, meshlet-local
    uploaded: bool


def read_processed_mesh(path, layout: str = "current") -> ProcessedMesh:
    """Decode a file of bevy's processed-asset cache (`imported_assets/Default/*.glb`).
    layout "current" = field order of assets/mod.rs:118-125 (meshlets, materials, vertices, indices, uploaded);
    layout "old"     = the order of the file the reference tree still ships: two leading vectors (both empty
                       there: meshlets and meshlet-local indices), materials, vertices, uploaded.
    Material = f16 BE metallic, f16 BE roughness, 3 x f16 BE colour, varint u16 texture offset (assets/mod.rs:61-73); the
    reference's decoder re-reads the metallic bytes as roughness (:88,110) -- that bug is not reproduced."""
    r = _Bincode(Path(path).read_bytes())

    def meshlets():
        n = r.varint()
        return np.array([[r.varint() for _ in range(4)] for _ in range(n)], np.uint32).reshape(n, 4)

    def materials():
        out = []
        for _ in range(r.varint()):
            metal, rough = r.f16(), r.f16()
            col = (r.f16(), r.f16(), r.f16())
            tex = r.varint()
            out.append(Material(col, metal, rough, (0.0, 0.0, 0.0), -1 if tex == 0xFFFF else tex))
        return out

    def vertices():
        n = r.varint()
        return np.frombuffer(r.take(32 * n), ">f4").reshape(n, 8).astype(np.float32)

    def bytes_vec():
        n = r.varint()
        return np.frombuffer(r.take(n), np.uint8).copy()

    if layout == "current":
        ml, mats, verts, idx = meshlets(), materials(), vertices(), bytes_vec()
    elif layout == "old":
        ml, idx = meshlets(), bytes_vec()
        mats, verts = materials(), vertices()
    else:
        raise ValueError("layout must be 'current' or 'old'")
    uploaded = bool(r.take(1)[0])
    if r.p != len(r.d):
        raise ValueError(f"bincode: {len(r.d) - r.p} trailing bytes (wrong layout?)")
    return ProcessedMesh(ml, mats, verts, idx, uploaded)


def write_processed_mesh(path, pm: ProcessedMesh) -> None:
    """Serialise in the CURRENT layout (MeshSaver, assets/mod.rs:299-314)."""

    def varint(v):
        if v < 251:
            return bytes([v])
        for tag, n in ((251, 2), (252, 4), (253, 8)):
            if v < (1 << (8 * n)):
                return bytes([tag]) + int(v).to_bytes(n, "big")
        raise ValueError("varint too large")

    out = bytearray(varint(len(pm.meshlets)))
    for m in pm.meshlets:
        for v in m:
            out += varint(int(v))
    out += varint(len(pm.materials))
    for m in pm.materials:
        out += np.array([m.metalic_factor, m.roughness_factor, *m.color], ">f2").tobytes()
        out += varint(0xFFFF if m.texture_offset < 0 else m.texture_offset)
    out += varint(len(pm.vertices)) + np.ascontiguousarray(pm.vertices, ">f4").tobytes()
    out += varint(len(pm.indices)) + bytes(np.asarray(pm.indices, np.uint8))
    out += bytes([1 if pm.uploaded else 0])
    Path(path).write_bytes(bytes(out))


def load_bluenoise(path=None) -> np.ndarray:
    """resources/bluenoise.png (256x256 RGBA8, shipped with the reference as data) -> (256,256,4) uint8."""
    from PIL import Image

    path = Path(path) if path else Path(__file__).resolve().parent.parent / "resources" / "bluenoise.png"
    return np.ascontiguousarray(np.array(Image.open(path).convert("RGBA"), np.uint8))
