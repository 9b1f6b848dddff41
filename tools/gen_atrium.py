#!/usr/bin/env python3
"""Materialise the synthetic stand-ins under the file names the reference loads: the atrium as `sponza_scene.glb`
(main.rs:93) and the procedural sky as `skybox2.exr` (main.rs:94, HALF pixels, PIZ), so that the file-based pipeline
(`tools/render.py --glb .. --exr ..`, `raytracer3_amd/host/asset_tool render ..`) runs on the same scene `bench.py` builds in memory.

  python tools/gen_atrium.py --out assets [--detail 1.0] [--seed 0x5F0A2A] [--sky 2048x1024]
"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from raytracer3_amd import assets, scenes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="assets")
    ap.add_argument("--detail", type=float, default=1.0)
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=scenes.ATRIUM_SEED)
    ap.add_argument("--sky", default="2048x1024")
    ap.add_argument("--sky-compression", default="piz", choices=["none", "rle", "zips", "zip", "piz"])
    args = ap.parse_args()
    out = Path(args.out)
    out.mkdir(parents=True, exist_ok=True)
    mesh = scenes.atrium(args.detail, args.seed)
    assets.write_glb(out / "sponza_scene.glb", mesh)
    w, h = (int(x) for x in args.sky.split("x"))
    assets.write_exr(out / "skybox2.exr", scenes.sky(w, h), args.sky_compression, half=True)
    back = assets.GltfMeshLoader.load(out / "sponza_scene.glb")
    assert back.n_triangles == mesh.n_triangles
    print(f"{out / 'sponza_scene.glb'}: {mesh.n_triangles} triangles, {len(mesh.geometries)} geometries, "
          f"{(out / 'sponza_scene.glb').stat().st_size >> 20} MiB; {out / 'skybox2.exr'}: {w}x{h} HALF {args.sky_compression}, "
          f"{(out / 'skybox2.exr').stat().st_size >> 10} KiB")


if __name__ == "__main__":
    main()
