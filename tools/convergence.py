#!/usr/bin/env python3
"""BASELINE.json configs[4] as specified: progressive accumulation at 3840x2160, 64 spp per pass x 64 passes =
4096 spp with PrevLight <- Light and blendfactor = 1 / (pass + 1) (refrence_mode.slang:59-65), per-pixel RMSE of LINEAR radiance at
64 * 2^k spp against an INDEPENDENT render of 16 384 spp (different `frame` seeds, so the curve does not go to zero by construction),
and a restart from the accumulation buffer dumped after pass 32 that must reproduce the remaining passes bit for bit (SURVEY 5,
checkpoint / resume).  One GPU; the same frames tile over N ranks unchanged (the image does not depend on the rank count).

  python tools/convergence.py --out profiles/r02_convergence_4k
"""
import argparse
import json
import math
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="3840x2160")
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--passes", type=int, default=64)
    ap.add_argument("--ref-passes", type=int, default=256, help="passes of the independent reference render (x --spp samples)")
    ap.add_argument("--checkpoint-pass", type=int, default=32)
    ap.add_argument("--bounces", type=int, default=4)
    ap.add_argument("--out", default="profiles/r02_convergence_4k")
    ap.add_argument("--scratch", default="/tmp")
    args = ap.parse_args()

    from raytracer3_amd import assets, scenes
    from raytracer3_amd.renderer import DEFAULT_FLAGS, Camera, PathTracer

    W, H = (int(x) for x in args.size.split("x"))
    mesh, sky, bn = scenes.atrium(1.0), scenes.sky(2048, 1024), assets.load_bluenoise()
    cam = Camera(scenes.ATRIUM_CAMERA["position"], scenes.ATRIUM_CAMERA["direction"], math.radians(scenes.ATRIUM_CAMERA["fov_deg"]), W / H)

    def progressive(first_pass, last_pass, seed0, start_image=None, on_pass=None):
        """passes first_pass .. last_pass-1 of a running mean; frame seed = seed0 + pass"""
        pt = PathTracer((W, H))
        pt.set_scene(mesh, sky, bn)
        if start_image is not None:
            pt.load_prev(start_image)
        t0 = time.perf_counter()
        for p in range(first_pass, last_pass):
            g = pt.make_gconst(cam, args.spp, args.bounces, frame=seed0 + p, blendfactor=1.0 / (p + 1), flags=DEFAULT_FLAGS)
            pt.render(g, postprocess=False, wait=False)
            if on_pass:
                on_pass(p, pt)
            if p != last_pass - 1:
                pt.swap_light_prev()
            if (p + 1) % 32 == 0:
                pt.ctx.wait()
                print(f"  pass {p + 1}/{last_pass} ({time.perf_counter() - t0:.1f} s)", flush=True)
        pt.ctx.wait()
        dt = time.perf_counter() - t0
        img, st = pt.light(), pt.ctx.stats()
        pt.close()
        return img, dt, st

    print(f"reference: {args.ref_passes} x {args.spp} = {args.ref_passes * args.spp} spp, independent seeds", flush=True)
    ref, ref_dt, _ = progressive(0, args.ref_passes, 1_000_000)
    ref64 = ref[..., :3].astype(np.float64)
    checkpoints = [1 << k for k in range(0, 32) if (1 << k) <= args.passes]
    curve, ckpt_file = [], Path(args.scratch) / "rt3_accum_checkpoint.npy"

    def on_pass(p, pt):
        if p + 1 in checkpoints:
            img = pt.light()
            rm = float(np.sqrt(np.mean((img[..., :3].astype(np.float64) - ref64) ** 2)))
            curve.append({"spp": (p + 1) * args.spp, "passes": p + 1, "rmse_vs_independent_reference": rm})
            print(f"  {(p + 1) * args.spp:5d} spp: RMSE {rm:.5f}", flush=True)
        if p + 1 == args.checkpoint_pass:
            np.save(ckpt_file, pt.light())  # the accumulation buffer + the pass counter are the whole state

    print(f"progressive: {args.passes} x {args.spp} spp", flush=True)
    final, dt, st = progressive(0, args.passes, 0, on_pass=on_pass)
    rays = st.extension_rays + st.shadow_rays
    print(f"restart from the checkpoint of pass {args.checkpoint_pass}", flush=True)
    resumed, _, _ = progressive(args.checkpoint_pass, args.passes, 0, start_image=np.load(ckpt_file))
    ckpt_file.unlink()
    bit_identical = bool(np.array_equal(resumed.view(np.uint32), final.view(np.uint32)))
    # the noise floor of the comparison: the reference itself has RMSE_1spp / sqrt(16384)
    out = {
        "config": "BASELINE configs[4]: atrium stand-in, %dx%d, %d spp per pass x %d passes = %d spp, B = %d, full estimator" % (W, H, args.spp, args.passes, args.spp * args.passes, args.bounces),
        "reference": {"spp": args.ref_passes * args.spp, "seeds": "frame = 1000000 + pass (the progressive run uses frame = pass)", "seconds": round(ref_dt, 2)},
        "curve": curve,
        "slope_log2": [round(math.log2(curve[i + 1]["rmse_vs_independent_reference"] / curve[i]["rmse_vs_independent_reference"]), 3) for i in range(len(curve) - 1)],
        "progressive_seconds": round(dt, 2), "ms_per_pass": round(dt / args.passes * 1e3, 2), "rays": int(rays), "Mrays_per_s_incl_checkpoint_downloads": round(rays / dt / 1e6, 1),
        "restart_from_pass": args.checkpoint_pass, "restart_bit_identical": bit_identical,
        "mean_radiance": float(final[..., :3].mean()), "gpus": 1,
    }
    Path(args.out + ".json").write_text(json.dumps(out, indent=1))
    md = ["# Progressive accumulation at 4K (BASELINE.json configs[4]; tools/convergence.py)", "", out["config"] + ", one MI355X.", "",
          f"RMSE of linear radiance against an independent {out['reference']['spp']}-spp render ({out['reference']['seeds']}); a Monte-Carlo estimator",
          "halves its RMSE per 4x samples (slope -0.5 per doubling) until the reference's own noise (1 / sqrt(16384 / spp) of the curve's value) shows.", "",
          "| spp | RMSE vs independent reference | log2 ratio to previous |", "|---|---|---|"]
    for i, c in enumerate(curve):
        md.append(f"| {c['spp']} | {c['rmse_vs_independent_reference']:.5f} | {out['slope_log2'][i - 1] if i else ''} |")
    md += ["", f"* {args.passes} passes in {out['progressive_seconds']} s = {out['ms_per_pass']} ms per {args.spp}-spp pass ({out['Mrays_per_s_incl_checkpoint_downloads']} Mrays/s including the {len(curve)} checkpoint downloads of 133 MB).",
           f"* Restart: the accumulation buffer dumped after pass {args.checkpoint_pass} (PrevLight + the pass counter are the whole state), loaded into a fresh context, "
           f"reproduces the remaining passes **{'bit for bit' if bit_identical else 'NOT bit for bit'}** (`restart_bit_identical: {str(bit_identical).lower()}`).",
           f"* Reference render: {out['reference']['seconds']} s.  Mean radiance {out['mean_radiance']:.4f}."]
    Path(args.out + ".md").write_text("\n".join(md) + "\n")
    print(json.dumps(out)[:600])
    if not bit_identical:
        sys.exit(1)


if __name__ == "__main__":
    main()
