#!/usr/bin/env python3
"""Time rt3_accel_build (GPU LBVH + SAH top on the device or on the host + four-wide quantised emit) for the bench scene and a larger one."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from raytracer3_amd import _lib as L  # noqa: E402
from raytracer3_amd import scenes  # noqa: E402
from raytracer3_amd.render_graph import Context  # noqa: E402

for detail in (1.0, 1.9):
    mesh = scenes.atrium(detail)
    ctx = Context(0)
    ctx.upload_mesh(mesh)
    for collapse, T, dev in ((2, 1, 1), (2, 2, 1), (2, 4, 1), (2, 0, 1), (1, 2, 1), (1, 2, 0), (1, 8, 1)):  # collapse 2 = cost-driven (default, T = 1)
        ctx.set_option(L.OPT_WIDE_COLLAPSE, collapse)
        ctx.set_option(L.OPT_SAH_TOP, T)
        ctx.set_option(L.OPT_SAH_TOP_DEVICE, dev)
        ts = []
        for _ in range(4):
            t0 = time.perf_counter()
            ctx.build_accel()
            ts.append(1e3 * (time.perf_counter() - t0))
        print(f"atrium({detail}): {mesh.n_triangles} triangles, collapse {collapse}, SAH_TOP {T} on the {'GPU' if dev else 'host'}: build {min(ts[1:]):.1f} ms (first {ts[0]:.1f}), {ctx.accel_info()[0]} nodes", flush=True)
    ctx.close()
