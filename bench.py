#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json: "Mrays/s + ms/frame, Sponza 1080p@64spp").

A step = one frame: gbuffer pass + refrence_mode pass (B = 4, full estimator: diffuse BSDF + sky NEE/MIS +
blue-noise shift) over this rank's 64x64 tiles, then ONE gather of the per-rank tile buffers to rank 0.
Workload = configs[2] (C3): atrium stand-in (sponza_scene.glb is not shipped), 1920x1080 @ 64 spp.
value = Mrays/s = (primary + bounce + shadow rays of all ranks) / max-over-ranks frame time; ms_per_step = ms/frame.
Strong scaling: the frame is fixed, tiles are split over the ranks.

Extra objects on the JSON line:
  roofline     the dominant kernel -- k_extend (k_trace if --fused-trace 1: extension queue then shadow queue in one
               launch): algorithmic bytes (48 + 64 n_nodes + 48 n_tris per ray) / HIP-event time of its
               launches inside the timed region, against the 8 TB/s HBM3E peak
  cpu_baseline the CPU oracle (a port; the reference cannot be built here) path tracing a centred crop of the same
               frame on the host cores; the crop also yields rmse_vs_oracle
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import math
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
DEFAULT_SAH_TOP = 2  # the library's RT3_OPT_SAH_TOP default (the oracle of the parity leg is built the same way)
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def usable_cpus() -> int:
    """Host cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (a container on a 256-thread
    host may be limited to far fewer; threads beyond the quota are only throttled)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()[:2]  # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read_text())
            p = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
            if q > 0 and p > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return max(1, min(n, 256))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--bounces", type=int, default=4)
    ap.add_argument("--detail", type=float, default=1.0)
    ap.add_argument("--batch-spp", type=int, default=0)
    ap.add_argument("--leaf-size", type=int, default=0, help="triangles per BVH leaf (0 = library default)")
    ap.add_argument("--node-width", type=int, default=0, help="2 | 4 (0 = library default)")
    ap.add_argument("--node-quant", type=int, default=-1, help="0 | 1 (-1 = library default)")
    ap.add_argument("--refill", type=int, default=-1, help="traversal tuning: idle lanes before a wave refills (RT3_OPT_EXTEND_VARIANT)")
    ap.add_argument("--fused-trace", type=int, default=-1, help="RT3_OPT_FUSED_TRACE: 1 one k_trace launch per bounce, 0 separate k_shadow / k_extend, -1 library default")
    ap.add_argument("--sah-top", type=int, default=-1, help="RT3_OPT_SAH_TOP: cluster size T of the SAH top (0 = plain LBVH, -1 = library default)")
    ap.add_argument("--pool-chunk", type=int, default=0, help="traversal tuning: rays per pool grab (RT3_OPT_POOL_CHUNK)")
    ap.add_argument("--flags", type=int, default=-1, help="GConst.pad[0] feature flags (-1 = the full estimator); experiments only")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-crop", type=str, default="1280x720")  # ~14 s of oracle time on the GPU box's 16 host cores
    args = ap.parse_args()
    default_workload = (args.gpus == 1 and (args.width, args.height, args.spp, args.bounces, args.detail) == (1920, 1080, 64, 4, 1.0)
                        and args.batch_spp == 0 and args.leaf_size == 0 and args.node_width == 0 and args.node_quant == -1
                        and args.refill == -1 and args.flags == -1 and args.pool_chunk == 0 and args.fused_trace == -1 and args.sah_top == -1)

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    # RT3_DIST_BACKEND=gloo is a REHEARSAL mode for a 1-GPU box: several ranks share GPU 0 and the gather goes through
    # host tensors.  The real multi-GPU run uses nccl (= RCCL over xGMI), one GPU per rank.
    backend = os.environ.get("RT3_DIST_BACKEND", "nccl")
    device_index = local_rank % max(torch.cuda.device_count(), 1)
    if world > 1:
        import torch.distributed as dist

        torch.cuda.set_device(device_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)

    from raytracer3_amd import _lib as L
    from raytracer3_amd import assets, scenes
    from raytracer3_amd.renderer import DEFAULT_FLAGS, Camera, PathTracer

    W, H = args.width, args.height
    mesh = scenes.atrium(args.detail)
    sky = scenes.sky(2048, 1024)
    bn = assets.load_bluenoise()
    pt = PathTracer((W, H), device=device_index if world > 1 else 0, rank=rank, n_ranks=world)
    pt.host_staged_gather = world > 1 and backend != "nccl"
    if args.leaf_size:
        pt.ctx.set_option(L.OPT_LEAF_SIZE, args.leaf_size)
    if args.node_width:
        pt.ctx.set_option(L.OPT_NODE_WIDTH, args.node_width)
    if args.node_quant >= 0:
        pt.ctx.set_option(L.OPT_NODE_QUANT, args.node_quant)
    if args.refill >= 0:
        pt.ctx.set_option(L.OPT_EXTEND_VARIANT, args.refill)
    if args.pool_chunk:
        pt.ctx.set_option(L.OPT_POOL_CHUNK, args.pool_chunk)
    if args.sah_top >= 0:
        pt.ctx.set_option(L.OPT_SAH_TOP, args.sah_top)
    if args.fused_trace >= 0:
        pt.ctx.set_option(L.OPT_FUSED_TRACE, args.fused_trace)
    pt.set_scene(mesh, sky, bn)
    if args.batch_spp:
        pt.ctx.set_option(L.OPT_BATCH_SPP, args.batch_spp)
    cam = Camera(scenes.ATRIUM_CAMERA["position"], scenes.ATRIUM_CAMERA["direction"], math.radians(scenes.ATRIUM_CAMERA["fov_deg"]), W / H)

    def barrier():
        if dist is not None:
            dist.barrier()
        pt.ctx.wait()
        torch.cuda.synchronize()

    def frame(i):
        g = pt.make_gconst(cam, args.spp, args.bounces, frame=i, flags=DEFAULT_FLAGS if args.flags < 0 else args.flags)
        pt.render(g, postprocess=False, wait=False)
        return pt.gather_light(dist, torch, download=False) if world > 1 else None, g  # the frame is assembled in rank 0's HBM; no host copy in the timed region

    for i in range(args.warmup):
        frame(i)
    pt.ctx.set_option(L.OPT_PROFILE, 1)  # HIP events around every kernel, on the context's own stream
    barrier()
    pt.ctx.stats_reset()
    t0 = time.perf_counter()
    for i in range(args.steps):
        _, g_last = frame(args.warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    st = pt.ctx.stats()
    pt.ctx.set_option(L.OPT_PROFILE, 0)

    rays_local = st.extension_rays + st.shadow_rays
    if world > 1:
        red_dev = "cuda" if backend == "nccl" else "cpu"
        tt = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        rr = torch.tensor([float(rays_local), float(st.extension_rays), float(st.shadow_rays)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(rr, op=dist.ReduceOp.SUM)
        dt = float(tt.item())
        rays_total, ext_total, sh_total = (float(x) for x in rr.tolist())
    else:
        rays_total, ext_total, sh_total = float(rays_local), float(st.extension_rays), float(st.shadow_rays)

    # ---- roofline of the dominant kernel on this rank: one untimed counting frame gives n_nodes / n_tris
    pt.ctx.set_option(L.OPT_COUNT_TRAVERSAL, 1)
    pt.ctx.stats_reset()
    pt.render(g_last, postprocess=False, wait=True)
    cst = pt.ctx.stats()
    pt.ctx.set_option(L.OPT_COUNT_TRAVERSAL, 0)
    n_nodes, n_tris, levels, node_bytes = pt.ctx.accel_info()
    steps = max(args.steps, 1)
    nb = float(node_bytes)
    ext_bytes = 48.0 * cst.extension_rays + nb * cst.nodes_visited + 48.0 * cst.tris_tested          # all closest-hit rays of the frame
    sh_bytes = 48.0 * cst.shadow_rays + nb * cst.shadow_nodes_visited + 48.0 * cst.shadow_tris_tested  # all any-hit rays
    # the dominant kernel: k_trace (one launch per bounce, extension queue then shadow queue) when the fused path is on,
    # k_extend otherwise.  Its algorithmic bytes come from its own counters of the counting frame.
    fused = st.trace_launches > 0
    if fused:
        dom_name = "k_trace"
        dom_rays = float(cst.trace_rays[0] + cst.trace_rays[1])
        dom_bytes = 48.0 * dom_rays + nb * (cst.trace_nodes[0] + cst.trace_nodes[1]) + 48.0 * (cst.trace_tris[0] + cst.trace_tris[1])
        dom_ms, dom_launches = st.trace_ms / steps, st.trace_launches / steps
    else:
        dom_name, dom_rays, dom_bytes, dom_ms, dom_launches = "k_extend", float(cst.extension_rays), ext_bytes, st.extend_ms / steps, st.extend_launches / steps
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    trav_ms = (st.trace_ms + st.extend_ms + st.shadow_ms) / steps  # every traversal launch of the frame
    peak = 8000.0
    # HBM-side bytes per launch cannot be counted from inside this process: they come from separate
    # `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes over this same command (tools/profile_round.sh), corrected as
    # MI355X_MICROARCH.md prescribes by tools/summarize_profile.py and committed under profiles/. Only quoted for the
    # default workload they were collected on; null otherwise.
    traffic, traffic_src = None, None
    tfiles = sorted((ROOT / "profiles").glob("*_traffic.json"))
    if tfiles and default_workload:
        tj = json.loads(tfiles[-1].read_text())
        kk = [v for k, v in tj["kernels"].items() if k.startswith(f"rt3::{dom_name}<false")]
        if kk:
            traffic = round(kk[0]["hbm_bytes_per_launch"])
            traffic_src = f"profiles/{tfiles[-1].name}: {tj['correction']}"
    roofline = {
        "kernel": dom_name, "bound": "hbm", "achieved": round(achieved, 1), "peak": peak, "unit": "GB/s", "frac": round(achieved / peak, 4),
        "frac_of_measured_copy_peak": round(achieved / 6290.0, 4),  # 6.29 TB/s float4 copy (MI355X_MICROARCH.md); SURVEY 8d asks for both
        "traffic": traffic, "traffic_source": traffic_src,
        "launches_per_frame": dom_launches, "avg_launch_ms": round(dom_ms / max(dom_launches, 1), 4),
        "algorithmic_bytes_per_launch": round(dom_bytes / max(dom_launches, 1)), "rays_per_launch": round(dom_rays / max(dom_launches, 1)),
        "bytes_per_ray_formula": f"48 + {node_bytes}*n_nodes + 48*n_tris", "bvh": {"nodes": n_nodes, "node_bytes": node_bytes, "tris": n_tris, "levels": levels},
        "rays_per_frame": int(cst.extension_rays), "nodes_per_ray": round(cst.nodes_visited / max(cst.extension_rays, 1), 2),
        "tris_per_ray": round(cst.tris_tested / max(cst.extension_rays, 1), 2),
        "shadow": {"rays_per_frame": int(cst.shadow_rays), "nodes_per_ray": round(cst.shadow_nodes_visited / max(cst.shadow_rays, 1), 2),
                   "tris_per_ray": round(cst.shadow_tris_tested / max(cst.shadow_rays, 1), 2)},
        "all_traversal": {"achieved": round((ext_bytes + sh_bytes) / max(trav_ms * 1e-3, 1e-12) / 1e9, 1), "ms_per_frame": round(trav_ms, 3),
                          "launches_per_frame": (st.trace_launches + st.extend_launches + st.shadow_launches) / steps},
        "ms_per_frame": {"k_trace": round(st.trace_ms / steps, 3), "k_extend": round(st.extend_ms / steps, 3), "k_shadow": round(st.shadow_ms / steps, 3),
                         "k_shade": round(st.shade_ms / steps, 3), "other": round(st.other_ms / steps, 3)},
    }

    out = {
        "metric": "Mrays/s", "value": round(rays_total / dt / 1e6, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / max(args.steps, 1) * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"C3: atrium stand-in ({mesh.n_triangles} tris) {W}x{H}@{args.spp}spp B={args.bounces} layered BSDF (diffuse + GGX) + sky NEE/MIS + bluenoise, "
                               f"64x64 tiles over {world} GPU(s), one gather", "rays_per_frame": int(rays_total / max(args.steps, 1)),
                   "extension_rays_per_frame": int(ext_total / max(args.steps, 1)), "shadow_rays_per_frame": int(sh_total / max(args.steps, 1)),
                   "device": pt.ctx.device_name},
        "roofline": roofline,
    }

    # ---- CPU baseline (rank 0, N=1 only): the oracle path-traces a centred crop of the SAME frame
    if rank == 0 and world == 1 and not args.no_cpu:
        import orc

        cw, ch = (int(x) for x in args.cpu_crop.split("x"))
        cw, ch = min(cw, W), min(ch, H)
        x0, y0 = (W - cw) // 2, (H - ch) // 2
        rect = (x0, y0, x0 + cw, y0 + ch)
        threads = usable_cpus()
        osc = orc.Scene(mesh, sky, bn, leaf_size=args.leaf_size or 2, node_width=args.node_width or 4, quantized=(1 if args.node_quant < 0 else args.node_quant), sah_top=(DEFAULT_SAH_TOP if args.sah_top < 0 else args.sah_top))
        og = orc.GConst()
        C.memmove(C.byref(og), C.byref(g_last), 304)
        ogb, odepth = osc.gbuffer(og, rect=rect, threads=threads)
        tc = time.perf_counter()
        olight, counts = osc.reference_mode(og, ogb, odepth, rect=rect, threads=threads)
        cdt = time.perf_counter() - tc
        crays = float(counts[0] + counts[1])
        glight = pt.light()
        a = glight[y0:y0 + ch, x0:x0 + cw, :3].astype(np.float64)
        b = olight[y0:y0 + ch, x0:x0 + cw, :3].astype(np.float64)
        out["cpu_baseline"] = {"value": round(crays / cdt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
                               "sample": f"oracle refrence_mode pass on the centred {cw}x{ch} crop of the same {W}x{H}@{args.spp}spp frame "
                                         f"({int(crays)} rays, {cdt:.1f} s)"}
        out["rmse_vs_oracle"] = float(np.sqrt(np.mean((a - b) ** 2)))
        out["crop_bit_exact"] = bool(np.array_equal(glight[y0:y0 + ch, x0:x0 + cw].view(np.uint32), olight[y0:y0 + ch, x0:x0 + cw].view(np.uint32)))
        # traversal-count cross-check: GPU counters vs oracle counters on the crop's primary rays
        ys, xs = np.mgrid[y0:y0 + ch, x0:x0 + cw]
        pr = orc.primary_rays(og, xs.ravel()[::7], ys.ravel()[::7])
        _, _, _, _, gn, gt, _ = pt.ctx.trace_rays(pr, counts=True)
        _, _, _, _, on, ot = osc.trace_closest(pr, threads=threads, counts=True)
        out["roofline"]["counts_match_oracle"] = bool(np.array_equal(gn, on) and np.array_equal(gt, ot))

    if world > 1 and os.environ.get("RT3_CHECK_GATHER"):  # rehearsal aid: the gathered frame must equal a 1-rank render of it
        full = pt.gather_light(dist, torch)
        if rank == 0:
            solo = PathTracer((W, H), device=device_index)
            solo.set_scene(mesh, sky, bn)
            solo.render(g_last, postprocess=False, wait=True)
            out["gather_bit_identical_to_single_rank"] = bool(np.array_equal(full.view(np.uint32), solo.light().view(np.uint32)))
            solo.close()
    if rank == 0:
        print(json.dumps(out))
    pt.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
