#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json: "Mrays/s + ms/frame, Sponza 1080p@64spp").

A step = one frame: gbuffer pass + refrence_mode pass (B = 4, full estimator: layered BSDF + sky NEE/MIS + blue-noise shift)
over this rank's 64x64 tiles, then the frame's ONE collective: rt3_gather_tiles (RCCL inside librt3, on librt3's stream).
Workload = configs[2] (C3): atrium stand-in (sponza_scene.glb is not shipped), 1920x1080 @ 64 spp.
value = Mrays/s = (primary + bounce + shadow rays of all ranks) / max-over-ranks frame time; ms_per_step = ms/frame.
Strong scaling: the frame is fixed, tiles are split over the ranks.

Launch: `python bench.py --gpus N` starts its N ranks itself (one child process per GPU, spawned BEFORE anything touches the
GPU in the parent, which never imports torch or librt3), prints rank 0's JSON line and exits non-zero if a rank fails or if
fewer than N devices are visible.  Under `torch.distributed.run` (WORLD_SIZE set) it is a rank.  RT3_DIST_BACKEND=gloo is a
REHEARSAL for a one-GPU box: the ranks share the visible GPU(s) and the gather's bytes move through host tensors.

Extra objects on the JSON line:
  roofline     the dominant kernel (k_extend).  `achieved` / `frac` are MEASURED fabric-side bytes (rocprofv3 FETCH_SIZE /
               WRITE_SIZE passes of this same command, committed under profiles/, quoted only while their kernel time agrees
               with this run) over the live HIP-event launch time, against the 8 TB/s HBM3E peak -- a fraction that cannot
               exceed 1.  The ALGORITHMIC bytes of SURVEY 8d (48 + node_bytes n_nodes + 48 n_tris per ray) are reported
               beside it: the BVH is cache resident, so that figure is a cache-side gather rate, not HBM traffic, and is
               priced against the measured rate of a pointer-chasing gather instead (profiles/r02_gather_cap.md: 9.65 TB/s of requested
               bytes from any cache level).  `binding` says what the counters show
               the kernel is limited by; `k_shade` carries the same for the one kernel that really is traffic bound.
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
DEFAULT_SAH_TOP = 1  # the library's RT3_OPT_SAH_TOP default (the oracle of the parity leg is built the same way)
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


def visible_gpus() -> int:
    """Devices a rank could open, counted in a short-lived CHILD: the launching parent must never initialise the GPU (a process
    that did cannot hand over to other programs on this pool) -- it does not even import torch."""
    import subprocess

    r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=600)
    try:
        return int(r.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        return 0


def deadline(seconds: float, what: str):
    """A watchdog for the steps of a rank that could wait for ever on another rank (communicator bootstrap, the first exchange, a
    collective): when the time is up the process says what it was doing and exits with status 3, so that the launcher (ours or
    torch.distributed.run) stops the other ranks instead of the whole job sitting in a collective.  Returns the timer; .cancel() it."""
    import threading

    def fire():
        print(f"bench.py: rank {os.environ.get('RANK', '0')}: '{what}' did not finish within {seconds:.0f} s -- giving up (exit 3)", file=sys.stderr, flush=True)
        os._exit(3)

    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t


def self_launch(n: int, argv: list) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes, relay rank 0's JSON line.
    The whole job has a wall-clock deadline (RT3_BENCH_DEADLINE_S, default 1500 s): past it the ranks are stopped and the exit
    status is non-zero -- a rank stuck in a collective must not hold the caller for ever."""
    import socket
    import subprocess

    limit = float(os.environ.get("RT3_BENCH_DEADLINE_S", "1500"))
    t_start = time.monotonic()

    backend = os.environ.get("RT3_DIST_BACKEND", "nccl")
    ndev = visible_gpus()
    if ndev < 1:
        print("bench.py: no GPU visible (librt3 has no CPU fallback)", file=sys.stderr)
        return 3
    if backend == "nccl" and ndev < n:
        print(f"bench.py: --gpus {n} needs {n} visible devices, found {ndev} (RCCL cannot put two ranks on one GPU; "
              f"RT3_DIST_BACKEND=gloo rehearses the N-rank path on fewer)", file=sys.stderr)
        return 3
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        # The build environment of this pool documents: "HSA_ENABLE_IPC_MODE_LEGACY=0 is already exported here and on the GPU box -- keep it in
        # any env you build for multi-process GPU work: the host driver only supports dmabuf IPC, and without it RCCL / CUDA-tensor sharing
        # across processes fails with hipIpcGetMemHandle: invalid argument".  setdefault: an operator's own value wins.
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("NCCL_DEBUG", "WARN")  # RCCL's own warnings of the first exchange go to stderr with the rank's output
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    rc = 0
    try:
        pending = set(range(n))
        lines = []
        import threading

        def pump():  # rank 0's stdout: keep everything, relay at the end
            for ln in procs[0].stdout:
                lines.append(ln)

        th = threading.Thread(target=pump, daemon=True)
        th.start()
        while pending:
            for r in list(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                    for q in pending:  # exact PIDs of our own children: they would wait in a collective for ever
                        procs[q].terminate()
            if pending and time.monotonic() - t_start > limit:
                print(f"bench.py: ranks {sorted(pending)} still running after {limit:.0f} s (RT3_BENCH_DEADLINE_S): stopping them", file=sys.stderr)
                rc = rc or 4
                for q in pending:
                    procs[q].terminate()
                time.sleep(2.0)
                break
            time.sleep(0.05)
        th.join(timeout=5)
        js = [ln for ln in lines if ln.startswith("{")]
        for ln in lines:
            if not ln.startswith("{"):
                sys.stderr.write(ln)
        if rc == 0 and not js:
            print("bench.py: rank 0 printed no result line", file=sys.stderr)
            rc = 1
        if rc == 0:
            print(js[-1].strip())
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    return rc


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
    ap.add_argument("--trace-blocks", type=int, default=0, help="traversal tuning: persistent workgroups per traversal launch (RT3_OPT_TRACE_BLOCKS)")
    ap.add_argument("--flags", type=int, default=-1, help="GConst.pad[0] feature flags (-1 = the full estimator); experiments only")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-crop", type=str, default="1280x720")  # ~14 s of oracle time on the GPU box's 16 host cores
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    default_workload = ((args.width, args.height, args.spp, args.bounces, args.detail) == (1920, 1080, 64, 4, 1.0)
                        and args.batch_spp == 0 and args.leaf_size == 0 and args.node_width == 0 and args.node_quant == -1
                        and args.refill == -1 and args.flags == -1 and args.pool_chunk == 0 and args.trace_blocks == 0 and args.fused_trace == -1 and args.sah_top == -1)

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:  # also when a launcher other than self_launch started this rank (see self_launch for where the first one comes from)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("NCCL_DEBUG", "WARN")
    run_watchdog = deadline(float(os.environ.get("RT3_BENCH_DEADLINE_S", "1500")), "the whole bench run") if world > 1 else None
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count and --gpus must agree")
    dist = None
    # RT3_DIST_BACKEND=gloo is a REHEARSAL mode for a 1-GPU box: several ranks share GPU 0 and the gather goes through
    # host tensors.  The real multi-GPU run uses nccl (= RCCL over xGMI), one GPU per rank.
    backend = os.environ.get("RT3_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py: no GPU visible (librt3 has no CPU fallback)")
    if world > 1 and backend == "nccl" and ndev < world:
        raise SystemExit(f"bench.py: {world} ranks need {world} visible devices, found {ndev} (RT3_DIST_BACKEND=gloo rehearses on fewer)")
    device_index = local_rank % ndev
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
    sky_w = int(os.environ.get("RT3_SKY_W", "2048"))  # (experiments: a smaller environment map)
    sky = scenes.sky(sky_w, sky_w // 2)
    bn = assets.load_bluenoise()
    pt = PathTracer((W, H), device=device_index if world > 1 else 0, rank=rank, n_ranks=world)
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
    if args.trace_blocks:
        pt.ctx.set_option(L.OPT_TRACE_BLOCKS, args.trace_blocks)
    if args.sah_top >= 0:
        pt.ctx.set_option(L.OPT_SAH_TOP, args.sah_top)
    if args.fused_trace >= 0:
        pt.ctx.set_option(L.OPT_FUSED_TRACE, args.fused_trace)
    pt.set_scene(mesh, sky, bn)
    if world > 1:
        # librt3's own RCCL communicator: rank 0's unique id travels through the process group's key-value store (the C ABI opens
        # no sockets; a Rust host would carry the 128 bytes over whatever channel it has).  The id exchange also runs in the gloo
        # rehearsal, so that everything but ncclCommInitRank / the exchange itself is exercised on a one-GPU box.
        store = dist.distributed_c10d._get_default_store()
        if rank == 0:
            store.set("rt3_comm_unique_id", pt.ctx.comm_unique_id())
        uid = bytes(store.get("rt3_comm_unique_id"))
        if len(uid) != L.COMM_ID_BYTES:
            raise SystemExit(f"bench.py: rank {rank} received a {len(uid)}-byte communicator id")
        gather_mode = "REHEARSAL: host-moved bytes (gloo), ranks share GPUs"
        if backend == "nccl" or os.environ.get("RT3_TRY_RCCL"):  # (RT3_TRY_RCCL: exercise this block in the one-GPU rehearsal, where RCCL must refuse)
            # rt3_comm_init has never met more than one GPU before the driver's node: if it fails on ANY rank, every rank
            # drops to the host-moved exchange over a gloo group, and the JSON line says so -- a labelled fallback instead of no number
            if rank == 0:
                ver = C.c_int(0)
                pt.ctx.lib.rt3_comm_version(C.byref(ver))
                print(f"bench.py: RCCL (ncclGetVersion) {ver.value // 10000}.{ver.value // 100 % 100}.{ver.value % 100}, NCCL_DEBUG={os.environ.get('NCCL_DEBUG', 'unset')}, "
                      f"HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', 'unset')}, {world} ranks", file=sys.stderr, flush=True)
            # ncclCommInitRank blocks until every rank has joined: if one of them cannot (its own init failed, its GPU is gone), the rest
            # would wait for ever -- the watchdog ends this rank instead, and the launcher stops the others
            wd = deadline(float(os.environ.get("RT3_COMM_DEADLINE_S", "240")), "rt3_comm_init + agreement of the ranks")
            try:
                pt.init_comm(uid)
                ok, why = 1, ""
            except L.Rt3Error as e:
                ok, why = 0, str(e)
                print(f"bench.py: rank {rank}: rt3_comm_init failed: {why}", file=sys.stderr, flush=True)
            flag = torch.tensor([ok], dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            wd.cancel()
            if int(flag.item()) == 1:
                gather_mode = "rt3_gather_tiles (RCCL)"
            else:
                if pt.comm_ready:
                    pt.ctx.comm_destroy()
                    pt.comm_ready = False
                pt.host_group = dist.new_group(backend="gloo")
                gather_mode = "FALLBACK: rt3_comm_init failed on at least one rank; bytes moved through host tensors (gloo)"
    if args.batch_spp:
        pt.ctx.set_option(L.OPT_BATCH_SPP, args.batch_spp)
    cam = Camera(scenes.ATRIUM_CAMERA["position"], scenes.ATRIUM_CAMERA["direction"], math.radians(scenes.ATRIUM_CAMERA["fov_deg"]), W / H)

    def barrier():
        pt.ctx.wait()  # librt3's own stream (passes + the gather's send / receives)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def frame(i):
        g = pt.make_gconst(cam, args.spp, args.bounces, frame=i, flags=DEFAULT_FLAGS if args.flags < 0 else args.flags)
        pt.render(g, postprocess=False, wait=False)
        # the frame is assembled in rank 0's HBM (its `Light` image); no host copy and, on RCCL, no host synchronisation in the timed region
        return pt.gather_light(dist, torch, download=False) if world > 1 else None, g

    for i in range(args.warmup):
        frame(i)
    if world > 1 and args.warmup == 0:
        # RCCL opens its peer connections at the first send / receive (tens to hundreds of ms): never inside the timed steps --
        # one untimed frame + gather stands in for the warm-up the caller asked not to have
        frame(0)
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
    by_rank = None
    if world > 1:
        red_dev = "cuda" if backend == "nccl" else "cpu"
        # every rank's per-kernel HIP-event times per frame: the slowest rank decides the frame, and which kernel makes it slow is what a
        # scaling curve needs beside it (the drain at the end of every traversal launch bounds C3 near 0.87 at N = 8)
        mine = torch.tensor([st.extend_ms, st.shadow_ms, st.shade_ms, st.gather_ms, st.other_ms, st.trace_ms], dtype=torch.float64, device=red_dev) / max(args.steps, 1)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        by_rank = [[round(float(x), 3) for x in t.tolist()] for t in allr]
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
    algo_gbps = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    # what the lanes REQUEST from the vector-memory path: 32 B of ray records per ray + 64 B per step (a step issues 4 x 16 B, node or
    # triangle) -- EXCEPT the node visits the kernel serves from its LDS copy of the top of the tree (counted by the counting kernels)
    if fused:
        dom_nodes, dom_tris = float(cst.trace_nodes[0] + cst.trace_nodes[1]), float(cst.trace_tris[0] + cst.trace_tris[1])
        dom_lds = float(cst.nodes_visited_lds + cst.shadow_nodes_visited_lds)
    else:
        dom_nodes, dom_tris, dom_lds = float(cst.nodes_visited), float(cst.tris_tested), float(cst.nodes_visited_lds)
    requested_gbps = (32.0 * dom_rays + 64.0 * (dom_nodes - dom_lds + dom_tris)) / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    trav_ms = (st.trace_ms + st.extend_ms + st.shadow_ms) / steps  # every traversal launch of the frame
    peak = 8000.0
    launches = max(dom_launches, 1)
    avg_ms = dom_ms / launches
    # ---- measured fabric-side traffic.  HBM-side bytes cannot be counted from inside this process: they come from separate
    # `rocprofv3 --pmc` passes over this same command (tools/profile_round.sh), summarised by tools/summarize_profile.py into
    # profiles/*_traffic.json TOGETHER WITH THE HASH OF THE KERNEL SOURCES they were taken on.  A profile is quoted only for the default
    # workload and only when that hash equals the hash of the sources this run was built from (counters of other kernels are not
    # evidence for these, however close their times).
    src_hash = L.kernel_source_hash()
    prof = None
    if default_workload and world == 1:
        for tf in sorted((ROOT / "profiles").glob("r*_traffic.json")):
            tj = json.loads(tf.read_text())
            if tj.get("source_hash") == src_hash:
                prof = dict(tj, _file=f"profiles/{tf.name}")  # (several matching files = several runs on the same kernels: the last one)

    def prof_kernel(prefix):
        if prof is None:
            return None
        kk = [v for k, v in prof["kernels"].items() if k.startswith(prefix)]
        return kk[0] if kk else None

    def hbm_side(entry, live_avg_ms):
        """{traffic, traffic_raw, achieved GB/s, frac} from a profile entry of THESE kernels, or Nones"""
        if not entry or live_avg_ms <= 0:
            return {"traffic": None, "traffic_raw": None, "achieved": None, "frac": None}
        t = entry["hbm_bytes_per_launch"]
        g = t / (live_avg_ms * 1e-3) / 1e9
        return {"traffic": round(t), "traffic_raw": round(entry["fetch_bytes_raw"] + entry["write_bytes"]), "achieved": round(g, 1), "frac": round(g / peak, 4)}

    pk = prof_kernel(f"rt3::{dom_name}<false") or {}
    ke = hbm_side(pk or None, avg_ms)
    # what the kernel MUST move through HBM whatever the caches do: its ray records in and hit records out (the queues are 21 GB)
    stream_bytes = 48.0 * dom_rays / launches
    stream_gbps = stream_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    binding = None
    if pk.get("valu_per_clk_per_simd") is not None:
        # a wave64 VALU instruction holds its SIMD-32 for two cycles: the issue ceiling is 0.5 per clock per SIMD
        binding = {"bound": "valu_issue", "achieved": pk["valu_per_clk_per_simd"], "peak": 0.5, "unit": "wave VALU instr / clk / SIMD",
                   "frac": round(pk["valu_per_clk_per_simd"] / 0.5, 4), "wait_any": pk.get("wait_any"), "wait_inst_any": pk.get("wait_inst_any"),
                   "active_inst_any": pk.get("active_inst_any"), "l2_hit_rate": pk.get("l2_hit_rate"), "source": prof["_file"]}
    # ---- k_shade.  Streaming bytes it must move per frame (record sizes of DESIGN.md 5):
    # first bounce: {pixel, blue noise} 8 + depth 4 + G-buffer 16 in, radiance slot 16 out; later bounces: ray records 32 (their .w
    # carry the path's pdf and id) + throughput 12 + hit 16 in; every extension ray out 44 (ray 32 + throughput 12), every shadow ray out 40
    n_first_paths = float(W * H * args.spp) / world
    later_in = max(float(cst.extension_rays) - float(W * H) / world, 0.0)  # every bounce ray traced is one path vertex shaded afterwards
    shade_stream = n_first_paths * (28.0 + 16.0) + later_in * 60.0 + float(cst.extension_rays - float(W * H) / world) * 44.0 + float(cst.shadow_rays) * 40.0
    shade_ms = st.shade_ms / steps
    shade_gbps = shade_stream / (shade_ms * 1e-3) / 1e9 if shade_ms > 0 else 0.0
    sh_first, sh_later = prof_kernel("rt3::k_shade<true"), prof_kernel("rt3::k_shade<false")
    shade_traffic = None
    if sh_first and sh_later:
        shade_traffic = sh_first["hbm_bytes_per_launch"] + sh_later["hbm_bytes_per_launch"] * (args.bounces - 1)
    gather_ceiling = 9650.0  # GB/s of requested bytes a pointer-chasing gather gets from any cache level (profiles/r02_gather_cap.md)
    roofline = {
        "kernel": dom_name, "bound": "hbm", "peak": peak, "unit": "GB/s",
        # frac: MEASURED fabric-side bytes of one launch (2 x FETCH_SIZE + WRITE_SIZE of a counter profile of these very kernels) / its
        # live duration / 8 TB/s -- the share of the HBM roofline the kernel really occupies; cannot exceed 1.  Without a profile of
        # these kernels: the compulsory queue bytes only (a lower bound), and `frac_is` says so.
        "achieved": ke["achieved"] if ke["achieved"] is not None else round(stream_gbps, 1),
        "frac": ke["frac"] if ke["frac"] is not None else round(stream_gbps / peak, 4),
        "frac_is": ("measured fabric-side bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, rocprofv3 --pmc, same kernel sources) / live HIP-event launch time / 8 TB/s"
                    if ke["achieved"] is not None else "LOWER BOUND: compulsory queue bytes (48 B per ray) / launch time / 8 TB/s -- no counter profile of these kernel sources is committed"),
        "traffic": ke["traffic"], "traffic_uncorrected": ke["traffic_raw"], "traffic_source": prof["_file"] if (prof and ke["traffic"] is not None) else None,
        # frac_8d: SURVEY 8d's contract figure -- ALGORITHMIC bytes (48 + 64 n_nodes + 48 n_tris per ray, counts equal to the oracle's) / the
        # same launch time / 8 TB/s.  Exceeds 1 when the tree is served from L2 / Infinity Cache instead of HBM: then it measures a
        # cache-side gather rate, not HBM use (see `vector_memory` for the ceiling that rate runs against)
        "achieved_8d": round(algo_gbps, 1), "frac_8d": round(algo_gbps / peak, 4),
        "frac_8d_is": f"algorithmic bytes per launch (48 + {node_bytes}*n_nodes + 48*n_tris per ray, SURVEY 8d) / live launch time / 8 TB/s; > 1 = cache-served",
        "algorithmic_bytes_per_launch": round(dom_bytes / launches),
        "compulsory_stream": {"bytes_per_launch": round(stream_bytes), "GBps": round(stream_gbps, 1), "frac": round(stream_gbps / peak, 4)},
        # bytes the lanes request from the vector-memory path (LDS-served node visits excluded) against the rate a dependent gather
        # reaches from any cache level in the microbenchmark: a fraction of THAT ceiling, <= 1
        "vector_memory": {"requested_GBps": round(requested_gbps, 1), "ceiling_GBps": gather_ceiling, "frac": round(requested_gbps / gather_ceiling, 4),
                          "node_visits_served_from_lds": round(dom_lds / max(dom_nodes, 1.0), 4),
                          "is": "(32 B per ray + 64 B per node / triangle step NOT served by the LDS top-of-tree copy) / launch time, against profiles/r02_gather_cap.md"},
        "binding": binding,
        "source_hash": src_hash,
        "launches_per_frame": dom_launches, "avg_launch_ms": round(avg_ms, 4), "rays_per_launch": round(dom_rays / launches),
        "bvh": {"nodes": n_nodes, "node_bytes": node_bytes, "tris": n_tris, "levels": levels},
        "rays_per_frame": int(cst.extension_rays), "nodes_per_ray": round(cst.nodes_visited / max(cst.extension_rays, 1), 2),
        "tris_per_ray": round(cst.tris_tested / max(cst.extension_rays, 1), 2),
        "shadow": {"rays_per_frame": int(cst.shadow_rays), "nodes_per_ray": round(cst.shadow_nodes_visited / max(cst.shadow_rays, 1), 2),
                   "tris_per_ray": round(cst.shadow_tris_tested / max(cst.shadow_rays, 1), 2)},
        "all_traversal": {"algorithmic_GBps": round((ext_bytes + sh_bytes) / max(trav_ms * 1e-3, 1e-12) / 1e9, 1), "ms_per_frame": round(trav_ms, 3),
                          "launches_per_frame": (st.trace_launches + st.extend_launches + st.shadow_launches) / steps},
        "k_shade": {"bound": "dependent-gather latency at 6 waves per SIMD (DESIGN.md 7); hbm share in hbm_side_frac", "streaming_bytes_per_frame": round(shade_stream),
                    "achieved": round(shade_gbps, 1), "peak": peak, "unit": "GB/s", "frac": round(shade_gbps / peak, 4),
                    "traffic_per_frame": round(shade_traffic) if shade_traffic else None,
                    "hbm_side_frac": round(shade_traffic / (shade_ms * 1e-3) / 1e9 / peak, 4) if shade_traffic else None,
                    "traffic_over_streaming": round(shade_traffic / shade_stream, 2) if shade_traffic else None,
                    "l2_hit_rate": (sh_later or {}).get("l2_hit_rate")},
        "ms_per_frame": {"k_trace": round(st.trace_ms / steps, 3), "k_extend": round(st.extend_ms / steps, 3), "k_shadow": round(st.shadow_ms / steps, 3),
                         "k_shade": round(st.shade_ms / steps, 3), "gather": round(st.gather_ms / steps, 3), "other": round(st.other_ms / steps, 3)},
    }

    out = {
        "metric": "Mrays/s", "value": round(rays_total / dt / 1e6, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / max(args.steps, 1) * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic", "source_hash": src_hash,
        "config": {"workload": f"C3: atrium stand-in ({mesh.n_triangles} tris) {W}x{H}@{args.spp}spp B={args.bounces} layered BSDF (diffuse + GGX) + sky NEE/MIS + bluenoise, "
                               f"64x64 tiles over {world} GPU(s), one gather", "rays_per_frame": int(rays_total / max(args.steps, 1)), "gather": ("none" if world == 1 else gather_mode),
                   "extension_rays_per_frame": int(ext_total / max(args.steps, 1)), "shadow_rays_per_frame": int(sh_total / max(args.steps, 1)),
                   "device": pt.ctx.device_name},
        "roofline": roofline,
    }
    if world > 1:
        # true only when the frame's bytes moved through rt3_gather_tiles on RCCL; a FALLBACK / REHEARSAL line is NOT a measurement of the gather
        out["gather_is_rccl"] = gather_mode.startswith("rt3_gather_tiles")
        keys = ["k_extend", "k_shadow", "k_shade", "gather", "other", "k_trace"]
        out["per_rank_ms_per_frame"] = {"keys": keys, "by_rank": by_rank, "max_over_ranks": {k: max(r[i] for r in by_rank) for i, k in enumerate(keys)},
                                        "slowest_rank_sum": round(max(sum(r) for r in by_rank), 3), "mean_rank_sum": round(sum(sum(r) for r in by_rank) / world, 3),
                                        "note": "HIP-event time of each kernel class per frame on every rank's own stream; gather = RCCL send / grouped receives (root: all peers)"}

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
        # SURVEY 8d's CPU baseline: the oracle's traversal of the SAME BVH arrays on the C2 primary batch (one ray per pixel) + the bounce-1
        # batch (the extension rays of sample 0 after the first shade), all usable host threads (static split), best of 5; the GPU
        # walks the same two batches through rt3_trace_rays for the ratio.  The crop's full path-traced pass stays beside it.
        og1 = orc.GConst()
        C.memmove(C.byref(og1), C.byref(og), 304)
        fgb, fdepth = osc.gbuffer(og1, threads=threads)
        ys_, xs_ = np.mgrid[0:H, 0:W]
        batch = np.ascontiguousarray(np.concatenate([orc.primary_rays(og1, xs_.ravel(), ys_.ravel()), osc.bounce1_rays(og1, fgb, fdepth)], axis=1))
        best = None
        for _ in range(5):
            tb = time.perf_counter()
            ohit = osc.trace_closest(batch, threads=threads)
            tb = time.perf_counter() - tb
            best = tb if best is None or tb < best else best
        ghit = pt.ctx.trace_rays(batch, repeat=5)
        trav = {"value": round(batch.shape[1] / best / 1e6, 3), "unit": "Mrays/s", "rays": int(batch.shape[1]), "best_of_5_s": round(best, 4),
                "gpu_same_batches_Mrays_s": round(batch.shape[1] / (ghit[4] * 1e-3) / 1e6, 1) if ghit[4] > 0 else None,
                "gpu_hits_equal_cpu_hits": bool(np.array_equal(ghit[3], ohit[3]) and np.array_equal(ghit[0], ohit[0]))}
        out["cpu_baseline"] = {"value": trav["value"], "unit": "Mrays/s", "cores": threads, "kind": "port",
                               "sample": f"oracle closest-hit traversal of the bench scene's BVH arrays: the C2 primary batch ({W * H} rays) + the bounce-1 batch "
                                         f"({batch.shape[1] - W * H} rays) at {W}x{H}, {threads} threads (static split), best of 5 = {best:.3f} s (SURVEY 8d)",
                               "traversal": trav,
                               "path_tracer": {"value": round(crays / cdt / 1e6, 3), "unit": "Mrays/s",
                                               "sample": f"oracle refrence_mode pass on the centred {cw}x{ch} crop of the same {W}x{H}@{args.spp}spp frame ({int(crays)} rays, {cdt:.1f} s)"}}
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
    if run_watchdog is not None:
        run_watchdog.cancel()
    pt.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
