"""N > 1 path on CPU: gloo ranks, each producing its interleaved 64x64 tiles (the oracle stands in for the GPU), ONE gather to
the root through the product's host-memory exchange (`exchange_tiles_host`: the layout of `rt3_gather_tiles` -- exact per-rank
counts at `gather_offsets`, nothing from the root itself), ONE untile over the whole receive buffer, compare with the
single-rank image."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path, W, H, dst):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import orc
    from raytracer3_amd import assets, scenes
    from raytracer3_amd.renderer import exchange_tiles_host, gather_offsets

    mesh, sky, bn = scenes.atrium(0.15), scenes.sky(128, 64), assets.load_bluenoise()
    osc = orc.Scene(mesh, sky, bn)
    g = orc.camera_gconst(width=W, height=H, **scenes.ATRIUM_CAMERA)
    g.bounces, g.samples, g.blendfactor, g.frame = 3, 2, 1.0, 5
    g.pad[0] = orc.F_NEE_SKY | orc.F_BLUENOISE | orc.F_FACEFORWARD
    gb, depth = osc.gbuffer(g, threads=2)
    full, _ = osc.reference_mode(g, gb, depth, threads=2)  # every rank can compute the full image; it only SHIPS its tiles
    lists = [orc.tile_pixels(W, H, r, world) for r in range(world)]
    off = gather_offsets([len(x) for x in lists], dst)
    assert off[-1] == W * H - len(lists[dst]) and off[dst + 1] == off[dst]
    mine_xy = lists[rank]
    image = np.zeros((H, W, 4), np.float32)
    image[mine_xy[:, 1], mine_xy[:, 0]] = full[mine_xy[:, 1], mine_xy[:, 0]]  # a rank renders its own tiles into its own image
    recv = exchange_tiles_host(dist, torch, rank, world, off, full[mine_xy[:, 1], mine_xy[:, 0]] if rank != dst else None, dst)
    assert (recv is not None) == (rank == dst)
    if rank == dst:
        xy = np.concatenate([lists[r] for r in range(world) if r != dst])  # the ONE untile over all received ranks
        assert len(xy) == len(recv)
        image[xy[:, 1], xy[:, 0]] = recv
        np.save(out_path, np.stack([image, full]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,dst", [(2, 0), (3, 1)])
def test_gather_reassembles_the_frame(tmp_path, world, dst):
    W, H = 200, 150  # 4 x 3 tiles with ragged right / bottom edges
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker, args=(world, _free_port(), out, W, H, dst), nprocs=world, join=True)
    image, full = np.load(out)
    assert np.array_equal(image.view(np.uint32), full.view(np.uint32))
    assert image[..., :3].mean() > 0


def test_gather_offsets_skip_the_root():
    from raytracer3_amd.renderer import gather_offsets

    assert gather_offsets([5, 7, 9], 0) == [0, 0, 7, 16]
    assert gather_offsets([5, 7, 9], 1) == [0, 5, 5, 14]
    assert gather_offsets([5, 0, 9], 2) == [0, 5, 5, 5]
    assert gather_offsets([4], 0) == [0, 0]


def _z(x, y):
    k = 0
    for b in range(16):
        k |= ((x >> b) & 1) << (2 * b) | ((y >> b) & 1) << (2 * b + 1)
    return k


def test_tile_partition_properties():
    import orc

    for (W, H) in ((1920, 1080), (200, 150), (64, 64), (65, 1)):
        for n in (1, 2, 3, 8):
            seen = np.zeros((H, W), np.int32)
            sizes = []
            for r in range(n):
                xy = orc.tile_pixels(W, H, r, n)
                seen[xy[:, 1], xy[:, 0]] += 1
                sizes.append(len(xy))
                # tile i (Z-order over the tile grid) belongs to rank i mod n: every pixel this rank lists lies in one of ITS tiles
                tx, ty = xy[:, 0].astype(np.uint32) // 64, xy[:, 1].astype(np.uint32) // 64
                tiles_x, tiles_y = (W + 63) // 64, (H + 63) // 64
                zkeys = sorted((_z(x, y), x, y) for y in range(tiles_y) for x in range(tiles_x))
                order = {(x, y): i for i, (_, x, y) in enumerate(zkeys)}
                own = np.array([order[(int(a), int(b))] % n for a, b in zip(tx[:: max(1, len(xy) // 500)], ty[:: max(1, len(xy) // 500)])])
                assert (own == r).all()
            assert (seen == 1).all()  # every pixel owned exactly once
            if W * H >= 64 * 64 * n * 4:
                assert max(sizes) - min(sizes) <= 2 * 64 * 64  # interleaving balances the load
    # 1080p: 30 x 17 = 510 tiles
    assert sum(len(orc.tile_pixels(1920, 1080, r, 8)) for r in range(8)) == 1920 * 1080
