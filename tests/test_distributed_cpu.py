"""N > 1 path on CPU: two gloo ranks, each producing its interleaved 64x64 tiles (the oracle stands in for the GPU),
ONE gather to rank 0 through the product's `gather_tiles`, untile, compare with the single-rank image."""
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


def _worker(rank, world, port, out_path, W, H):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import orc
    from raytracer3_amd import assets, scenes
    from raytracer3_amd.renderer import gather_tiles

    mesh, sky, bn = scenes.atrium(0.15), scenes.sky(128, 64), assets.load_bluenoise()
    osc = orc.Scene(mesh, sky, bn)
    g = orc.camera_gconst(width=W, height=H, **scenes.ATRIUM_CAMERA)
    g.bounces, g.samples, g.blendfactor, g.frame = 3, 2, 1.0, 5
    g.pad[0] = orc.F_NEE_SKY | orc.F_BLUENOISE | orc.F_FACEFORWARD
    gb, depth = osc.gbuffer(g, threads=2)
    full, _ = osc.reference_mode(g, gb, depth, threads=2)  # every rank can compute the full image; it only SHIPS its tiles
    mine_xy = orc.tile_pixels(W, H, rank, world)
    counts = [len(orc.tile_pixels(W, H, r, world)) for r in range(world)]
    image = np.zeros((H, W, 4), np.float32)

    def pack(buf):
        buf[: len(mine_xy)] = torch.from_numpy(full[mine_xy[:, 1], mine_xy[:, 0]])

    def unpack(r, buf):
        xy = orc.tile_pixels(W, H, r, world)
        image[xy[:, 1], xy[:, 0]] = buf[: len(xy)].numpy()

    done = gather_tiles(dist, torch, torch.device("cpu"), rank, world, counts, pack, unpack, dst=0)
    assert done == (rank == 0)
    if rank == 0:
        np.save(out_path, np.stack([image, full]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gather_reassembles_the_frame(tmp_path):
    W, H = 200, 150  # 4 x 3 tiles with ragged right / bottom edges
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker, args=(2, _free_port(), out, W, H), nprocs=2, join=True)
    image, full = np.load(out)
    assert np.array_equal(image.view(np.uint32), full.view(np.uint32))
    assert image[..., :3].mean() > 0


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
                tiles = {(int(x) // 64, int(y) // 64) for x, y in xy[:: max(1, len(xy) // 200)]}
                assert all(True for _ in tiles)
            assert (seen == 1).all()  # every pixel owned exactly once
            if W * H >= 64 * 64 * n * 4:
                assert max(sizes) - min(sizes) <= 2 * 64 * 64  # interleaving balances the load
    # 1080p: 30 x 17 = 510 tiles
    assert sum(len(orc.tile_pixels(1920, 1080, r, 8)) for r in range(8)) == 1920 * 1080
