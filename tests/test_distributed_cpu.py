"""CPU suite, part 3: the N > 1 path (tile deal + the one gather) on gloo, world_size 2."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fujiyama_renderer_amd import distributed as fjdist


def test_tile_deal_partitions_every_tile_once():
    for n, w in ((2040, 8), (2040, 3), (64, 2), (5, 8), (1, 1)):
        seen = sorted(t for r in range(w) for t in fjdist.tiles_of_rank(n, r, w))
        assert seen == list(range(n))
        sizes = [len(fjdist.tiles_of_rank(n, r, w)) for r in range(w)]
        assert max(sizes) - min(sizes) <= 1


def test_pack_unpack_roundtrip_with_ragged_edge_tiles():
    H, W, tw, th = 70, 100, 32, 32            # 4 x 3 tiles, last column 4 px, last row 6 px
    fb = torch.arange(H * W * 4, dtype=torch.float32).reshape(H, W, 4)
    ids_a, ids_b = fjdist.tiles_of_rank(12, 0, 2), fjdist.tiles_of_rank(12, 1, 2)
    a, b = fjdist.pack_tiles(fb, ids_a, tw, th), fjdist.pack_tiles(fb, ids_b, tw, th)
    assert a.shape == (6, 32, 32, 4)
    back = fjdist.unpack_tiles([a, b], [ids_a, ids_b], W, H, tw, th)
    assert torch.equal(back, fb)


def _worker(rank, world, port, H, W, out_path, mode=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tw = th = 32
    nx, ny = -(-W // tw), -(-H // th)
    n_tiles = nx * ny
    # each rank "renders" only its own tiles: pixel value encodes (y, x, channel)
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    full = torch.stack([yy * 1000.0 + xx, yy * 1.0, xx * 1.0, torch.ones_like(yy) * 1.0], dim=-1).float()
    fb = torch.zeros_like(full)
    for t in fjdist.tiles_of_rank(n_tiles, rank, world):
        x0, y0 = (t % nx) * tw, (t // nx) * th
        fb[y0:y0 + th, x0:x0 + tw] = full[y0:y0 + th, x0:x0 + tw]
    frame = fjdist.gather_frame(fb, n_tiles, tw, th, rank, world, mode=mode)
    if rank == 0:
        assert torch.equal(frame, full)
        np.save(out_path, frame.numpy())
    else:
        assert frame is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,world", [("gather", 2), ("send_recv", 2), ("all_gather", 2), ("gather", 3)])
def test_gather_frame_gloo(tmp_path, mode, world):
    """every spelling of the one exchange (gather / point-to-point / all_gather), world sizes 2 and 3 (ragged deal)"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, port, 54, 100, out, mode), nprocs=world, join=True)
    f = np.load(out)
    assert f.shape == (54, 100, 4) and f[53, 99, 0] == 53 * 1000.0 + 99
