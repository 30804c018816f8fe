"""The N > 1 path on CPU: tile ownership + the per-step gather to rank 0, with
two gloo processes (no GPU compute: each rank fills the pixels it owns with a
known function of the pixel index, as its GPU would render them)."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, width, height, out_path):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    tiles = importlib.import_module("path-tracer_amd.tiles")
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    g = tiles.TileGather(width, height, rank, world, torch.device("cpu"))
    n = width * height
    want = (torch.arange(n * 3, dtype=torch.float32) * 0.5 + 1.0)
    for step in range(2):  # two steps: the buffers are reused
        rgb = torch.zeros(n * 3)
        mine = g.idx[rank]
        rgb.view(-1, 3)[mine] = (want * (step + 1)).view(-1, 3)[mine]
        g.gather(rgb)
        if rank == 0:
            assert torch.equal(rgb, want * (step + 1)), "assembled frame differs"
        else:
            others = torch.ones(n, dtype=torch.bool)
            others[mine] = False
            assert torch.count_nonzero(rgb.view(-1, 3)[others]) == 0  # non-root ranks receive nothing
    # the 8-bit frame (after mipt_pack_rgb8_device) through the same collective, a quarter of the bytes
    want8 = (torch.arange(n * 3, dtype=torch.int64) * 7 % 251).to(torch.uint8)
    for step in range(2):
        rgb8 = torch.zeros(n * 3, dtype=torch.uint8)
        rgb8.view(-1, 3)[g.idx[rank]] = want8.view(-1, 3)[g.idx[rank]]
        g.gather_rgb8(rgb8)
        if rank == 0:
            assert torch.equal(rgb8, want8), "assembled 8-bit frame differs"
    dist.barrier()
    if rank == 0:
        open(out_path, "w").write("ok")
    dist.destroy_process_group()


@pytest.mark.parametrize("width,height", [(1920, 1080), (200, 120)])
def test_two_rank_gather_assembles_the_frame(tmp_path, width, height):
    import torch.multiprocessing as mp
    out = str(tmp_path / "ok.txt")
    mp.spawn(_worker, args=(2, _free_port(), width, height, out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def test_owner_map_properties():
    sys.path.insert(0, ROOT)
    tiles = importlib.import_module("path-tracer_amd.tiles")
    for parts in (1, 2, 4, 8):
        own = tiles.owned_pixel_indices(1920, 1080, parts)
        allpix = np.sort(np.concatenate(own))
        assert np.array_equal(allpix, np.arange(1920 * 1080))  # a partition: every pixel exactly once
        sizes = np.array([len(o) for o in own])
        assert sizes.max() - sizes.min() <= 0.01 * sizes.mean()  # balanced to 1 %
    m = tiles.tile_owner_map(1920, 1080, 8)
    # interleave: every 8 consecutive tiles of a tile row go to 8 different GPUs, and rows are shifted
    assert len(set(m[0, ::64][:8].tolist())) == 8 and m[0, 0] != m[8, 0]


@pytest.mark.gpu
def test_python_owner_map_equals_library_partition(B, sf):
    """the gather indices (tiles.py) and the pixels the C library renders for a part are the same set"""
    tiles = importlib.import_module("path-tracer_amd.tiles")
    scene = sf.product_scene_from_mesh_arrays(*sf.load_scene_arrays("cornellbox"), env=sf.procedural_env(32))
    ctx = B.Context(scene)
    w, h, parts = 200, 120, 3
    cam = B.default_camera(w, h)
    own = tiles.owned_pixel_indices(w, h, parts)
    for r in range(parts):
        img = ctx.render(cam, mode=B.MODE_PRIMARY, part_index=r, part_count=parts)
        wrote = np.flatnonzero(np.any(img.reshape(-1, 3) != 0, axis=1))
        assert np.array_equal(wrote, own[r])  # every pixel sees box or env, so nothing renders as exact zero
