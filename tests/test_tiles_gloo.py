"""The N>1 path on CPU: two ranks (gloo), each renders its band of rows (with the oracle standing
in for the GPU renderer, which cannot run here), tiles gathered by the product's tiles.gather_tiles,
rank 0 compares with the full frame."""
import importlib
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def _worker(rank, world_size, port, H, W, out_path):
    sys.path[:0] = [str(ROOT), str(ROOT / "oracle")]
    import torch
    import torch.distributed as dist
    import oracle as O
    from _bootstrap import package
    rtc = package()
    scenes = importlib.import_module(rtc.__name__ + ".scenes")
    tiles = importlib.import_module(rtc.__name__ + ".tiles")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    w, cam = scenes.synthetic(12, W, H)
    y0, y1 = tiles.row_range(H, world_size, rank)
    per = tiles.rows_per_rank(H, world_size)
    tile = torch.zeros((per, W, 3), dtype=torch.float64)
    tile[: y1 - y0] = torch.from_numpy(O.render(w.array(), len(w), w.light, cam, mode=1, y0=y0, y1=y1))
    canvas = torch.empty((world_size * per, W, 3), dtype=torch.float64) if rank == 0 else None
    tiles.gather_tiles(tile, canvas, world_size, rank)
    if rank == 0:
        full = O.render(w.array(), len(w), w.light, cam, mode=1)
        np.save(out_path, np.array([np.array_equal(tiles.assemble(canvas, H).numpy(), full)]))
    dist.barrier()
    dist.destroy_process_group()


def _worker_bands(rank, world_size, port, H, W, out_path):
    """Interleaved 8-row bands: rank r renders bands r, r+N, ... packed (the oracle stands in for
    rtc_render_bands), gather, rank 0 un-deals them (tiles.deinterleave)."""
    sys.path[:0] = [str(ROOT), str(ROOT / "oracle")]
    import torch
    import torch.distributed as dist
    import oracle as O
    from _bootstrap import package
    rtc = package()
    scenes = importlib.import_module(rtc.__name__ + ".scenes")
    tiles = importlib.import_module(rtc.__name__ + ".tiles")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    w, cam = scenes.synthetic(12, W, H)
    per = tiles.packed_rows(H, world_size)
    tile = torch.zeros((per, W, 3), dtype=torch.float64)
    for k, b in enumerate(tiles.bands_of_rank(H, world_size, rank)):
        y0, y1 = 8 * b, min(H, 8 * b + 8)
        tile[8 * k: 8 * k + (y1 - y0)] = torch.from_numpy(O.render(w.array(), len(w), w.light, cam, mode=1, y0=y0, y1=y1))
    gathered = torch.empty((world_size * per, W, 3), dtype=torch.float64) if rank == 0 else None
    tiles.gather_tiles(tile, gathered, world_size, rank)
    if rank == 0:
        canvas = tiles.deinterleave(gathered, torch.empty_like(gathered), world_size)
        full = O.render(w.array(), len(w), w.light, cam, mode=1)
        np.save(out_path, np.array([np.array_equal(tiles.assemble(canvas, H).numpy(), full)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("H", [48, 37, 5])  # even split; a short last band and an odd band count; fewer bands than ranks
def test_two_rank_interleaved_bands_and_gather(tmp_path, H):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = tmp_path / "ok.npy"
    mp.spawn(_worker_bands, args=(2, port, H, 48, str(out)), nprocs=2, join=True)
    assert np.load(out)[0]


def test_bands_partition_the_canvas(rtc):
    tiles = importlib.import_module(rtc.__name__ + ".tiles")
    for H in (1, 7, 8, 9, 1080, 1083, 4096):
        for n in (1, 2, 3, 4, 8):
            owned = sorted(b for r in range(n) for b in tiles.bands_of_rank(H, n, r))
            assert owned == list(range(tiles.n_bands(H)))
            assert all(len(tiles.bands_of_rank(H, n, r)) <= tiles.bands_per_rank(H, n) for r in range(n))
            assert tiles.packed_rows(H, n) * n >= H


@pytest.mark.parametrize("H", [40, 37])  # even split, and a short last band
def test_two_rank_row_tiling_and_gather(tmp_path, H):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = tmp_path / "ok.npy"
    mp.spawn(_worker, args=(2, port, H, 48, str(out)), nprocs=2, join=True)
    assert np.load(out)[0]


def test_row_ranges_partition_the_canvas(rtc):
    tiles = importlib.import_module(rtc.__name__ + ".tiles")
    for H in (1, 7, 8, 1080, 4096, 8192):
        for n in (1, 2, 3, 4, 8):
            rows = [tiles.row_range(H, n, r) for r in range(n)]
            assert rows[0][0] == 0 and rows[-1][1] == H
            assert all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
            assert all(y1 - y0 <= tiles.rows_per_rank(H, n) for y0, y1 in rows)
