"""The N > 1 path on the CPU: real ranks (gloo, world_size 2 / 3 / 8) pack their 8-row bands, ONE gather delivers the
chunks to rank 0 in rank order (what ncclGather does in rtc_group_render), and rank 0 un-deals them with the product's own
index function — csrc/rtc_bands.h, the header rtc_group.cpp and k_undeal are built from, reached through the [host]
entries of include/rtc.h (rtc_group_packed_rows, _bands_owned, _packed_row_to_image, _undeal_host). The oracle stands in
for the render kernel (it cannot run here); which rows a rank renders is decided by the product's arithmetic."""
import importlib
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def _worker(rank, world_size, port, H, W, nframes, use_oracle, out_path):
    sys.path[:0] = [str(ROOT), str(ROOT / "oracle")]
    import torch
    import torch.distributed as dist
    from _bootstrap import package
    rtc = package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    N = world_size
    if use_oracle:
        import oracle as O
        scenes = importlib.import_module(rtc.__name__ + ".scenes")
        w, cam = scenes.synthetic(12, W, H)
        frames = [None] * nframes

        def rows_of(f, y0, y1):   # the oracle stands in for rtc_render_views' kernel on these rows
            return O.render(w.array(), len(w), w.light, cam, mode=1, y0=y0, y1=y1)
        full = np.stack([O.render(w.array(), len(w), w.light, cam, mode=1)] * nframes) if rank == 0 else None
    else:
        yy, xx = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
        frames = [np.stack([f * 1e6 + yy * 1e3 + xx, -(yy + f), xx * 0.5 + f], axis=-1) for f in range(nframes)]

        def rows_of(f, y0, y1):
            return frames[f][y0:y1]
        full = np.stack(frames) if rank == 0 else None
    rows = rtc.group_packed_rows(H, N)                      # rows of every member's packed tile (one gather chunk per frame)
    mine = rtc.group_bands_owned(H, N, rank)
    tile = np.full((nframes, rows, W, 3), -7.0)            # padding must never reach the canvas
    for f in range(nframes):
        for k in range(mine):                               # the k-th band of this member's tile: which image rows?
            y0 = rtc.group_packed_row_to_image(rank, 8 * k, N)
            y1 = min(H, y0 + 8)
            assert y0 == (rank + k * N) * 8 and y0 < H
            tile[f, 8 * k: 8 * k + (y1 - y0)] = rows_of(f, y0, y1)
    t = torch.from_numpy(tile)
    chunks = [torch.empty_like(t) for _ in range(N)] if rank == 0 else None
    dist.gather(t, chunks, dst=0)                           # ONE exchange step, chunk p = member p's tile (ncclGather's layout)
    if rank == 0:
        staging = torch.stack(chunks).numpy()               # (N, nframes, rows, W, 3)
        canvas = rtc.group_undeal_host(staging, N, nframes, H)
        ok = np.array_equal(canvas, full)
        # every image row has exactly one owner, and the forward and inverse maps agree
        seen = np.zeros(H, dtype=np.int64)
        for p in range(N):
            for r in range(8 * rtc.group_bands_owned(H, N, p)):
                y = rtc.group_packed_row_to_image(p, r, N)
                if y < H:
                    seen[y] += 1
                    ok = ok and rtc.group_row_owner(y, N) == (p, r)
        ok = ok and bool((seen == 1).all())
        np.save(out_path, np.array([ok]))
    dist.barrier()
    dist.destroy_process_group()


def _run(tmp_path, N, H, W, nframes, use_oracle):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = tmp_path / "ok.npy"
    mp.spawn(_worker, args=(N, port, H, W, nframes, use_oracle, str(out)), nprocs=N, join=True)
    assert np.load(out)[0]


@pytest.mark.parametrize("H", [48, 37, 5])   # even split; a short last band and an odd band count; fewer bands than ranks
def test_two_ranks_oracle_rows_gather_undeal(tmp_path, H):
    _run(tmp_path, 2, H, 48, 1, True)


@pytest.mark.parametrize("N,H,nframes", [(2, 93, 3), (3, 37, 2), (3, 16, 1), (8, 203, 2), (8, 41, 1)])
def test_ranks_frames_and_short_bands(tmp_path, N, H, nframes):
    """Several frames per batch (the staging index's frame term), 3 and 8 ranks, short last bands, members that own one
    band less than member 0, members that own none."""
    _run(tmp_path, N, H, 24, nframes, False)


def test_bands_partition_the_canvas(rtc):
    for H in (1, 7, 8, 9, 1080, 1083, 4096):
        nb = -(-H // 8)
        for n in (1, 2, 3, 4, 8):
            owned = [rtc.group_bands_owned(H, n, r) for r in range(n)]
            assert sum(owned) == nb and max(owned) * 8 == rtc.group_packed_rows(H, n) and max(owned) - min(owned) <= 1
            assert owned == [len(range(r, nb, n)) for r in range(n)]
            for y in (0, H // 2, H - 1):
                p, r = rtc.group_row_owner(y, n)
                assert rtc.group_packed_row_to_image(p, r, n) == y and r < rtc.group_packed_rows(H, n)
