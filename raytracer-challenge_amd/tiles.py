"""Row tiling of the Canvas across ranks (one process per GPU) and the gather of tiles to rank 0.

The path shards over pixels with no data dependency (camera.rs:151-156: every pixel depends only
on the read-only World), so each rank renders a contiguous band of rows and the only exchange is
one gather of the bands into the reference's row-major Canvas layout (canvas.rs:44) on rank 0 —
`torch.distributed.gather`, which is RCCL send/recv over xGMI with backend "nccl" on ROCm.
"""
from __future__ import annotations


def rows_per_rank(height: int, world_size: int) -> int:
    return -(-height // world_size)


def row_range(height: int, world_size: int, rank: int) -> tuple[int, int]:
    """Rows [y0, y1) owned by `rank`: contiguous bands of ceil(H / world_size) rows."""
    per = rows_per_rank(height, world_size)
    y0 = min(height, rank * per)
    return y0, min(height, y0 + per)


def band_views(canvas, world_size: int):
    """The per-rank bands of the gathered canvas as a list of views (build once, reuse every frame)."""
    return list(canvas.chunk(world_size, dim=0))


def gather_tiles(tile, canvas, world_size: int, rank: int, group=None, async_op: bool = False, bands=None):
    """Gather every rank's (rows_per_rank, W, 3) tile into `canvas` ((world_size*rows_per_rank, W, 3))
    on rank 0. Bands are contiguous, so the gather lands each tile at its final place. With
    async_op=True returns the work handle (wait() before reusing `tile` / reading `canvas`).
    `bands` = band_views(canvas, world_size) prebuilt by the caller (saves host time per frame)."""
    import torch.distributed as dist

    if rank == 0:
        return dist.gather(tile, bands if bands is not None else band_views(canvas, world_size), dst=0, group=group, async_op=async_op)
    return dist.gather(tile, None, dst=0, group=group, async_op=async_op)


def assemble(canvas, height: int):
    """The full Canvas: the first `height` rows of the gathered buffer (the last band may be short)."""
    return canvas[:height]
