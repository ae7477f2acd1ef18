"""Row tiling of the Canvas across ranks (one process per GPU) and the gather of tiles to rank 0.

The path shards over pixels with no data dependency (camera.rs:151-156: every pixel depends only
on the read-only World), so each rank renders a contiguous band of rows and the only exchange is
one gather of the bands into the reference's row-major Canvas layout (canvas.rs:44) on rank 0 —
`torch.distributed.gather`, which is RCCL send/recv over xGMI with backend "nccl" on ROCm.

Two ways to cut the rows:
* contiguous ranges (`row_range`): rank r owns rows [r*ceil(H/N), ...). The gather lands every tile
  at its final place, but the ranks' work is uneven (sky at the top, everything else below) and
  the job runs at the pace of the slowest rank.
* interleaved bands (`bands_of_rank`, the default of bench.py): the canvas is cut into bands of 8
  rows (the kernel's tile height) dealt round-robin, rank r owns bands r, r+N, r+2N, ... and
  renders them packed into one buffer (`rtc_render_bands`); every rank gets an even share of the
  image. After the gather rank 0 un-deals the bands with one strided device copy (`deinterleave`).
"""
from __future__ import annotations

BAND_ROWS = 8  # RTC_BAND_ROWS, include/rtc.h


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


# ---- interleaved bands --------------------------------------------------------------------
def n_bands(height: int) -> int:
    return -(-height // BAND_ROWS)


def bands_per_rank(height: int, world_size: int) -> int:
    """Bands in every rank's packed buffer (ranks that own one band less leave the last slot unused)."""
    return -(-n_bands(height) // world_size)


def bands_of_rank(height: int, world_size: int, rank: int) -> list[int]:
    return list(range(rank, n_bands(height), world_size))


def packed_rows(height: int, world_size: int) -> int:
    """Rows of every rank's packed tile buffer (equal on all ranks, as a gather needs)."""
    return bands_per_rank(height, world_size) * BAND_ROWS


def deinterleave_views(gathered, canvas, world_size: int):
    """(dst, src) such that dst.copy_(src) un-deals the bands: gathered is (world_size *
    packed_rows, W, 3) as the gather delivers it (rank-major), canvas the same shape in image order
    (band b = rank b % N, slot b // N). Build once per buffer pair, copy every frame."""
    per = gathered.shape[0] // (world_size * BAND_ROWS)
    W, ch = gathered.shape[1], gathered.shape[2]
    return (canvas.view(per, world_size, BAND_ROWS, W, ch),
            gathered.view(world_size, per, BAND_ROWS, W, ch).permute(1, 0, 2, 3, 4))


def deinterleave(gathered, canvas, world_size: int):
    """One strided copy on the tensors' device; the image is canvas[:height]."""
    dst, src = deinterleave_views(gathered, canvas, world_size)
    dst.copy_(src)
    return canvas
