#!/usr/bin/env python3
"""bench.py — the hot path's headline metric on MI355X.

Metric (BASELINE.json): Mrays/sec (primary+shadow), 1920x1080 x 100 spheres; 1/2/4/8 MI355X.
A step = one frame: Camera::render_async over the synthetic N-sphere scene (SURVEY.md §8d,
SplitMix64 seed 13) with the World and the Canvas tile resident in HBM. One launch renders 8
consecutive frames (rtc_render_views; --views-per-launch 1 makes a launch a frame). With --gpus N
the frame is row-tiled (8-row bands dealt round-robin, rank r renders bands r, r+N, ...), the tiles
of 32 frames are gathered to rank 0 with one RCCL gather and un-dealt there — total work is fixed,
so scaling is "strong". DESIGN.md §6/§7 explain every one of these choices with measurements.

Prints ONE JSON line on rank 0. `roofline` is the HBM view the north star asks for (algorithmic
bytes / kernel time vs 8 TB/s). `valu_roofline` prices the reference's brute-force arithmetic
(54 f64 flop per ray x sphere) against 39.3 T f64-instr/s (FMA contraction off); the
culled kernel skips most of that arithmetic, so this figure can exceed 1 — it is the algorithmic
rate delivered, not a hardware utilisation.
`cpu_baseline` times the CPU oracle (a port of the Rust path; the Rust sources cannot be built
here) on the host cores, rank 0, N=1 only.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path[:0] = [str(ROOT), str(ROOT / "oracle")]

def baseline_metric():
    """BASELINE.json's metric string (the headline the line is checked against)."""
    try:
        return json.loads((ROOT / "BASELINE.json").read_text())["metric"]
    except Exception:
        return "Mrays/sec (primary+shadow), 1920\u00d71080 \u00d7 100 spheres; 1/2/4/8 MI355X"


HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TINSTR = 39.3      # 78.6 TFLOP/s FP64 vector counts FMA as 2; contraction is off here


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spheres", type=int, default=100)
    ap.add_argument("--no-plane", action="store_true")
    ap.add_argument("--reflective", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline budget (bounded sample)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo is a rehearsal aid for boxes with fewer GPUs than ranks (tiles hop through host memory)")
    ap.add_argument("--no-overlap", action="store_true", help="wait for each gather before rendering the next frame")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the distributed code path (process group, gather, reductions) even with one rank: "
                         "exercises RCCL on a 1-GPU box")
    ap.add_argument("--gather", default="u8", choices=["u8", "f64"],
                    help="what rank 0 collects: the 8-bit frame (Color::scale, 3 B/pixel - what every file writer of the "
                         "reference consumes) or the raw f64 canvas (24 B/pixel; xGMI-ingest bound at this frame size)")
    ap.add_argument("--frames-per-exchange", type=int, default=32,
                    help="N>1: how many frames' tiles each rank sends per RCCL gather (1 = a gather per frame)")
    ap.add_argument("--streams", type=int, default=0,
                    help="frames in flight: consecutive frames are launched round-robin on this many HIP streams, so a "
                         "launch that cannot fill the chip (a rank's 1/N of the frame) overlaps the next one. "
                         "0 = 1 stream for the single-GPU run (clean per-kernel timing), 2 for N > 1 "
                         "(if two are measured to run side by side)")
    ap.add_argument("--pipelined-probe", action="store_true",
                    help="single GPU: after the measured run, also time the same frames with two in flight on two HIP "
                         "streams and report it as the informational `pipelined` block (off by default so that a "
                         "profiler sees only the measured run's launches)")
    ap.add_argument("--views-per-launch", type=int, default=0,
                    help="how many consecutive frames one launch renders (rtc_render_views; every frame of this static "
                         "benchmark has the same camera, an animation would pass its camera path). Default 8: the "
                         "launch overheads, the ramp-up and the tail of a launch are paid once per 8 frames, and a "
                         "rank's launch stays a whole frame's worth of work at 8 GPUs. 1 = a launch is a frame. "
                         "Needs --tiling bands when the exchange is on")
    ap.add_argument("--time-every", type=int, default=0,
                    help="take the kernel duration on every n-th launch of each stream (a launch's start/stop events "
                         "cost ~9 us of host and ~5 us of GPU time). 0 = every launch on one GPU (what rocprofv3's "
                         "kernel trace is compared with), every 8th for N > 1 (launch-bound)")
    ap.add_argument("--tiling", default="bands", choices=["bands", "rows"],
                    help="how the rows are cut across ranks: 8-row bands dealt round-robin (even work per rank; rank 0 "
                         "un-deals them after the gather) or one contiguous range of rows per rank")
    return ap.parse_args()


def algorithmic_bytes(W, rows, world):
    """24 B per pixel written once + the scene read once."""
    n = len(world)
    n_pat = sum(1 for s in world.shapes if s.material.pattern_kind != 0)
    return 24 * W * rows + 400 * n + 128 * n_pat + 200


def measured_traffic(workload_prefix, frames_per_launch=1):
    """HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 cannot be run
    from inside the timed process); newest summary whose workload matches, else None."""
    best = None
    for f in sorted((ROOT / "profiles").glob("r*_pmc.json")):
        try:
            d = json.loads(f.read_text())
            if workload_prefix.startswith(d.get("workload", "\0")) and d.get("frames_per_launch", 1) == frames_per_launch:
                best = (d["hbm_traffic"]["traffic_bytes"], f"profiles/{f.name}")
        except Exception:
            continue
    return best


def algorithmic_flops(world, rays_total, hits):
    """54 f64 flop per (ray, sphere), 33 per (ray, plane), ~200 per hit."""
    per_ray = sum(54 if s.kind == 0 else 33 if s.kind == 1 else 60 for s in world.shapes)
    return rays_total * per_ray + hits * 200


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world_size:
        if world_size == 1 and args.gpus > 1:
            sys.exit(f"--gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py ...`")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no HIP device is visible (there is no CPU fallback for the render path)")
    dev_index = local_rank % torch.cuda.device_count()  # == local_rank on a full node
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    gloo = args.dist_backend == "gloo"
    dist_on = world_size > 1 or args.force_dist
    if args.force_dist and world_size == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if gloo:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm

    from _bootstrap import package
    rtc = package()
    scenes = importlib.import_module(rtc.__name__ + ".scenes")
    tiles = importlib.import_module(rtc.__name__ + ".tiles")

    W, H = args.width, args.height
    world, cam = scenes.synthetic(args.spheres, W, H, with_plane=not args.no_plane, reflective=args.reflective)
    banded = args.tiling == "bands" and (world_size > 1 or args.force_dist)
    y0, y1 = tiles.row_range(H, world_size, rank)
    V = args.views_per_launch if args.views_per_launch > 0 else (8 if (banded or not dist_on) else 1)
    V = max(1, min(V, 8))
    if V > 1 and dist_on and not banded:
        sys.exit("--views-per-launch > 1 needs --tiling bands when the exchange is on")
    rows_max = tiles.packed_rows(H, world_size) if (banded or V > 1) else tiles.rows_per_rank(H, world_size)
    V = max(1, min(V, int(4e9 // (rows_max * W * 24))))     # a launch's V f64 tiles stay within ~4 GB (C5-sized frames: V = 2)
    rows_mine = (sum(min(8, H - 8 * b) for b in tiles.bands_of_rank(H, world_size, rank)) if banded else y1 - y0)

    # Stream 0 is torch's current stream (RCCL orders against it); further streams carry every S-th
    # frame. One context (= one stream, one event ring, one counter block) per stream, the World uploaded
    # into each (it is ~50 KB).
    S = args.streams if args.streams > 0 else (2 if dist_on else 1)
    stream = torch.cuda.current_stream(dev)
    # HIP multiplexes streams onto 4 hardware queues (GPU_MAX_HW_QUEUES; raising it made things slower
    # here): a second render stream easily lands on the queue of the first and the two then serialise.
    # So create a few candidates and MEASURE which of them run beside stream 0 (pick_overlapping_streams).
    pool = 1 if S == 1 else S + 3
    streams = [stream] + [torch.cuda.Stream(dev) for _ in range(pool - 1)]
    ctxs = [rtc.Context(dev_index, stream=st_.cuda_stream) for st_ in streams]
    dworlds = [c.upload(world) for c in ctxs]
    ctx, dworld = ctxs[0], dworlds[0]
    everything = list(zip(dworlds, ctxs))   # closed at the end, whichever streams end up in use
    # The exchange: K frames' tiles per rank go out in ONE gather (fewer, larger collectives: a gather
    # costs ~35 us of fixed enqueue / cross-stream work, half a frame at this size; one rank's 1/8 of the
    # frame: 20.4 us per frame at K = 8, 16.2 at 16, 14.3 at 32, 13.5 at 64, 12.9 without any exchange). Two batch buffers:
    # the RCCL gather of batch j runs (on RCCL's own stream) while batch j+1 renders; a buffer is reused
    # only after the gather that reads or fills it has completed.
    K = max(1, args.frames_per_exchange) if dist_on else S * V   # frames in flight never share an output slot
    if dist_on:                     # a batch keeps K f64 tiles (+ their 8-bit frames) per buffer: stay within ~2 GB
        K = max(1, min(K, int(2e9 // (rows_max * W * 24))))
        V = min(V, K)
        K -= K % V                  # whole launches per batch
    nbuf = 1 if (not dist_on or args.no_overlap) else 2
    gdev = torch.device("cpu") if gloo else dev
    # every step renders the f64 canvas tile (resident in HBM, Canvas::get_pixel semantics) AND its
    # 8-bit quantisation; the exchange moves one of the two
    gdtype = torch.uint8 if args.gather == "u8" else torch.float64
    tile_bufs = [torch.zeros((K, rows_max, W, 3), dtype=torch.float64, device=dev) for _ in range(nbuf)]
    tile8_bufs = [torch.zeros((K, rows_max, W, 3), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    root = rank == 0 and dist_on
    # gather destination on rank 0: rank-major, (N, K, rows_max, W, 3) flattened over the first three axes
    canvases = [torch.empty((world_size * K * rows_max, W, 3), dtype=gdtype, device=gdev) if root else None for _ in range(nbuf)]
    bands = [tiles.band_views(c, world_size) if c is not None else None for c in canvases]
    # rank 0 turns what the gather delivers into K row-major frames with one strided device copy
    # (not needed when K == 1 and the ranks own contiguous rows: the gather then lands every tile in place)
    need_copy = root and (banded or K > 1)
    frames = [torch.empty((K, world_size * rows_max, W, 3), dtype=gdtype, device=gdev) for _ in range(nbuf)] if need_copy else None
    if need_copy:
        per = rows_max // tiles.BAND_ROWS
        if banded:   # (N, K, per, 8) -> (K, per, N, 8): frame-major, bands back in image order
            undeal = [(f.view(K, per, world_size, tiles.BAND_ROWS, W, 3),
                       c.view(world_size, K, per, tiles.BAND_ROWS, W, 3).permute(1, 2, 0, 3, 4, 5)) for c, f in zip(canvases, frames)]
        else:        # (N, K, rows) -> (K, N, rows)
            undeal = [(f.view(K, world_size, rows_max, W, 3), c.view(world_size, K, rows_max, W, 3).permute(1, 0, 2, 3, 4))
                      for c, f in zip(canvases, frames)]
    pending = [None] * nbuf
    state = {"k": 0, "pend": 0, "launches": 0}
    cam_arrays = {n_: (type(cam) * n_)(*([cam] * n_)) for n_ in range(1, V + 1)}   # the frames' cameras, per launch size

    def step():
        k = state["k"]
        state["k"] = k + 1
        slot, b = k % K, (k // K) % nbuf
        if slot == 0:
            if pending[b] is not None:
                finish(b)           # stream 0 waits for the gather that last used batch buffer b ...
            for st_ in streams[1:]:
                st_.wait_stream(stream)   # ... and the other streams wait for stream 0 (once per batch)
        state["pend"] += 1
        if state["pend"] == V or slot == K - 1:
            launch(b, slot)
        if dist_on and slot == K - 1:
            exchange(b)

    def launch(b, last_slot):
        """One launch for the frames accumulated since the last one (V of them, fewer at a batch's end)."""
        cnt = state["pend"]
        if cnt == 0:
            return
        state["pend"] = 0
        lane = state["launches"] % len(dworlds)
        state["launches"] += 1
        # single GPU: every stream owns V output slots (frames on one stream are ordered); with the exchange
        # on, a frame's slot is its place in the batch
        first = last_slot - cnt + 1 if dist_on else lane * V
        tile, tile8 = tile_bufs[b][first], tile8_bufs[b][first]
        if V == 1:
            render_on(lane, tile, tile8)
        else:
            dworlds[lane].render_views(cam_arrays[cnt], rank if banded else 0, world_size if banded else 1, tile.data_ptr(),
                                       rows_max, rtc.MODE_RENDER_ASYNC, d_ptr8=tile8.data_ptr())

    # rank 0's un-deal copy (a batch's worth of frames, read + written once) runs on its own stream so that
    # rank 0's renders do not queue behind it: it depends on the gather only. (Per batch, not per frame: the
    # extra host calls no longer matter.)
    copy_stream = torch.cuda.Stream(dev) if (need_copy and not gloo) else None

    def exchange(b):
        for st_ in streams[1:]:
            stream.wait_stream(st_)       # the gather (ordered after stream 0) needs every frame of the batch
        if copy_stream is not None:
            stream.wait_stream(copy_stream)   # ... and overwrites canvases[b]: its last un-deal copy must be done
        src = (tile8_bufs[b] if args.gather == "u8" else tile_bufs[b]).view(K * rows_max, W, 3)
        src = src.cpu() if gloo else src
        work = tiles.gather_tiles(src, canvases[b], world_size, rank, async_op=not args.no_overlap, bands=bands[b])
        if work is not None:
            pending[b] = work
        elif need_copy:
            undeal[b][0].copy_(undeal[b][1])

    def finish(b):
        work, pending[b] = pending[b], None
        work.wait()                       # stream 0: the batch's tile buffer is rendered into next
        if need_copy:
            if copy_stream is not None:
                with torch.cuda.stream(copy_stream):
                    work.wait()
                    undeal[b][0].copy_(undeal[b][1])
            else:
                undeal[b][0].copy_(undeal[b][1])

    def render_on(i, tile, tile8):
        if banded:
            dworlds[i].render_bands(cam, rank, world_size, tile.data_ptr(), rtc.MODE_RENDER_ASYNC, d_ptr8=tile8.data_ptr())
        else:
            dworlds[i].render_rows(cam, y0, y1, tile.data_ptr(), rtc.MODE_RENDER_ASYNC, d_ptr8=tile8.data_ptr())

    def pick_overlapping_streams(S=S):
        """Keep S of the candidate streams: stream 0 plus those whose launches really run beside
        stream 0's (two streams that share a hardware queue serialise). Measured, not assumed: pairs of
        this rank's own launches on (stream 0, candidate) against pairs on stream 0 alone."""
        if S == 1 or len(streams) == 1:
            return 1
        ta = [torch.zeros((rows_max, W, 3), dtype=torch.float64, device=dev) for _ in range(2)]
        tq = [torch.zeros((rows_max, W, 3), dtype=torch.uint8, device=dev) for _ in range(2)]

        def pairs2(a, b, reps=24):
            best = float("inf")
            for _ in range(3):
                torch.cuda.synchronize(dev)
                t = time.perf_counter()
                for _ in range(reps):
                    render_on(a, ta[0], tq[0])
                    render_on(b, ta[1], tq[1])
                torch.cuda.synchronize(dev)
                best = min(best, time.perf_counter() - t)
            return best

        pairs2(0, 0)
        serial = pairs2(0, 0)
        ratio0 = {j: pairs2(0, j) / serial for j in range(1, len(streams))}
        keep = [0]
        for j in sorted(ratio0, key=ratio0.get):        # greedy: a stream joins if it overlaps with every one kept so far
            if len(keep) == S:
                break
            if ratio0[j] < 0.95 and all(pairs2(x, j) / serial < 0.95 for x in keep[1:]):
                keep.append(j)
        if os.environ.get("RTC_BENCH_DEBUG"):
            print(f"rank {rank}: serial pair {serial * 1e3:.3f} ms; candidate/serial vs stream 0: "
                  + ", ".join(f"{j}:{r:.2f}" for j, r in sorted(ratio0.items())) + f"; kept {keep}", file=sys.stderr)
        streams[:] = [streams[j] for j in keep]
        ctxs[:] = [ctxs[j] for j in keep]
        dworlds[:] = [dworlds[j] for j in keep]
        return len(keep)

    def drain():
        k = state["k"]
        state["last"] = k - 1
        if state["pend"]:            # frames handed to step() but not launched yet
            launch(((k - 1) // K) % nbuf, (k - 1) % K)
        if k % K != 0:              # a partly filled batch: exchange it as it is (the unused slots carry old frames)
            if dist_on:
                exchange(((k - 1) // K) % nbuf)
            state["k"] = k + (K - k % K)   # the next frame starts a batch (and a launch)
        for b in range(nbuf):
            if pending[b] is not None:
                finish(b)

    for _ in range(V):              # first-use costs (code object load, communicator set-up) never land in the
        step()                      # timed region, whatever --warmup is: one launch of the usual size ...
    drain()                         # ... and, with the exchange on, the (partly filled) batch it belongs to
    torch.cuda.synchronize(dev)
    in_flight = pick_overlapping_streams()   # streams (= frames in flight) actually used from here on
    for _ in range(args.warmup):
        step()
    drain()
    time_every = args.time_every if args.time_every > 0 else (8 if dist_on else 1)
    launches_before = state["launches"]
    for c in ctxs:
        c.reset_stats()
        c.set_timing(time_every)   # from here on: the timed region's launches only
    # kernel duration: every launch carries its own pair of HIP events on the launch stream
    # (hipExtLaunchKernel start/stop events inside rtc_render_rows, read back after the timed region),
    # on every --time-every-th launch
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(args.steps):
        step()
    drain()
    torch.cuda.synchronize(dev)
    if dist_on:
        dist.barrier()
    elapsed = time.perf_counter() - t0

    st = {}
    for c in ctxs:
        for key, v in c.stats().items():
            st[key] = st.get(key, 0) + v
    # the timed steps (the newest 1024 per stream if more)
    times = np.concatenate([c.kernel_times_ms(1024) for c in ctxs])   # the sampled launches of the timed region
    kernel_ms = float(times.mean()) if len(times) else 0.0
    last_ms = float(times[-1]) if len(times) else 0.0
    # Single GPU, for information only (the line's value / roofline stay those of the one-stream run, whose
    # per-kernel durations are what rocprofv3 shows): the same frames with two in flight on two HIP streams,
    # so that one launch's tail and the next one's ramp-up overlap.
    pipelined = None
    if not dist_on and S == 1 and args.pipelined_probe:
        extra = [torch.cuda.Stream(dev) for _ in range(4)]
        streams.extend(extra)
        ctxs.extend(rtc.Context(dev_index, stream=st_.cuda_stream) for st_ in extra)
        dworlds.extend(c.upload(world) for c in ctxs[1:])
        everything.extend(zip(dworlds[1:], ctxs[1:]))
        if pick_overlapping_streams(2) == 2:
            outs = [(torch.zeros((rows_max, W, 3), dtype=torch.float64, device=dev), torch.zeros((rows_max, W, 3), dtype=torch.uint8, device=dev))
                    for _ in range(2)]
            for k in range(args.warmup):
                render_on(k % 2, *outs[k % 2])
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for k in range(args.steps):
                render_on(k % 2, *outs[k % 2])
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t1
            pipelined = {"streams": 2, "ms_per_step": round(dt / max(1, args.steps) * 1e3, 4),
                         "value": round((st["rays_primary"] + st["rays_shadow"]) / dt / 1e6, 3), "unit": "Mrays/s",
                         "note": "two frames in flight on two HIP streams (bench.py --streams 2 makes this the measured run)"}
    # outside the timed region: the last frame rank 0 assembled from the gathered tiles must be the
    # frame one GPU renders on its own, bit for bit
    exchange_check = None
    if root and state.get("last", -1) >= 0:
        k = state["last"]
        slot, b = k % K, (k // K) % nbuf
        got = (frames[b][slot] if need_copy else canvases[b])[:H]
        ref = torch.zeros((H, W, 3), dtype=torch.float64, device=dev)
        ref8 = torch.zeros((H, W, 3), dtype=torch.uint8, device=dev)
        dworld.render_rows(cam, 0, H, ref.data_ptr(), rtc.MODE_RENDER_ASYNC, d_ptr8=ref8.data_ptr())
        torch.cuda.synchronize(dev)
        want = ref8 if args.gather == "u8" else ref
        exchange_check = "ok" if torch.equal(got.to(want.device), want) else "MISMATCH"
    agg = torch.tensor([elapsed, float(st["rays_primary"]), float(st["rays_shadow"]), float(st["rays_reflect"] + st["rays_refract"]),
                        kernel_ms], dtype=torch.float64, device=torch.device("cpu") if gloo else dev)
    if dist_on:
        tmax = agg[[0, 4]].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        elapsed, kernel_ms_max = float(tmax[0]), float(tmax[1])
    else:
        kernel_ms_max = kernel_ms
    rays_ps = float(agg[1] + agg[2])          # primary + shadow, whole job, over the timed steps
    rays_other = float(agg[3])

    if rank == 0:
        steps = max(1, args.steps)
        value = rays_ps / elapsed / 1e6
        rows = rows_mine
        frames_per_launch = args.steps / max(1, state["launches"] - launches_before)   # V, or a little less if steps % V != 0
        abytes = algorithmic_bytes(W, rows * frames_per_launch, world)   # per LAUNCH: that many frames' worth of this rank's rows, the scene once
        rays_rank = (st["rays_primary"] + st["rays_shadow"] + st["rays_reflect"] + st["rays_refract"]) / steps
        hits_rank = st["rays_shadow"] / steps  # one shadow ray per shaded hit (shape.rs:688)
        aflops = algorithmic_flops(world, rays_rank, hits_rank) * frames_per_launch     # per launch, like kernel_ms
        workload = (f"{W}x{H}, {args.spheres} spheres" + ("" if args.no_plane else " + checker floor plane") +
                    ", 1 point light, render_async, SplitMix64 seed 13" + (", reflective depth 5" if args.reflective else ""))
        traffic = measured_traffic(workload, V) if world_size == 1 else None
        out = {
            "metric": baseline_metric(),
            "value": round(value, 3),
            "unit": "Mrays/s",
            "n_gpus": world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": workload,
                "objects": len(world), "rows_per_gpu": rows, "frames_per_launch": V, "parallelism": (f"{'8-row bands dealt round-robin' if banded else 'contiguous row tiles'} x{world_size} + {'gloo (rehearsal)' if gloo else 'RCCL'} gather of the "
                                                                          f"{'8-bit frame (Color::scale)' if args.gather == 'u8' else 'f64 canvas'} to rank 0"
                                                                          + (f", one gather per {K} frames" if K > 1 else "")
                                                                          + (f", {in_flight} frames in flight (HIP streams measured to run side by side)" if in_flight > 1 else "")
                                                                          + ("" if args.no_overlap else ", gather of batch j overlapped with the renders of batch j+1")) if dist_on else ("single GPU" + (f", {in_flight} frames in flight (HIP streams measured to run side by side)" if in_flight > 1 else "")),
                "exchange_bytes_per_frame": (W * H * (3 if args.gather == "u8" else 24) * (world_size - 1) // world_size) if dist_on else 0,
                "rays_per_frame_primary_shadow": int(round(rays_ps / steps)), "rays_per_frame_other": int(round(rays_other / steps)),
            },
            "roofline": {
                "bound": "hbm", "kernel": "k_trace", "achieved": round(abytes / (kernel_ms * 1e-3) / 1e9, 3), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(abytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                "traffic": int(traffic[0]) if traffic else None, "traffic_source": traffic[1] if traffic else None,
                "algorithmic_bytes_per_launch": int(abytes), "kernel_ms_avg": round(kernel_ms, 5), "kernel_ms_last_launch": round(last_ms, 5),
                "kernel_launches_timed": int(len(times)), "kernel_timed_every": time_every,
                "note": "one launch writes the f64 canvas tile once and reads the ~50 KB scene; the kernel is f64-VALU/latency bound, see DESIGN.md",
            },
            "valu_roofline": {
                "bound": "fp64_valu_no_fma", "achieved": round(aflops / (kernel_ms * 1e-3) / 1e12, 4), "peak": FP64_PEAK_TINSTR,
                "unit": "T f64-instr/s", "frac": round(aflops / (kernel_ms * 1e-3) / 1e12 / FP64_PEAK_TINSTR, 5),
                "algorithmic_flops_per_launch": int(aflops),
            },
            "device": ctx.device_info(),
        }
        if world_size > 1:
            out["config"]["kernel_ms_max_over_ranks"] = round(kernel_ms_max, 5)
        if exchange_check is not None:
            out["config"]["gathered_frame_vs_single_gpu_render"] = exchange_check
        if pipelined is not None:
            out["pipelined"] = pipelined
        if world_size == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(world, cam, args.cpu_seconds)
        print(json.dumps(out), flush=True)

    for d_, c in everything:
        d_.close()
        c.close()
    if dist_on:
        dist.destroy_process_group()


def cpu_baseline(world, cam, budget_s):
    """The CPU oracle (a C port of the Rust path, literal sorted-list form) on the host cores,
    same scene; bounded sample: as many whole frames (or one band of rows) as fit the budget."""
    import oracle as O
    O.build()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 64))
    arr = world.array()
    H = cam.vsize
    # probe: a thin band to estimate the rate, then size the sample
    band = max(1, H // 60)
    t = time.perf_counter()
    _, st = O.render(arr, len(world), world.light, cam, mode=1, y0=H // 2, y1=H // 2 + band, nthreads=cores, want_stats=True)
    dt = max(1e-6, time.perf_counter() - t)
    est_frame = dt * H / band
    if est_frame <= budget_s:
        t = time.perf_counter()
        rays = frames = 0
        while frames == 0 or (time.perf_counter() - t) + est_frame <= budget_s:   # whole frames until the budget is used
            t1 = time.perf_counter()
            _, st = O.render(arr, len(world), world.light, cam, mode=1, nthreads=cores, want_stats=True)
            est_frame = time.perf_counter() - t1
            rays += st["rays_primary"] + st["rays_shadow"]
            frames += 1
        dt = time.perf_counter() - t
        sample = f"{frames} full frame(s) {cam.hsize}x{cam.vsize}"
    else:
        rows = max(band, int(H * budget_s / est_frame))
        ya = (H - rows) // 2
        t = time.perf_counter()
        _, st = O.render(arr, len(world), world.light, cam, mode=1, y0=ya, y1=ya + rows, nthreads=cores, want_stats=True)
        dt = time.perf_counter() - t
        rays = st["rays_primary"] + st["rays_shadow"]
        sample = f"rows [{ya},{ya + rows}) of {cam.hsize}x{cam.vsize} (centre band)"
    out = {"value": round(rays / dt / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port", "sample": sample,
           "seconds": round(dt, 2), "form": "literal sorted-list oracle (oracle/rtc_oracle.c), f64, -O2 -ffp-contract=off"}

    # BASELINE.md §3's two other variants, on small bounded samples (a band of rows each, ~2 s):
    # the literal form on ONE thread (analogue of Camera::render) and the streaming form on all cores
    def band_rate(nthreads, streaming, seconds):
        rows = max(1, int(H * seconds / max(est_frame * (cores / nthreads if not streaming else 1.0), 1e-6)))
        rows = min(rows, H)
        ya = (H - rows) // 2
        t = time.perf_counter()
        rays_, reps = 0, 0
        while reps == 0 or (rows == H and time.perf_counter() - t < seconds):   # whole frames: repeat until the time is used
            _, st_ = O.render(arr, len(world), world.light, cam, mode=1, y0=ya, y1=ya + rows, nthreads=nthreads, streaming=streaming,
                              want_stats=True)
            rays_ += st_["rays_primary"] + st_["rays_shadow"]
            reps += 1
        d = time.perf_counter() - t
        return {"value": round(rays_ / d / 1e6, 4), "unit": "Mrays/s", "cores": nthreads,
                "sample": (f"{reps} full frame(s)" if rows == H else f"rows [{ya},{ya + rows}) of {cam.hsize}x{cam.vsize}"), "seconds": round(d, 2)}

    out["literal_1_thread"] = band_rate(1, False, 2.0)
    out["streaming_all_cores"] = band_rate(cores, True, 2.0)
    return out


if __name__ == "__main__":
    main()
