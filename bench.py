#!/usr/bin/env python3
"""bench.py — the hot path's headline metric on MI355X.

Metric (BASELINE.json): Mrays/sec (primary+shadow), 1920x1080 x 100 spheres; 1/2/4/8 MI355X.
A step = one frame: Camera::render_async (camera.rs:144-160) over a synthetic scene (SURVEY.md §8d,
SplitMix64 seed 13) with the World resident in HBM and the f64 Canvas left in HBM. One launch renders
`--views-per-launch` (default 8) consecutive frames (rtc_render_views).

--gpus N (one process per GPU, torch.distributed.run): the frame is row-tiled behind the C-ABI
(rtc_group, include/rtc.h) — 8-row bands dealt round-robin, member r renders bands r, r+N, ..., the
f64 tiles are gathered to member 0 with ONE RCCL gather per batch (ncclGather over xGMI, issued by
librtc.so itself) and un-dealt there into the reference's row-major Canvas. Total work is fixed, so
scaling is "strong". The headline value for N > 1 is measured WITH the f64 Canvas exchange (the path's
own output, 24 B/pixel); the same run then repeats a shorter timed loop with the 8-bit frame
(Color::scale, 3 B/pixel) and with no exchange, reported as labelled secondary records.

Prints ONE JSON line on rank 0:
  roofline      HBM view: algorithmic bytes of one launch / that launch's duration (HIP events on the
                launch stream) vs 8 TB/s; `traffic` = HBM bytes per launch from the committed PMC profile
  valu_issue    the figure that actually binds: VALU wave-instructions (PMC) x 4 cycles over the SIMD-cycles
                the kernel had — an issue-slot utilisation, always <= 1
  single_view   the drop-in launch shape: ONE camera per launch (Camera::render_async is one camera per call)
  dropin        N = 1: the host-canvas call a Rust caller makes — context create, world upload, rtc_render
                into pageable / registered / page-locked canvases (PCIe-inclusive; never `value`)
  cpu_baseline  the CPU oracle (a C port of the Rust path) on the host cores, N = 1 only.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time
from pathlib import Path

# RCCL's peer transport needs dmabuf IPC on this driver (the legacy mode fails with hipIpcGetMemHandle: invalid argument);
# the pool exports this already — keep it set for every rank whatever launched us. Must precede the first HIP call.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = Path(__file__).resolve().parent
sys.path[:0] = [str(ROOT), str(ROOT / "oracle")]

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SIMDS, CLOCK_HZ = 1024, 2.4e9   # 256 CUs x 4 SIMDs, max clock (MI355X_MICROARCH.md chip table)

# BASELINE.json configs. `ns` = the point the metric is quoted on.
WORKLOADS = {
    "ns": dict(width=1920, height=1080, spheres=100, desc="north star: 1920x1080, 100 spheres + checker floor plane"),
    "c2": dict(width=1920, height=1080, scene="test7", desc="C2: 1920x1080, the reference's test7 scene (main.rs:204-251), 3 spheres + 1 plane"),
    "c3": dict(width=1920, height=1080, spheres=10000, no_plane=True, desc="C3: 1920x1080, 10 000 random spheres"),
    "c4": dict(width=4096, height=4096, spheres=100, reflective=True, desc="C4: 4096x4096, 100 spheres + floor, reflective depth 5"),
    "c5": dict(width=8192, height=8192, spheres=1000, desc="C5: 8192x8192, 1000 spheres + checker floor plane"),
}


def baseline_metric():
    try:
        return json.loads((ROOT / "BASELINE.json").read_text())["metric"]
    except Exception:
        return "Mrays/sec (primary+shadow), 1920×1080 × 100 spheres; 1/2/4/8 MI355X"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="ns", choices=sorted(WORKLOADS), help="BASELINE.json configuration (default: the north-star point)")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spheres", type=int, default=-1)
    ap.add_argument("--no-plane", action="store_true")
    ap.add_argument("--reflective", action="store_true")
    ap.add_argument("--views-per-launch", type=int, default=8,
                    help="consecutive frames one launch renders (rtc_render_views; this static benchmark repeats one camera, "
                         "an animation passes its camera path); clamped so that a launch's canvases stay within ~4 GB")
    ap.add_argument("--exchange", default="f64", choices=["f64", "u8", "none"],
                    help="N > 1: what member 0 collects in the MEASURED run: the f64 Canvas (default, the path's output), "
                         "the 8-bit frame, or nothing. The other two are reported as secondary records")
    ap.add_argument("--force-group", action="store_true",
                    help="take the multi-GPU code path (rtc_group in rank mode, RCCL communicator, gather, un-deal) with one rank")
    ap.add_argument("--time-every", type=int, default=1, help="HIP-event pair on every n-th launch (1 = all)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline budget (bounded sample)")
    ap.add_argument("--no-dropin", action="store_true", help="skip the host-canvas (PCIe-inclusive) drop-in measurements")
    ap.add_argument("--dropin-multi", action="store_true",
                    help="N > 1: also measure the shared host canvas that every rank fills over its own PCIe link (off by default: "
                         "it has only been rehearsed with one rank, and nothing may endanger the headline line of a multi-GPU run)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the single-view / other-exchange secondary runs")
    ap.add_argument("--lean", action="store_true", help="= --no-cpu-baseline --no-dropin --no-secondary (profiling runs)")
    a = ap.parse_args()
    if a.lean:
        a.no_cpu_baseline = a.no_dropin = a.no_secondary = True
    return a


def build_scene(args, scenes):
    cfg = dict(WORKLOADS[args.workload])
    if args.width:
        cfg["width"] = args.width
    if args.height:
        cfg["height"] = args.height
    custom = args.spheres >= 0 or args.no_plane or args.reflective
    if custom:
        cfg = dict(width=cfg["width"], height=cfg["height"], spheres=args.spheres if args.spheres >= 0 else cfg.get("spheres", 100),
                   no_plane=args.no_plane, reflective=args.reflective)
    W, H = cfg["width"], cfg["height"]
    if cfg.get("scene") == "test7":
        world, cam = scenes.test7(W, H)
        desc = f"{W}x{H}, reference scene test7 (3 spheres + 1 plane)"
    else:
        world, cam = scenes.synthetic(cfg["spheres"], W, H, with_plane=not cfg.get("no_plane", False), reflective=cfg.get("reflective", False))
        desc = (f"{W}x{H}, {cfg['spheres']} spheres" + ("" if cfg.get("no_plane") else " + checker floor plane") +
                (", reflective depth 5" if cfg.get("reflective") else ""))
    key = args.workload if not (custom or args.width or args.height) else "custom"
    return world, cam, key, desc + ", 1 point light, render_async, SplitMix64 seed 13"


def algorithmic_bytes(W, rows, world):
    """24 B per pixel written once + the scene read once."""
    n = len(world)
    n_pat = sum(1 for s in world.shapes if s.material.pattern_kind != 0)
    return 24 * W * rows + 400 * n + 128 * n_pat + 200


def committed_profile(key):
    """Newest PMC summary under profiles/ for this workload (rocprofv3 cannot run inside the timed process):
    per-FRAME HBM traffic and VALU wave-instructions of the dominant kernel."""
    best = None
    for f in sorted((ROOT / "profiles").glob("r*_pmc.json")):
        try:
            d = json.loads(f.read_text())
            if d.get("workload_key") == key and "per_frame" in d:
                best = (d["per_frame"], f"profiles/{f.name}")
        except Exception:
            continue
    return best


class FrameQueue:
    """Frames are handed over one at a time (a step = a frame); a launch goes out every V frames."""

    def __init__(self, V, launch):
        self.V, self.launch, self.pending, self.sizes = V, launch, 0, []

    def step(self):
        self.pending += 1
        if self.pending == self.V:
            self.flush()

    def flush(self):
        if self.pending:
            self.launch(self.pending)
            self.sizes.append(self.pending)
            self.pending = 0


def timed(fn_step, fn_flush, fn_sync, steps, barrier=None):
    if barrier:
        barrier()
    fn_sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn_step()
    fn_flush()
    fn_sync()
    if barrier:
        barrier()
    return time.perf_counter() - t0


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world_size and world_size == 1 and args.gpus > 1:
        sys.exit(f"--gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py ...`")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no HIP device is visible (there is no CPU fallback for the render path)")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    grouped = world_size > 1 or args.force_group
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)   # nccl == RCCL on ROCm: barrier + reductions of the report

    from _bootstrap import package
    rtc = package()
    scenes = importlib.import_module(rtc.__name__ + ".scenes")
    world, cam, wkey, wdesc = build_scene(args, scenes)
    W, H = cam.hsize, cam.vsize
    N = world_size
    nbands = -(-H // 8)
    rows_max = -(-nbands // N) * 8                      # rows of one member's packed tile (rtc_group)
    rows_mine = sum(min(8, H - 8 * b) for b in range(rank, nbands, N))
    V = max(1, min(args.views_per_launch, 8, int(4e9 // (rows_max * W * 24))))
    cams = {n_: (type(cam) * n_)(*([cam] * n_)) for n_ in range(1, V + 1)}
    barrier = (lambda: dist.barrier()) if grouped else None

    if grouped:
        uid = torch.zeros(rtc.GROUP_ID_BYTES, dtype=torch.uint8, device=dev)
        if rank == 0:
            uid.copy_(torch.tensor(list(rtc.group_unique_id()), dtype=torch.uint8))
        dist.broadcast(uid, 0)
        group = rtc.Group(device=dev_index, nranks=N, rank=rank, uid=bytes(uid.cpu().tolist()))
        ctx = group.contexts[0]
        gworld = group.upload(world)
        # member 0's destination: two batches of V canvases (a consumer reads batch j while batch j+1 is assembled)
        canv = [torch.zeros((V, H, W, 3), dtype=torch.float64, device=dev) for _ in range(2)] if rank == 0 else [None, None]
        canv8 = [torch.zeros((V, H, W, 3), dtype=torch.uint8, device=dev) for _ in range(2)] if rank == 0 else [None, None]
        torch.cuda.synchronize(dev)
        state = {"batch": 0, "what": {"f64": rtc.GATHER_F64, "u8": rtc.GATHER_U8, "none": rtc.GATHER_NONE}[args.exchange]}

        def launch(n):
            b = state["batch"] & 1
            state["batch"] += 1
            gworld.render(cams[n], state["what"], canv[b].data_ptr() if rank == 0 else None,
                          canv8[b].data_ptr() if rank == 0 else None)
        sync = group.synchronize
        stats, reset_stats = group.stats, group.reset_stats
    else:
        ctx = rtc.Context(dev_index, stream=torch.cuda.current_stream(dev).cuda_stream)
        dworld = ctx.upload(world)
        tile = torch.zeros((V * rows_max, W, 3), dtype=torch.float64, device=dev)
        tile8 = torch.zeros((V * rows_max, W, 3), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize(dev)

        def launch(n):
            dworld.render_views(cams[n], 0, 1, tile.data_ptr(), rows_max, rtc.MODE_RENDER_ASYNC, d_ptr8=tile8.data_ptr())
        sync = ctx.synchronize
        stats, reset_stats = ctx.stats, ctx.reset_stats

    q = FrameQueue(V, launch)
    # first-use costs (code object load, communicator set-up) never land in the timed region
    for _ in range(V):
        q.step()
    q.flush()
    sync()
    for _ in range(args.warmup):
        q.step()
    q.flush()
    sync()
    reset_stats()
    ctx.set_timing(args.time_every)
    q.sizes.clear()
    elapsed = timed(q.step, q.flush, sync, args.steps, barrier)
    st = stats()
    sizes = list(q.sizes)
    times = ctx.kernel_times_ms(1024)
    # launches that carried an event pair: every time_every-th, in order
    timed_sizes = sizes[::args.time_every][-len(times):] if len(times) else []
    full = [t for t, n_ in zip(times, timed_sizes) if n_ == V]
    if full:
        kernel_ms, kernel_frames = float(np.mean(full)), V
    elif len(times):
        kernel_ms, kernel_frames = float(times[-1]), timed_sizes[-1]
    else:
        kernel_ms, kernel_frames = 0.0, V
    kernel_ms_per_frame = float(np.sum(times) / max(1, sum(timed_sizes))) if len(times) else 0.0

    # ---- secondary records (outside the headline's timed region; every rank runs the same sequence)
    secondary = {}
    if not args.no_secondary:
        n2 = max(V, min(args.steps, 64))
        if grouped:
            for name, what in (("f64", rtc.GATHER_F64), ("u8", rtc.GATHER_U8), ("none", rtc.GATHER_NONE)):
                if name == args.exchange:
                    continue
                state["what"] = what
                for _ in range(V):
                    q.step()
                q.flush()
                sync()
                reset_stats()
                dt = timed(q.step, q.flush, sync, n2, barrier)
                s2 = stats()
                secondary["exchange_" + name] = (dt, s2["rays_primary"] + s2["rays_shadow"], n2)
            state["what"] = {"f64": rtc.GATHER_F64, "u8": rtc.GATHER_U8, "none": rtc.GATHER_NONE}[args.exchange]
        else:
            q1 = FrameQueue(1, launch)     # ONE camera per launch: the shape of Camera::render_async(&World) -> Canvas
            for _ in range(4):
                q1.step()
            sync()
            reset_stats()
            ctx.set_timing(1)
            dt = timed(q1.step, q1.flush, sync, n2, None)
            s2 = stats()
            t1 = ctx.kernel_times_ms(1024)
            secondary["single_view"] = (dt, s2["rays_primary"] + s2["rays_shadow"], n2, float(np.mean(t1)) if len(t1) else 0.0)

    # ---- outside the timed region: the frame member 0 assembled must be the frame one GPU renders, bit for bit
    exchange_check = None
    if grouped and rank == 0 and args.exchange != "none":
        state["what"] = rtc.GATHER_F64 | rtc.GATHER_U8
        launch(1)
        sync()
        b = (state["batch"] - 1) & 1
        c1 = rtc.Context(dev_index)
        d1 = c1.upload(world)
        ref = torch.zeros((H, W, 3), dtype=torch.float64, device=dev)
        ref8 = torch.zeros((H, W, 3), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize(dev)
        d1.render_rows(cam, 0, H, ref.data_ptr(), rtc.MODE_RENDER_ASYNC, d_ptr8=ref8.data_ptr())
        c1.synchronize()
        exchange_check = "ok" if (torch.equal(canv[b][0], ref) and torch.equal(canv8[b][0], ref8)) else "MISMATCH"
        d1.close()
        c1.close()
    elif grouped and args.exchange != "none":
        state["what"] = rtc.GATHER_F64 | rtc.GATHER_U8
        launch(1)
        sync()

    # ---- host-canvas drop-in (PCIe-inclusive), all ranks fill ONE shared host canvas side by side
    dropin = None
    if not args.no_dropin and (not grouped or N == 1 or args.dropin_multi):
        dropin = measure_dropin(rtc, np, torch, dev_index, world, cam, grouped, group if grouped else None,
                                gworld if grouped else None, rank, N, barrier)

    agg = torch.tensor([elapsed, float(st["rays_primary"]), float(st["rays_shadow"]), float(st["rays_reflect"] + st["rays_refract"]),
                        kernel_ms_per_frame] + [x for v in secondary.values() for x in v[:2]], dtype=torch.float64, device=dev)
    if grouped:
        tmax = agg.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        elapsed, kpf_max = float(tmax[0]), float(tmax[4])
    else:
        tmax, kpf_max = agg, kernel_ms_per_frame
    rays_ps, rays_other = float(agg[1] + agg[2]), float(agg[3])

    if rank == 0:
        steps = max(1, args.steps)
        value = rays_ps / elapsed / 1e6
        abytes = algorithmic_bytes(W, rows_mine * kernel_frames, world)   # one launch: kernel_frames frames of this member's rows, the scene once
        prof = committed_profile(wkey) if N == 1 else None
        kernel_s = kernel_ms * 1e-3
        roof = {"bound": "hbm", "kernel": "k_trace", "achieved": round(abytes / kernel_s / 1e9, 3) if kernel_s else None, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(abytes / kernel_s / 1e9 / HBM_PEAK_GBS, 6) if kernel_s else None,
                "traffic": int(prof[0]["traffic_bytes"] * kernel_frames) if prof else None, "traffic_source": prof[1] if prof else None,
                "algorithmic_bytes_per_launch": int(abytes), "frames_per_launch": kernel_frames,
                "kernel_ms_avg": round(kernel_ms, 5), "kernel_ms_per_frame": round(kernel_ms_per_frame, 6),
                "kernel_launches_timed": int(len(times)), "kernel_launches_of_full_size": len(full), "kernel_timed_every": args.time_every,
                "note": "algorithmic bytes of one launch (its f64 canvases written once + the scene read once) / that launch's duration by "
                        "HIP events on the launch stream; launches of fewer frames (steps % frames_per_launch) are left out of the average. "
                        "The kernel is f64-VALU/latency bound, not HBM bound: see valu_issue and DESIGN.md"}
        out = {
            "metric": baseline_metric(), "value": round(value, 3), "unit": "Mrays/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": wdesc, "workload_key": wkey, "objects": len(world), "rows_per_gpu": rows_mine, "frames_per_launch": V,
                "parallelism": ("single GPU" if not grouped else
                                f"8-row bands dealt round-robin over {N} GPUs behind the C-ABI (rtc_group), one RCCL gather (ncclGather) per batch of "
                                f"{V} frames of the {'f64 Canvas' if args.exchange == 'f64' else '8-bit frame' if args.exchange == 'u8' else 'nothing (tiles stay put)'}"
                                " to member 0 + un-deal kernel; exchange of batch j overlapped with the render of batch j+1"),
                "exchange": args.exchange if grouped else None,
                "exchange_bytes_per_frame": (W * H * {"f64": 24, "u8": 3, "none": 0}[args.exchange] * (N - 1) // N) if grouped else 0,
                "rays_per_frame_primary_shadow": int(round(rays_ps / steps)), "rays_per_frame_other": int(round(rays_other / steps)),
            },
            "roofline": roof,
            "device": ctx.device_info(),
        }
        if prof and kernel_ms_per_frame:
            insts = prof[0]["insts_valu"]
            out["valu_issue"] = {"bound": "valu_issue_slots", "insts_valu_per_frame": int(insts), "cycles_per_inst": 4,
                                 "frac": round(insts * 4 / (SIMDS * kernel_ms_per_frame * 1e-3 * CLOCK_HZ), 4), "source": prof[1],
                                 "note": "SQ_INSTS_VALU (wave-instructions, PMC pass of the committed profile) x 4 issue cycles / (1024 SIMDs x kernel "
                                         "time per frame x 2.4 GHz); a lower bound of the VALU pipes' occupancy (f64 divides and square roots take longer)"}
        if N > 1:
            out["config"]["kernel_ms_per_frame_max_over_ranks"] = round(kpf_max, 6)
        if exchange_check is not None:
            out["config"]["gathered_frame_vs_single_gpu_render"] = exchange_check
        k = 5
        for name, v in secondary.items():
            dt, rays = float(tmax[k]), float(agg[k + 1])
            k += 2
            rec = {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "ms_per_step": round(dt / v[2] * 1e3, 4), "steps": v[2]}
            if name == "single_view":
                rec.update(frames_per_launch=1, kernel_ms_per_frame=round(v[3], 6),
                           note="one camera per launch, the shape of Camera::render_async(&World) -> Canvas; canvas left in HBM")
            else:
                rec["note"] = {"exchange_u8": "same run, member 0 collects the 8-bit frame (Color::scale, 3 B/pixel) instead of the f64 Canvas",
                               "exchange_f64": "same run, member 0 collects the f64 Canvas (24 B/pixel)",
                               "exchange_none": "same run, no exchange: every GPU keeps its tile (the render side alone)"}[name]
            out[name] = rec
        if dropin is not None:
            out["dropin"] = dropin
        if N == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(world, cam, args.cpu_seconds)
        print(json.dumps(out), flush=True)

    if grouped:
        gworld.close()
        group.close()
        dist.destroy_process_group()
    else:
        dworld.close()
        ctx.close()


def measure_dropin(rtc, np, torch, dev_index, world, cam, grouped, group, gworld, rank, N, barrier):
    """What a caller of Camera::render_async(&World) -> Canvas pays when the Canvas lives in host memory
    (canvas.rs:16-22): context + world set-up, then one synchronous call per frame."""
    W, H = cam.hsize, cam.vsize
    nbytes = W * H * 24
    frames = 3 if nbytes < 1 << 28 else 1
    out = {"canvas_bytes": nbytes, "frames_timed": frames}

    def per_frame(fn):
        fn()
        if barrier:
            barrier()
        t = time.perf_counter()
        for _ in range(frames):
            fn()
        if barrier:
            barrier()
        return round((time.perf_counter() - t) / frames * 1e3, 4)

    if not grouped:
        ts = []
        for _ in range(5):
            t = time.perf_counter()
            c = rtc.Context(dev_index)
            ts.append(time.perf_counter() - t)
            c.close()
        out["context_create_ms"] = round(sorted(ts)[2] * 1e3, 4)
        c = rtc.Context(dev_index)
        ts = []
        for _ in range(3):
            t = time.perf_counter()
            d = c.upload(world)
            ts.append(time.perf_counter() - t)
            d.close()
        out["world_upload_ms"] = round(sorted(ts)[1] * 1e3, 4)
        d = c.upload(world)
        pageable = np.empty((H, W, 3), dtype=np.float64)
        pageable[...] = 0.0
        out["rtc_render_fresh_canvas_ms"] = per_frame(lambda: d.render(cam))   # a NEW host canvas per frame (Canvas::new per call): first-touch page faults included
        out["rtc_render_pageable_ms"] = per_frame(lambda: d.render(cam, out=pageable))
        rtc.host_register(pageable)
        out["rtc_render_registered_ms"] = per_frame(lambda: d.render(cam, out=pageable))
        rtc.host_unregister(pageable)
        pinned = rtc.host_canvas(H, W)
        out["rtc_render_pinned_ms"] = per_frame(lambda: d.render(cam, out=pinned))
        out["note"] = ("ms per 1-camera frame INCLUDING the copy of the f64 canvas to host memory over PCIe (never `value`): fresh_canvas = a new "
                       "allocation every frame (what `Canvas::new` per call costs: first-touch page faults), pageable = one plain allocation reused, "
                       "registered = the same after rtc_host_register, pinned = rtc_host_alloc")
        d.close()
        c.close()
    else:
        # every rank maps ONE shared-memory canvas and DMAs its own bands into it over its own PCIe link. Every barrier below
        # is executed by every rank whatever happens locally (a rank that fails must not leave the others waiting).
        path = f"/dev/shm/rtc_bench_canvas_{os.environ.get('MASTER_PORT', '0')}"
        err = None
        shared, registered = None, False
        try:
            if rank == 0:
                with open(path, "wb") as f:
                    f.truncate(nbytes)
        except Exception as e:   # noqa: BLE001
            err = f"create: {e}"
        barrier()
        try:
            shared = np.memmap(path, dtype=np.float64, mode="r+", shape=(H, W, 3))
            try:
                rtc.host_register(shared)
                registered = True
            except Exception:    # noqa: BLE001 - an unregistered mapping still works (bounce buffers)
                registered = False
        except Exception as e:   # noqa: BLE001
            err = err or f"map: {e}"

        def frame():
            if shared is not None:
                gworld.render_host(cam, shared)
        try:
            frame()
        except Exception as e:   # noqa: BLE001
            err = err or f"render_host: {e}"
            shared = None
        barrier()
        t = time.perf_counter()
        try:
            for _ in range(frames):
                frame()
        except Exception as e:   # noqa: BLE001
            err = err or f"render_host: {e}"
        barrier()
        out["rtc_group_render_host_ms"] = round((time.perf_counter() - t) / frames * 1e3, 4)
        out["host_canvas_registered"] = registered
        ok = None
        if rank == 0 and shared is not None and err is None:
            try:
                c = rtc.Context(dev_index)
                ok = bool(np.array_equal(np.asarray(shared), c.upload(world).render(cam)))
                c.close()
            except Exception as e:   # noqa: BLE001
                err = f"check: {e}"
        out["shared_canvas_vs_single_gpu_render"] = "ok" if ok else ("MISMATCH" if ok is False else "not checked")
        if err:
            out["error"] = err
        out["note"] = (f"ms per frame for Camera::render_async into ONE host canvas (shared memory, page-locked in every process): each of the {N} "
                       "GPUs DMAs its bands straight to their rows over its own PCIe link (rtc_group_render_host); no gather")
        try:
            if registered:
                rtc.host_unregister(shared)
        except Exception:        # noqa: BLE001
            pass
        del shared
        barrier()
        if rank == 0:
            try:
                os.unlink(path)
            except OSError:
                pass
    return out


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
