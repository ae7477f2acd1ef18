#!/usr/bin/env python3
"""bench.py — the hot path's headline metric on MI355X.

Metric (BASELINE.json): Mrays/sec (primary+shadow), 1920x1080 x 100 spheres; 1/2/4/8 MI355X.
A step = one frame = ONE call of Camera::render_async(&World) -> Canvas (camera.rs:144-160; the reference's Criterion
bench calls it in a loop on one camera, benches/render.rs:63-85): ONE camera per launch, every frame into a fresh device
canvas (a ring of --canvases f64 canvases: render_async returns a new Canvas per call), World resident in HBM. The context
is pipelined (rtc_context_set_pipeline, --pipeline 3): consecutive launches go round-robin to three streams, so launch i+1
fills the CUs launch i's last waves leave idle and its binning kernel runs beside launch i's render.

--gpus N: the frame is row-tiled behind the C-ABI (rtc_group, include/rtc.h) — 8-row bands dealt round-robin over one
process per GPU, the f64 tiles gathered to member 0 with ONE RCCL gather per frame (ncclGather over xGMI, issued by
librtc.so itself) and un-dealt there into the reference's row-major Canvas. Total work is fixed: scaling "strong".
`python bench.py --gpus N` with N > 1 starts the N ranks itself (torch.distributed.run as a child process, before any HIP
call); under torchrun it is one of the ranks.

Prints ONE JSON line on rank 0:
  value / ms_per_step   the pipelined one-camera-per-launch loop (wall clock, barrier + synchronize on both sides)
  roofline      HBM view of the dominant kernel k_trace: algorithmic bytes of one launch / its SOLO duration (a leg of
                launches with a synchronize after each, HIP events on the launch's own stream; the quantity rocprofv3's
                kernel trace of `bench.py --profile-leg` reports) vs 8 TB/s; `traffic` from the committed PMC profile
  valu_issue    the figure that actually binds: VALU wave-instructions (PMC) x 4 cycles over the SIMD-cycles of the kernel
  launch        which object source / lists the headline launches ran with (rtc_context_last_launch_info)
  serial_single_view, batched_views, brute_force_lds   labelled secondary records (N = 1)
  exchange_*, host_canvas                               labelled secondary records (N > 1)
  dropin        N = 1: the host-canvas calls a Rust caller makes, PCIe-inclusive (never `value`): rtc_render (f64) and
                rtc_render_rgb8 (the 8-bit frame the reference's file writers consume)
  cpu_baseline  the CPU oracle (a C port of the Rust path) on the host cores, N = 1 only.
"""
from __future__ import annotations

import argparse
import importlib
import json
import math
import os
import subprocess
import sys
import time
from pathlib import Path

# RCCL's peer transport needs dmabuf IPC on this driver (the legacy mode fails with hipIpcGetMemHandle: invalid argument);
# the pool exports this already — keep it set for every rank whatever launched us. Must precede the first HIP call.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = Path(__file__).resolve().parent
sys.path[:0] = [str(ROOT), str(ROOT / "oracle")]

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F64_VALU_PEAK = 39.3e12      # f64 vector instructions/s without FMA contraction (78.6 TFLOP/s counts FMA = 2), SURVEY.md §8(d)
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
    ap.add_argument("--pipeline", type=int, default=3, help="N = 1: streams the context deals consecutive launches over (1 = in order on one stream)")
    ap.add_argument("--canvases", type=int, default=4, help="N = 1: device canvases the frames rotate through (>= --pipeline)")
    ap.add_argument("--views-per-launch", type=int, default=1,
                    help="cameras per launch in the HEADLINE loop (1 = the reference's call shape; > 1 = rtc_render_views with that many "
                         "DISTINCT cameras of an orbit, a labelled batch); clamped so that a launch's canvases stay within ~4 GB")
    ap.add_argument("--exchange", default="f64", choices=["f64", "u8", "none"],
                    help="N > 1: what member 0 collects in the MEASURED run: the f64 Canvas (default, the path's output), "
                         "the 8-bit frame, or nothing. The other two are reported as secondary records")
    ap.add_argument("--force-group", action="store_true",
                    help="take the multi-GPU code path (rtc_group in rank mode, RCCL communicator, gather, un-deal) with one rank")
    ap.add_argument("--time-every", type=int, default=4, help="headline loop: HIP-event pair on every n-th launch (0 = none)")
    ap.add_argument("--solo-launches", type=int, default=24, help="launches of the roofline leg (each followed by a synchronize)")
    ap.add_argument("--profile-leg", action="store_true", help="run ONLY the roofline leg (what tools/profile_final.sh puts under rocprofv3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline budget (bounded sample)")
    ap.add_argument("--no-dropin", action="store_true", help="skip the host-canvas (PCIe-inclusive) drop-in measurements")
    ap.add_argument("--no-host-canvas", action="store_true", help="N > 1: skip the shared-host-canvas record (every rank DMAs its bands over its own PCIe link)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary timed loops")
    ap.add_argument("--lean", action="store_true", help="= --no-cpu-baseline --no-dropin --no-secondary (profiling runs)")
    a = ap.parse_args()
    if a.lean:
        a.no_cpu_baseline = a.no_dropin = a.no_secondary = True
    a.pipeline = max(1, min(4, a.pipeline))
    a.canvases = max(a.canvases, a.pipeline)
    return a


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` outside torchrun: start the N ranks as a CHILD process (torch.distributed.run) before this
    process has made any HIP call, relay rank 0's JSON line, return the child's exit code."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    else:
        sys.stdout.write(p.stdout)
    return p.returncode if p.returncode != 0 else (0 if lines else 1)


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


def orbit_cameras(rtc, cam, n):
    """n DISTINCT cameras of the workload's size: the workload's own camera and n-1 more along a small orbit around its
    subject (the reference's AddFrame loop moves the camera between frames, lua.rs:34-41 / functions.lua:3-11)."""
    cams = [cam]
    for i in range(1, n):
        a = 0.05 * i
        frm = (8.0 * math.sin(a), 2.0 + 0.1 * i, 5.0 - 13.0 * math.cos(a))
        cams.append(rtc.camera(cam.hsize, cam.vsize, cam.fov, rtc.Matrix.make_view_transform(frm, (0.0, 1.0, 5.0), (0.0, 1.0, 0.0)), cam.samples))
    return cams


def scene_bytes(world):
    """the scene read once (400 B per object, + 128 per patterned one, + camera / light)."""
    n = len(world)
    n_pat = sum(1 for s in world.shapes if s.material.pattern_kind != 0)
    return 400 * n + 128 * n_pat + 200


def algorithmic_bytes(W, rows, world, with_u8=False):
    """24 B per pixel written once (+ 3 B when the launch also writes the 8-bit frame) + the scene read once."""
    return (27 if with_u8 else 24) * W * rows + scene_bytes(world)


def brute_force_flops(world, st):
    """SURVEY.md §8(d) for the brute-force kernel: every ray tests every object (54 f64 flop per (ray, sphere) — 18 + 15
    transform, 21 quadratic; 33 per (ray, plane); 70 per (ray, cube)) + ~200 per hit for shading. An UPPER bound of what
    the kernel executes: a wave's shadow pass stops once every lane is shadowed."""
    per_ray = sum({0: 54, 1: 33, 2: 70}[s.kind] for s in world.shapes)
    rays = st["rays_primary"] + st["rays_shadow"] + st["rays_reflect"] + st["rays_refract"]
    return rays * per_ray + st["rays_shadow"] * 200


def committed_profile(key):
    """Newest PMC summary under profiles/ for this workload (rocprofv3 cannot run inside the timed process):
    per-FRAME HBM traffic and VALU wave-instructions of the dominant kernel."""
    best = None
    for f in sorted((ROOT / "profiles").glob("r*_pmc.json")):
        try:
            d = json.loads(f.read_text())
            if d.get("workload_key") == key and "per_frame" in d:
                best = (d["per_frame"], f"profiles/{f.name}", d.get("head"))
        except Exception:
            continue
    return best


def timed(fn_step, fn_sync, steps, barrier=None):
    if barrier:
        barrier()
    fn_sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn_step()
    fn_sync()
    if barrier:
        barrier()
    return time.perf_counter() - t0


def ps(st):
    return st["rays_primary"] + st["rays_shadow"]


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))       # before any HIP call: this process never touches the GPU
    import numpy as np
    import torch
    import torch.distributed as dist

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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
    rows_max = rtc.group_packed_rows(H, N)                 # rows of one member's packed tile (csrc/rtc_bands.h)
    rows_mine = sum(min(8, H - 8 * b) for b in range(rank, -(-H // 8), N))
    V = max(1, min(args.views_per_launch, 8, int(4e9 // (rows_max * W * 24))))
    VB = max(1, min(8, int(4e9 // (rows_max * W * 24))))   # the batched secondary record
    cam_list = orbit_cameras(rtc, cam, max(V, VB))
    cam_arr = {n_: (type(cam) * n_)(*cam_list[:n_]) for n_ in {1, V, VB}}
    barrier = (lambda: dist.barrier()) if grouped else None
    out = None

    if grouped:
        out = run_group(args, rtc, np, torch, dist, dev, dev_index, world, cam, cam_arr, wkey, wdesc, rank, N, V, VB, rows_mine, barrier)
    else:
        out = run_single(args, rtc, np, torch, dev, dev_index, world, cam, cam_arr, wkey, wdesc, V, VB)
    bad = False
    if rank == 0 and out is not None:
        print(json.dumps(out), flush=True)
        bad = out.get("config", {}).get("gathered_frame_vs_single_gpu_render") == "MISMATCH"
    if grouped:
        dist.destroy_process_group()
    if bad:
        sys.exit("the frame member 0 assembled differs from a single-GPU render")


# ------------------------------------------------------------------------------------------------------------------
# N = 1
# ------------------------------------------------------------------------------------------------------------------
def run_single(args, rtc, np, torch, dev, dev_index, world, cam, cam_arr, wkey, wdesc, V, VB):
    W, H = cam.hsize, cam.vsize
    ctx = rtc.Context(dev_index, stream=torch.cuda.current_stream(dev).cuda_stream)
    dworld = ctx.upload(world)
    R = args.canvases
    HP = -(-H // 8) * 8     # rows one view occupies in a multi-view launch (whole 8-row bands)
    ring = [torch.zeros(((V * HP) if V > 1 else H, W, 3), dtype=torch.float64, device=dev) for _ in range(R)]   # Camera::render_async returns a NEW Canvas per call
    ptrs = [t.data_ptr() for t in ring]
    torch.cuda.synchronize(dev)
    state = {"i": 0}

    def launch():
        i = state["i"]
        state["i"] = i + 1
        if V == 1:
            dworld.render_rows(cam, 0, H, ptrs[i % R], rtc.MODE_RENDER_ASYNC)
        else:
            dworld.render_views(cam_arr[V], 0, 1, ptrs[i % R], HP, rtc.MODE_RENDER_ASYNC)
    def sync():   # the context's lanes and stream, then the device (the driver's contract: torch.cuda.synchronize() on both sides)
        ctx.synchronize()
        torch.cuda.synchronize(dev)

    def solo_leg(n):
        """n launches, a synchronize after each: every kernel has the GPU to itself (its own binning kernel in front of it on the
        same lane). HIP events on the launch's own stream = rocprofv3's kernel-trace durations of `--profile-leg`."""
        ctx.set_timing(1)
        for _ in range(n):
            launch()
            sync()
        t, b = ctx.kernel_times_ms(1024)[-n:], ctx.binning_times_ms(1024)[-n:]
        ctx.set_timing(0)
        return (float(np.mean(t)) if len(t) else 0.0, float(np.mean(b)) if len(b) else 0.0, len(t))

    ctx.set_pipeline(args.pipeline)
    ctx.set_timing(0)
    if args.profile_leg:
        # ONLY solo launches in this process (what tools/profile_final.sh puts under rocprofv3: its kernel-trace average of k_trace
        # is then the quantity `roofline.kernel_ms_avg` reports)
        solo_leg(args.pipeline + 1)           # first-use costs: code object, every lane's list buffers
        k_ms, b_ms, n = solo_leg(max(args.steps, 1))
        out = {"profile_leg": True, "kernel_ms_avg": round(k_ms, 5), "binning_ms_avg": round(b_ms, 5), "launches": n,
               "launch": ctx.last_launch_info(),
               "config": {"workload": wdesc, "workload_key": wkey, "frames_per_launch": V, "pipeline_streams": args.pipeline}}
        dworld.close()
        ctx.close()
        return out
    for _ in range(max(2 * R, 4)):        # first-use costs (code object load, list buffers of every lane) never land in a timed region
        launch()
    sync()
    info = ctx.last_launch_info()

    # ---- the headline: K steps = K launches, pipelined, one camera each
    for _ in range(args.warmup):
        launch()
    sync()
    ctx.reset_stats()
    ctx.set_timing(args.time_every)
    elapsed = timed(launch, sync, args.steps)
    st = ctx.stats(extended=True)
    t_over = ctx.kernel_times_ms(1024)
    ctx.set_timing(0)
    steps = max(1, args.steps)
    frames = steps * V
    k_ms, b_ms, n_solo = solo_leg(max(1, min(args.solo_launches, 64)))

    prof = committed_profile(wkey)
    abytes = algorithmic_bytes(W, H * V, world)
    kernel_s = k_ms * 1e-3
    roof = {"bound": "hbm", "kernel": "k_trace", "achieved": round(abytes / kernel_s / 1e9, 3) if kernel_s else None, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(abytes / kernel_s / 1e9 / HBM_PEAK_GBS, 6) if kernel_s else None,
            "traffic": int(prof[0]["traffic_bytes"] * V) if prof else None, "traffic_source": prof[1] if prof else None,
            "traffic_profile_head": prof[2] if prof else None,
            "algorithmic_bytes_per_launch": int(abytes), "frames_per_launch": V,
            "kernel_ms_avg": round(k_ms, 5), "kernel_ms_per_frame": round(k_ms / V, 6), "kernel_launches_timed": n_solo,
            "binning_kernel_ms_avg": round(b_ms, 5),
            "frac_counting_the_binning_kernel": round(abytes / ((k_ms + b_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 6) if kernel_s else None,
            "pipelined_gbs": round(abytes * steps / elapsed / 1e9, 3),
            "kernel_ms_avg_while_overlapped": round(float(np.mean(t_over)), 5) if len(t_over) else None,
            "note": "algorithmic bytes of one launch (its f64 canvas written once + the scene read once; the headline launch writes no 8-bit "
                    "frame) / the render kernel's SOLO duration: a leg of launches with a synchronize after each, HIP events "
                    "(hipExtLaunchKernel begin/end) on the launch's own stream — what `rocprofv3 --kernel-trace -- python3 bench.py --profile-leg` "
                    "reports. binning_kernel_ms_avg = the launch's k_bin_tiles (same leg; in the pipelined loop it runs beside the other "
                    "lane's render). pipelined_gbs = the same bytes over the headline loop's wall clock. The kernel is f64-VALU-issue / "
                    "latency bound, not HBM bound: see valu_issue and DESIGN.md"}
    out = {
        "metric": baseline_metric(), "value": round(ps(st) / elapsed / 1e6, 3), "unit": "Mrays/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / steps * 1e3, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": wdesc, "workload_key": wkey, "objects": len(world), "rows_per_gpu": H, "frames_per_launch": V,
            "call_shape": ("ONE camera per launch = one Camera::render_async(&World) -> Canvas per step" if V == 1 else
                           f"{V} DISTINCT cameras of an orbit per launch (rtc_render_views): a batch, not the reference's call shape"),
            "pipeline_streams": args.pipeline, "device_canvases": R, "parallelism": "single GPU",
            "exchange": None, "exchange_bytes_per_frame": 0,
            "rays_per_frame_primary_shadow": int(round(ps(st) / frames)),
            "rays_per_frame_other": int(round((st["rays_reflect"] + st["rays_refract"]) / frames)),
            "primary_rays_per_frame_answered_by_the_tile_proof": int(round(st["rays_primary_proven_miss"] / frames)),
            "tile_proof_note": "of the primary rays: those of 8-row tile bands the launch's binning kernel PROVED to hit nothing (empty candidate lists, cones clear "
                               "of every plane: the sky) - counted, as the reference casts them, but no ray is generated (rtc_stats.rays_primary_proven_miss)",
        },
        "roofline": roof,
        "launch": info,
        "device": ctx.device_info(),
    }
    if prof and k_ms:
        insts = prof[0]["insts_valu"]
        out["valu_issue"] = {"bound": "valu_issue_slots", "insts_valu_per_frame": int(insts), "cycles_per_inst": 4,
                             "frac": round(insts * 4 / (SIMDS * (k_ms / V) * 1e-3 * CLOCK_HZ), 4), "source": prof[1],
                             "note": "SQ_INSTS_VALU (wave-instructions, PMC pass of the committed profile) x 4 issue cycles / (1024 SIMDs x solo kernel "
                                     "time per frame x 2.4 GHz); a lower bound of the VALU pipes' occupancy (f64 divides and square roots take longer)"}

    if not args.no_secondary:
        n2 = max(8, min(args.steps, 64))

        def loop(fn, n, warm=4):
            for _ in range(warm):
                fn()
            sync()
            ctx.reset_stats()
            dt = timed(fn, sync, n)
            s2 = ctx.stats()
            return dt, s2

        # (1) the same launches IN ORDER on one stream (pipeline depth 1): what the pipelining buys
        ctx.set_pipeline(1)
        dt, s2 = loop(launch, n2)
        k1, b1, _ = solo_leg(8)
        out["serial_single_view"] = {"value": round(ps(s2) / dt / 1e6, 3), "unit": "Mrays/s", "ms_per_step": round(dt / n2 * 1e3, 4), "steps": n2,
                                     "frames_per_launch": V, "kernel_ms_per_frame": round(k1 / V, 6), "launch": ctx.last_launch_info(),
                                     "note": "the headline's launches in order on ONE stream (rtc_context_set_pipeline(1), round 2's `single_view`)"}
        # (2) a batch: VB distinct cameras per launch (rtc_render_views), in order, binning on the side stream
        if VB > 1 and V == 1:
            big = torch.zeros((VB * HP, W, 3), dtype=torch.float64, device=dev)
            torch.cuda.synchronize(dev)

            def launch_b():
                dworld.render_views(cam_arr[VB], 0, 1, big.data_ptr(), HP, rtc.MODE_RENDER_ASYNC)
            nb = max(2, n2 // VB)
            dt, s2 = loop(launch_b, nb)
            ctx.set_timing(1)
            launch_b()
            sync()
            kb = float(ctx.kernel_times_ms(1)[-1])
            ctx.set_timing(0)
            out["batched_views"] = {"value": round(ps(s2) / dt / 1e6, 3), "unit": "Mrays/s", "ms_per_frame": round(dt / (nb * VB) * 1e3, 4), "frames": nb * VB,
                                    "frames_per_launch": VB, "kernel_ms_per_frame": round(kb / VB, 6), "launch": ctx.last_launch_info(),
                                    "rays_per_frame_primary_shadow": int(round(ps(s2) / (nb * VB))),
                                    "note": f"{VB} DISTINCT cameras (an orbit around the scene) per launch, rtc_render_views: a labelled batch, never the headline"}
            del big
        # (3) BASELINE.json north_star's literal kernel: every ray loops over ALL objects, the table staged in LDS
        flags = rtc.FLAG_NO_CULL | rtc.FLAG_LDS_TABLE

        def launch_bf():
            i = state["i"]
            state["i"] = i + 1
            dworld.render_rows(cam, 0, H, ptrs[i % R], rtc.MODE_RENDER_ASYNC, flags=flags)
        nbf = 4 if len(world) * W * H < 1e9 else 1   # (1000 objects at 8192^2 take most of a second per frame)
        dt, s2 = loop(launch_bf, nbf, warm=1)
        ctx.set_timing(1)
        launch_bf()
        sync()
        kbf = float(ctx.kernel_times_ms(1)[-1])
        ctx.set_timing(0)
        fl = brute_force_flops(world, s2) / nbf
        out["brute_force_lds"] = {"value": round(ps(s2) / dt / 1e6, 3), "unit": "Mrays/s", "ms_per_frame": round(dt / nbf * 1e3, 4), "frames": nbf,
                                  "kernel_ms_per_frame": round(kbf, 5), "launch": ctx.last_launch_info(),
                                  "flops_per_frame": int(fl), "f64_valu": {"achieved_tflops": round(fl / (kbf * 1e-3) / 1e12, 3),
                                                                           "peak_tflops_no_fma": F64_VALU_PEAK / 1e12,
                                                                           "frac": round(fl / (kbf * 1e-3) / F64_VALU_PEAK, 4)},
                                  "note": "RTC_FLAG_NO_CULL | RTC_FLAG_LDS_TABLE: one thread per pixel looping ray-sphere / ray-plane tests over ALL objects "
                                          "staged in LDS (BASELINE.json north_star's literal kernel), same pixels bit for bit. flops = SURVEY.md §8(d): rays x "
                                          "sum over objects (54 per sphere, 33 per plane) + 200 per hit — the one kernel for which that roofline is meaningful"}
        ctx.set_pipeline(args.pipeline)

    if not args.no_dropin:
        out["dropin"] = measure_dropin(rtc, np, torch, dev_index, world, cam, False, None, None, 0, 1, None)
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(world, cam, args.cpu_seconds)
    dworld.close()
    ctx.close()
    return out


# ------------------------------------------------------------------------------------------------------------------
# N > 1 (one process per GPU), or --force-group with one rank
# ------------------------------------------------------------------------------------------------------------------
def run_group(args, rtc, np, torch, dist, dev, dev_index, world, cam, cam_arr, wkey, wdesc, rank, N, V, VB, rows_mine, barrier):
    W, H = cam.hsize, cam.vsize
    uid = torch.zeros(rtc.GROUP_ID_BYTES, dtype=torch.uint8, device=dev)
    if rank == 0:
        uid.copy_(torch.tensor(list(rtc.group_unique_id()), dtype=torch.uint8))
    dist.broadcast(uid, 0)
    group = rtc.Group(device=dev_index, nranks=N, rank=rank, uid=bytes(uid.cpu().tolist()))
    ctx = group.contexts[0]
    gworld = group.upload(world)
    VM = max(V, VB)
    # member 0's destination: two batches of canvases (a consumer reads batch j while batch j+1 is assembled)
    canv = [torch.zeros((VM, H, W, 3), dtype=torch.float64, device=dev) for _ in range(2)] if rank == 0 else [None, None]
    canv8 = [torch.zeros((VM, H, W, 3), dtype=torch.uint8, device=dev) for _ in range(2)] if rank == 0 else [None, None]
    torch.cuda.synchronize(dev)
    WHAT = {"f64": rtc.GATHER_F64, "u8": rtc.GATHER_U8, "none": rtc.GATHER_NONE}
    state = {"batch": 0, "what": WHAT[args.exchange], "v": V}

    def launch():
        b = state["batch"] & 1
        state["batch"] += 1
        gworld.render(cam_arr[state["v"]], state["what"], canv[b].data_ptr() if rank == 0 else None,
                      canv8[b].data_ptr() if rank == 0 else None)
    def sync():   # every member stream of this rank, then the device
        group.synchronize()
        torch.cuda.synchronize(dev)
    stats, reset_stats = group.stats, group.reset_stats

    for _ in range(4):     # first-use costs (code object load, communicator set-up, tile buffers) never land in a timed region
        launch()
    sync()
    for _ in range(args.warmup):
        launch()
    sync()
    reset_stats()
    ctx.set_timing(max(1, args.time_every))
    elapsed = timed(launch, sync, args.steps, barrier)
    st = stats()
    times = ctx.kernel_times_ms(1024)
    ctx.set_timing(0)
    steps = max(1, args.steps)
    kernel_ms = float(np.mean(times)) if len(times) else 0.0
    info = ctx.last_launch_info()

    secondary = {}
    if not args.no_secondary:
        n2 = max(8, min(args.steps, 64))

        def loop(n):
            for _ in range(2):
                launch()
            sync()
            reset_stats()
            dt = timed(launch, sync, n, barrier)
            return dt, ps(stats()), n * state["v"]
        for name in ("f64", "u8", "none"):
            if name != args.exchange:
                state["what"] = WHAT[name]
                secondary["exchange_" + name] = loop(n2)
        state["what"] = WHAT[args.exchange]
        if VB > 1 and V == 1:
            state["v"] = VB
            secondary["batched_views"] = loop(max(2, n2 // VB))
            state["v"] = V

    # ---- outside the timed region: the frame member 0 assembled must be the frame one GPU renders, bit for bit (mandatory)
    exchange_check = None
    state["what"], state["v"] = rtc.GATHER_F64 | rtc.GATHER_U8, 1
    launch()
    sync()
    if rank == 0:
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

    # ---- host canvases (PCIe-inclusive), all ranks fill ONE shared host canvas side by side: the exchange-free delivery
    host = None
    if not args.no_host_canvas and not args.no_dropin:
        host = measure_dropin(rtc, np, torch, dev_index, world, cam, True, group, gworld, rank, N, barrier)

    sec_flat = [x for v in secondary.values() for x in v[:2]]
    agg = torch.tensor([elapsed, float(st["rays_primary"]), float(st["rays_shadow"]), float(st["rays_reflect"] + st["rays_refract"]), kernel_ms] + sec_flat,
                       dtype=torch.float64, device=dev)
    tmax = agg.clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.all_reduce(agg, op=dist.ReduceOp.SUM)
    elapsed, kms_max = float(tmax[0]), float(tmax[4])
    rays_ps, rays_other = float(agg[1] + agg[2]), float(agg[3])
    out = None
    if rank == 0:
        frames = steps * V
        abytes = algorithmic_bytes(W, rows_mine * V, world)
        kernel_s = kernel_ms * 1e-3
        roof = {"bound": "hbm", "kernel": "k_trace", "achieved": round(abytes / kernel_s / 1e9, 3) if kernel_s else None, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(abytes / kernel_s / 1e9 / HBM_PEAK_GBS, 6) if kernel_s else None, "traffic": None,
                "algorithmic_bytes_per_launch": int(abytes), "frames_per_launch": V, "kernel_ms_avg": round(kernel_ms, 5),
                "kernel_ms_avg_max_over_ranks": round(kms_max, 5), "kernel_launches_timed": int(len(times)),
                "note": "member 0's share: algorithmic bytes of one of its launches (its rows of the f64 canvas + the scene) / that launch's duration "
                        "by HIP events on its render stream (the exchange of the previous frame runs beside it)"}
        out = {
            "metric": baseline_metric(), "value": round(rays_ps / elapsed / 1e6, 3), "unit": "Mrays/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": wdesc, "workload_key": wkey, "objects": len(world), "rows_per_gpu": rows_mine, "frames_per_launch": V,
                "call_shape": ("ONE camera per rtc_group_render call" if V == 1 else f"{V} distinct cameras per rtc_group_render call (a batch)"),
                "parallelism": (f"8-row bands dealt round-robin over {N} GPUs behind the C-ABI (rtc_group), one RCCL gather (ncclGather) per call of the "
                                f"{'f64 Canvas' if args.exchange == 'f64' else '8-bit frame' if args.exchange == 'u8' else 'nothing (tiles stay put)'}"
                                " to member 0 + un-deal kernel; exchange of call j overlapped with the render of call j+1"),
                "exchange": args.exchange,
                "exchange_bytes_per_frame": W * H * {"f64": 24, "u8": 3, "none": 0}[args.exchange] * (N - 1) // N,
                "rays_per_frame_primary_shadow": int(round(rays_ps / frames)), "rays_per_frame_other": int(round(rays_other / frames)),
                "gathered_frame_vs_single_gpu_render": exchange_check,
            },
            "roofline": roof,
            "launch": info,
            "device": ctx.device_info(),
        }
        k = 5
        notes = {"exchange_u8": "same run, member 0 collects the 8-bit frame (Color::scale, 3 B/pixel) instead of the f64 Canvas",
                 "exchange_f64": "same run, member 0 collects the f64 Canvas (24 B/pixel)",
                 "exchange_none": "same run, no exchange: every GPU keeps its tile (the render side alone)",
                 "batched_views": f"same run and exchange, {VB} DISTINCT cameras per call (one launch per member and batch): a labelled batch"}
        for name, v in secondary.items():
            dt, rays = float(tmax[k]), float(agg[k + 1])
            k += 2
            out[name] = {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "ms_per_frame": round(dt / v[2] * 1e3, 4), "frames": v[2], "note": notes[name]}
        if host is not None:
            out["host_canvas"] = host
    gworld.close()
    group.close()
    return out


def measure_dropin(rtc, np, torch, dev_index, world, cam, grouped, group, gworld, rank, N, barrier):
    """What a caller of Camera::render_async(&World) -> Canvas pays when the Canvas lives in host memory
    (canvas.rs:16-22): context + world set-up, then one synchronous call per frame."""
    W, H = cam.hsize, cam.vsize
    nbytes = W * H * 24
    frames = 3 if nbytes < 1 << 28 else 1
    out = {"canvas_bytes": nbytes, "rgb8_bytes": W * H * 3, "frames_timed": frames}

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
        # the 8-bit frame: what the reference's file writers consume (canvas.rs:86-109, 61-79) — 3 B/pixel over PCIe, no f64 canvas written
        page8 = np.zeros((H, W, 3), dtype=np.uint8)
        out["rtc_render_rgb8_pageable_ms"] = per_frame(lambda: d.render_rgb8(cam, out=page8))
        pin8 = rtc.host_canvas_rgb8(H, W)
        out["rtc_render_rgb8_ms"] = per_frame(lambda: d.render_rgb8(cam, out=pin8))
        out["rgb8_equals_color_scale_of_the_f64_canvas"] = bool(np.array_equal(pin8, rtc.color_scale255(pinned).reshape(H, W, 3)))
        out["note"] = ("ms per 1-camera frame INCLUDING the copy to host memory over PCIe (never `value`). rtc_render = the f64 canvas (24 B/pixel): fresh_canvas "
                       "= a new allocation every frame (what `Canvas::new` per call costs: first-touch page faults), pageable = one plain allocation "
                       "reused, registered = the same after rtc_host_register, pinned = rtc_host_alloc. rtc_render_rgb8 = only the 8-bit frame "
                       "(Color::scale on the device, 3 B/pixel; pinned unless it says pageable)")
        d.close()
        c.close()
    else:
        # every rank maps ONE shared-memory canvas and DMAs its own bands into it over its own PCIe link. Every barrier below
        # is executed by every rank whatever happens locally (a rank that fails must not leave the others waiting).
        path = f"/dev/shm/rtc_bench_canvas_{os.environ.get('MASTER_PORT', '0')}"
        err = None
        shared, shared8, registered = None, None, False
        try:
            if rank == 0:
                with open(path, "wb") as f:
                    f.truncate(nbytes + W * H * 3)
        except Exception as e:   # noqa: BLE001
            err = f"create: {e}"
        barrier()
        try:
            whole = np.memmap(path, dtype=np.uint8, mode="r+", shape=(nbytes + W * H * 3,))
            shared = whole[:nbytes].view(np.float64).reshape(H, W, 3)
            shared8 = whole[nbytes:].reshape(H, W, 3)
            try:
                rtc.host_register(whole)
                registered = True
            except Exception:    # noqa: BLE001 - an unregistered mapping still works (bounce buffers)
                registered = False
        except Exception as e:   # noqa: BLE001
            err = err or f"map: {e}"

        def run(fn, key):   # exactly two barriers, whatever happens locally
            nonlocal err
            try:
                fn()
            except Exception as e:   # noqa: BLE001
                err = err or f"{key}: {e}"
            barrier()
            t = time.perf_counter()
            try:
                if err is None:
                    for _ in range(frames):
                        fn()
            except Exception as e:   # noqa: BLE001
                err = err or f"{key}: {e}"
            barrier()
            out[key] = round((time.perf_counter() - t) / frames * 1e3, 4)

        if shared is not None:
            run(lambda: gworld.render_host(cam, shared), "rtc_group_render_host_ms")
            run(lambda: gworld.render_host_rgb8(cam, shared8), "rtc_group_render_host_rgb8_ms")
        else:
            for _ in range(4):
                barrier()
        out["host_canvas_registered"] = registered
        ok = None
        if rank == 0 and shared is not None and err is None:
            try:
                c = rtc.Context(dev_index)
                ref = c.upload(world).render(cam)
                ok = bool(np.array_equal(np.asarray(shared), ref)) and bool(np.array_equal(np.asarray(shared8), rtc.color_scale255(ref).reshape(H, W, 3)))
                c.close()
            except Exception as e:   # noqa: BLE001
                err = f"check: {e}"
        out["shared_canvas_vs_single_gpu_render"] = "ok" if ok else ("MISMATCH" if ok is False else "not checked")
        if err:
            out["error"] = err
        out["note"] = (f"ms per frame for Camera::render_async into ONE host canvas (shared memory, page-locked in every process): each of the {N} "
                       "GPUs DMAs its bands straight to their rows over its own PCIe link (rtc_group_render_host: the f64 Canvas, 24 B/pixel; "
                       "_rgb8: the 8-bit frame, 3 B/pixel); no gather, no xGMI")
        try:
            if registered:
                rtc.host_unregister(whole)
        except Exception:        # noqa: BLE001
            pass
        shared = shared8 = whole = None
        barrier()
        if rank == 0:
            try:
                os.unlink(path)
            except OSError:
                pass
    return out


def cpu_baseline(world, cam, budget_s):
    """The CPU oracle (a C port of the Rust path, literal sorted-list form) on ALL host cores the process may use,
    same scene; bounded sample: as many whole frames (or one band of rows) as fit the budget."""
    import oracle as O
    O.build()
    machine = os.cpu_count() or 1
    allowed = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else machine
    allowed = max(1, allowed)
    arr = world.array()
    H = cam.vsize
    # probe: a thin band to estimate the rate and to pick the thread count — the box may give this job a share of the machine's
    # cores (a cgroup quota the affinity mask does not show): more threads than that share only slow the oracle down
    band = max(1, H // 60)
    t = time.perf_counter()
    O.render(arr, len(world), world.light, cam, mode=1, y0=H // 2, y1=H // 2 + band, nthreads=min(allowed, 32))
    rough = max(1e-6, time.perf_counter() - t) / band                      # seconds per row, roughly
    pband = max(band, min(H, int(0.08 / rough)))                           # ~0.08 s per candidate: long enough to tell them apart
    tried = {}
    for nt in sorted({n_ for n_ in (16, 32, 64, 128, allowed) if n_ <= allowed}):
        best_t = None
        for _ in range(3):                                                   # best of three (the box's cores are shared: single timings scatter)
            t = time.perf_counter()
            O.render(arr, len(world), world.light, cam, mode=1, y0=(H - pband) // 2, y1=(H - pband) // 2 + pband, nthreads=nt)
            d_ = max(1e-6, time.perf_counter() - t)
            best_t = d_ if best_t is None else min(best_t, d_)
        tried[nt] = best_t
    fastest = min(tried.values())
    cores = min(nt for nt, v in tried.items() if v <= fastest * 1.05)       # the fewest threads within 5 % of the fastest

    def sample(nthreads, seconds, est):
        """Whole frames (or one centre band) for about `seconds`: (rays, seconds taken, description)."""
        if est <= seconds:
            t0 = time.perf_counter()
            rays_ = frames = 0
            while frames == 0 or (time.perf_counter() - t0) + est <= seconds:   # whole frames until the time is used
                t1 = time.perf_counter()
                _, st_ = O.render(arr, len(world), world.light, cam, mode=1, nthreads=nthreads, want_stats=True)
                est = time.perf_counter() - t1
                rays_ += st_["rays_primary"] + st_["rays_shadow"]
                frames += 1
            return rays_, time.perf_counter() - t0, f"{frames} full frame(s) {cam.hsize}x{cam.vsize}", est
        rows = max(pband, int(H * seconds / est))
        ya = (H - rows) // 2
        t0 = time.perf_counter()
        _, st_ = O.render(arr, len(world), world.light, cam, mode=1, y0=ya, y1=ya + rows, nthreads=nthreads, want_stats=True)
        d_ = time.perf_counter() - t0
        return st_["rays_primary"] + st_["rays_shadow"], d_, f"rows [{ya},{ya + rows}) of {cam.hsize}x{cam.vsize} (centre band)", d_ * H / rows

    # A thin band does not always predict the sustained rate (a box whose cores are shared: the band once favoured 256 threads
    # that then ran whole frames at half the rate of 64): the two best candidates each get a short sustained sample first.
    ranked = sorted(tried, key=lambda n_: (tried[n_], n_))
    finalists = [cores] + [n_ for n_ in ranked if n_ != cores][:1]
    trial = {}
    for nt in finalists:
        r_, d_, _, e_ = sample(nt, min(2.0, budget_s / 6.0), tried[nt] * H / pband)
        trial[nt] = (r_ / d_, e_)
    cores = max(trial, key=lambda n_: (trial[n_][0], -n_))
    est_frame = trial[cores][1]
    band = pband
    rays, dt, sample_desc, est_frame = sample(cores, budget_s * 2.0 / 3.0, est_frame)
    sample = sample_desc
    out = {"value": round(rays / dt / 1e6, 4), "unit": "Mrays/s", "cores": cores, "host_cores_online": machine, "host_cores_allowed": allowed,
           "threads_tried_band_seconds": {str(k): round(v, 4) for k, v in tried.items()},
           "finalists_sustained_mrays_s": {str(k): round(v[0] / 1e6, 3) for k, v in trial.items()}, "kind": "port", "sample": sample,
           "seconds": round(dt, 2), "form": "literal sorted-list oracle (oracle/rtc_oracle.c), f64, -O2 -ffp-contract=off; `cores` = the thread count: of those tried on a "
                                            "band of rows (up to every core this process may run on) the two best get a short sustained sample each and the better one the rest of the budget; rows handed out dynamically, like rayon"}

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
