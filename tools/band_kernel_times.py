"""Kernel time of one rank's share of the north-star frame (interleaved bands, stride N) on ONE GPU:
what each rank's launch costs at N = 1, 2, 4, 8 (the strong-scaling ceiling of the render itself)."""
import importlib
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT)]
from _bootstrap import package  # noqa: E402

rtc = package()
scenes = importlib.import_module(rtc.__name__ + ".scenes")
tiles = importlib.import_module(rtc.__name__ + ".tiles")
import torch  # noqa: E402

W, H = 1920, 1080
w, cam = scenes.synthetic(100, W, H)
ctx = rtc.Context(0)
dw = ctx.upload(w)
for N in (1, 2, 4, 8):
    rows = tiles.packed_rows(H, N)
    f = torch.zeros((rows, W, 3), dtype=torch.float64, device="cuda:0")
    q = torch.zeros((rows, W, 3), dtype=torch.uint8, device="cuda:0")
    res = []
    for r in range(N):
        for _ in range(60):
            dw.render_bands(cam, r, N, f.data_ptr(), d_ptr8=q.data_ptr())
        res.append(float(ctx.kernel_times_ms(50).mean()))
    print(f"N={N}: kernel ms per rank: min {min(res):.4f} max {max(res):.4f}  -> ceiling {0.0735 / max(res):.2f}x of one GPU's 0.0735 ms frame")

# Frames in flight: one rank's launches issued round-robin on S streams (wall clock per frame).
import time  # noqa: E402

pool = [torch.cuda.Stream() for _ in range(6)]
ctxs = [ctx] + [rtc.Context(0, stream=s.cuda_stream) for s in pool]
dws = [dw] + [c.upload(w) for c in ctxs[1:]]
for N in (1, 2, 4, 8):
    rows = tiles.packed_rows(H, N)
    bufs = [(torch.zeros((rows, W, 3), dtype=torch.float64, device="cuda:0"), torch.zeros((rows, W, 3), dtype=torch.uint8, device="cuda:0")) for _ in range(len(dws))]

    def run(idx, reps=200):
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            t = time.perf_counter()
            for k in range(reps):
                i = idx[k % len(idx)]
                dws[i].render_bands(cam, 0, N, bufs[i][0].data_ptr(), d_ptr8=bufs[i][1].data_ptr())
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t) / reps)
        return best * 1e3

    base = run([0])
    # order the candidates by how well they overlap with stream 0 (two streams may share a hardware queue)
    order = sorted(range(1, len(dws)), key=lambda j: run([0, j], 60))
    line = [f"S=1 {base:.4f}"]
    for S in (2, 3, 4):
        line.append(f"S={S} {run([0] + order[:S - 1]):.4f}")
    print(f"N={N}: ms per frame of one rank, wall clock: " + ", ".join(line))

# 8 frames per launch (rtc_render_views) of one rank's bands, on 1..3 streams
cams8 = (type(cam) * 8)(*([cam] * 8))
for N in (1, 2, 4, 8):
    rows = tiles.packed_rows(H, N)
    bufs = [(torch.zeros((8 * rows, W, 3), dtype=torch.float64, device="cuda:0"), torch.zeros((8 * rows, W, 3), dtype=torch.uint8, device="cuda:0")) for _ in range(len(dws))]

    def run8(idx, reps=60):
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            t = time.perf_counter()
            for k in range(reps):
                i = idx[k % len(idx)]
                dws[i].render_views(cams8, 0, N, bufs[i][0].data_ptr(), rows, d_ptr8=bufs[i][1].data_ptr())
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t) / (reps * 8))
        return best * 1e3

    base = run8([0])
    order = sorted(range(1, len(dws)), key=lambda j: run8([0, j], 20))
    print(f"N={N}: ms per frame of one rank with 8 frames per launch: S=1 {base:.4f}, S=2 {run8([0] + order[:1]):.4f}, S=3 {run8([0] + order[:2]):.4f}")
