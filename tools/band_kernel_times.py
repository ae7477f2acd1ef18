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
