"""The reference's only benchmark definition (Criterion `simple render/f1`, benches/render.rs:63-85:
400x300, three spheres of which two are glass, a checker plane, render_async) on the GPU and on the CPU
oracle. Not a pytest module; run on the GPU box: python tests/criterion_scene.py. (Lives under tests/
because it uses the oracle.) The reference times scene construction inside its closure; here the World is
built once and only the render is timed."""
import importlib
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "oracle")]
import oracle as O  # noqa: E402
from _bootstrap import package  # noqa: E402

rtc = package()
scenes = importlib.import_module(rtc.__name__ + ".scenes")
import torch  # noqa: E402

w, cam = scenes.criterion(400, 300)
ctx = rtc.Context(0)
dw = ctx.upload(w)
f = torch.zeros((300, 400, 3), dtype=torch.float64, device="cuda:0")
for _ in range(20):
    dw.render_rows(cam, 0, 300, f.data_ptr())
ctx.set_timing(1)
n = 200
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(n):
    dw.render_rows(cam, 0, 300, f.data_ptr())
torch.cuda.synchronize()
wall = (time.perf_counter() - t) / n
kern = float(ctx.kernel_times_ms(n).mean())
# the same loop on a pipelined context (round 3): Criterion's `b.iter(|| camera.render_async(&world))` returns a new Canvas per
# iteration, so consecutive frames may overlap — a lone 400x300 frame is 1875 waves for 4096 wave slots
piped = {}
ring = [torch.zeros((300, 400, 3), dtype=torch.float64, device="cuda:0") for _ in range(4)]
ctx.set_timing(0)
for depth in (2, 3, 4):
    ctx.set_pipeline(depth)
    for i in range(20):
        dw.render_rows(cam, 0, 300, ring[i % 4].data_ptr())
    ctx.synchronize()
    t = time.perf_counter()
    for i in range(n):
        dw.render_rows(cam, 0, 300, ring[i % 4].data_ptr())
    ctx.synchronize()
    piped[depth] = (time.perf_counter() - t) / n
ctx.set_pipeline(1)
got, st = dw.render(cam, with_stats=True)
t = time.perf_counter()
want, ost = O.render(w.array(), len(w), w.light, cam, mode=1, nthreads=1, want_stats=True)
cpu1 = time.perf_counter() - t
t = time.perf_counter()
for _ in range(5):
    O.render(w.array(), len(w), w.light, cam, mode=1, nthreads=64)
cpu64 = (time.perf_counter() - t) / 5
rays = sum(st[k] for k in ("rays_primary", "rays_shadow", "rays_reflect", "rays_refract"))
print(f"criterion scene 400x300: {rays} rays/frame ({st}); parity max|d|={np.max(np.abs(got - want)):.2e}, counts equal: {st == ost}")
print(f"  GPU: kernel {kern * 1e3:.1f} us, {wall * 1e6:.1f} us per frame back to back  ({rays / wall / 1e6:.0f} Mrays/s all rays)")
print("  GPU, pipelined context (one camera per launch, ring of 4 canvases): " + ", ".join(f"depth {d}: {v * 1e6:.1f} us per frame ({rays / v / 1e6:.0f} Mrays/s)" for d, v in piped.items()))
print(f"  CPU oracle: 1 thread {cpu1 * 1e3:.1f} ms ({rays / cpu1 / 1e6:.2f} Mrays/s), 64 threads {cpu64 * 1e3:.2f} ms ({rays / cpu64 / 1e6:.1f} Mrays/s)")
