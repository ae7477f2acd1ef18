"""PCIe-inclusive rate: rtc_render() hands the f64 canvas back in host memory every frame
(device malloc + render + D2H into pageable numpy memory + free). Reported in DESIGN.md, never as
bench.py's `value`."""
import importlib
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT)]
from _bootstrap import package  # noqa: E402

rtc = package()
scenes = importlib.import_module(rtc.__name__ + ".scenes")
w, cam = scenes.synthetic(100, 1920, 1080)
ctx = rtc.Context(0)
dw = ctx.upload(w)
for _ in range(3):
    dw.render(cam)
n = 20
t = time.perf_counter()
for _ in range(n):
    img, st = dw.render(cam, with_stats=True)
dt = (time.perf_counter() - t) / n
rays = st["rays_primary"] + st["rays_shadow"]
print(f"rtc_render (pageable host canvas) {dt * 1e3:.3f} ms/frame -> {rays / dt / 1e6:.0f} Mrays/s; canvas {img.nbytes / 1e6:.1f} MB -> {img.nbytes / dt / 1e9:.1f} GB/s effective")
pinned = rtc.host_canvas(cam.vsize, cam.hsize)
for _ in range(3):
    dw.render(cam, out=pinned)
t = time.perf_counter()
for _ in range(n):
    dw.render(cam, out=pinned)
dt = (time.perf_counter() - t) / n
assert (pinned == img).all()
print(f"rtc_render (rtc_host_alloc canvas) {dt * 1e3:.3f} ms/frame -> {rays / dt / 1e6:.0f} Mrays/s; {img.nbytes / dt / 1e9:.1f} GB/s effective")
