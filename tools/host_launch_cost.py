"""Host cost of one render launch through the Python binding (ctypes + rtc_render_bands + HIP launch):
many launches of a tiny frame, so the GPU is never the limit."""
import importlib
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT)]
from _bootstrap import package  # noqa: E402

rtc = package()
scenes = importlib.import_module(rtc.__name__ + ".scenes")
import torch  # noqa: E402

w, cam = scenes.synthetic(100, 64, 16)
ctx = rtc.Context(0)
dw = ctx.upload(w)
f = torch.zeros((16, 64, 3), dtype=torch.float64, device="cuda:0")
q = torch.zeros((16, 64, 3), dtype=torch.uint8, device="cuda:0")
fp, qp = f.data_ptr(), q.data_ptr()
for n in (2000, 20000):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        dw.render_bands(cam, 0, 1, fp, d_ptr8=qp)
    t_enq = time.perf_counter() - t
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t
    print(f"{n} launches: {t_enq / n * 1e6:.2f} us per call to enqueue, {t_all / n * 1e6:.2f} us per launch until done")
