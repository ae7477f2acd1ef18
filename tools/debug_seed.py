import importlib, sys, os
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "oracle"), str(ROOT / "tests")]
import oracle as O
from _bootstrap import package
rtc = package()
from test_gpu_parity import adversarial_scene, hit_fields
seed = int(sys.argv[1])
w, cam = adversarial_scene(rtc, seed)
want = O.render(w.array(), len(w), w.light, cam, mode=1, nthreads=16)
for src in (None, 0, 1, 3, 4):
    if src is None: os.environ.pop("RTC_SRC", None)
    else: os.environ["RTC_SRC"] = str(src)
    ctx = rtc.Context(0)
    dw = ctx.upload(w)
    got = dw.render(cam)
    bad = np.argwhere(np.abs(got - want).max(axis=2) > 1e-12)
    print("src", src, "bad pixels", len(bad), bad[:5].tolist())
    if len(bad) and src in (None, 3):
        y, x = bad[0]
        ray = rtc.ray_for_pixel(cam, int(x), int(y))
        for rem in range(0, 6):
            g = dw.color_at(np.array([ray]), rem)[0]
            o = O.color_at(w.array(), len(w), w.light, ray, rem)
            print("  rem", rem, "gpu", g, "oracle", o, "diff", np.abs(g - o).max())
    dw.close(); ctx.close()
kinds = [(s.kind, round(s.material.reflective, 3), round(s.material.transparency, 3)) for s in w.shapes]
print(kinds)
