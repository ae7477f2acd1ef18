"""Probe the 64 primary rays of one 8x8 tile together (same wave composition as the render kernel)
at increasing `remaining`, against the oracle, culled vs brute."""
import importlib, os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "oracle"), str(ROOT / "tests")]
import oracle as O
from _bootstrap import package
rtc = package()
from test_gpu_parity import adversarial_scene
seed, tx0, ty0 = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
w, cam = adversarial_scene(rtc, seed)
rays = np.array([rtc.ray_for_pixel(cam, tx0 + (l & 7), ty0 + (l >> 3)) for l in range(64)])
ctx = rtc.Context(0); dw = ctx.upload(w)
for rem in range(6):
    g = dw.color_at(rays, rem)
    b = dw.color_at(rays, rem, flags=1)
    o = np.array([O.color_at(w.array(), len(w), w.light, r, rem) for r in rays])
    bad = np.argwhere(np.abs(g - o).max(axis=1) > 1e-12).ravel()
    print("rem", rem, "culled-vs-oracle bad lanes", bad.tolist(), "brute-vs-oracle bad", np.argwhere(np.abs(b - o).max(axis=1) > 1e-12).ravel().tolist(),
          "maxdiff", np.abs(g - o).max())
print([(s.kind, round(s.material.reflective, 2), round(s.material.transparency, 2), round(s.material.refractive_index, 2)) for s in w.shapes])
print("light", list(w.light.position))
