"""One-off stress campaign (not part of the test suite): many random adversarial worlds, the culled
kernels (one- and two-level) against plain brute force on the GPU, bit for bit (canvas + ray counts);
every 10th world also against the CPU oracle. Usage: python tests/stress_parity.py [n_worlds] [seed0] [pipeline_depth]
(RTC_BIN_SMALL_PIXELS=0 / RTC_BIN_SMALL_PIXELS_PIPELINED=0 in the environment force the binned primary pass on these small frames.)"""
import importlib
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "oracle"), str(ROOT / "tests")]
import oracle as O  # noqa: E402
from _bootstrap import package  # noqa: E402

rtc = package()
from test_gpu_parity import adversarial_scene  # noqa: E402

n_worlds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
pipeline = int(sys.argv[3]) if len(sys.argv) > 3 else 1


def big_world(seed):
    """Up to ~2000 small objects (forces the two-level cull) with a few big / odd ones mixed in."""
    rng = np.random.default_rng(seed)
    u = lambda a, b: float(rng.uniform(a, b))
    w = rtc.World(rtc.light((u(-8, 8), u(2, 12), u(-10, 0))))
    n = int(rng.integers(257, 2000))
    for i in range(n):
        r = u(0.02, 0.4) if rng.random() < 0.97 else u(1.0, 6.0)
        t = rtc.Matrix.identity().scaling(r, r * u(0.5, 1.5), r).rotation_y(u(0, 3)).translation(u(-12, 12), u(-1, 8), u(-6, 30))
        glass = rng.random() < 0.03
        m = rtc.material(color=(u(0, 1), u(0, 1), u(0, 1)), ambient=u(0, 0.3), diffuse=u(0.3, 0.9), specular=u(0, 0.5), shininess=u(5, 100),
                         reflective=(u(0.1, 0.6) if rng.random() < 0.05 else 0.0), transparency=(u(0.3, 0.9) if glass else 0.0),
                         refractive_index=(u(1.1, 1.9) if glass else 1.0))
        w.add_shape((rtc.cube if rng.random() < 0.1 else rtc.sphere)(t, m))
    if rng.random() < 0.6:
        w.add_shape(rtc.plane(rtc.Matrix.identity(), rtc.material(specular=0.0, pattern=("checker", (0.3,) * 3, (0.7,) * 3, None))))
    cam = rtc.camera(64, 40, u(0.5, 1.4), rtc.Matrix.make_view_transform((u(-3, 3), u(0.5, 5), u(-10, -4)), (u(-1, 1), u(0, 2), u(2, 8)), (0, 1, 0)))
    return w, cam


def far_world(seed):
    """The f32 wave-level cull's worst cases: the whole scene (objects, camera, light) translated far from the origin
    (centres and apex lose up to 2^-24 of 1e3..1e7 when rounded to f32 — more than many of the radii), the scene scaled
    by 1e-3..1e3, tiny spheres far away, a few hundred objects now and then (two-level walk)."""
    rng = np.random.default_rng(seed)
    u = lambda a, b: float(rng.uniform(a, b))
    off = [0.0, 0.0, 0.0]
    if rng.random() < 0.8:
        mag = 10.0 ** u(2, 7)
        off = [mag * u(-1, 1), mag * u(-1, 1) * 0.3, mag * u(-1, 1)]
    sc = 10.0 ** u(-3, 3) if rng.random() < 0.5 else 1.0
    P = lambda x, y, z: (off[0] + sc * x, off[1] + sc * y, off[2] + sc * z)
    w = rtc.World(rtc.light(P(u(-8, 8), u(2, 12), u(-10, 0))))
    n = int(rng.integers(300, 700)) if rng.random() < 0.25 else int(rng.integers(5, 60))
    for i in range(n):
        r = sc * (u(0.001, 0.02) if rng.random() < 0.3 else u(0.05, 1.5))
        t = rtc.Matrix.identity().scaling(r, r * u(0.3, 1.7), r).rotation_z(u(0, 3)).translation(*P(u(-10, 10), u(-1, 6), u(-6, 40)))
        m = rtc.material(color=(u(0, 1), u(0, 1), u(0, 1)), ambient=u(0, 0.3), diffuse=u(0.3, 0.9), specular=u(0, 0.5), shininess=u(5, 100),
                         reflective=(u(0.1, 0.6) if rng.random() < 0.15 else 0.0))
        try:
            w.add_shape((rtc.cube if rng.random() < 0.15 else rtc.sphere)(t, m))
        except rtc.RtcError:
            pass  # singular by the reference's 1e-8 determinant rule
    if rng.random() < 0.5:
        w.add_shape(rtc.plane(rtc.Matrix.identity().translation(*P(0, u(-1, 0), 0)), rtc.material(specular=0.0, reflective=u(0, 0.4))))
    cam = rtc.camera(56, 40, u(0.4, 1.6), rtc.Matrix.make_view_transform(P(u(-3, 3), u(0.5, 5), u(-10, -4)), P(u(-1, 1), u(0, 2), u(2, 8)), (0, 1, 0)))
    return w, cam


def mirror_world(seed):
    """One-level worlds (<= 256 objects) of small MIRRORS: reflection rays off a sphere of a few pixels fan out over a
    hemisphere, no cone holds them, and the pass takes the per-lane walk over groups of 8 (walk_per_lane, round 3) with its
    distance limit; now and then glass (refraction rays likewise), cubes, flattened ellipsoids, a mirror floor, far offsets."""
    rng = np.random.default_rng(seed)
    u = lambda a, b: float(rng.uniform(a, b))
    off = [0.0, 0.0, 0.0]
    if rng.random() < 0.2:
        mag = 10.0 ** u(1, 5)
        off = [mag * u(-1, 1), mag * u(-1, 1) * 0.2, mag * u(-1, 1)]
    P = lambda x, y, z: (off[0] + x, off[1] + y, off[2] + z)
    w = rtc.World(rtc.light(P(u(-8, 8), u(3, 12), u(-10, 0))))
    n = int(rng.integers(3, 250))
    for i in range(n):
        r = u(0.03, 0.25) if rng.random() < 0.8 else u(0.4, 2.0)
        flat = u(0.05, 1.0) if rng.random() < 0.2 else 1.0
        t = rtc.Matrix.identity().scaling(r, r * flat, r).rotation_x(u(0, 3)).translation(*P(u(-6, 6), u(0, 4), u(-3, 14)))
        glass = rng.random() < 0.15
        m = rtc.material(color=(u(0, 1), u(0, 1), u(0, 1)), ambient=u(0, 0.3), diffuse=u(0.2, 0.9), specular=u(0, 0.9), shininess=u(5, 300),
                         reflective=(u(0.05, 1.0) if rng.random() < 0.8 else 0.0), transparency=(u(0.2, 1.0) if glass else 0.0),
                         refractive_index=(u(1.0, 2.0) if glass else 1.0))
        try:
            w.add_shape((rtc.cube if rng.random() < 0.15 else rtc.sphere)(t, m))
        except rtc.RtcError:
            pass
    if rng.random() < 0.7:
        w.add_shape(rtc.plane(rtc.Matrix.identity().translation(*P(0, u(-0.5, 0), 0)),
                              rtc.material(specular=0.0, reflective=u(0, 0.6), pattern=("checker", (0.3,) * 3, (0.7,) * 3, None))))
    cam = rtc.camera(64, 40, u(0.5, 1.4), rtc.Matrix.make_view_transform(P(u(-3, 3), u(0.5, 5), u(-10, -4)), P(u(-1, 1), u(0, 2), u(2, 8)), (0, 1, 0)))
    return w, cam


ctx = rtc.Context(0)
if pipeline > 1:
    ctx.set_pipeline(pipeline)
bad = 0
t0 = time.time()
for k in range(n_worlds):
    seed = seed0 + k
    w, cam = mirror_world(seed) if k % 5 == 4 else (far_world(seed) if k % 4 == 3 else (big_world(seed) if k % 3 == 2 else adversarial_scene(rtc, seed)))
    dw = ctx.upload(w)
    got, st = dw.render(cam, rtc.MODE_RENDER_ASYNC, with_stats=True)
    brute, sb = dw.render(cam, rtc.MODE_RENDER_ASYNC, flags=1, with_stats=True)
    ok = np.array_equal(got, brute) and st == sb
    if ok and k % 10 == 0:
        want, ost = O.render(w.array(), len(w), w.light, cam, mode=1, nthreads=16, want_stats=True)
        ok = float(np.max(np.abs(got - want))) <= 1e-12 and st == ost
    if not ok:
        bad += 1
        print(f"MISMATCH seed {seed} objects {len(w)} max|d|={np.max(np.abs(got - brute)):.3e} stats {st} vs {sb}", flush=True)
    dw.close()
    if k % 2000 == 1999:
        print(f"{k + 1} worlds, {bad} mismatches, {time.time() - t0:.0f}s", flush=True)
print(f"done: {n_worlds} worlds, {bad} mismatches")
sys.exit(1 if bad else 0)
