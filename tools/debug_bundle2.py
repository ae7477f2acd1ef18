"""Emulate, on the CPU, every shadow bundle of one 8x8 tile through the whole recursion (the GPU's
lock-step order: iteration t = each lane's t-th color_at call in depth-first order) and report
objects that occlude a lane's shadow ray but fail the emulated cull test."""
import ctypes as C, sys
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
arr = w.array(); n = len(w)
L = np.array(list(w.light.position)); f32 = np.float32

def bounds():
    out = []
    for s in w.shapes:
        if s.kind == 1: out.append(None); continue
        inv = np.array(list(s.inv)).reshape(4, 4)
        F = np.linalg.inv(inv[:3, :3]); c = -F @ inv[:3, 3]
        r = np.linalg.svd(F, compute_uv=False).max() if s.kind == 0 else max(np.linalg.norm(F @ np.array([a, b, d])) for a in (-1, 1) for b in (-1, 1) for d in (-1, 1))
        out.append((c, r * (1 + 1e-6) + 1e-9 * (1 + np.linalg.norm(c)) + 1e-7 * np.abs(F).max()))
    return out
B = bounds()

def calls(ray, rem, out):
    """depth-first sequence of color_at calls of one pixel: list of hit records (or None)"""
    rgb, h = O.color_at(arr, n, w.light, ray, rem, want_hit=True)
    if h.hit_index < 0: out.append(None); return
    out.append((h, list(ray)))
    m = arr[h.hit_index].material
    if rem != 0 and m.reflective > 0:
        calls(list(h.over_point) + list(h.reflectv), rem - 1, out)
    if rem != 0 and m.transparency != 0:
        nr = h.n1 / h.n2; ci = sum(a * b for a, b in zip(h.eyev, h.normal)); s2 = nr * nr * (1 - ci * ci)
        if not s2 > 1:
            ct = (1 - s2) ** 0.5
            calls(list(h.under_point) + [h.normal[k] * (nr * ci - ct) - h.eyev[k] * nr for k in range(3)], rem - 1, out)

seqs = []
for l in range(64):
    out = []; calls(rtc.ray_for_pixel(cam, tx0 + (l & 7), ty0 + (l >> 3)), 5, out); seqs.append(out)
T = max(len(s) for s in seqs)
print("iterations", T)
for t in range(T):
    lanes = [i for i in range(64) if t < len(seqs[i]) and seqs[i][t] is not None]
    if not lanes: continue
    over = {i: np.array(list(seqs[i][t][0].over_point)) for i in lanes}
    dist = {i: np.linalg.norm(L - over[i]) for i in lanes}
    sdir = {i: (L - over[i]) / dist[i] for i in lanes}
    f = {}
    for i in lanes:
        x = (-sdir[i]).astype(f32); f[i] = x / f32(np.sqrt((x * x).sum()))
    hi = [i for i in lanes if i >= 27]; lane0 = hi[0] if hi else lanes[-1]
    a = f[lane0]
    dot = {i: float((f[i] * a).sum()) for i in lanes}; q2 = {i: float((np.cross(a, f[i]) ** 2).sum()) for i in lanes}
    narrow = all(dot[i] > 0.7 for i in lanes)
    if narrow:
        sinT = f32(np.sqrt(f32(max(q2.values())))) * f32(1.001) + f32(4e-6); cosT = f32(np.sqrt(max(0, 1 - sinT * sinT)))
    else:
        cmin = 1 - max(max(0, 1 - dot[i]) for i in lanes); cosT = f32(cmin - 1e-3); sinT = f32(np.sqrt(max(0, 1 - cosT * cosT)))
    off = (not narrow and not cmin > 0.2) or not sinT < 0.98
    tmax = float(max(f32(dist[i]) * f32(1.0001) + f32(1e-30) for i in lanes))
    sinT, cosT, ax = float(sinT), float(cosT), a.astype(np.float64)
    for j in range(n):
        if B[j] is None or off: continue
        c, r = B[j]; wv = c - L
        Re = r * 1.00001 + 1e-6 * np.abs(wv).sum() + 1e-12
        d2 = wv @ wv; wa = wv @ ax
        if d2 <= Re * Re: touch, why = True, "apex inside"
        elif wa < -Re: touch, why = False, "behind"
        elif d2 > (tmax + Re) ** 2: touch, why = False, f"beyond reach tmax={tmax:.6g}"
        else:
            rhs = Re + (wa + abs(wa) * 1e-5) * sinT; perp2 = d2 - wa * wa * 1.00001
            touch = not (rhs < 0 or perp2 * cosT * cosT > rhs * rhs); why = f"cone lhs={perp2 * cosT * cosT:.8g} rhs2={rhs * rhs:.8g} sinT={sinT:.6g}"
        for i in lanes:
            ts = (C.c_double * 2)()
            k = O.lib().orc_shape_intersect(C.byref(arr[j]), O.Ray6(*(list(over[i]) + list(sdir[i]))), ts)
            if any(0.0 <= ts[q] < dist[i] for q in range(k)) and not touch:
                print(f"iter {t}: OBJECT {j} kind {arr[j].kind} occludes lane {i} t={[ts[q] for q in range(k)]} dist={dist[i]:.6g} but CULLED: {why}; d2={d2:.6g} wa={wa:.6g} Re={Re:.6g} lanes={len(lanes)} axis_lane={lane0} narrow={narrow}")
print("done")
t = 2
lanes = [i for i in range(64) if t < len(seqs[i]) and seqs[i][t] is not None]
over = {i: np.array(list(seqs[i][t][0].over_point)) for i in lanes}
print("lane 11 over", over[11], "hit idx", seqs[11][t][0].hit_index, "t", seqs[11][t][0].t)
print("lane 27 over", over[27])
d11 = (over[11] - L); d27 = (over[27] - L)
print("angle between lane 11 and lane 27 directions from the light (deg):", np.degrees(np.arccos(d11 @ d27 / np.linalg.norm(d11) / np.linalg.norm(d27))))
c, r = B[2]; wv = c - L
print("object 2 centre", c, "r", r, "angle centre-vs-lane11 dir", np.degrees(np.arccos(wv @ d11 / np.linalg.norm(wv) / np.linalg.norm(d11))))
tt = wv @ d11 / np.linalg.norm(d11); print("closest approach of lane 11's segment to centre:", np.linalg.norm(wv - tt * d11 / np.linalg.norm(d11)), "at distance", tt, "from light")
s = w.shapes[2]
inv = np.array(list(s.inv)).reshape(4, 4)
print("inv =\n", inv)
ov = over[11]; sd = (L - ov) / np.linalg.norm(L - ov)
for tt in (1141411.1384748667, 1141411.1652308572):
    P = ov + sd * tt
    q = inv[:3, :3] @ P + inv[:3, 3]
    print("hit point", P, "dist from light", np.linalg.norm(P - L), "object-space norm", np.linalg.norm(q), "dist from bound centre", np.linalg.norm(P - B[2][0]))
import mpmath as mp
mp.mp.dps = 60
o = [mp.mpf(float(x)) for x in ov]; d = [mp.mpf(float(x)) for x in sd]
M = [[mp.mpf(float(inv[r, c])) for c in range(4)] for r in range(3)]
op = [M[r][0] * o[0] + M[r][1] * o[1] + M[r][2] * o[2] + M[r][3] for r in range(3)]
dp = [M[r][0] * d[0] + M[r][1] * d[1] + M[r][2] * d[2] for r in range(3)]
a = sum(x * x for x in dp); b = 2 * sum(x * y for x, y in zip(dp, op)); c = sum(x * x for x in op) - 1
print("exact disc (mpmath):", b * b - 4 * a * c, " a", a, "b", b, "c", c)
