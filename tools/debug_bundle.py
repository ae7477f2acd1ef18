"""Emulate the shadow bundle of one 8x8 tile on the CPU (numpy f32/f64 mirroring make_bundle /
bundle_touches) and report objects that occlude some lane's shadow ray but fail the cull test."""
import ctypes as C, importlib, sys
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
L = np.array(list(w.light.position))
f32 = np.float32

def bounds():
    # mirror of bound_of() is in C++; recompute a generous bound here from inv numerically
    out = []
    for s in w.shapes:
        if s.kind == 1: out.append(None); continue
        inv = np.array(list(s.inv)).reshape(4, 4)
        F = np.linalg.inv(inv[:3, :3]); c = -F @ inv[:3, 3]
        if s.kind == 0: r = np.linalg.svd(F, compute_uv=False).max()
        else: r = max(np.linalg.norm(F @ np.array([sx, sy, sz])) for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1))
        out.append((c, r * (1 + 1e-6) + 1e-9 * (1 + np.linalg.norm(c)) + 1e-7 * np.abs(F).max()))
    return out
B = bounds()

def trace_level(rays, active, depth):
    hits = []
    for i in range(64):
        if not active[i]: hits.append(None); continue
        rgb, h = O.color_at(arr, n, w.light, rays[i], 5, want_hit=True)
        hits.append(h if h.hit_index >= 0 else None)
    # shadow bundle
    good = [h is not None for h in hits]
    if not any(good): return
    over = np.array([list(h.over_point) if h else [0, 0, 0] for h in hits])
    v = L - over; dist = np.linalg.norm(v, axis=1); sdir = v / np.where(dist > 0, dist, 1)[:, None]
    d = -sdir
    f = d.astype(f32); l2 = (f * f).sum(axis=1); f = f / np.sqrt(np.where(l2 > 0, l2, 1))[:, None].astype(f32)
    gmask = [i for i in range(64) if good[i]]
    hi = [i for i in gmask if i >= 27]
    lane0 = hi[0] if hi else gmask[-1]
    a = f[lane0]
    dot = (f * a).sum(axis=1); cr = np.cross(np.tile(a, (64, 1)), f); q2 = (cr * cr).sum(axis=1)
    q2max = max(q2[i] for i in gmask); narrow = all(dot[i] > 0.7 for i in gmask)
    if narrow:
        sinT = f32(np.sqrt(q2max)) * f32(1.001) + f32(4e-6); cosT = f32(np.sqrt(max(0, 1 - sinT * sinT)))
    else:
        cmin = 1 - max(max(0, 1 - dot[i]) for i in gmask); cosT = f32(cmin - 1e-3); sinT = f32(np.sqrt(max(0, 1 - cosT * cosT)) + 1e-3)
    tmax = max(f32(dist[i]) * f32(1.0001) + f32(1e-30) for i in gmask)
    sinT, cosT, tmax, ax = float(sinT), float(cosT), float(tmax), a.astype(np.float64)
    print(f"depth {depth}: {len(gmask)} hit lanes, axis lane {lane0}, narrow {narrow}, sinT {sinT:.6g} cosT {cosT:.6g} tmax {tmax:.6g}")
    for j in range(n):
        if B[j] is None: continue
        c, r = B[j]
        Re = r * 1.00001 + 1e-12
        wv = c - L; d2 = wv @ wv; wa = wv @ ax
        if d2 <= Re * Re: touch = True; why = "apex inside"
        elif wa < -Re: touch = False; why = "behind"
        elif d2 > (tmax + Re) ** 2: touch = False; why = "beyond reach"
        else:
            rhs = Re + (wa + abs(wa) * 1e-5) * sinT
            perp2 = d2 - wa * wa * 1.00001
            touch = not (rhs < 0 or perp2 * cosT * cosT > rhs * rhs); why = f"cone perp2*c2={perp2 * cosT * cosT:.6g} rhs2={rhs * rhs:.6g}"
        for i in gmask:
            ts = (C.c_double * 2)()
            ray = list(over[i]) + list(sdir[i])
            k = O.lib().orc_shape_intersect(C.byref(arr[j]), O.Ray6(*ray), ts)
            occ = any(0.0 <= ts[q] < dist[i] for q in range(k))
            if occ and not touch:
                print(f"  OBJECT {j} kind {arr[j].kind} occludes lane {i} (t={[ts[q] for q in range(k)]}, dist={dist[i]:.6g}) but is CULLED: {why}; d2={d2:.6g} wa={wa:.6g} Re={Re:.6g}")
    return hits

rays = []
for ly in range(8):
    for lx in range(8):
        rays.append(rtc.ray_for_pixel(cam, tx0 + lx, ty0 + ly))
hits = trace_level(rays, [True] * 64, 0)
# one level of secondary rays (reflection, then refraction) for lanes that have them
for name in ("reflect", "refract"):
    rays2, act = [], []
    for i, h in enumerate(hits):
        m = arr[h.hit_index].material if h else None
        if h and name == "reflect" and m.reflective > 0:
            rays2.append(list(h.over_point) + list(h.reflectv)); act.append(True)
        elif h and name == "refract" and m.transparency != 0:
            nr = h.n1 / h.n2; ci = sum(a * b for a, b in zip(h.eyev, h.normal)); s2 = nr * nr * (1 - ci * ci)
            if s2 > 1: rays2.append([0] * 6); act.append(False); continue
            ct = (1 - s2) ** 0.5
            dvec = [h.normal[k] * (nr * ci - ct) - h.eyev[k] * nr for k in range(3)]
            rays2.append(list(h.under_point) + dvec); act.append(True)
        else:
            rays2.append([0] * 6); act.append(False)
    print(name, sum(act), "lanes")
    if any(act): trace_level(rays2, act, 1)

# ---- extra diagnostics for the depth-0 shadow bundle
over = np.array([list(h.over_point) for h in hits])
v = L - over; dist = np.linalg.norm(v, axis=1); sdir = v / dist[:, None]; d = -sdir
a = d[27] / np.linalg.norm(d[27])
ang = np.degrees(np.arccos(np.clip(d @ a, -1, 1)))
print("angles to axis: max", ang.max(), "lane62", ang[62], "min dot", (d @ a).min())
c, r = B[22]
wv = c - L
print("sphere centre dist", np.linalg.norm(wv), "r", r, "angle of centre to axis", np.degrees(np.arccos(wv @ a / np.linalg.norm(wv))))
# distance from centre to lane 62's ray from the light
t = wv @ d[62]; perp = np.linalg.norm(wv - t * d[62]); print("lane62 ray: closest approach to centre", perp, "at t", t)
