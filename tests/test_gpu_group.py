"""GPU tests of the round-2 additions to the C-ABI: row tiles across GPUs behind the boundary (rtc_group),
the anti-aliasing branch with its resample test, world-id semantics of compute_refractive, page-locked
caller canvases. Everything through librtc.so; the oracle is the checker."""
import importlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TIGHT_TOL = 1e-12


@pytest.fixture(scope="module")
def scenes(rtc):
    return importlib.import_module(rtc.__name__ + ".scenes")


def _orbit(rtc, W, H, n):
    import math
    return [rtc.camera(W, H, 0.7 + 0.03 * i, rtc.Matrix.make_view_transform((2.5 * math.sin(0.5 * i), 2.0, -8.0 + 0.5 * i), (0.0, 1.0, 5.0), (0.0, 1.0, 0.0)))
            for i in range(n)]


# ------------------------------------------------------------------ rtc_group
@pytest.mark.parametrize("kind", ["rccl_in_process_1", "rccl_rank_1", "p2p_3", "p2p_8", "p2p_2_short_last_band"])
def test_group_render_assembles_the_single_gpu_frame(rtc, gpu, scenes, kind):
    """rtc_group_render (bands dealt over N members, gather of the f64 tiles to member 0, un-deal kernel) must
    deliver exactly the canvases one GPU renders (camera.rs:144-160: pixels are independent), f64 and 8-bit,
    several frames per call, two calls back to back (double-buffered tiles), ray counts summed over members.
    N > 1 is rehearsed on the 1-GPU box with the peer-copy exchange, which accepts one device several times;
    the RCCL exchange (ncclCommInitAll / ncclCommInitRank + ncclGather) runs with N = 1."""
    import torch
    W, H = (176, 93) if kind != "p2p_2_short_last_band" else (96, 43)
    w, _ = scenes.synthetic(30, W, H, reflective=(kind == "p2p_3"))
    cams = _orbit(rtc, W, H, 3)
    if kind == "rccl_in_process_1":
        g = rtc.Group(devices=[0], exchange=rtc.EXCHANGE_RCCL)
    elif kind == "rccl_rank_1":
        g = rtc.Group(device=0, nranks=1, rank=0, uid=rtc.group_unique_id())
    else:
        g = rtc.Group(devices=[0] * int(kind.split("_")[1]), exchange=rtc.EXCHANGE_P2P)
    assert g.size == g.local_size == len(g.contexts)
    gw = g.upload(w)
    dw = gpu.upload(w)
    singles, total = [], {}
    for c in cams:
        img, st = dw.render(c, rtc.MODE_RENDER_ASYNC, with_stats=True)
        singles.append(img)
        for k, v in st.items():
            total[k] = total.get(k, 0) + v
    canvas = torch.full((3, H, W, 3), -1.0, dtype=torch.float64, device="cuda:0")
    frame8 = torch.full((3, H, W, 3), 9, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    g.reset_stats()
    gw.render(cams, rtc.GATHER_F64 | rtc.GATHER_U8, canvas.data_ptr(), frame8.data_ptr())
    g.synchronize()
    assert g.stats() == total
    ch, qh = canvas.cpu().numpy(), frame8.cpu().numpy()
    for v in range(3):
        assert np.array_equal(ch[v], singles[v]), (kind, v)
        assert np.array_equal(qh[v], rtc.color_scale255(singles[v]).reshape(H, W, 3)), (kind, v)
    # back to back batches without synchronising in between (tile buffers alternate, exchange overlaps the next render)
    outs = [torch.zeros((2, H, W, 3), dtype=torch.float64, device="cuda:0") for _ in range(4)]
    for k, o in enumerate(outs):
        gw.render([cams[k % 3], cams[(k + 1) % 3]], rtc.GATHER_F64, o.data_ptr())
    g.synchronize()
    for k, o in enumerate(outs):
        oh = o.cpu().numpy()
        assert np.array_equal(oh[0], singles[k % 3]) and np.array_equal(oh[1], singles[(k + 1) % 3]), (kind, k)
    # the 8-bit frame alone (3 B/pixel exchange)
    frame8.fill_(3)
    gw.render(cams[:1], rtc.GATHER_U8, None, frame8.data_ptr())
    g.synchronize()
    assert np.array_equal(frame8[0].cpu().numpy(), rtc.color_scale255(singles[0]).reshape(H, W, 3))
    with pytest.raises(rtc.RtcError):
        gw.render(cams, rtc.GATHER_F64, None)          # member 0 needs a canvas
    with pytest.raises(rtc.RtcError):
        gw.render(cams * 3, rtc.GATHER_F64, canvas.data_ptr())   # more than 8 frames per call
    gw.close()
    g.close()
    dw.close()


@pytest.mark.parametrize("n", [1, 4])
def test_group_render_host_fills_the_callers_canvas(rtc, gpu, scenes, n):
    """rtc_group_render_host: every member DMAs its bands straight into the caller's host canvas (no gather);
    page-locked (rtc_host_alloc), registered (rtc_host_register on a caller allocation) and plain pageable
    memory all receive the frame one GPU renders."""
    W, H = 200, 117   # 15 bands, the last one 5 rows
    w, cam = scenes.synthetic(25, W, H)
    want, wst = gpu.upload(w).render(cam, with_stats=True)
    g = rtc.Group(devices=[0] * n, exchange=rtc.EXCHANGE_P2P)
    gw = g.upload(w)
    pinned = rtc.host_canvas(H, W)
    got, st = gw.render_host(cam, pinned, with_stats=True)
    assert np.array_equal(got, want) and st == wst
    plain = np.full((H, W, 3), -2.0)
    assert np.array_equal(gw.render_host(cam, plain), want)
    mine = np.full((H, W, 3), -3.0)
    rtc.host_register(mine)
    try:
        assert np.array_equal(gw.render_host(cam, mine), want)
        # the same registered canvas through the single-GPU entry point
        assert np.array_equal(gpu.upload(w).render(cam, out=mine), want)
    finally:
        rtc.host_unregister(mine)
    gw.close()
    g.close()


def test_group_argument_errors(rtc):
    with pytest.raises(rtc.RtcError):
        rtc.Group(devices=[0, 0], exchange=rtc.EXCHANGE_RCCL)   # RCCL refuses one device twice
    with pytest.raises(rtc.RtcError):
        rtc.Group(devices=[], exchange=rtc.EXCHANGE_P2P)
    with pytest.raises(rtc.RtcError):
        rtc.Group(device=0, nranks=2, rank=2, uid=bytes(128))


# ------------------------------------------------------------------ context cost
def test_context_creation_is_cheap(rtc):
    """A context no longer creates its 2048 timing events up front: creating and destroying one, and rendering a
    first frame through a fresh one, must stay far below a frame's PCIe copy (drop-in call path)."""
    import time
    rtc.Context(0).close()     # runtime warm
    t = time.perf_counter()
    for _ in range(20):
        rtc.Context(0).close()
    per = (time.perf_counter() - t) / 20
    assert per < 2e-3, per


# ------------------------------------------------------------------ anti-aliasing branch
def test_antialiasing_branch_trigger_and_resample(rtc, gpu, O, scenes):
    """render_pixel camera.rs:94-114 on the device: samples == 0 and samples == 4 take the 4-sub-sample branch and
    count the pixels that trip the resample test (bit-identical mask); with FLAG_AA_RESAMPLE the extra rays use
    the documented counter-based offsets, which the oracle restates — canvases equal to 1e-12, ray counts exact.
    (Against the reference itself the resample is only statistically comparable: its offsets are thread_rng.)"""
    W, H = 96, 54
    for refl in (False, True):
        w, cam = scenes.synthetic(30, W, H, reflective=refl)
        dw = gpu.upload(w)
        arr = w.array()
        for samples, flags in ((0, 0), (4, 0), (0, rtc.FLAG_AA_RESAMPLE), (4, rtc.FLAG_AA_RESAMPLE), (7, rtc.FLAG_AA_RESAMPLE),
                               (4, rtc.FLAG_AA_RESAMPLE | rtc.FLAG_NO_CULL)):
            cam.samples = samples
            got, st = dw.render(cam, rtc.MODE_RENDER_ASYNC, flags=flags, with_stats=True)
            want, ost = O.render(arr, len(w), w.light, cam, mode=1, nthreads=8, want_stats=True, flags=flags)
            assert st == ost, (refl, samples, flags, st, ost)
            assert np.max(np.abs(got - want)) <= TIGHT_TOL, (refl, samples, flags)
            assert 0 < st["pixels_resample"] < W * H
            extra = samples if flags & rtc.FLAG_AA_RESAMPLE else 0
            assert st["rays_primary"] == 4 * W * H + extra * st["pixels_resample"]
        cam.samples = 256
        with pytest.raises(rtc.RtcError):
            dw.render(cam)
        dw.close()


def test_antialiasing_through_every_object_source(rtc, O, scenes):
    """The sub-sample store lives behind the object tiles in dynamic LDS: run the AA branch with the resample
    through the LDS-tile variants too (RTC_SRC=1, 2) and the two-level cull (RTC_SRC=4)."""
    from test_gpu_parity import make_ctx
    w, cam = scenes.synthetic(40, 80, 45, samples=5)
    want, ost = O.render(w.array(), len(w), w.light, cam, mode=1, nthreads=8, want_stats=True, flags=rtc.FLAG_AA_RESAMPLE)
    for src, cap in ((1, None), (2, 16), (4, None), (0, None)):
        ctx = make_ctx(rtc, src, cap)
        got, st = ctx.upload(w).render(cam, flags=rtc.FLAG_AA_RESAMPLE, with_stats=True)
        assert st == ost and np.max(np.abs(got - want)) <= TIGHT_TOL, src
        ctx.close()


# ------------------------------------------------------------------ world ids
def test_world_ids_assigned_and_shared_ids_honoured(rtc, gpu, O, scenes):
    """compute_refractive keys on world_id (shape.rs:127). (1) Shapes from rtc_shape_init carry id 0; when every id
    is 0 rtc_world_create numbers them like World::add_shape (shape.rs:661-667) — same canvas as explicit 1..n.
    (2) Ids that collide (modulo 3 / 7 here; modulo 256 = the reference's own u8 wrap at 300 shapes) are ONE
    container: hit records (n1, n2) bit-identical to the literal sorted-list oracle, canvases to 1e-12."""
    w, cam = scenes.glass_cluster(30, 64, 48)
    explicit = gpu.upload(w).render(cam)
    w0, _ = scenes.glass_cluster(30, 64, 48)
    for s in w0.shapes:
        s.world_id = 0
    assert np.array_equal(gpu.upload(w0).render(cam), explicit)
    for n, mod, size in ((24, 3, (64, 48)), (40, 7, (64, 48)), (300, 256, (40, 24))):
        w, cam = scenes.glass_cluster(n, size[0], size[1], id_modulus=mod)
        dw = gpu.upload(w)
        arr = w.array()
        got, st = dw.render(cam, with_stats=True)
        want, ost = O.render(arr, len(w), w.light, cam, mode=1, nthreads=8, want_stats=True)
        assert st == ost and np.max(np.abs(got - want)) <= TIGHT_TOL, (n, mod)
        rays = np.array([rtc.ray_for_pixel(cam, x, y) for y in range(0, size[1], 3) for x in range(0, size[0], 3)])
        rgb, hits = dw.color_at(rays, 5, want_hits=True)
        differs = 0
        for i, r in enumerate(rays):
            orgb, oh = O.color_at(arr, len(w), w.light, r, 5, want_hit=True)
            assert (hits[i].hit_index, hits[i].t, hits[i].n1, hits[i].n2) == (oh.hit_index, oh.t, oh.n1, oh.n2), (n, mod, i)
            assert np.max(np.abs(rgb[i] - orgb)) <= TIGHT_TOL
        # shared ids are not a no-op: the same shapes with unique ids give another picture
        wu, _ = scenes.glass_cluster(n, size[0], size[1], id_modulus=0)
        assert not np.array_equal(gpu.upload(wu).render(cam), got)
        dw.close()


# ------------------------------------------------------------------ binned primary pass
def _ctx_env(rtc, **env):
    old = {k: os.environ.get(k) for k in env}
    try:
        for k, v in env.items():
            os.environ[k] = str(v)
        return rtc.Context(0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_binned_primary_pass_and_light_lists_equal_the_group_walk(rtc, O, scenes):
    """Two-level worlds take their primary rays' candidates from per-view tile lists (k_bin_tiles) and their shadow rays'
    candidates from the light-space direction-cell lists built once per World
    (k_light_cells / k_light_bin), instead of walking the groups. Same conservative predicate, so the canvases must equal
    the walk's (RTC_BINNING=0 RTC_LIGHT_LISTS=0) and brute force bit for bit with identical ray counts — including the cases
    the lists cannot serve:
    tiles whose list overflows (many objects behind few pixels), objects that cover most of the screen, unbounded objects (planes: the sorted tables' prefix), row ranges that do not start on a tile row (no
    binning), Camera::render's untraced last row/column, interleaved bands, several views per launch, and a reflective
    world (first pass binned, secondary passes not)."""
    import torch
    rng = np.random.default_rng(77)
    u = lambda a, b: float(rng.uniform(a, b))
    w = rtc.World(rtc.light((-6.0, 9.0, -8.0)))
    for i in range(700):
        r = u(0.03, 0.3)
        w.add_shape(rtc.sphere(rtc.Matrix.identity().scaling(r, r, r).translation(u(-8, 8), u(0, 6), u(-2, 25)),
                               rtc.material(color=(u(0, 1), u(0, 1), u(0, 1)), specular=0.3, shininess=40.0,
                                            reflective=(0.3 if i % 9 == 0 else 0.0))))
    w.add_shape(rtc.sphere(rtc.Matrix.identity().scaling(30, 30, 30).translation(0, 0, 70), rtc.material(color=(0.2, 0.3, 0.9))))   # fills the view
    w.add_shape(rtc.sphere(rtc.Matrix.identity().scaling(4, 4, 4).translation(1, 2, 6), rtc.material(color=(0.9, 0.3, 0.2))))       # wide
    w.add_shape(rtc.plane(rtc.Matrix.identity(), rtc.material(specular=0.0, pattern=("checker", (0.3,) * 3, (0.7,) * 3, None))))
    view = rtc.Matrix.make_view_transform((0.0, 2.0, -8.0), (0.0, 1.0, 5.0), (0.0, 1.0, 0.0))
    ctx_bin, ctx_walk = rtc.Context(0), _ctx_env(rtc, RTC_BINNING=0, RTC_LIGHT_LISTS=0)
    dwb, dww = ctx_bin.upload(w), ctx_walk.upload(w)
    for (W, H) in ((640, 360), (72, 45)):       # 72x45: hundreds of objects behind every tile -> lists overflow, tiles walk
        cam = rtc.camera(W, H, 0.8, view)
        for mode in (rtc.MODE_RENDER_ASYNC, rtc.MODE_RENDER):
            a, sa = dwb.render(cam, mode, with_stats=True)
            b, sb = dww.render(cam, mode, with_stats=True)
            c, sc = dwb.render(cam, mode, flags=rtc.FLAG_NO_CULL, with_stats=True)
            assert np.array_equal(a, b) and sa == sb and np.array_equal(a, c) and sa == sc, (W, H, mode)
        # rows that do not start on a tile row (no binning), and one rank's bands (binning with a band stride)
        cam = rtc.camera(W, H, 0.8, view)
        full = dww.render(cam)
        t = torch.zeros((H - 13, W, 3), dtype=torch.float64, device="cuda:0")
        torch.cuda.synchronize()
        dwb.render_rows(cam, 13, H, t.data_ptr())
        ctx_bin.synchronize()
        assert np.array_equal(t.cpu().numpy(), full[13:])
        nb = -(-H // 8)
        per = -(-nb // 3) * 8
        t = torch.zeros((2 * per, W, 3), dtype=torch.float64, device="cuda:0")
        cam2 = rtc.camera(W, H, 0.75, rtc.Matrix.make_view_transform((1.0, 2.5, -7.0), (0.0, 1.0, 5.0), (0.0, 1.0, 0.0)))
        dwb.render_views([cam, cam2], 1, 3, t.data_ptr(), per)
        ctx_bin.synchronize()
        th, full2 = t.cpu().numpy(), dww.render(cam2)
        for k, band in enumerate(range(1, nb, 3)):
            rows = min(8, H - band * 8)
            assert np.array_equal(th[k * 8:k * 8 + rows], full[band * 8:band * 8 + rows])
            assert np.array_equal(th[per + k * 8:per + k * 8 + rows], full2[band * 8:band * 8 + rows])
    # anti-aliased renders are binned too (the tile cones span the pixel areas): sub-samples and the resample
    cam = rtc.camera(200, 120, 0.8, view, samples=3)
    for flags in (0, rtc.FLAG_AA_RESAMPLE):
        a, sa = dwb.render(cam, flags=flags, with_stats=True)
        b, sb = dww.render(cam, flags=flags, with_stats=True)
        assert np.array_equal(a, b) and sa == sb and sa["pixels_resample"] > 0, flags
    want, ost = O.render(w.array(), len(w), w.light, cam, mode=1, nthreads=8, want_stats=True, flags=rtc.FLAG_AA_RESAMPLE)
    assert sa == ost and np.max(np.abs(a - want)) <= TIGHT_TOL
    # against the oracle on sampled pixels of the larger frame
    cam = rtc.camera(640, 360, 0.8, view)
    a = dwb.render(cam)
    arr = w.array()
    for _ in range(300):
        x, y = int(rng.integers(0, 640)), int(rng.integers(0, 360))
        want = O.color_at(arr, len(w), w.light, rtc.ray_for_pixel(cam, x, y), 5)
        assert np.max(np.abs(a[y, x] - want)) <= TIGHT_TOL, (x, y)
    ctx_bin.close()
    ctx_walk.close()


def test_tile_lists_of_worlds_beyond_65536_objects(rtc):
    """Tile-list entries carry the upper half of the object's key above a 16-bit index (RTC_BIN_PACKED); Worlds with more
    than 65 536 objects fall back to plain indices and keys computed from the bounds. Both forms must give the walk's and
    brute force's canvas bit for bit (this one is the fallback; every other binned test runs the packed form). Most of the
    spheres sit behind the camera so that the tiles' lists stay below their capacity and are actually used."""
    rng = np.random.default_rng(5)
    u = lambda a, b: float(rng.uniform(a, b))
    w = rtc.World(rtc.light((-6.0, 9.0, -8.0)))
    mat = rtc.material(color=(0.4, 0.7, 0.5), specular=0.3, shininess=40.0)
    for i in range(70_000):
        front = i % 16 == 0
        r = u(0.03, 0.15)
        z = u(-2, 25) if front else u(-60, -12)
        w.add_shape(rtc.sphere(rtc.Matrix.identity().scaling(r, r, r).translation(u(-8, 8), u(0, 6), z),
                               mat if i % 3 else rtc.material(color=(u(0, 1), u(0, 1), u(0, 1)), specular=0.2)))
    w.add_shape(rtc.plane(rtc.Matrix.identity(), rtc.material(specular=0.0, pattern=("checker", (0.3,) * 3, (0.7,) * 3, None))))
    cam = rtc.camera(640, 360, 0.8, rtc.Matrix.make_view_transform((0.0, 2.0, -8.0), (0.0, 1.0, 5.0), (0.0, 1.0, 0.0)))
    ctx_bin, ctx_walk = rtc.Context(0), _ctx_env(rtc, RTC_BINNING=0, RTC_LIGHT_LISTS=0)
    dwb, dww = ctx_bin.upload(w), ctx_walk.upload(w)
    a, sa = dwb.render(cam, with_stats=True)
    b, sb = dww.render(cam, with_stats=True)
    c, sc = dwb.render(cam, flags=rtc.FLAG_NO_CULL, with_stats=True)
    assert np.array_equal(a, b) and sa == sb and np.array_equal(a, c) and sa == sc
    assert sa["rays_shadow"] > 100_000
    ctx_bin.close()
    ctx_walk.close()


def test_render_paths_stop_allocating_after_the_first_launch(rtc, scenes):
    """A frame sequence must not call hipMalloc / hipFree after its first launch (they wait for the device: 0.2-3 ms in the
    middle of a sequence). The case that did: a launch of FEWER views (a 5-frame warm-up) between launches of 8 left one of
    the two binning sets too small, and the next 8-view launch re-allocated it inside the timed region of
    `bench.py --steps 20 --warmup 5` (0.09-0.22 ms per frame instead of 0.07). Both sets are now made ready by the first
    binned launch, for RTC_MAX_VIEWS views. rtc_render's scratch canvas: allocated once."""
    import ctypes as C
    import torch
    lib = rtc.lib()
    lib.rtc_debug_render_allocs.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]

    def allocs(ctx):
        out = C.c_ulonglong(0)
        assert lib.rtc_debug_render_allocs(ctx._h, C.byref(out)) == 0
        return out.value

    for n in (100, 2000):                 # one-level world (binned in long launches: forced here), two-level world (always binned)
        w, cam = scenes.synthetic(n, 320, 184)
        ctx = _ctx_env(rtc, RTC_BIN_SMALL_PIXELS=0)
        dw = ctx.upload(w)
        buf = torch.zeros((8 * 184, 320, 3), dtype=torch.float64, device="cuda:0")
        torch.cuda.synchronize()
        dw.render_views([cam] * 4, 0, 1, buf.data_ptr(), 184)
        ctx.synchronize()
        first = allocs(ctx)
        assert first > 0
        for views in (8, 5, 8, 8, 1, 8, 4, 8):
            dw.render_views([cam] * views, 0, 1, buf.data_ptr(), 184)
        ctx.synchronize()
        assert allocs(ctx) == first, (n, first, allocs(ctx))
        a = dw.render(cam)
        mid = allocs(ctx)
        b = dw.render(cam)
        assert allocs(ctx) == mid and np.array_equal(a, b)
        ctx.close()


def test_one_level_worlds_binned_equal_the_walk(rtc, O, scenes):
    """Worlds of up to 256 objects (one-level cull) take the binned primary pass only in long launches (views x pixels >=
    RTC_BIN_SMALL_PIXELS: the bench's 8-view 1080p launches, 4096^2 frames); forced here on small canvases. Same lists, same
    consumer as for large worlds, but the other kernel instantiations (k_trace<3,...>): flat, reflective (LDS frame stack)
    and glass; whole frames only (row ranges and bands walk)."""
    import torch
    rng = np.random.default_rng(11)
    u = lambda a, b: float(rng.uniform(a, b))
    view = rtc.Matrix.make_view_transform((0.0, 2.0, -8.0), (0.0, 1.0, 5.0), (0.0, 1.0, 0.0))
    for kind in ("flat", "reflective", "glass"):
        w = rtc.World(rtc.light((-6.0, 9.0, -8.0)))
        for i in range(150):
            r = u(0.1, 0.5)
            mat = dict(color=(u(0, 1), u(0, 1), u(0, 1)), specular=0.3, shininess=40.0)
            if kind == "reflective" and i % 3 == 0:
                mat["reflective"] = 0.4
            if kind == "glass" and i % 4 == 0:
                mat.update(transparency=0.8, refractive_index=1.5, reflective=0.2)
            w.add_shape(rtc.sphere(rtc.Matrix.identity().scaling(r, r, r).translation(u(-7, 7), u(0.2, 4), u(-2, 20)), rtc.material(**mat)))
        w.add_shape(rtc.sphere(rtc.Matrix.identity().scaling(5, 5, 5).translation(2, 3, 14), rtc.material(color=(0.9, 0.3, 0.2))))   # covers many tiles
        w.add_shape(rtc.plane(rtc.Matrix.identity(), rtc.material(specular=0.0, reflective=(0.3 if kind == "reflective" else 0.0),
                                                              pattern=("checker", (0.3,) * 3, (0.7,) * 3, None))))
        ctx_bin, ctx_walk = _ctx_env(rtc, RTC_BIN_SMALL_PIXELS=0), _ctx_env(rtc, RTC_BINNING=0, RTC_LIGHT_LISTS=0)
        dwb, dww = ctx_bin.upload(w), ctx_walk.upload(w)
        for (W, H) in ((400, 232), (61, 37)):
            cam = rtc.camera(W, H, 0.8, view)
            for mode in (rtc.MODE_RENDER_ASYNC, rtc.MODE_RENDER):
                a, sa = dwb.render(cam, mode, with_stats=True)
                b, sb = dww.render(cam, mode, with_stats=True)
                c, sc = dwb.render(cam, mode, flags=rtc.FLAG_NO_CULL, with_stats=True)
                assert np.array_equal(a, b) and sa == sb and np.array_equal(a, c) and sa == sc, (kind, W, H, mode)
        # several views per launch (different cameras), and the anti-aliased branch
        cam = rtc.camera(400, 232, 0.8, view)
        cam2 = rtc.camera(400, 232, 0.7, rtc.Matrix.make_view_transform((1.5, 2.5, -7.0), (0.0, 1.0, 5.0), (0.0, 1.0, 0.0)))
        t = torch.zeros((3 * 232, 400, 3), dtype=torch.float64, device="cuda:0")
        torch.cuda.synchronize()
        dwb.render_views([cam, cam2, cam], 0, 1, t.data_ptr(), 232)
        ctx_bin.synchronize()
        th = t.cpu().numpy()
        assert np.array_equal(th[:232], dww.render(cam)) and np.array_equal(th[232:464], dww.render(cam2)) and np.array_equal(th[464:], th[:232])
        cam_aa = rtc.camera(160, 96, 0.8, view, samples=3)
        a, sa = dwb.render(cam_aa, flags=rtc.FLAG_AA_RESAMPLE, with_stats=True)
        b, sb = dww.render(cam_aa, flags=rtc.FLAG_AA_RESAMPLE, with_stats=True)
        assert np.array_equal(a, b) and sa == sb
        # against the oracle on sampled pixels
        cam = rtc.camera(400, 232, 0.8, view)
        a = dwb.render(cam)
        arr = w.array()
        for _ in range(120):
            x, y = int(rng.integers(0, 400)), int(rng.integers(0, 232))
            want = O.color_at(arr, len(w), w.light, rtc.ray_for_pixel(cam, x, y), 5)
            assert np.max(np.abs(a[y, x] - want)) <= TIGHT_TOL, (kind, x, y)
        ctx_bin.close()
        ctx_walk.close()
