"""GPU tests of the round-3 additions to the C-ABI: pipelined launches (rtc_context_set_pipeline), the host-side 8-bit
delivery (rtc_render_rgb8, rtc_group_render_host_rgb8, rtc_canvas_*_ppm_rgb8), rtc_context_last_launch_info, the
LDS-table brute force by flag, and the forced one-level cull beyond 256 objects."""
import importlib
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scenes(rtc):
    return importlib.import_module(rtc.__name__ + ".scenes")


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


@pytest.mark.parametrize("kind", ["flat100", "reflective", "glass", "large"])
def test_pipelined_launches_equal_in_order_launches(rtc, scenes, kind):
    """rtc_context_set_pipeline(2..4): consecutive launches on alternating streams of the context's own, each with its own
    tile lists in front of it — canvases, 8-bit frames and ray counts must be those of the in-order context bit for bit,
    for one camera per launch (distinct cameras, a ring of canvases), for bands, and with the lists forced on and off."""
    import torch
    W, H = 320, 203
    if kind == "flat100":
        w, _ = scenes.synthetic(100, W, H)
    elif kind == "reflective":
        w, _ = scenes.synthetic(60, W, H, reflective=True)
    elif kind == "glass":
        w, _ = scenes.glass_cluster(40, W, H)
    else:
        w, _ = scenes.synthetic(700, W, H, with_plane=True)
    cams = [rtc.camera(W, H, 0.7 + 0.03 * i, rtc.Matrix.make_view_transform((2.5 * math.sin(0.3 * i), 2.0, -8.0 + 0.5 * i), (0.0, 1.0, 5.0), (0.0, 1.0, 0.0)))
            for i in range(7)]
    ref_ctx = rtc.Context(0)
    dref = ref_ctx.upload(w)
    want, total = [], {}
    for c in cams:
        img, st = dref.render(c, rtc.MODE_RENDER_ASYNC, with_stats=True)
        want.append(img)
        for k, v in st.items():
            total[k] = total.get(k, 0) + v
    dref.close()
    ref_ctx.close()
    for depth, small in ((2, 0), (3, 10**12), (4, 0)):   # lists forced for every launch / never (one-level worlds)
        ctx = _ctx_env(rtc, RTC_BIN_SMALL_PIXELS_PIPELINED=small)
        dw = ctx.upload(w)
        ctx.set_pipeline(depth)
        ring = [torch.full((H, W, 3), -1.0, dtype=torch.float64, device="cuda:0") for _ in range(len(cams))]
        ring8 = [torch.full((H, W, 3), 7, dtype=torch.uint8, device="cuda:0") for _ in range(len(cams))]
        torch.cuda.synchronize()
        ctx.reset_stats()
        lanes = []
        for i, c in enumerate(cams):
            dw.render_rows(c, 0, H, ring[i].data_ptr(), rtc.MODE_RENDER_ASYNC, d_ptr8=ring8[i].data_ptr())
            info = ctx.last_launch_info()
            lanes.append(info["lane"])
            assert info["binned_primary_pass"] == (kind == "large" or small == 0), (kind, depth, info)
        assert lanes == [i % depth for i in range(len(cams))]
        assert ctx.stats() == total, (kind, depth)        # synchronises every lane
        for i in range(len(cams)):
            assert np.array_equal(ring[i].cpu().numpy(), want[i]), (kind, depth, i)
            assert np.array_equal(ring8[i].cpu().numpy(), rtc.color_scale255(want[i]).reshape(H, W, 3)), (kind, depth, i)
        # one rank's bands through the pipelined context (rank 1 of 3), and the fence: work on the caller's stream afterwards
        per = rtc.group_packed_rows(H, 3)
        tiles = [torch.zeros((per, W, 3), dtype=torch.float64, device="cuda:0") for _ in range(3)]
        torch.cuda.synchronize()
        for i in range(3):
            dw.render_bands(cams[i], 1, 3, tiles[i].data_ptr())
        ctx.fence()
        ctx.synchronize()
        for i in range(3):
            got = tiles[i].cpu().numpy()
            for k in range(rtc.group_bands_owned(H, 3, 1)):
                y0 = rtc.group_packed_row_to_image(1, 8 * k, 3)
                y1 = min(H, y0 + 8)
                assert np.array_equal(got[8 * k: 8 * k + (y1 - y0)], want[i][y0:y1]), (kind, depth, i, k)
        # back to in-order launches on the same context and World
        ctx.set_pipeline(1)
        one = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda:0")
        torch.cuda.synchronize()
        dw.render_rows(cams[3], 0, H, one.data_ptr(), rtc.MODE_RENDER_ASYNC)
        ctx.synchronize()
        assert np.array_equal(one.cpu().numpy(), want[3])
        assert np.array_equal(dw.render(cams[5]), want[5])   # rtc_render (host canvas) on a context that was pipelined
        dw.close()
        ctx.close()
    with pytest.raises(rtc.RtcError):
        c = rtc.Context(0)
        try:
            c.set_pipeline(5)
        finally:
            c.close()


def test_rgb8_delivery_and_ppm_equal_the_f64_path_and_the_oracle(rtc, gpu, scenes, O, tmp_path):
    """rtc_render_rgb8 (only the device's 8-bit frame crosses PCIe; no f64 canvas is written) == Color::scale of the f64 render,
    and the PPM written from it == rtc_canvas_format_ppm of the f64 render == the PPM of the oracle's canvas
    (canvas.rs:86-109, color.rs:100-114), for Camera::render and render_async, AA on, odd sizes (the unaligned store path)."""
    for (name, W, H, samples) in (("mixed", 160, 120, 1), ("mixed", 50, 37, 1), ("test8", 96, 72, 4), ("default", 33, 9, 1)):
        if name == "mixed":
            w, cam = scenes.mixed(W, H)
        elif name == "test8":
            w, cam = scenes.test8(W, H, samples=samples)
        else:
            w, cam = scenes.default_scene(W, H)
        dw = gpu.upload(w)
        for mode in (rtc.MODE_RENDER, rtc.MODE_RENDER_ASYNC):
            f64, st64 = dw.render(cam, mode, with_stats=True)
            u8, st8 = dw.render_rgb8(cam, mode, with_stats=True)
            assert st8 == st64
            assert u8.dtype == np.uint8 and u8.shape == (H, W, 3)
            assert np.array_equal(u8, rtc.color_scale255(f64).reshape(H, W, 3)), (name, W, H, mode)
            ppm8 = rtc.format_ppm_rgb8(u8)
            assert ppm8 == rtc.format_ppm(f64)
            want = O.render(w.array(), len(w), w.light, cam, mode=mode)
            assert np.max(np.abs(want - f64)) <= 1e-12
            # the oracle's PPM: identical unless a component sits within 1e-12 of a quantisation step (pow is <= 4 ulp)
            oppm = O.format_ppm(want) if hasattr(O, "format_ppm") else rtc.format_ppm(want)
            if oppm != ppm8:
                a = np.frombuffer(rtc.color_scale255(want).tobytes(), dtype=np.uint8).reshape(H, W, 3).astype(int)
                d = np.abs(a - u8.astype(int))
                assert d.max() <= 1 and (d != 0).sum() <= 3, (name, W, H, mode, int(d.max()), int((d != 0).sum()))
            p = tmp_path / f"{name}_{W}_{mode}.ppm"
            rtc.write_ppm_rgb8(p, u8)
            assert p.read_bytes() == ppm8
        # a page-locked 8-bit frame, reused
        pin = rtc.host_canvas_rgb8(H, W)
        dw.render_rgb8(cam, out=pin)
        assert np.array_equal(pin, rtc.color_scale255(dw.render(cam)).reshape(H, W, 3))
        with pytest.raises(ValueError):
            dw.render_rgb8(cam, out=np.zeros((H, W, 4), dtype=np.uint8))
        dw.close()


def test_rows_bands_views_with_only_the_8bit_output(rtc, gpu, scenes):
    """d_rgb = NULL, d_rgb8 given: rtc_render_rows / _bands / _views write the 8-bit rows only; both NULL is an error."""
    import torch
    W, H = 160, 93
    w, cam = scenes.synthetic(25, W, H, reflective=True)
    dw = gpu.upload(w)
    full8 = rtc.color_scale255(dw.render(cam)).reshape(H, W, 3)
    q = torch.full((H, W, 3), 9, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    dw.render_rows(cam, 0, H, None, d_ptr8=q.data_ptr())
    gpu.synchronize()
    assert np.array_equal(q.cpu().numpy(), full8)
    per = rtc.group_packed_rows(H, 2)
    t = torch.full((2, 1, per, W, 3), 9, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    for r in range(2):
        dw.render_bands(cam, r, 2, None, d_ptr8=t[r].data_ptr())
    gpu.synchronize()
    assert np.array_equal(rtc.group_undeal_host(t.cpu().numpy(), 2, 1, H)[0], full8)
    v = torch.full((2 * 96, W, 3), 9, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    dw.render_views([cam, cam], 0, 1, None, 96, d_ptr8=v.data_ptr())
    gpu.synchronize()
    vh = v.cpu().numpy()
    assert np.array_equal(vh[:H], full8) and np.array_equal(vh[96:96 + H], full8)
    with pytest.raises(rtc.RtcError):
        dw.render_rows(cam, 0, H, None)
    dw.close()


def test_group_host_rgb8(rtc, scenes):
    """rtc_group_render_host_rgb8: every member DMAs its 8-bit bands straight to their rows of ONE host frame
    (members rehearsed on one device with peer copies, as in test_gpu_group.py)."""
    W, H = 200, 117
    w, cam = scenes.synthetic(30, W, H)
    c = rtc.Context(0)
    ref = c.upload(w).render(cam)
    c.close()
    want8 = rtc.color_scale255(ref).reshape(H, W, 3)
    for devices in ([0], [0, 0], [0, 0, 0]):
        g = rtc.Group(devices=devices, exchange=rtc.EXCHANGE_P2P)
        gw = g.upload(w)
        for out in (np.zeros((H, W, 3), dtype=np.uint8), rtc.host_canvas_rgb8(H, W)):
            got, st = gw.render_host_rgb8(cam, out, with_stats=True)
            assert np.array_equal(got, want8), len(devices)
            assert st["rays_primary"] == W * H
        f64 = np.zeros((H, W, 3))
        assert np.array_equal(gw.render_host(cam, f64), ref)     # the f64 host canvas after the 8-bit one: buffers are separate
        assert [x.device for x in g.contexts] == devices
        gw.close()
        g.close()


def test_launch_info_and_the_lds_table_flag(rtc, gpu, scenes):
    """rtc_context_last_launch_info names the object source; RTC_FLAG_NO_CULL | RTC_FLAG_LDS_TABLE selects the LDS-staged
    brute force (BASELINE.json north_star's literal kernel) for any World size — same pixels, same ray counts."""
    import torch
    for n, want_default, want_lds in ((100, 3, 1), (700, 4, 1), (1500, 4, 2)):
        W, H = 96, 54
        w, cam = scenes.synthetic(n, W, H)
        dw = gpu.upload(w)
        a, sa = dw.render(cam, with_stats=True)
        assert gpu.last_launch_info()["source"] == want_default
        b, sb = dw.render(cam, flags=rtc.FLAG_NO_CULL | rtc.FLAG_LDS_TABLE, with_stats=True)
        info = gpu.last_launch_info()
        assert info["source"] == want_lds and info["dynamic_lds_bytes"] > 0 and not info["binned_primary_pass"], info
        c, sc = dw.render(cam, flags=rtc.FLAG_NO_CULL, with_stats=True)
        assert gpu.last_launch_info()["source"] in (0, 1, 2)
        assert np.array_equal(a, b) and np.array_equal(a, c) and sa == sb == sc, n
        dw.close()


def test_forced_one_level_cull_beyond_256_objects(rtc, scenes):
    """RTC_SRC=3 (the one-level cull forced for an A/B run) on a World of 700 objects with light lists: the listed shadow
    branch keeps a 256-bit 'done' set and must not be taken (objects j and j + 256 would share a bit); canvases equal the
    default context's (two-level cull) and the walk's (RTC_LIGHT_LISTS=0), bit for bit."""
    W, H = 240, 135
    w, cam = scenes.synthetic(700, W, H)
    c0 = rtc.Context(0)
    want, st0 = c0.upload(w).render(cam, with_stats=True)
    c0.close()
    for env in (dict(RTC_SRC=3), dict(RTC_SRC=3, RTC_LIGHT_LISTS=0), dict(RTC_SRC=3, RTC_BINNING=0)):
        c = _ctx_env(rtc, **env)
        got, st = c.upload(w).render(cam, with_stats=True)
        assert c.last_launch_info()["source"] == 3
        assert np.array_equal(got, want) and st == st0, env
        c.close()


def test_shared_divisor_normalize_is_bit_identical(gpu):
    """Vector::normalize (vec.rs:65-76) with the three divisions sharing the divisor-only part of hipcc's f64 division
    expansion (rtc_kernels.hip vnormalize_shared, behind RTC_SHARED_NORMALIZE; an experiment) must equal three IEEE divisions
    bit for bit — and both must equal the host: ordinary directions, zeros and signed zeros, tiny and huge components (the
    guard's fall-back), denormals, infinities."""
    rng = np.random.default_rng(11)
    n = 2_000_000
    v = rng.normal(size=(n, 3))
    v[:200000] *= np.exp(rng.uniform(-700, 700, (200000, 1)))            # whole vectors far outside the guard's range
    v[200000:400000] *= np.exp(rng.uniform(-60, 60, (200000, 3)))        # components of very different size
    v[400000:500000, rng.integers(0, 3)] = 0.0
    v[500000:520000] = np.where(rng.random((20000, 3)) < 0.5, 0.0, -0.0) + rng.normal(size=(20000, 3)) * (rng.random((20000, 3)) < 0.4)
    v[520000:540000] *= 1e-310                                             # denormals
    v[540000:540010] = [[np.inf, 1, 1], [1, -np.inf, 0], [0, 0, 0], [-0.0, 0.0, -0.0], [np.nan, 1, 2], [1e308, 1e308, 1e308],
                        [5e-324, 0, 0], [1, 5e-324, 0], [2.0 ** -500, 2.0 ** -501, 1], [2.0 ** 400, 2.0 ** 400, 0]]
    flat = np.ascontiguousarray(v.reshape(-1))
    shared = gpu.device_arith(5, flat)
    plain = gpu.device_arith(6, flat)
    assert np.array_equal(shared.view(np.uint64), plain.view(np.uint64))
    with np.errstate(all="ignore"):
        mag = np.sqrt(v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1] + v[:, 2] * v[:, 2])
        host = v / mag[:, None]
    ok = (plain.reshape(-1, 3).view(np.uint64) == host.view(np.uint64)) | (np.isnan(plain.reshape(-1, 3)) & np.isnan(host))
    assert ok.all()


def test_tile_rows_proven_black_are_skipped_exactly(rtc, scenes, O):
    """k_bin_tiles proves tile rows black (empty candidate list + the tile's cone clear of every plane) and k_trace then
    generates no ray for them: canvases, 8-bit frames and ray counts must equal the same render with the proof switched off
    (RTC_SKY_ROWS=0), without binning (RTC_BINNING=0) and the oracle — floor scenes (sky above the horizon), no plane at all,
    a wall behind the scene (no sky left: nothing may be skipped), a tilted floor, the camera under the floor, the camera ON
    the plane (no proof possible), Camera::render's exclusive edge, bands, several views, a reflective world."""
    import torch
    M = rtc.Matrix
    W, H = 320, 203

    def world(kind):
        w, cam = scenes.synthetic(60, W, H, with_plane=(kind != "noplane"), reflective=(kind == "reflective"))
        if kind == "wall":
            w.add_shape(rtc.plane(M.identity().rotation_x(math.pi / 2.0).translation(0.0, 0.0, 40.0), rtc.material(color=(0.4, 0.5, 0.7))))
        if kind == "tilted":
            w.shapes[-1] = rtc.plane(M.identity().rotation_z(0.2).rotation_x(-0.1), rtc.material(specular=0.0, pattern=("checker", (0.3,) * 3, (0.7,) * 3, None)))
            w.shapes[-1].world_id = len(w.shapes)
        if kind == "under":
            cam = rtc.camera(W, H, 0.9, M.make_view_transform((0.0, -3.0, -8.0), (0.0, 2.0, 5.0), (0.0, 1.0, 0.0)))
        if kind == "onplane":
            cam = rtc.camera(W, H, 0.9, M.make_view_transform((0.0, 0.0, -8.0), (0.0, 1.0, 5.0), (0.0, 1.0, 0.0)))
        if kind == "lookup":
            cam = rtc.camera(W, H, 1.2, M.make_view_transform((0.0, 1.0, -8.0), (0.0, 9.0, 5.0), (0.0, 1.0, 0.0)))
        return w, cam

    expect_sky = {"floor": True, "noplane": True, "wall": False, "tilted": None, "under": None, "onplane": False, "lookup": True, "reflective": True}
    for kind, sky in expect_sky.items():
        w, cam = world(kind)
        base = _ctx_env(rtc, RTC_BINNING=0)
        want, st_want = base.upload(w).render(cam, with_stats=True)
        base.close()
        for env in (dict(RTC_BIN_SMALL_PIXELS=0), dict(RTC_BIN_SMALL_PIXELS=0, RTC_SKY_ROWS=0)):
            ctx = _ctx_env(rtc, **env)
            dw = ctx.upload(w)
            for mode in (rtc.MODE_RENDER_ASYNC, rtc.MODE_RENDER):
                ref_ctx = _ctx_env(rtc, RTC_BINNING=0)
                ref, st_ref = ref_ctx.upload(w).render(cam, mode, with_stats=True)
                ref_ctx.close()
                f = torch.full((H, W, 3), -1.0, dtype=torch.float64, device="cuda:0")
                q = torch.full((H, W, 3), 9, dtype=torch.uint8, device="cuda:0")
                torch.cuda.synchronize()
                ctx.reset_stats()
                dw.render_rows(cam, 0, H, f.data_ptr(), mode, d_ptr8=q.data_ptr())
                st = ctx.stats(extended=True)
                proven = st.pop("rays_primary_proven_miss")
                assert ctx.last_launch_info()["binned_primary_pass"]
                assert np.array_equal(f.cpu().numpy(), ref) and st == st_ref, (kind, env, mode)
                assert np.array_equal(q.cpu().numpy(), rtc.color_scale255(ref).reshape(H, W, 3)), (kind, env, mode)
                if "RTC_SKY_ROWS" in env:
                    assert proven == 0
                elif sky is True:
                    assert proven > 0 and proven % (W - (1 if mode == rtc.MODE_RENDER else 0)) == 0, (kind, proven)   # whole tile rows
                    assert proven <= st["rays_primary"] - np.count_nonzero(ref.reshape(-1, 3).any(axis=1))      # never more than the black pixels
                elif sky is False:
                    assert proven == 0, (kind, proven)
            # one rank's bands (rank 2 of 3) and three views in one launch
            per = rtc.group_packed_rows(H, 3)
            t = torch.zeros((per, W, 3), dtype=torch.float64, device="cuda:0")
            torch.cuda.synchronize()
            dw.render_bands(cam, 2, 3, t.data_ptr())
            ctx.synchronize()
            got = t.cpu().numpy()
            for k in range(rtc.group_bands_owned(H, 3, 2)):
                y0 = rtc.group_packed_row_to_image(2, 8 * k, 3)
                assert np.array_equal(got[8 * k: 8 * k + min(8, H - y0)], want[y0:y0 + 8]), (kind, env, k)
            cams = [cam, rtc.camera(W, H, 0.8, M.make_view_transform((1.0, 3.0, -9.0), (0.0, 0.5, 5.0), (0.0, 1.0, 0.0))), cam]
            HP = -(-H // 8) * 8
            v = torch.zeros((3 * HP, W, 3), dtype=torch.float64, device="cuda:0")
            torch.cuda.synchronize()
            dw.render_views(cams, 0, 1, v.data_ptr(), HP)
            ctx.synchronize()
            vh = v.cpu().numpy()
            assert np.array_equal(vh[:H], want) and np.array_equal(vh[2 * HP:2 * HP + H], want), (kind, env)
            other_ctx = _ctx_env(rtc, RTC_BINNING=0)
            assert np.array_equal(vh[HP:HP + H], other_ctx.upload(w).render(cams[1])), (kind, env)
            other_ctx.close()
            dw.close()
            ctx.close()
        o_img, o_st = O.render(w.array(), len(w), w.light, cam, mode=1, nthreads=8, want_stats=True)
        assert float(np.max(np.abs(o_img - want))) <= 1e-12 and o_st == st_want, kind


@pytest.mark.parametrize("which", ["c2_test7", "north_star"])
def test_whole_frame_1080p_against_the_oracle(rtc, scenes, O, which):
    """Configs C2 (the reference's `test7` scene, main.rs:204-251) and the north-star world at their FULL 1920x1080 size, every
    pixel against the CPU oracle (not a sample) with exact ray counts — as bench.py times them: one camera per launch on a
    pipelined context (depth 3, ring of canvases, binned primary pass, light lists, tile rows proven black), and in order."""
    import torch
    W, H = 1920, 1080
    w, cam = scenes.test7(W, H) if which == "c2_test7" else scenes.synthetic(100, W, H)
    want, ost = O.render(w.array(), len(w), w.light, cam, mode=1, nthreads=16, want_stats=True)
    ctx = rtc.Context(0)
    dw = ctx.upload(w)
    got, st = dw.render(cam, rtc.MODE_RENDER_ASYNC, with_stats=True)
    assert st == ost
    assert float(np.max(np.abs(got - want))) <= 1e-12
    ctx.set_pipeline(3)
    ring = [torch.full((H, W, 3), -1.0, dtype=torch.float64, device="cuda:0") for _ in range(4)]
    torch.cuda.synchronize()
    ctx.reset_stats()
    for i in range(8):
        dw.render_rows(cam, 0, H, ring[i % 4].data_ptr())
    ctx.synchronize()
    pst = ctx.stats()
    assert all(pst[k] == 8 * st[k] for k in st), (pst, st)
    info = ctx.last_launch_info()
    assert info["binned_primary_pass"] and info["light_lists"] == (which == "north_star")   # test7's four objects get no lists
    for r in ring:
        assert np.array_equal(r.cpu().numpy(), got)
    dw.close()
    ctx.close()


def test_lua_program_render_orbit_animation(rtc, scenes, O):
    """rtc_lua_program_render (lua.rs's render_lua on the GPU): the orbit script's 12 AddFrame jobs and its Render job, one
    launch each on the context's lanes with the frame copies behind them, frames delivered in job order — equal to the
    same jobs rendered one by one through rtc_render_rgb8 and to Color::scale of the oracle's canvases; ray counts add up;
    the context is in order again afterwards; a callback can stop the run; a second world in the script is a new upload."""
    from pathlib import Path
    data = Path(rtc.__file__).resolve().parent / "data"
    text = "FRAMES = 5 BALLS = 9 WIDTH, HEIGHT = 200, 136\n" + (data / "orbit_animation.lua").read_text()
    text += "\ntable.remove(world.shapes, 2)\nRender(world, camera, 'fewer.ppm')\n"          # a different world at the end: no cube
    prog = rtc.LuaProgram(text=text, base_dir=data)
    jobs = prog.jobs
    assert len(jobs) == 7 and [j.same_world_as_previous for j in jobs] == [False, True, True, True, True, True, False]
    ctx = rtc.Context(0)
    frames, st = prog.render(ctx, with_stats=True)
    assert len(frames) == 7 and all(f.shape == (136, 200, 3) and f.dtype == np.uint8 for f in frames)
    assert ctx.last_launch_info()["lane"] in (0, 1, 2)
    total = {}
    for j, f in zip(jobs, frames):
        dw = ctx.upload(j.world)
        one, s1 = dw.render_rgb8(j.camera, with_stats=True)
        dw.close()
        assert np.array_equal(f, one), j.index
        for k, v in s1.items():
            total[k] = total.get(k, 0) + v
        if j.index in (0, 3, 6):
            want = O.render(j.world.array(), len(j.world), j.world.light, j.camera, mode=1, nthreads=8)
            q = rtc.color_scale255(want).reshape(136, 200, 3)
            diff = np.abs(f.astype(np.int16) - q.astype(np.int16))
            assert diff.max() <= 1 and np.count_nonzero(diff) <= 4, (j.index, int(diff.max()), int(np.count_nonzero(diff)))   # 1e-12 on a quantisation edge
    assert all(st[k] == total[k] for k in total if k in st), (st, total)
    assert not np.array_equal(frames[0], frames[1]) and not np.array_equal(frames[5], frames[6])
    # in order again: a plain render after it equals the frame
    dw = ctx.upload(jobs[2].world)
    assert np.array_equal(dw.render_rgb8(jobs[2].camera), frames[2])
    dw.close()
    # a pipelined context keeps its depth
    ctx.set_pipeline(2)
    seen = []
    prog.render(ctx, on_frame=lambda i, frame, outfile, kind: seen.append((i, outfile, kind, frame.copy())) or i == 2)
    assert [s[0] for s in seen] == [0, 1, 2] and seen[0][1:3] == ("orbit.gif", "AddFrame")
    assert all(np.array_equal(s[3], frames[s[0]]) for s in seen)
    ctx.set_pipeline(1)
    with pytest.raises(ZeroDivisionError):
        prog.render(ctx, on_frame=lambda *a: 1 // 0)
    # the files: AddFrame frames numbered under the animation's name, Render by extension (.ppm: the P3 writer; other: PNG)
    from test_gpu_facade import decode_png
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        paths = prog.render_to_files(ctx, d)
        assert [p.name for p in paths] == [f"orbit.gif.{k:04d}.png" for k in range(5)] + ["orbit_top.ppm", "fewer.ppm"]
        assert np.array_equal(decode_png(paths[3].read_bytes()), frames[3])
        assert paths[5].read_bytes() == rtc.format_ppm_rgb8(frames[5])
    ctx.close()


def test_guided_chunks_render_the_same_frames(rtc, scenes):
    """Guided chunks (RenderParams::chunk_wgs): a launch's first workgroups render eight, four, three, two tiles each and its last
    ones one — only the tile -> workgroup mapping changes, so canvases, 8-bit frames and ray counts must equal the one-tile-per-
    workgroup launch bit for bit. RTC_TILES_SLOTS makes small launches take every chunk level (the real threshold is three
    rounds of workgroups): flat, reflective, glass, anti-aliased, two-level worlds; rows, one rank's bands, several views;
    Camera::render's exclusive edge; a frame whose tile count divides by none of the chunk sizes."""
    import torch
    M = rtc.Matrix
    cases = []
    for kind in ("flat", "reflective", "glass", "large", "aa"):
        W, H = (331, 203) if kind != "aa" else (160, 96)
        if kind == "glass":
            w, cam = scenes.glass_cluster(30, W, H)
        else:
            w, cam = scenes.synthetic(700 if kind == "large" else 50, W, H, reflective=(kind == "reflective"))
        if kind == "aa":
            cam.samples = 4
        cases.append((kind, w, cam))
    for kind, w, cam in cases:
        W, H = cam.hsize, cam.vsize
        ref_ctx = _ctx_env(rtc, RTC_TILES_GUIDED=0)
        dref = ref_ctx.upload(w)
        want = {m: dref.render(cam, m, with_stats=True) for m in (rtc.MODE_RENDER_ASYNC, rtc.MODE_RENDER)}
        assert ref_ctx.last_launch_info()["multi_tile_workgroups"] == 0
        for slots, kmax in ((5, 8), (16, 4), (3, 2)):
            ctx = _ctx_env(rtc, RTC_TILES_SLOTS=slots, RTC_TILES_KMAX=kmax, RTC_BIN_SMALL_PIXELS=0)
            dw = ctx.upload(w)
            for mode, (img, st) in want.items():
                f = torch.full((H, W, 3), -1.0, dtype=torch.float64, device="cuda:0")
                q = torch.full((H, W, 3), 9, dtype=torch.uint8, device="cuda:0")
                torch.cuda.synchronize()
                ctx.reset_stats()
                dw.render_rows(cam, 0, H, f.data_ptr(), mode, d_ptr8=q.data_ptr())
                got_st = ctx.stats(extended=True)
                assert ctx.last_launch_info()["multi_tile_workgroups"] > 0, (kind, slots, kmax)
                assert np.array_equal(f.cpu().numpy(), img) and all(got_st[k] == st[k] for k in st if k in got_st), (kind, slots, kmax, mode, got_st, st)
                assert np.array_equal(q.cpu().numpy(), rtc.color_scale255(img).reshape(H, W, 3)), (kind, slots, kmax, mode)
            # one rank's bands (rank 1 of 3) and three views in one launch
            per = rtc.group_packed_rows(H, 3)
            t = torch.zeros((per, W, 3), dtype=torch.float64, device="cuda:0")
            torch.cuda.synchronize()
            dw.render_bands(cam, 1, 3, t.data_ptr())
            ctx.synchronize()
            got = t.cpu().numpy()
            full = want[rtc.MODE_RENDER_ASYNC][0]
            for k in range(rtc.group_bands_owned(H, 3, 1)):
                y0 = rtc.group_packed_row_to_image(1, 8 * k, 3)
                assert np.array_equal(got[8 * k: 8 * k + min(8, H - y0)], full[y0:y0 + 8]), (kind, slots, k)
            HP = -(-H // 8) * 8
            v = torch.zeros((3 * HP, W, 3), dtype=torch.float64, device="cuda:0")
            torch.cuda.synchronize()
            dw.render_views([cam, cam, cam], 0, 1, v.data_ptr(), HP)
            ctx.synchronize()
            vh = v.cpu().numpy()
            for k in range(3):
                assert np.array_equal(vh[k * HP: k * HP + H], full), (kind, slots, "view", k)
            dw.close()
            ctx.close()
        dref.close()
        ref_ctx.close()
