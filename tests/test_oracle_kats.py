"""Pin the CPU oracle against the reference's own known-answer tests.

Every test here is a transcription (as data) of a #[test] in /root/reference/ch1/src/*.rs —
the name and file:line are in the docstring. Tolerance is the reference's own
floats_equal: |a-b| < 1e-4 (lib.rs:13-15); `==` where the reference uses assert_eq!.
No GPU is used; this is what makes the oracle trustworthy before it checks the HIP path.
"""
import ctypes as C
import math

import numpy as np
import pytest

PI = math.pi
SQ2 = math.sqrt(2.0) / 2.0


def feq(a, b):  # floats_equal lib.rs:13-15
    return abs(a - b) < 0.0001


def veq(a, b):
    return all(feq(x, y) for x, y in zip(a, b))


def isect(O, s, ray):
    ts = (C.c_double * 2)()
    n = O.lib().orc_shape_intersect(C.byref(s), O.Ray6(*ray), ts)
    return [ts[i] for i in range(n)]


def normal_at(O, s, p):
    out = O.Vec3()
    O.lib().orc_normal_at(C.byref(s), O.Vec3(*p), out)
    return list(out)


def world_intersect(O, arr, n, ray):
    ts = (C.c_double * (2 * n))()
    idx = (C.c_int32 * (2 * n))()
    k = O.lib().orc_world_intersect(arr, n, O.Ray6(*ray), ts, idx)
    return [ts[i] for i in range(k)], [idx[i] for i in range(k)]


def comps(O, arr, ray, ts, idxs, pos):
    """Intersection::compute_vectors(ray, &list) for list entry `pos`."""
    k = len(ts)
    T = (C.c_double * k)(*ts)
    I = (C.c_int32 * k)(*idxs)
    h = O.RtcHit()
    rc = O.lib().orc_compute_vectors(arr, O.Ray6(*ray), T, I, k, pos, C.byref(h))
    assert rc == 0
    return h


def shade_hit(O, arr, n, lgt, h, remaining):
    out = O.Vec3()
    O.lib().orc_shade_hit(arr, n, C.byref(lgt), C.byref(h), remaining, out)
    return list(out)


# ---------------------------------------------------------------- vec.rs / transform.rs
def test_magnitude_dot_cross_reflect(O):
    """vec.rs:326-413 test_magnitude, test_vector_reflect1/2 through the oracle's view/normal paths."""
    # magnitude(2,5,4) = 6.708204 is exercised through normalize in view_transform; check reflect:
    h = O.RtcHit()
    s = O.shape(1)  # plane, normal (0,1,0)
    arr = O.world([s])
    hh = comps(O, arr, (0, 1, 0, 1, -1, 0), [1.0], [0], 0)  # reflect (1,-1,0) about (0,1,0)
    assert veq(list(hh.reflectv), (1, 1, 0))


def test_transform_point1_translate_ray(O):
    """vec.rs:353-391 test_transform_point1, test_translate_ray1/2 (exact)."""
    out = O.Vec3()
    O.lib().orc_transform_point(O.chain(("scaling", 2, 3, 4)), O.Vec3(-4, 6, 8), out)
    assert list(out) == [-8.0, 18.0, 32.0]
    O.lib().orc_transform_point(O.chain(("translation", 3, 4, 5)), O.Vec3(1, 2, 3), out)
    assert list(out) == [4.0, 6.0, 8.0]
    O.lib().orc_transform_vector(O.chain(("translation", 3, 4, 5)), O.Vec3(0, 1, 0), out)
    assert list(out) == [0.0, 1.0, 0.0]
    O.lib().orc_transform_point(O.chain(("scaling", 2, 3, 4)), O.Vec3(1, 2, 3), out)
    assert list(out) == [2.0, 6.0, 12.0]
    O.lib().orc_transform_vector(O.chain(("scaling", 2, 3, 4)), O.Vec3(0, 1, 0), out)
    assert list(out) == [0.0, 3.0, 0.0]


def test_array_mul_transpose_determinant(O):
    """transform.rs:269-336 test_array_mul, test_array_transpose, test_array_determinant1."""
    a = O.mat([[1, 2, 3, 4], [5, 6, 7, 8], [9, 8, 7, 6], [5, 4, 3, 2]])
    b = O.mat([[-2, 1, 2, 3], [3, 2, 1, -1], [4, 3, 6, 5], [1, 2, 7, 8]])
    out = O.Mat16()
    O.lib().orc_matrix_multiply(a, b, out)
    assert list(out) == [20, 22, 50, 48, 44, 54, 114, 108, 40, 58, 110, 102, 16, 26, 46, 42]
    # 3x3 determinant -196 embedded in a 4x4 with a unit last row/col
    m = O.mat([[1, 2, 6, 0], [-5, 8, -4, 0], [2, 6, 4, 0], [0, 0, 0, 1]])
    assert feq(O.lib().orc_matrix_determinant(m), -196.0)
    t = O.Mat16()
    O.lib().orc_matrix_transpose(O.mat(np.arange(16.0)), t)
    assert list(t) == list(np.arange(16.0).reshape(4, 4).T.reshape(16))


def test_array_invert(O):
    """transform.rs:343-362: 5-decimal strings of the inverse."""
    m = O.mat([[-5, 2, 6, -8], [1, -5, 1, 8], [7, 7, -6, -7], [1, -3, 7, 4]])
    inv = O.inverse(m)
    want = ["0.21805", "0.45113", "0.24060", "-0.04511", "-0.80827", "-1.45677", "-0.44361", "0.52068",
            "-0.07895", "-0.22368", "-0.05263", "0.19737", "-0.52256", "-0.81391", "-0.30075", "0.30639"]
    assert [f"{v:.5f}" for v in inv] == want


def test_singular_matrix_is_rejected(O):
    """transform.rs:35-38,175-177: |det| <= 1e-8 -> inverse panics."""
    with pytest.raises(ValueError):
        O.inverse(O.chain(("scaling", 0.001, 0.001, 0.001)))  # det = 1e-9
    O.inverse(O.chain(("scaling", 0.003, 0.003, 0.003)))      # det = 2.7e-8 is fine


def test_transform(O):
    """transform.rs:365-376: identity.rotation_x(pi/2).scaling(5,5,5).translation(10,5,7) * (1,0,1) = (15,0,7)."""
    m = O.chain(("rotation_x", PI / 2), ("scaling", 5, 5, 5), ("translation", 10, 5, 7))
    out = O.Vec3()
    O.lib().orc_transform_point(m, O.Vec3(1, 0, 1), out)
    assert veq(list(out), (15, 0, 7))


def test_view_transform1_2_3(O):
    """transform.rs:380-404 (exact matrix equality)."""
    assert list(O.view_transform((0, 0, 0), (0, 0, -1), (0, 1, 0))) == list(O.mat())
    assert list(O.view_transform((0, 0, 0), (0, 0, 1), (0, 1, 0))) == list(O.chain(("scaling", -1, 1, -1)))
    assert list(O.view_transform((0, 0, 8), (0, 0, 0), (0, 1, 0))) == list(O.chain(("translation", 0, 0, -8)))


def test_view_transform4(O):
    """transform.rs:406-419."""
    m = O.view_transform((1, 3, 2), (4, -2, 8), (1, 1, 0))
    want = ["-0.50709", "0.50709", "0.67612", "-2.36643", "0.76772", "0.60609", "0.12122", "-2.82843",
            "-0.35857", "0.59761", "-0.71714", "0.00000", "0.00000", "0.00000", "0.00000", "1.00000"]
    got = [f"{v:.5f}".replace("-0.00000", "0.00000") for v in m]
    assert got == want


# ---------------------------------------------------------------- camera.rs
def test_pixel_size1_2(O):
    """camera.rs:174-185."""
    assert feq(O.camera(200, 125, PI / 2).pixel_size, 0.01)
    assert feq(O.camera(125, 200, PI / 2).pixel_size, 0.01)


def ray_for_pixel(O, cam, x, y):
    r = O.Ray6()
    O.lib().orc_camera_ray_for_pixel(C.byref(cam), x, 0.5, y, 0.5, r)
    return list(r)


def test_camera1_2_3(O):
    """camera.rs:187-214."""
    cam = O.camera(201, 101, PI / 2)
    r = ray_for_pixel(O, cam, 100, 50)
    assert veq(r[:3], (0, 0, 0)) and veq(r[3:], (0, 0, -1))
    r = ray_for_pixel(O, cam, 0, 0)
    assert veq(r[:3], (0, 0, 0)) and veq(r[3:], (0.66519, 0.33259, -0.66851))
    cam = O.camera(201, 101, PI / 2, O.chain(("translation", 0, -2, 5), ("rotation_y", PI / 4)))
    r = ray_for_pixel(O, cam, 100, 50)
    assert veq(r[:3], (0, 2, -5)) and veq(r[3:], (SQ2, 0, -SQ2))


def test_render1(O):
    """camera.rs:216-231: default world 11x11, Camera::render, pixel (5,5)."""
    arr = O.world(O.default_world())
    cam = O.camera(11, 11, PI / 2, O.view_transform((0, 0, -5), (0, 0, 0), (0, 1, 0)))
    img = O.render(arr, 2, O.light(), cam, mode=0)
    assert veq(img[5, 5], (0.38066, 0.47583, 0.2855))


def test_render_off_by_one_vs_render_async(O):
    """camera.rs:120-121 vs :149: render leaves the last row/column black; render_async does not;
    elsewhere the two agree exactly (unpinned by a reference test)."""
    arr = O.world(O.default_world())
    cam = O.camera(11, 11, PI / 2, O.view_transform((0, 0, -5), (0, 0, 0), (0, 1, 0)))
    a = O.render(arr, 2, O.light(), cam, mode=0)
    b = O.render(arr, 2, O.light(), cam, mode=1)
    assert np.array_equal(a[:10, :10], b[:10, :10])
    assert not a[10].any() and not a[:, 10].any()


# ---------------------------------------------------------------- shape.rs: spheres
def test_intersect1_to_5(O):
    """shape.rs:807-865."""
    s = O.shape(0)
    assert veq(isect(O, s, (0, 0, -5, 0, 0, 1)), (4, 6))
    assert veq(isect(O, s, (0, 1, -5, 0, 0, 1)), (5, 5))
    assert isect(O, s, (0, 2, -5, 0, 0, 1)) == []
    assert veq(isect(O, s, (0, 0, 0, 0, 0, 1)), (-1, 1))
    assert veq(isect(O, s, (0, 0, 5, 0, 0, 1)), (-6, -4))


def test_hits1_2(O):
    """shape.rs:884-904."""
    L = O.lib()
    ts, ix, k = (C.c_double * 4)(), (C.c_int32 * 4)(), C.c_uint32(0)
    L.orc_list_insert_sorted(ts, ix, C.byref(k), 1.0, 0)
    L.orc_list_insert_sorted(ts, ix, C.byref(k), 2.0, 0)
    assert ts[L.orc_list_get_hit(ts, k.value)] == 1.0
    k = C.c_uint32(0)
    L.orc_list_insert_sorted(ts, ix, C.byref(k), -1.0, 0)
    L.orc_list_insert_sorted(ts, ix, C.byref(k), -2.0, 0)
    assert L.orc_list_get_hit(ts, k.value) == -1


def test_shape_transform1_2(O):
    """shape.rs:906-926."""
    assert veq(isect(O, O.shape(0, O.chain(("scaling", 2, 2, 2))), (0, 0, -5, 0, 0, 1)), (3, 7))
    assert isect(O, O.shape(0, O.chain(("translation", 5, 0, 0))), (0, 0, -5, 0, 0, 1)) == []


def test_circle_normal1_to_5(O):
    """shape.rs:928-974."""
    s = O.shape(0)
    assert veq(normal_at(O, s, (1, 0, 0)), (1, 0, 0))
    assert veq(normal_at(O, s, (0, 1, 0)), (0, 1, 0))
    assert veq(normal_at(O, s, (0, 0, 1)), (0, 0, 1))
    n = math.sqrt(3) / 3
    got = normal_at(O, s, (n, n, n))
    assert veq(got, (n, n, n)) and feq(math.sqrt(sum(c * c for c in got)), 1.0)


def test_translate_normal1_2(O):
    """shape.rs:976-1002."""
    s = O.shape(0, O.chain(("translation", 0, 1, 0)))
    assert veq(normal_at(O, s, (0, 1.70711, -0.70711)), (0, 0.70711, -0.70711))
    s = O.shape(0, O.chain(("rotation_z", PI / 5), ("scaling", 1, 0.5, 1)))
    assert veq(normal_at(O, s, (0, SQ2, -SQ2)), (0, 0.97014, -0.24254))


# ---------------------------------------------------------------- shape.rs: world / shading
def test_world1_2(O):
    """shape.rs:1004-1027 (exact, in order)."""
    arr = O.world(O.default_world())
    ts, _ = world_intersect(O, arr, 2, (0, 0, -5, 0, 0, 1))
    assert ts == [4.0, 4.5, 5.5, 6.0]


def test_world3(O):
    """shape.rs:1029-1039 (exact)."""
    arr = O.world([O.shape(0)])
    ts, ix = world_intersect(O, arr, 1, (0, 0, 0, 0, 0, 1))
    h = comps(O, arr, (0, 0, 0, 0, 0, 1), ts, ix, ts.index(1.0))
    assert h.inside == 1 and list(h.point) == [0.0, 0.0, 1.0] and list(h.eyev) == [0.0, 0.0, -1.0]


def test_world4(O):
    """shape.rs:1041-1054."""
    arr = O.world(O.default_world())
    ray = (0, 0, -5, 0, 0, 1)
    ts = isect(O, arr[0], ray)
    h = comps(O, arr, ray, ts, [0, 0], ts.index(4.0))
    assert veq(shade_hit(O, arr, 2, O.light(), h, 1), (0.38066, 0.47583, 0.2855))


def test_world5(O):
    """shape.rs:1056-1071."""
    arr = O.world(O.default_world())
    ray = (0, 0, 0, 0, 0, 1)
    ts = isect(O, arr[1], ray)
    h = comps(O, arr, ray, ts, [1, 1], ts.index(0.5))
    assert veq(shade_hit(O, arr, 2, O.light((0, 0.25, 0)), h, 1), (0.90498, 0.90498, 0.90498))


def test_color_at1_2_3(O):
    """shape.rs:1073-1112."""
    arr = O.world(O.default_world())
    assert veq(O.color_at(arr, 2, O.light(), (0, 0, -5, 0, 1, 0), 1), (0, 0, 0))
    assert veq(O.color_at(arr, 2, O.light(), (0, 0, -5, 0, 0, 1), 1), (0.38066, 0.47583, 0.2855))
    shapes = O.default_world()
    shapes[0].material.ambient = 1.0
    shapes[1].material.ambient = 1.0
    arr = O.world(shapes)
    assert veq(O.color_at(arr, 2, O.light(), (0, 0, 0.75, 0, 0, -1), 1), (1, 1, 1))


def test_color_at_jd1_and_intersect_jd1(O):
    """shape.rs:1114-1130 (grazing ray on a sphere scaled by 1.00001: colour must not be black) and
    shape.rs:867-882 (the same ray against the unit sphere; the reference only prints the count)."""
    n = math.sin(PI * 3.0 / 4.0)
    d = 1.0 / math.sqrt(2.0)
    ray = (0, 0, -2.0 * n, 0, d, d)
    s = O.shape(0, O.chain(("scaling", 1.00001, 1.00001, 1.00001)), O.material(color=(1, 0, 0)))
    c = O.color_at(O.world([s]), 1, O.light(), ray, 1)
    assert tuple(c) != (0.0, 0.0, 0.0)
    assert len(isect(O, s, ray)) == 2
    assert len(isect(O, O.shape(0, O.chain(("scaling", 1.0, 1.0, 1.0))), ray)) in (0, 2)


def test_render_jd1_and_async1(O):
    """camera.rs:233-247 (101x101 render of a scaled, lifted sphere; the reference prints pixel
    (50, 50)) and camera.rs:250-254 (render_async of the default World, 20x10: runs, every pixel)."""
    s = O.shape(0, O.chain(("scaling", 2.0, 2.0, 2.0), ("translation", 0.0, 1.01, 0.0)))
    cam = O.camera(101, 101, PI / 2, O.view_transform((0, 0, -5), (0, 0, 0), (0, 1, 0)))
    img = O.render(O.world([s]), 1, O.light(), cam, mode=0)
    assert img.shape == (101, 101, 3) and img[50, 50].any() and not img[100].any() and not img[:, 100].any()
    img = O.render(O.world(O.default_world()), 2, O.light(), O.camera(20, 10, 1.5), mode=1)
    assert img.shape == (10, 20, 3) and img.any()


def test_shadow1(O):
    """shape.rs:1132-1144: exact (0.1, 0.1, 0.1)."""
    arr = O.world([O.shape(0), O.shape(0, O.chain(("translation", 0, 0, 10)))])
    lgt = O.light((0, 0, -10))
    ray = (0, 0, 5, 0, 0, 1)
    ts, ix = world_intersect(O, arr, 2, ray)
    assert ts.count(4.0) == 1
    h = comps(O, arr, ray, ts, ix, ts.index(4.0))
    assert shade_hit(O, arr, 2, lgt, h, 1) == [0.1, 0.1, 0.1]


def test_shadow2(O):
    """shape.rs:1146-1156."""
    s = O.shape(0, O.chain(("translation", 0, 0, 1)))
    arr = O.world([s])
    ray = (0, 0, -5, 0, 0, 1)
    ts = isect(O, s, ray)
    h = comps(O, arr, ray, ts, [0, 0], ts.index(5.0))
    assert h.over_point[2] < -1e-8 / 2 and h.point[2] > h.over_point[2]


def test_material_shadow2_to_5(O):
    """material.rs:459-481: World::is_shadowed on the default world."""
    arr = O.world(O.default_world())
    lgt = O.light()
    f = lambda p: O.lib().orc_is_shadowed(arr, 2, C.byref(lgt), O.Vec3(*p))
    assert f((0, 10, 0)) == 0 and f((10, -10, 10)) == 1 and f((-20, 20, -20)) == 0 and f((-2, 2, -2)) == 0


# ---------------------------------------------------------------- planes
def test_plane1_to_5(O):
    """shape.rs:1158-1203 (exact)."""
    p = O.shape(1)
    for pt in ((0, 0, 0), (10, 0, -10), (-5, 0, 150)):
        assert normal_at(O, p, pt) == [0.0, 1.0, 0.0]
    assert isect(O, p, (0, 10, 0, 0, 0, 1)) == []
    assert isect(O, p, (0, 0, 0, 0, 0, 1)) == []
    assert isect(O, p, (0, 1, 0, 0, -1, 0)) == [1.0]
    assert isect(O, p, (0, -1, 0, 0, 1, 0)) == [1.0]


# ---------------------------------------------------------------- reflection
def _world_with_floor(O, **matkw):
    shapes = O.default_world()
    shapes.append(O.shape(1, O.chain(("translation", 0, -1, 0)), O.material(**matkw)))
    return shapes


def test_reflect1(O):
    """shape.rs:1205-1213 (exact)."""
    arr = O.world([O.shape(1)])
    h = comps(O, arr, (0, 0, -1, 0, -SQ2, SQ2), [SQ2], [0], 0)
    assert list(h.reflectv) == [0.0, SQ2, SQ2]


def test_reflect2(O):
    """shape.rs:1215-1229."""
    shapes = O.default_world()
    shapes[1].material.ambient = 1.0
    arr = O.world(shapes)
    h = comps(O, arr, (0, 0, 0, 0, 0, 1), [1.0], [1], 0)
    out = O.Vec3()
    lgt = O.light()
    O.lib().orc_reflected_color(arr, 2, C.byref(lgt), C.byref(h), 1, out)
    assert list(out) == [0.0, 0.0, 0.0]


def test_reflect3_4(O):
    """shape.rs:1231-1266."""
    arr = O.world(_world_with_floor(O, reflective=0.5))
    ray = (0, 0, -3, 0, -SQ2, SQ2)
    h = comps(O, arr, ray, [SQ2 * 2.0], [2], 0)
    out = O.Vec3()
    lgt = O.light()
    O.lib().orc_reflected_color(arr, 3, C.byref(lgt), C.byref(h), 1, out)
    assert veq(list(out), (0.19032, 0.2379, 0.14274))
    assert veq(shade_hit(O, arr, 3, lgt, h, 1), (0.87677, 0.92436, 0.82918))


def _glass(O, xf, ior):
    return O.shape(0, xf, O.material(transparency=1.0, refractive_index=ior))


def test_reflect5(O):
    """shape.rs:1268-1304: n1/n2 at six intersections (exact)."""
    A = _glass(O, O.chain(("scaling", 2, 2, 2)), 1.5)
    B = _glass(O, O.chain(("translation", 0, 0, -0.25)), 2.0)
    Cc = _glass(O, O.chain(("translation", 0, 0, 0.25)), 2.5)
    arr = O.world([A, B, Cc])
    ts = [2.0, 2.75, 3.25, 4.75, 5.25, 6.0]
    ix = [0, 1, 2, 1, 2, 0]
    want = [(1.0, 1.5), (1.5, 2.0), (2.0, 2.5), (2.5, 2.5), (2.5, 1.5), (1.5, 1.0)]
    for i in range(6):
        h = comps(O, arr, (0, 0, -4, 0, 0, 1), ts, ix, i)
        assert (h.n1, h.n2) == want[i]


def test_under_point(O):
    """shape.rs:1306-1315."""
    s = _glass(O, O.chain(("translation", 0, 0, 1)), 1.5)
    arr = O.world([s])
    h = comps(O, arr, (0, 0, -5, 0, 0, 1), [5.0], [0], 0)
    assert h.under_point[2] > 1e-8 / 2 and h.point[2] < h.under_point[2]


# ---------------------------------------------------------------- refraction
def _refracted(O, arr, n, h, remaining):
    out = O.Vec3()
    lgt = O.light()
    O.lib().orc_refracted_color(arr, n, C.byref(lgt), C.byref(h), remaining, out)
    return list(out)


def test_refract_1_2_3(O):
    """shape.rs:1317-1365 (exact BLACK): opaque; remaining 0; total internal reflection."""
    arr = O.world(O.default_world())
    ray = (0, 0, -5, 0, 0, 1)
    h = comps(O, arr, ray, [4.0, 6.0], [0, 0], 0)
    assert _refracted(O, arr, 2, h, 5) == [0.0, 0.0, 0.0]
    shapes = O.default_world()
    shapes[0].material.transparency = 1.0
    shapes[0].material.refractive_index = 1.5
    arr = O.world(shapes)
    h = comps(O, arr, ray, [4.0, 6.0], [0, 0], 0)
    assert _refracted(O, arr, 2, h, 0) == [0.0, 0.0, 0.0]
    h = comps(O, arr, (0, 0, SQ2, 0, 1, 0), [-SQ2, SQ2], [0, 0], 1)
    assert _refracted(O, arr, 2, h, 5) == [0.0, 0.0, 0.0]


def test_refract_4(O):
    """shape.rs:1367-1395: TestPattern outer sphere seen through a glass inner sphere."""
    shapes = O.default_world()
    shapes[0] = O.shape(0, O.mat(), O.material(color=(0.8, 1.0, 0.6), diffuse=0.7, specular=0.2, ambient=1.0,
                                               pattern=("test", (0, 0, 0), (0, 0, 0), None)))
    shapes[1].material.transparency = 1.0
    shapes[1].material.refractive_index = 1.5
    arr = O.world(shapes)
    h = comps(O, arr, (0, 0, 0.1, 0, 1, 0), [-0.9899, -0.4899, 0.4899, 0.9899], [0, 1, 1, 0], 2)
    assert veq(_refracted(O, arr, 2, h, 5), (0.0, 0.99888, 0.04725))


def _floor_ball_world(O, **floor_kw):
    shapes = O.default_world()
    floor = O.shape(1, O.chain(("translation", 0, -1, 0)), O.material(**floor_kw))
    ball = O.shape(0, O.chain(("translation", 0, -3.5, -0.5)),
                   O.material(color=(1.0, 0.0, 0.0), ambient=0.5))
    return shapes + [floor, ball]


def test_refract_5(O):
    """shape.rs:1397-1420."""
    arr = O.world(_floor_ball_world(O, transparency=0.5, refractive_index=1.5))
    h = comps(O, arr, (0, 0, -3, 0, -SQ2, SQ2), [SQ2 * 2.0], [2], 0)
    assert veq(shade_hit(O, arr, 4, O.light(), h, 5), (0.93642, 0.68642, 0.68642))


# ---------------------------------------------------------------- Schlick
def test_schlick_1_2_3(O):
    """shape.rs:1422-1450."""
    g = _glass(O, O.mat(), 1.5)
    arr = O.world([g])
    h = comps(O, arr, (0, 0, SQ2, 0, 1, 0), [-SQ2, SQ2], [0, 0], 1)
    assert O.lib().orc_reflectance(C.byref(h)) == 1.0
    h = comps(O, arr, (0, 0, 0, 0, 1, 0), [-1.0, 1.0], [0, 0], 1)
    assert feq(O.lib().orc_reflectance(C.byref(h)), 0.04)
    h = comps(O, arr, (0, 0.99, -2, 0, 0, 1), [1.8589], [0], 0)
    assert feq(O.lib().orc_reflectance(C.byref(h)), 0.48873)


def test_schlick_4(O):
    """shape.rs:1452-1476."""
    arr = O.world(_floor_ball_world(O, reflective=0.5, transparency=0.5, refractive_index=1.5))
    h = comps(O, arr, (0, 0, -3, 0, -SQ2, SQ2), [SQ2 * 2.0], [2], 0)
    assert veq(shade_hit(O, arr, 4, O.light(), h, 5), (0.93391, 0.69643, 0.69243))


# ---------------------------------------------------------------- cubes
def test_cube1_2_normals(O):
    """shape.rs:1478-1538."""
    c = O.shape(2)
    hits = [((5, 0.5, 0), (-1, 0, 0), 4, 6), ((-5, 0.5, 0), (1, 0, 0), 4, 6), ((0.5, 5, 0), (0, -1, 0), 4, 6),
            ((0.5, -5, 0), (0, 1, 0), 4, 6), ((0.5, 0, 5), (0, 0, -1), 4, 6), ((0.5, 0, -5), (0, 0, 1), 4, 6),
            ((0, 0.5, 0), (0, 0, 1), -1, 1)]
    for o, d, t0, t1 in hits:
        assert veq(isect(O, c, (*o, *d)), (t0, t1))
    misses = [((-2, 0, 0), (0.2673, 0.5345, 0.8018)), ((0, -2, 0), (0.8018, 0.2673, 0.5345)),
              ((0, 0, -2), (0.5345, 0.8018, 0.2673)), ((2, 0, 2), (0, 0, -1)), ((0, 2, 2), (0, -1, 0)), ((2, 2, 0), (-1, 0, 0))]
    for o, d in misses:
        assert isect(O, c, (*o, *d)) == []
    normals = [((1, 0.5, -0.8), (1, 0, 0)), ((-1, -0.2, 0.9), (-1, 0, 0)), ((-0.4, 1, -0.1), (0, 1, 0)),
               ((0.3, -1, -0.7), (0, -1, 0)), ((-0.6, 0.3, 1), (0, 0, 1)), ((0.4, 0.4, -1), (0, 0, -1)),
               ((1, 1, 1), (1, 0, 0)), ((-1, -1, -1), (-1, 0, 0))]
    for p, n in normals:  # normal_at_local; identity transform => normal_at normalises the same axis
        assert normal_at(O, c, p) == [float(v) for v in n]


# ---------------------------------------------------------------- material.rs
def _lighting(O, m, lgt, point, eye, normal, in_shadow, shape=None):
    out = O.Vec3()
    rc = O.lib().orc_lighting(C.byref(m), C.byref(shape) if shape is not None else None, C.byref(lgt), O.Vec3(*point),
                              O.Vec3(*eye), O.Vec3(*normal), 1 if in_shadow else 0, out)
    assert rc == 0
    return list(out)


def test_lighting1_to_5_and_shadow1(O):
    """material.rs:379-457."""
    m = O.material()
    n = (0, 0, -1)
    assert veq(_lighting(O, m, O.light((0, 0, -10)), (0, 0, 0), (0, 0, -1), n, False), (1.9,) * 3)
    assert veq(_lighting(O, m, O.light((0, 0, -10)), (0, 0, 0), (0, SQ2, -SQ2), n, False), (1.0,) * 3)
    assert veq(_lighting(O, m, O.light((0, 10, -10)), (0, 0, 0), (0, 0, -1), n, False), (0.7364,) * 3)
    assert veq(_lighting(O, m, O.light((0, 10, -10)), (0, 0, 0), (0, -SQ2, -SQ2), n, False), (1.6364,) * 3)
    assert veq(_lighting(O, m, O.light((0, 0, 10)), (0, 0, 0), (0, 0, -1), n, False), (0.1,) * 3)
    assert _lighting(O, m, O.light((0, 0, -10)), (0, 0, 0), (0, 0, -1), n, True) == [0.1, 0.1, 0.1]


def test_lighting_without_colour_or_pattern_is_an_error(O):
    """material.rs:328-331: expect() panics -> error code."""
    m = O.material(color=None)
    out = O.Vec3()
    lgt = O.light()
    rc = O.lib().orc_lighting(C.byref(m), None, C.byref(lgt), O.Vec3(0, 0, 0), O.Vec3(0, 0, -1), O.Vec3(0, 0, -1), 0, out)
    assert rc == 2


def _pattern_at_shape(O, s, p):
    out = O.Vec3()
    O.lib().orc_pattern_at_shape(C.byref(s.material), C.byref(s), O.Vec3(*p), out)
    return list(out)


def test_pattern1_2_3(O):
    """material.rs:483-518 (exact)."""
    z = (0, 0, 0)
    s = O.shape(0, O.chain(("scaling", 2, 2, 2)), O.material(pattern=("test", z, z, None)))
    assert _pattern_at_shape(O, s, (2, 3, 4)) == [1.0, 1.5, 2.0]
    s = O.shape(0, O.mat(), O.material(pattern=("test", z, z, O.chain(("scaling", 2, 2, 2)))))
    assert _pattern_at_shape(O, s, (2, 3, 4)) == [1.0, 1.5, 2.0]
    s = O.shape(0, O.chain(("scaling", 2, 2, 2)), O.material(pattern=("test", z, z, O.chain(("translation", 0.5, 1, 1.5)))))
    assert _pattern_at_shape(O, s, (2.5, 3, 3.5)) == [0.75, 0.5, 0.25]


def test_stripe_pattern1_2_3(O):
    """material.rs:520-555."""
    W, B = (1, 1, 1), (0, 0, 0)
    s = O.shape(0, O.chain(("scaling", 2, 2, 2)), O.material(pattern=("stripe", W, B, None)))
    assert _pattern_at_shape(O, s, (1.5, 0, 0)) == [1.0, 1.0, 1.0]
    s = O.shape(0, O.mat(), O.material(pattern=("stripe", W, B, O.chain(("scaling", 2, 2, 2)))))
    assert _pattern_at_shape(O, s, (1.5, 0, 0)) == [1.0, 1.0, 1.0]
    s = O.shape(0, O.chain(("scaling", 2, 2, 2)), O.material(pattern=("stripe", W, B, O.chain(("translation", 0.5, 0, 0)))))
    assert _pattern_at_shape(O, s, (2.5, 0, 0)) == [1.0, 1.0, 1.0]


def test_gradient_pattern(O):
    """material.rs:557-565 (exact)."""
    m = O.material(pattern=("gradient", (1, 1, 1), (0, 0, 0), None))
    out = O.Vec3()
    for x, want in ((0.0, 1.0), (0.25, 0.75), (0.5, 0.5), (0.75, 0.25)):
        O.lib().orc_pattern_at(C.byref(m), O.Vec3(x, 0, 0), out)
        assert list(out) == [want] * 3


def test_negative_coordinates_use_fmod_sign_of_dividend(O):
    """material.rs:98,199: Rust `%` keeps the dividend's sign: floor(-0.5) % 2 = -1 != 0 -> colour b
    (unpinned by a reference test; source reading)."""
    m = O.material(pattern=("checker", (1, 1, 1), (0, 0, 0), None))
    out = O.Vec3()
    O.lib().orc_pattern_at(C.byref(m), O.Vec3(-0.5, 0.5, 0.5), out)
    assert list(out) == [0.0, 0.0, 0.0]
    O.lib().orc_pattern_at(C.byref(m), O.Vec3(-1.5, 0.5, 0.5), out)  # floor = -2 -> 0 -> a
    assert list(out) == [1.0, 1.0, 1.0]


# ---------------------------------------------------------------- color.rs / canvas.rs
def test_color_scale_and_ppm(O):
    """color.rs:100-114 (truncating saturating cast, clamp) and canvas.rs:86-109 (unpinned: no
    reference test; format by source reading)."""
    sc = O.lib().orc_color_scale
    assert sc(1.0, 255) == 255 and sc(0.5, 255) == 127 and sc(-0.2, 255) == 0 and sc(7.0, 255) == 255
    assert sc(float("nan"), 255) == 0 and sc(0.999999, 255) == 254 and sc(1e300, 255) == 255
    img = np.zeros((2, 3, 3))
    img[0, 0] = (1, 0, 0)
    img[1, 2] = (0.5, 1.5, -1)
    assert O.format_ppm(img) == b"P3\n3 2\n255\n255 0 0 0 0 0 0 0 0\n0 0 0 0 0 0 127 255 0\n"
