"""ctypes binding of liboracle.so — the CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
It reuses the plain-data ctypes struct mirrors of include/rtc.h from the package's abi.py.
"""
from __future__ import annotations

import ctypes as C
import subprocess
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent
sys.path.insert(0, str(ROOT))
from _bootstrap import package  # noqa: E402

_abi = __import__("importlib").import_module(package().__name__ + ".abi")
RtcCamera, RtcHit, RtcLight, RtcMaterial, RtcShape, RtcStats = (_abi.RtcCamera, _abi.RtcHit, _abi.RtcLight,
                                                               _abi.RtcMaterial, _abi.RtcShape, _abi.RtcStats)
Mat16, Vec3 = _abi.Mat16, _abi.Vec3
Ray6 = C.c_double * 6
PD = C.POINTER(C.c_double)

LIB_PATH = HERE / "liboracle.so"
_lib = None


def build(force: bool = False) -> Path:
    src = [HERE / "rtc_oracle.c", HERE / "rtc_oracle.h", ROOT / "include" / "rtc.h"]
    if force or not LIB_PATH.exists() or any(s.stat().st_mtime > LIB_PATH.stat().st_mtime for s in src):
        subprocess.run(["make", "-C", str(HERE), "-B" if force else "-s", "liboracle.so"], check=True)
    return LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            build()
        L = C.CDLL(str(LIB_PATH))
        D, U32, I32 = C.c_double, C.c_uint32, C.c_int32
        PS, PL, PC, PH = C.POINTER(RtcShape), C.POINTER(RtcLight), C.POINTER(RtcCamera), C.POINTER(RtcHit)
        sig = {
            "orc_matrix_identity": (None, [Mat16]),
            "orc_matrix_multiply": (None, [Mat16, Mat16, Mat16]),
            "orc_matrix_translation": (None, [Mat16, D, D, D, Mat16]),
            "orc_matrix_scaling": (None, [Mat16, D, D, D, Mat16]),
            "orc_matrix_rotation_x": (None, [Mat16, D, Mat16]),
            "orc_matrix_rotation_y": (None, [Mat16, D, Mat16]),
            "orc_matrix_rotation_z": (None, [Mat16, D, Mat16]),
            "orc_matrix_shearing": (None, [Mat16, D, D, D, D, D, D, Mat16]),
            "orc_matrix_determinant": (D, [Mat16]),
            "orc_matrix_inverse": (C.c_int, [Mat16, Mat16]),
            "orc_matrix_transpose": (None, [Mat16, Mat16]),
            "orc_view_transform": (None, [Vec3, Vec3, Vec3, Mat16]),
            "orc_transform_point": (None, [Mat16, Vec3, Vec3]),
            "orc_transform_vector": (None, [Mat16, Vec3, Vec3]),
            "orc_camera_init": (C.c_int, [U32, U32, D, Mat16, PC]),
            "orc_camera_ray_for_pixel": (None, [PC, U32, D, U32, D, Ray6]),
            "orc_material_default": (None, [C.POINTER(RtcMaterial)]),
            "orc_light_default": (None, [PL]),
            "orc_shape_init": (C.c_int, [U32, Mat16, C.POINTER(RtcMaterial), PS]),
            "orc_shape_intersect": (C.c_int, [PS, Ray6, C.c_double * 2]),
            "orc_normal_at": (None, [PS, Vec3, Vec3]),
            "orc_list_insert_sorted": (None, [PD, C.POINTER(I32), C.POINTER(U32), D, I32]),
            "orc_list_get_hit": (I32, [PD, U32]),
            "orc_world_intersect": (U32, [PS, U32, Ray6, PD, C.POINTER(I32)]),
            "orc_compute_vectors": (C.c_int, [PS, Ray6, PD, C.POINTER(I32), U32, U32, PH]),
            "orc_is_shadowed": (C.c_int, [PS, U32, PL, Vec3]),
            "orc_shade_hit": (None, [PS, U32, PL, PH, U32, Vec3]),
            "orc_color_at": (None, [PS, U32, PL, Ray6, U32, Vec3, PH]),
            "orc_color_at_streaming": (None, [PS, U32, PL, Ray6, U32, Vec3, PH]),
            "orc_reflected_color": (None, [PS, U32, PL, PH, U32, Vec3]),
            "orc_refracted_color": (None, [PS, U32, PL, PH, U32, Vec3]),
            "orc_reflectance": (D, [PH]),
            "orc_lighting": (C.c_int, [C.POINTER(RtcMaterial), PS, PL, Vec3, Vec3, Vec3, C.c_int, Vec3]),
            "orc_pattern_at": (None, [C.POINTER(RtcMaterial), Vec3, Vec3]),
            "orc_pattern_at_shape": (None, [C.POINTER(RtcMaterial), PS, Vec3, Vec3]),
            "orc_render": (None, [PS, U32, PL, PC, U32, U32, U32, PD, U32, C.c_int, C.POINTER(RtcStats)]),
            "orc_render_flags": (None, [PS, U32, PL, PC, U32, U32, U32, U32, PD, U32, C.c_int, C.POINTER(RtcStats)]),
            "orc_format_ppm": (C.c_size_t, [PD, U32, U32, C.c_char_p, C.c_size_t]),
            "orc_color_scale": (I32, [D, I32]),
            "orc_canvas_to_rgba8": (None, [PD, U32, U32, C.c_float, C.POINTER(C.c_uint8)]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


# ---- small Pythonic helpers used by the tests ---------------------------------------------
def mat(m=None) -> Mat16:
    out = Mat16()
    if m is None:
        lib().orc_matrix_identity(out)
    else:
        flat = np.asarray(m, dtype=np.float64).reshape(16)
        for i in range(16):
            out[i] = flat[i]
    return out


def chain(*ops) -> Mat16:
    """chain(("scaling", x, y, z), ("translation", x, y, z), ...) == identity().scaling(..).translation(..)"""
    m = mat()
    for op in ops:
        out = Mat16()
        getattr(lib(), "orc_matrix_" + op[0])(m, *[C.c_double(v) for v in op[1:]], out)
        m = out
    return m


def view_transform(frm, to, up) -> Mat16:
    out = Mat16()
    lib().orc_view_transform(Vec3(*frm), Vec3(*to), Vec3(*up), out)
    return out


def inverse(m: Mat16) -> Mat16:
    out = Mat16()
    if lib().orc_matrix_inverse(m, out):
        raise ValueError("Matrix is not invertable")
    return out


def material(**kw) -> RtcMaterial:
    m = RtcMaterial()
    lib().orc_material_default(C.byref(m))
    pattern = kw.pop("pattern", None)
    color = kw.pop("color", (1.0, 1.0, 1.0))
    if color is None:
        m.has_color = 0
    else:
        for i in range(3):
            m.color[i] = float(color[i])
    for k, v in kw.items():
        setattr(m, k, float(v))
    if pattern is not None:
        kind, a, b, xf = pattern
        m.pattern_kind = _abi.PATTERNS[kind]
        for i in range(3):
            m.pat_a[i] = float(a[i])
            m.pat_b[i] = float(b[i])
        inv = inverse(xf if xf is not None else mat())
        for i in range(16):
            m.pat_inv[i] = inv[i]
    return m


def shape(kind: int, transform: Mat16 | None = None, mat_: RtcMaterial | None = None) -> RtcShape:
    s = RtcShape()
    if lib().orc_shape_init(kind, transform if transform is not None else mat(), C.byref(mat_) if mat_ is not None else None, C.byref(s)):
        raise ValueError("Matrix is not invertable")
    return s


def light(position=(-10.0, 10.0, -10.0), intensity=(1.0, 1.0, 1.0)) -> RtcLight:
    l = RtcLight()
    for i in range(3):
        l.position[i] = float(position[i])
        l.intensity[i] = float(intensity[i])
    return l


def world(shapes: list[RtcShape], u8_ids: bool = False):
    """World::add_shape numbering (shape.rs:661-667). u8_ids=True reproduces the reference's own id type:
    `last_world_id: u8` wraps in a release build, so shape 256 gets id 0, shape 257 id 1, ... and
    compute_refractive (shape.rs:127) treats shapes that share an id as one container."""
    arr = (RtcShape * max(1, len(shapes)))()
    for i, s in enumerate(shapes):
        arr[i] = s
        arr[i].world_id = ((i + 1) & 0xFF) if u8_ids else i + 1
    return arr


def default_world():
    """impl Default for World (shape.rs:784-795)."""
    s1 = shape(0, mat(), material(color=(0.8, 1.0, 0.6), diffuse=0.7, specular=0.2))
    s2 = shape(0, chain(("scaling", 0.5, 0.5, 0.5)))
    return [s1, s2]


def camera(hsize, vsize, fov, view: Mat16 | None = None, samples: int = 1) -> RtcCamera:
    cam = RtcCamera()
    if lib().orc_camera_init(hsize, vsize, float(fov), view if view is not None else mat(), C.byref(cam)):
        raise ValueError("Matrix is not invertable")
    cam.samples = samples
    return cam


def color_at(shapes_arr, n, lgt, ray, remaining=5, streaming=False, want_hit=False):
    rgb = Vec3()
    hit = RtcHit()
    fn = lib().orc_color_at_streaming if streaming else lib().orc_color_at
    fn(shapes_arr, n, C.byref(lgt), Ray6(*ray), remaining, rgb, C.byref(hit))
    out = np.array(list(rgb))
    return (out, hit) if want_hit else out


def render(shapes_arr, n, lgt, cam, mode=1, y0=0, y1=None, nthreads=1, streaming=False, want_stats=False, flags=0):
    y1 = cam.vsize if y1 is None else y1
    out = np.zeros((y1 - y0, cam.hsize, 3), dtype=np.float64)
    st = RtcStats()
    lib().orc_render_flags(shapes_arr, n, C.byref(lgt), C.byref(cam), mode, flags, y0, y1, out.ctypes.data_as(PD), nthreads,
                           1 if streaming else 0, C.byref(st))
    if want_stats:
        d = {"rays_primary": st.rays_primary, "rays_shadow": st.rays_shadow, "rays_reflect": st.rays_reflect,
             "rays_refract": st.rays_refract, "pixels": st.pixels}
        if cam.samples != 1:
            d["pixels_resample"] = st.pixels_resample
        return out, d
    return out


def format_ppm(rgb: np.ndarray) -> bytes:
    a = np.ascontiguousarray(rgb, dtype=np.float64)
    need = lib().orc_format_ppm(a.ctypes.data_as(PD), a.shape[1], a.shape[0], None, 0)
    buf = C.create_string_buffer(need + 1)
    lib().orc_format_ppm(a.ctypes.data_as(PD), a.shape[1], a.shape[0], buf, need + 1)
    return buf.raw[:need]
