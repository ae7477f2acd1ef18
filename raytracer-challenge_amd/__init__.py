"""raytracer-challenge_amd — MI355X-native renderer for the hot path of joedane/raytracer-challenge.

This package is a thin ctypes binding of ``librtc.so`` (``include/rtc.h``): HIP kernels for
``Camera::render -> World::color_at -> World::intersect -> shade_hit`` behind a C-ABI, plus the
host-side setup arithmetic (Matrix / Camera / Shape constructors, the jamis.yml loader, the PPM
writer). The directory name carries a hyphen; import it through ``_bootstrap.py`` at the repo
root, which registers it as ``raytracer_challenge_amd``.

There is no CPU fallback: if ``librtc.so`` is missing, or no gfx950 device is usable, the
calls raise.
"""
from __future__ import annotations

import ctypes as C
import weakref
from pathlib import Path

import numpy as np

from .abi import (LUA_FRAME_FN, RtcLuaJob, RtcCamera, RtcHit, RtcLaunchInfo, RtcLight, RtcMaterial, RtcShape, RtcStats, Mat16, Vec3, SOURCE_NAMES,
                  SPHERE, PLANE, CUBE, MODE_RENDER, MODE_RENDER_ASYNC, FLAG_NONE, FLAG_NO_CULL, FLAG_AA_RESAMPLE, FLAG_LDS_TABLE,
                  EXCHANGE_RCCL, EXCHANGE_P2P, GATHER_NONE, GATHER_F64, GATHER_U8, GROUP_ID_BYTES, PATTERNS, STATUS_NAMES, declare)

PKG = Path(__file__).resolve().parent
LIB_PATH = PKG / "librtc.so"

_lib = None


class RtcError(RuntimeError):
    def __init__(self, status: int, where: str, detail: str = ""):
        self.status = status
        name = STATUS_NAMES.get(status, str(status))
        super().__init__(f"{where}: {name}" + (f" ({detail})" if detail else ""))


def lib() -> C.CDLL:
    """Load librtc.so (built by ``build.build()`` / ``__graft_entry__.build()``). Fails loudly."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise RuntimeError(f"{LIB_PATH} is missing: run `python __graft_entry__.py build` (hipcc, gfx950). "
                               "There is no CPU fallback for the render path.")
        # One HIP runtime per process: PyTorch wheels bundle their own libamdhip64.so.7 (same soname
        # as /opt/rocm's). Import torch first so that librtc.so binds to the runtime torch uses —
        # otherwise two HSA runtimes race for the device and the second one sees "no GPUs".
        try:
            import torch  # noqa: F401
        except Exception:  # torch is plumbing (device tensors, streams, RCCL), not a hard dependency
            pass
        _lib = C.CDLL(str(LIB_PATH))
        declare(_lib)
    return _lib


_fn_cache: dict = {}


def _render_rows():
    f = _fn_cache.get("rows")
    if f is None:
        f = _fn_cache["rows"] = lib().rtc_render_rows
    return f


def _render_bands():
    f = _fn_cache.get("bands")
    if f is None:
        f = _fn_cache["bands"] = lib().rtc_render_bands
    return f


def _check(status: int, where: str, detail: str = "") -> None:
    if status != 0:
        raise RtcError(status, where, detail)


# ---------------------------------------------------------------------------------------
# host-side mirror of the reference's setup API (names follow ch1/src/*.rs)
# ---------------------------------------------------------------------------------------
class Matrix:
    """Matrix (transform.rs:23-27) with the fluent, LEFT-multiplying builders (transform.rs:53-105)."""

    __slots__ = ("m",)

    def __init__(self, m=None):
        self.m = Mat16()
        if m is None:
            lib().rtc_matrix_identity(self.m)
        else:
            flat = np.asarray(m, dtype=np.float64).reshape(16)
            for i in range(16):
                self.m[i] = flat[i]

    @staticmethod
    def identity() -> "Matrix":
        return Matrix()

    def _apply(self, fn, *args) -> "Matrix":
        out = Matrix.__new__(Matrix)
        out.m = Mat16()
        fn(self.m, *[C.c_double(a) for a in args], out.m)
        return out

    def translation(self, x, y, z): return self._apply(lib().rtc_matrix_translation, x, y, z)
    def scaling(self, x, y, z): return self._apply(lib().rtc_matrix_scaling, x, y, z)
    def rotation_x(self, r): return self._apply(lib().rtc_matrix_rotation_x, r)
    def rotation_y(self, r): return self._apply(lib().rtc_matrix_rotation_y, r)
    def rotation_z(self, r): return self._apply(lib().rtc_matrix_rotation_z, r)
    def shearing(self, xy, xz, yx, yz, zx, zy): return self._apply(lib().rtc_matrix_shearing, xy, xz, yx, yz, zx, zy)

    def multiply(self, other: "Matrix") -> "Matrix":
        out = Matrix.__new__(Matrix)
        out.m = Mat16()
        lib().rtc_matrix_multiply(self.m, other.m, out.m)
        return out

    def inverse(self) -> "Matrix":
        out = Matrix.__new__(Matrix)
        out.m = Mat16()
        _check(lib().rtc_matrix_inverse(self.m, out.m), "Matrix.inverse")
        return out

    def transpose(self) -> "Matrix":
        out = Matrix.__new__(Matrix)
        out.m = Mat16()
        lib().rtc_matrix_transpose(self.m, out.m)
        return out

    def determinant(self) -> float:
        return float(lib().rtc_matrix_determinant(self.m))

    @staticmethod
    def make_view_transform(frm, to, up) -> "Matrix":
        out = Matrix.__new__(Matrix)
        out.m = Mat16()
        lib().rtc_view_transform(Vec3(*frm), Vec3(*to), Vec3(*up), out.m)
        return out

    def numpy(self) -> np.ndarray:
        return np.array(list(self.m), dtype=np.float64).reshape(4, 4)


def material(color=(1.0, 1.0, 1.0), ambient=0.1, diffuse=0.9, specular=0.9, shininess=200.0, reflective=0.0,
             transparency=0.0, refractive_index=1.0, pattern=None) -> RtcMaterial:
    """Material (material.rs:244-254) starting from Material::default() (white). `pattern` is
    (kind_name, color_a, color_b, Matrix|None) or None; `color=None` means Material.color = None."""
    m = RtcMaterial()
    lib().rtc_material_default(C.byref(m))
    if color is None:
        m.has_color = 0
    else:
        m.has_color = 1
        for i in range(3):
            m.color[i] = float(color[i])
    m.ambient, m.diffuse, m.specular, m.shininess = float(ambient), float(diffuse), float(specular), float(shininess)
    m.reflective, m.transparency, m.refractive_index = float(reflective), float(transparency), float(refractive_index)
    if pattern is not None:
        kind, a, b, xf = pattern
        xf = xf if xf is not None else Matrix.identity()
        _check(lib().rtc_material_set_pattern(C.byref(m), PATTERNS[kind], Vec3(*a), Vec3(*b), xf.m), "Pattern.set_transform")
    return m


def _shape(kind: int, transform: Matrix | None, mat: RtcMaterial | None) -> RtcShape:
    s = RtcShape()
    t = transform if transform is not None else Matrix.identity()
    _check(lib().rtc_shape_init(kind, t.m, C.byref(mat) if mat is not None else None, C.byref(s)), "Shape.new_with_transform_and_material")
    return s


def sphere(transform=None, mat=None) -> RtcShape: return _shape(SPHERE, transform, mat)   # shape.rs:308
def plane(transform=None, mat=None) -> RtcShape: return _shape(PLANE, transform, mat)     # shape.rs:436
def cube(transform=None, mat=None) -> RtcShape: return _shape(CUBE, transform, mat)       # shape.rs:525


def light(position=(-10.0, 10.0, -10.0), intensity=(1.0, 1.0, 1.0)) -> RtcLight:
    l = RtcLight()
    for i in range(3):
        l.position[i] = float(position[i])
        l.intensity[i] = float(intensity[i])
    return l


class World:
    """World (shape.rs:633-637): host-side list of shapes + one light; `add_shape` assigns
    world ids like the reference (shape.rs:661-667)."""

    def __init__(self, lgt: RtcLight | None = None):
        self.light = lgt if lgt is not None else light()
        self.shapes: list[RtcShape] = []

    def add_shape(self, s: RtcShape) -> "World":
        s.world_id = len(self.shapes) + 1
        self.shapes.append(s)
        return self

    @staticmethod
    def default() -> "World":
        """impl Default for World (shape.rs:784-795)."""
        w = World()
        w.add_shape(sphere(Matrix.identity(), material(color=(0.8, 1.0, 0.6), diffuse=0.7, specular=0.2)))
        w.add_shape(sphere(Matrix.identity().scaling(0.5, 0.5, 0.5)))
        return w

    def array(self):
        arr = (RtcShape * max(1, len(self.shapes)))()
        for i, s in enumerate(self.shapes):
            arr[i] = s
        return arr

    def __len__(self):
        return len(self.shapes)


def camera(hsize: int, vsize: int, fov: float, view: Matrix | None = None, samples: int = 1) -> RtcCamera:
    """Camera::new_with_transform (camera.rs:33-58)."""
    cam = RtcCamera()
    v = view if view is not None else Matrix.identity()
    _check(lib().rtc_camera_init(hsize, vsize, float(fov), v.m, C.byref(cam)), "Camera.new_with_transform")
    cam.samples = samples
    return cam


def ray_for_pixel(cam: RtcCamera, x: int, y: int, xo: float = 0.5, yo: float = 0.5) -> np.ndarray:
    out = (C.c_double * 6)()
    lib().rtc_camera_ray_for_pixel(C.byref(cam), x, C.c_double(xo), y, C.c_double(yo), out)
    return np.array(list(out))


def load_yaml(text: str | None = None, path: str | None = None):
    """jamis.yml-vocabulary loader -> (World, RtcCamera)."""
    shapes = C.POINTER(RtcShape)()
    n = C.c_uint32(0)
    lgt, cam = RtcLight(), RtcCamera()
    err = C.create_string_buffer(512)
    if path is not None:
        st = lib().rtc_scene_load_yaml_file(str(path).encode(), C.byref(shapes), C.byref(n), C.byref(lgt), C.byref(cam), err, 512)
    else:
        st = lib().rtc_scene_load_yaml(text.encode(), C.byref(shapes), C.byref(n), C.byref(lgt), C.byref(cam), err, 512)
    _check(st, "rtc_scene_load_yaml", err.value.decode(errors="replace"))
    w = World(lgt)
    for i in range(n.value):
        s = RtcShape()
        C.memmove(C.byref(s), C.byref(shapes[i]), C.sizeof(RtcShape))
        w.shapes.append(s)
    lib().rtc_free(shapes)
    return w, cam


class LuaJob:
    """One Render(world, camera, file) or encoder:AddFrame(world, camera) call of a script (rtc_lua_job), converted by
    lua.rs's *_from_table rules at the moment of the call."""

    def __init__(self, j: RtcLuaJob, index: int):
        self.index = index
        self.kind = "AddFrame" if j.kind == 1 else "Render"
        self.outfile = (j.outfile or b"").decode(errors="replace")
        self.animation, self.frame, self.line = j.animation, j.frame, j.line
        self.same_world_as_previous = bool(j.same_world_as_previous)
        self.camera = RtcCamera()
        C.memmove(C.byref(self.camera), C.byref(j.camera), C.sizeof(RtcCamera))
        lgt = RtcLight()
        C.memmove(C.byref(lgt), C.byref(j.light), C.sizeof(RtcLight))
        self.world = World(lgt)
        for i in range(j.n_shapes):
            s = RtcShape()
            C.memmove(C.byref(s), C.byref(j.shapes[i]), C.sizeof(RtcShape))
            self.world.shapes.append(s)


class LuaProgram:
    """A scene script of the reference's Lua front-end, interpreted (rtc_lua_run: csrc/host_lua.cpp carries its own
    interpreter of the Lua 5.3 subset those scripts use): `jobs` = its Render / AddFrame calls in order, `output` = what it
    print()ed. render(ctx) is lua.rs's render_lua: every job rendered on the GPU, 8-bit frames back in job order."""

    def __init__(self, text: str | None = None, path=None, base_dir=None, step_limit: int = 0):
        h = C.c_void_p()
        err = C.create_string_buffer(1024)
        if path is not None:
            st = lib().rtc_lua_run_file(str(path).encode(), step_limit, C.byref(h), err, 1024)
        else:
            st = lib().rtc_lua_run(text.encode(), None if base_dir is None else str(base_dir).encode(), step_limit, C.byref(h), err, 1024)
        _check(st, "rtc_lua_run", err.value.decode(errors="replace"))
        self._h = h
        self.output = lib().rtc_lua_program_output(h).decode(errors="replace")

    def __len__(self):
        return lib().rtc_lua_program_jobs(self._h) if self._h else 0

    def job(self, index: int) -> LuaJob:
        j = RtcLuaJob()
        _check(lib().rtc_lua_program_job(self._h, index, C.byref(j)), "rtc_lua_program_job")
        return LuaJob(j, index)

    @property
    def jobs(self):
        return [self.job(i) for i in range(len(self))]

    def render(self, ctx: "Context", on_frame=None, mode: int = MODE_RENDER_ASYNC, flags: int = 0, with_stats: bool = False):
        """rtc_lua_program_render: returns the list of (vsize, hsize, 3) uint8 frames in job order — or, with `on_frame`
        (called as on_frame(job_index, frame, outfile, kind); a true return value stops), nothing is kept."""
        frames = []
        raised = []

        def cb(_user, jp, index, rgb8):
            try:
                cam = jp.contents.camera
                frame = np.ctypeslib.as_array(rgb8, shape=(cam.vsize, cam.hsize, 3))
                if on_frame is None:
                    frames.append(frame.copy())
                    return 0
                j = jp.contents
                return 1 if on_frame(index, frame, (j.outfile or b"").decode(errors="replace"), "AddFrame" if j.kind == 1 else "Render") else 0
            except BaseException as e:  # never unwind through the C frames
                raised.append(e)
                return 1

        st = RtcStats()
        fn = LUA_FRAME_FN(cb)
        rc = lib().rtc_lua_program_render(ctx._h, self._h, mode, flags, fn, None, C.byref(st) if with_stats else None)
        if raised:
            raise raised[0]
        _check(rc, "rtc_lua_program_render")
        if with_stats:
            return frames, _stats_dict(st, True)
        return frames

    def render_to_files(self, ctx: "Context", out_dir, mode: int = MODE_RENDER_ASYNC, flags: int = 0) -> list:
        """render_lua with the files written: Render(world, camera, "x.png") -> out_dir/x.png, "x.ppm" -> the P3 file of
        Canvas::write_to_file_simple; any other extension (the reference's JPEG / GIF codecs are not rebuilt) -> the name
        + ".png". AddFrame frames of StartAnimation("a.gif") -> out_dir/a.gif.0000.png, a.gif.0001.png, ...
        Only the file's base name is used. Returns the paths in job order."""
        out = Path(out_dir)
        out.mkdir(parents=True, exist_ok=True)
        paths = []

        def on_frame(index, frame, outfile, kind):
            name = Path(outfile).name or f"job{index}"
            if kind == "AddFrame":
                target = out / f"{name}.{self.job(index).frame:04d}.png"
            elif name.lower().endswith((".png", ".ppm")):
                target = out / name
            else:
                target = out / (name + ".png")
            if target.suffix.lower() == ".ppm":
                write_ppm_rgb8(target, frame)
            else:
                write_png(target, frame)
            paths.append(target)
            return False

        self.render(ctx, on_frame=on_frame, mode=mode, flags=flags)
        return paths

    def close(self):
        if getattr(self, "_h", None):
            lib().rtc_lua_program_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def load_lua(text: str | None = None, path: str | None = None, render_index: int = 0):
    """One job of a Lua scene script (rtc_scene_load_lua) -> (World, RtcCamera, outfile, n_jobs)."""
    shapes = C.POINTER(RtcShape)()
    n, renders = C.c_uint32(0), C.c_uint32(0)
    lgt, cam = RtcLight(), RtcCamera()
    err, outfile = C.create_string_buffer(512), C.create_string_buffer(512)
    if path is not None:
        st = lib().rtc_scene_load_lua_file(str(path).encode(), render_index, C.byref(shapes), C.byref(n), C.byref(lgt), C.byref(cam), outfile, 512,
                                           C.byref(renders), err, 512)
    else:
        st = lib().rtc_scene_load_lua(text.encode(), render_index, C.byref(shapes), C.byref(n), C.byref(lgt), C.byref(cam), outfile, 512,
                                      C.byref(renders), err, 512)
    _check(st, "rtc_scene_load_lua", err.value.decode(errors="replace"))
    w = World(lgt)
    for i in range(n.value):
        s = RtcShape()
        C.memmove(C.byref(s), C.byref(shapes[i]), C.sizeof(RtcShape))
        w.shapes.append(s)
    lib().rtc_free(shapes)
    return w, cam, outfile.value.decode(errors="replace"), renders.value


def format_ppm(rgb: np.ndarray) -> bytes:
    """Canvas::write_to_file_simple (canvas.rs:86-109) into memory."""
    a = np.ascontiguousarray(rgb, dtype=np.float64)
    h, w = a.shape[0], a.shape[1]
    p = a.ctypes.data_as(C.POINTER(C.c_double))
    need = lib().rtc_canvas_format_ppm(p, w, h, None, 0)
    buf = C.create_string_buffer(need + 1)
    lib().rtc_canvas_format_ppm(p, w, h, buf, need + 1)
    return buf.raw[:need]


def to_rgba8(canvas: np.ndarray, gamma: float = 1.0) -> np.ndarray:
    """Canvas::to_imgbuf (canvas.rs:61-79): (H, W, 4) uint8, gamma-corrected, alpha 255."""
    c = np.ascontiguousarray(canvas, dtype=np.float64)
    h, w = c.shape[0], c.shape[1]
    out = np.empty((h, w, 4), dtype=np.uint8)
    lib().rtc_canvas_to_rgba8(c.ctypes.data_as(C.POINTER(C.c_double)), w, h, gamma, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def color_scale255(rgb: np.ndarray) -> np.ndarray:
    """Color::scale(c, 255) (color.rs:100-114) element-wise on the host."""
    a = np.ascontiguousarray(rgb, dtype=np.float64)
    out = np.empty(a.shape, dtype=np.uint8)
    lib().rtc_color_scale255(a.ctypes.data_as(C.POINTER(C.c_double)), a.size, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def format_ppm_rgb8(rgb8: np.ndarray) -> bytes:
    """The PPM of a frame that is already quantised ((H, W, 3) uint8, Color::scale'd: rtc_render_rgb8)."""
    a = np.ascontiguousarray(rgb8, dtype=np.uint8)
    h, w = a.shape[0], a.shape[1]
    p = a.ctypes.data_as(C.POINTER(C.c_uint8))
    need = lib().rtc_canvas_format_ppm_rgb8(p, w, h, None, 0)
    buf = C.create_string_buffer(need + 1)
    lib().rtc_canvas_format_ppm_rgb8(p, w, h, buf, need + 1)
    return buf.raw[:need]


def write_ppm_rgb8(path, rgb8: np.ndarray) -> None:
    a = np.ascontiguousarray(rgb8, dtype=np.uint8)
    _check(lib().rtc_canvas_write_ppm_rgb8(str(path).encode(), a.ctypes.data_as(C.POINTER(C.c_uint8)), a.shape[1], a.shape[0]),
           "Canvas.write_to_file_simple (rgb8)")


def format_png(pixels: np.ndarray) -> bytes:
    """An 8-bit PNG (rtc_canvas_format_png8) of a (H, W, 3) or (H, W, 4) uint8 frame — Canvas::write_to_file's ".png" case."""
    a = np.ascontiguousarray(pixels, dtype=np.uint8)
    if a.ndim != 3 or a.shape[2] not in (3, 4):
        raise ValueError("pixels must be (H, W, 3) or (H, W, 4) uint8")
    h, w, c = a.shape
    P8 = C.POINTER(C.c_uint8)
    need = lib().rtc_canvas_format_png8(a.ctypes.data_as(P8), w, h, c, None, 0)
    if need == 0:
        raise ValueError("empty frame")
    buf = np.empty(need, dtype=np.uint8)
    lib().rtc_canvas_format_png8(a.ctypes.data_as(P8), w, h, c, buf.ctypes.data_as(P8), need)
    return buf.tobytes()


def write_png(path, pixels: np.ndarray) -> None:
    a = np.ascontiguousarray(pixels, dtype=np.uint8)
    if a.ndim != 3 or a.shape[2] not in (3, 4):
        raise ValueError("pixels must be (H, W, 3) or (H, W, 4) uint8")
    _check(lib().rtc_canvas_write_png8(str(path).encode(), a.ctypes.data_as(C.POINTER(C.c_uint8)), a.shape[1], a.shape[0], a.shape[2]), "rtc_canvas_write_png8")


def write_ppm(path, rgb: np.ndarray) -> None:
    a = np.ascontiguousarray(rgb, dtype=np.float64)
    _check(lib().rtc_canvas_write_ppm(str(path).encode(), a.ctypes.data_as(C.POINTER(C.c_double)), a.shape[1], a.shape[0]), "Canvas.write_to_file_simple")


# ---------------------------------------------------------------------------------------
# device side
# ---------------------------------------------------------------------------------------
def _stats_dict(s: RtcStats, with_resample: bool = False) -> dict:
    d = {"rays_primary": s.rays_primary, "rays_shadow": s.rays_shadow, "rays_reflect": s.rays_reflect,
         "rays_refract": s.rays_refract, "pixels": s.pixels}
    if with_resample or s.pixels_resample:
        d["pixels_resample"] = s.pixels_resample
    return d


class Context:
    """One GPU + one stream (rtc_context). `stream` is a raw hipStream_t value (int) or None."""

    def __init__(self, device: int = 0, stream: int | None = None, _borrowed=None):
        self._worlds = []  # weak references to the worlds uploaded through this context
        self._owned = _borrowed is None
        if _borrowed is not None:   # a group member's context: owned by the group
            self._h = C.c_void_p(_borrowed)
            self.device = device
            return
        self._h = C.c_void_p()
        _check(lib().rtc_context_create(device, C.c_void_p(stream or None), C.byref(self._h)), "rtc_context_create",
               "no usable MI355X (gfx950); this library has no CPU fallback")
        self.device = device

    def close(self):
        if self._h:
            for ref in self._worlds:
                w = ref()
                if w is not None:
                    w.close()
            self._worlds = []
            if self._owned:
                lib().rtc_context_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _check(lib().rtc_context_synchronize(self._h), "rtc_context_synchronize")

    def device_info(self):
        name = C.create_string_buffer(128)
        cu, mhz = C.c_int32(), C.c_int32()
        _check(lib().rtc_context_device_info(self._h, name, 128, C.byref(cu), C.byref(mhz)), "rtc_context_device_info")
        return {"name": name.value.decode(), "compute_units": cu.value, "clock_mhz": mhz.value}

    def upload(self, world: World) -> "DeviceWorld":
        return DeviceWorld(self, world)

    def stats(self, extended: bool = False) -> dict:
        """Ray counters since the last reset. extended=True adds `rays_primary_proven_miss` (of rays_primary: primary rays of
        tiles the binning kernel proved black — counted as cast, answered without generating a ray; include/rtc.h)."""
        s = RtcStats()
        _check(lib().rtc_stats_read(self._h, C.byref(s)), "rtc_stats_read")
        d = _stats_dict(s)
        if extended:
            d["rays_primary_proven_miss"] = s.rays_primary_proven_miss
        return d

    def reset_stats(self):
        _check(lib().rtc_stats_reset(self._h), "rtc_stats_reset")

    def last_kernel_ms(self) -> float:
        ms = C.c_float()
        _check(lib().rtc_last_kernel_ms(self._h, C.byref(ms)), "rtc_last_kernel_ms")
        return ms.value

    def set_timing(self, every: int) -> None:
        """Time every `every`-th render launch (1 = all, 0 = none); see include/rtc.h."""
        _check(lib().rtc_context_set_timing(self._h, every), "rtc_context_set_timing")

    def kernel_times_ms(self, last: int = 1024) -> np.ndarray:
        """Durations (ms) of the most recent `last` render launches, oldest first (include/rtc.h)."""
        buf = (C.c_float * max(1, last))()
        n = C.c_uint32()
        _check(lib().rtc_kernel_times_ms(self._h, buf, last, C.byref(n)), "rtc_kernel_times_ms")
        return np.frombuffer(buf, dtype=np.float32, count=n.value).copy()

    def set_pipeline(self, depth: int) -> None:
        """Deal consecutive render launches over `depth` streams of the context's own (rtc_context_set_pipeline):
        outputs of `depth` consecutive launches must not overlap; results are complete after synchronize()."""
        _check(lib().rtc_context_set_pipeline(self._h, depth), "rtc_context_set_pipeline")

    def fence(self) -> None:
        """Make the context's stream wait for every launch enqueued so far (no host wait)."""
        _check(lib().rtc_context_fence(self._h), "rtc_context_fence")

    def last_launch_info(self) -> dict:
        """Object source, lists and lane the most recent render launch ran with (rtc_context_last_launch_info)."""
        i = RtcLaunchInfo()
        _check(lib().rtc_context_last_launch_info(self._h, C.byref(i)), "rtc_context_last_launch_info")
        return {"source": i.source, "source_name": SOURCE_NAMES.get(i.source, "?"), "reflective": bool(i.reflective),
                "refractive": bool(i.refractive), "binned_primary_pass": bool(i.binned), "light_lists": bool(i.light_lists),
                "lane": i.lane, "threads_per_workgroup": i.block, "dynamic_lds_bytes": i.lds_bytes,
                "tiles_per_workgroup": i.tiles_per_workgroup, "multi_tile_workgroups": i.multi_tile_workgroups}

    def binning_times_ms(self, last: int = 1024) -> np.ndarray:
        """Durations (ms) of the binning kernels of the most recent `last` timed launches (0 where a launch had none)."""
        buf = (C.c_float * max(1, last))()
        n = C.c_uint32()
        _check(lib().rtc_binning_times_ms(self._h, buf, last, C.byref(n)), "rtc_binning_times_ms")
        return np.frombuffer(buf, dtype=np.float32, count=n.value).copy()

    def device_arith(self, op: int, a: np.ndarray, b: np.ndarray | None = None) -> np.ndarray:
        a = np.ascontiguousarray(a, dtype=np.float64)
        bb = np.ascontiguousarray(b if b is not None else a, dtype=np.float64)
        out = np.empty_like(a)
        P = C.POINTER(C.c_double)
        _check(lib().rtc_device_arith(self._h, op, a.ctypes.data_as(P), bb.ctypes.data_as(P), a.size, out.ctypes.data_as(P)), "rtc_device_arith")
        return out


class _Pinned:
    """Owner of one rtc_host_alloc block; freed when the last array viewing it is gone."""

    def __init__(self, nbytes: int):
        self.ptr = C.c_void_p()
        _check(lib().rtc_host_alloc(nbytes, C.byref(self.ptr)), "rtc_host_alloc")
        self.buf = (C.c_char * nbytes).from_address(self.ptr.value)
        self.buf._owner = self  # numpy views hold `buf`; the cycle keeps this owner exactly as long

    def __del__(self):
        try:
            if self.ptr:
                lib().rtc_host_free(self.ptr)
                self.ptr = C.c_void_p()
        except Exception:
            pass


def host_canvas(vsize: int, hsize: int) -> np.ndarray:
    """A zeroed (vsize, hsize, 3) float64 canvas in page-locked memory (rtc_host_alloc)."""
    arr = np.frombuffer(_Pinned(vsize * hsize * 24).buf, dtype=np.float64).reshape(vsize, hsize, 3)
    arr[...] = 0.0
    return arr


class DeviceWorld:
    """Flattened World resident in HBM (rtc_world)."""

    def __init__(self, ctx: Context, world: World):
        self.ctx = ctx
        self._h = C.c_void_p()
        arr = world.array()
        _check(lib().rtc_world_create(ctx._h, arr, len(world.shapes), C.byref(world.light), C.byref(self._h)), "rtc_world_create")
        self.n = len(world.shapes)
        ctx._worlds.append(weakref.ref(self))

    def close(self):
        if self._h:
            lib().rtc_world_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, cam: RtcCamera, mode: int = MODE_RENDER_ASYNC, flags: int = 0, with_stats: bool = False,
               out: np.ndarray | None = None):
        """Camera::render(&World) -> Canvas as a (vsize, hsize, 3) float64 array (host). `out` = a
        canvas to reuse, e.g. one from host_canvas() (page-locked: the copy runs at link speed)."""
        if out is None:
            out = np.empty((cam.vsize, cam.hsize, 3), dtype=np.float64)
        elif out.shape != (cam.vsize, cam.hsize, 3) or out.dtype != np.float64 or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous (vsize, hsize, 3) float64 array")
        st = RtcStats()
        _check(lib().rtc_render(self.ctx._h, self._h, C.byref(cam), mode, flags, out.ctypes.data_as(C.POINTER(C.c_double)),
                                C.byref(st) if with_stats else None), "rtc_render")
        if with_stats:
            return out, _stats_dict(st, cam.samples != 1)
        return out

    def render_rgb8(self, cam: RtcCamera, mode: int = MODE_RENDER_ASYNC, flags: int = 0, with_stats: bool = False,
                    out: np.ndarray | None = None):
        """Camera::render for a caller that only writes the image: the (vsize, hsize, 3) uint8 frame of
        Color::scale(c, 255), quantised on the device — 3 bytes per pixel cross PCIe (rtc_render_rgb8)."""
        if out is None:
            out = np.empty((cam.vsize, cam.hsize, 3), dtype=np.uint8)
        elif out.shape != (cam.vsize, cam.hsize, 3) or out.dtype != np.uint8 or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous (vsize, hsize, 3) uint8 array")
        st = RtcStats()
        _check(lib().rtc_render_rgb8(self.ctx._h, self._h, C.byref(cam), mode, flags, out.ctypes.data_as(C.POINTER(C.c_uint8)),
                                     C.byref(st) if with_stats else None), "rtc_render_rgb8")
        if with_stats:
            return out, _stats_dict(st, cam.samples != 1)
        return out

    def render_rows(self, cam: RtcCamera, y0: int, y1: int, d_ptr: int, mode: int = MODE_RENDER_ASYNC, flags: int = 0,
                    d_ptr8: int | None = None) -> None:
        """Enqueue rows [y0, y1) into the DEVICE buffer at address `d_ptr` (no synchronisation);
        `d_ptr8` optionally receives the rows quantised to 8 bits (Color::scale)."""
        st = _render_rows()(self.ctx._h, self._h, cam, mode, y0, y1, d_ptr, d_ptr8, flags)  # launch path: no temporaries
        if st != 0:
            raise RtcError(st, "rtc_render_rows")

    def render_bands(self, cam: RtcCamera, first_band: int, band_stride: int, d_ptr: int, mode: int = MODE_RENDER_ASYNC,
                     flags: int = 0, d_ptr8: int | None = None) -> None:
        """Enqueue the 8-row bands first_band, first_band + band_stride, ... packed one after the
        other into the DEVICE buffer at `d_ptr` (interleaved row tiles, include/rtc.h)."""
        st = _render_bands()(self.ctx._h, self._h, cam, mode, first_band, band_stride, d_ptr, d_ptr8, flags)
        if st != 0:
            raise RtcError(st, "rtc_render_bands")

    def render_views(self, cams, first_band: int, band_stride: int, d_ptr: int, view_rows: int, mode: int = MODE_RENDER_ASYNC,
                     flags: int = 0, d_ptr8: int | None = None) -> None:
        """Enqueue ONE launch that renders every camera of `cams` (a list of RtcCamera, or a prebuilt
        ctypes array of them) onto this World; view v lands `v * view_rows` rows below view 0
        (rtc_render_views, include/rtc.h)."""
        arr = cams if isinstance(cams, C.Array) else (RtcCamera * len(cams))(*cams)
        st = lib().rtc_render_views(self.ctx._h, self._h, arr, len(arr), mode, first_band, band_stride, d_ptr, d_ptr8, view_rows, flags)
        if st != 0:
            raise RtcError(st, "rtc_render_views")

    def color_at(self, rays: np.ndarray, remaining: int = 5, want_hits: bool = False, flags: int = 0):
        """World::color_at for an (n, 6) array of rays; returns rgb (n,3) [and the rtc_hit array]."""
        r = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        n = r.shape[0]
        rgb = np.empty((n, 3), dtype=np.float64)
        hits = (RtcHit * max(1, n))() if want_hits else None
        P = C.POINTER(C.c_double)
        _check(lib().rtc_color_at(self.ctx._h, self._h, r.ctypes.data_as(P), n, remaining, flags, rgb.ctypes.data_as(P), hits), "rtc_color_at")
        return (rgb, hits) if want_hits else rgb


def host_canvas_rgb8(vsize: int, hsize: int) -> np.ndarray:
    """A zeroed (vsize, hsize, 3) uint8 frame in page-locked memory (rtc_host_alloc)."""
    arr = np.frombuffer(_Pinned(vsize * hsize * 3).buf, dtype=np.uint8).reshape(vsize, hsize, 3)
    arr[...] = 0
    return arr


def host_register(arr: np.ndarray) -> None:
    """rtc_host_register: page-lock a canvas the caller allocated (a Vec<Color> on the Rust side)."""
    _check(lib().rtc_host_register(C.c_void_p(arr.ctypes.data), arr.nbytes), "rtc_host_register")


def host_unregister(arr: np.ndarray) -> None:
    _check(lib().rtc_host_unregister(C.c_void_p(arr.ctypes.data)), "rtc_host_unregister")


def group_unique_id() -> bytes:
    """ncclGetUniqueId through the C-ABI (rank 0 calls it and ships the 128 bytes to the other ranks)."""
    buf = (C.c_uint8 * GROUP_ID_BYTES)()
    _check(lib().rtc_group_unique_id(buf), "rtc_group_unique_id", "RCCL not loadable")
    return bytes(buf)


class Group:
    """N GPUs rendering one frame together (rtc_group): 8-row bands dealt round-robin, RCCL gather of the
    f64 tiles to member 0, un-deal on member 0's device. Group(devices=[...]) drives all devices from this
    process; Group(device=d, nranks=N, rank=r, uid=...) is one member of a one-process-per-GPU group."""

    def __init__(self, devices=None, exchange: int = EXCHANGE_RCCL, device: int | None = None, nranks: int | None = None,
                 rank: int | None = None, uid: bytes | None = None):
        self._h = C.c_void_p()
        if devices is not None:
            arr = (C.c_int32 * len(devices))(*devices)
            _check(lib().rtc_group_create(arr, len(devices), exchange, C.byref(self._h)), "rtc_group_create")
        else:
            idb = (C.c_uint8 * GROUP_ID_BYTES)(*uid)
            _check(lib().rtc_group_create_rank(device, nranks, rank, idb, C.byref(self._h)), "rtc_group_create_rank")
        self.size = lib().rtc_group_size(self._h)
        self.local_size = lib().rtc_group_local_size(self._h)
        self.contexts = []
        for i in range(self.local_size):   # member i's context, on member i's device (in-process: devices[i]; rank mode: device)
            ptr = lib().rtc_group_context(self._h, i)
            if not ptr:
                raise RtcError(4, "rtc_group_context", f"member {i} has no context")
            self.contexts.append(Context(device=(devices[i] if devices is not None else device), _borrowed=ptr))
        self._worlds = []

    def close(self):
        if self._h:
            for ref in self._worlds:
                w = ref()
                if w is not None:
                    w.close()
            self._worlds = []
            for c in self.contexts:
                c.close()
            lib().rtc_group_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _check(lib().rtc_group_synchronize(self._h), "rtc_group_synchronize")

    def upload(self, world: World) -> "GroupWorld":
        return GroupWorld(self, world)

    def stats(self) -> dict:
        s = RtcStats()
        _check(lib().rtc_group_stats_read(self._h, C.byref(s)), "rtc_group_stats_read")
        return _stats_dict(s)

    def reset_stats(self):
        _check(lib().rtc_group_stats_reset(self._h), "rtc_group_stats_reset")


class GroupWorld:
    """World replicated on every local member of a Group (rtc_group_world)."""

    def __init__(self, group: Group, world: World):
        self.group = group
        self._h = C.c_void_p()
        _check(lib().rtc_group_world_create(group._h, world.array(), len(world.shapes), C.byref(world.light), C.byref(self._h)),
               "rtc_group_world_create")
        group._worlds.append(weakref.ref(self))

    def close(self):
        if self._h:
            lib().rtc_group_world_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, cams, what: int = GATHER_F64, d_canvas: int | None = None, d_rgb8: int | None = None,
               mode: int = MODE_RENDER_ASYNC, flags: int = 0) -> None:
        """Enqueue one batch (<= 8 cameras of one size): every member renders its bands, the tiles are gathered to
        member 0 and un-dealt into `d_canvas` (DEVICE address on member 0's device; frames back to back)."""
        arr = cams if isinstance(cams, C.Array) else (RtcCamera * len(cams))(*cams)
        st = lib().rtc_group_render(self.group._h, self._h, arr, len(arr), mode, flags, what, d_canvas, d_rgb8)
        if st != 0:
            raise RtcError(st, "rtc_group_render")

    def render_host(self, cam: RtcCamera, out: np.ndarray, mode: int = MODE_RENDER_ASYNC, flags: int = 0, with_stats: bool = False):
        """Camera::render_async -> host Canvas: every local member DMAs its bands straight into `out`."""
        if out.shape != (cam.vsize, cam.hsize, 3) or out.dtype != np.float64 or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous (vsize, hsize, 3) float64 array")
        st = RtcStats()
        _check(lib().rtc_group_render_host(self.group._h, self._h, C.byref(cam), mode, flags, out.ctypes.data_as(C.POINTER(C.c_double)),
                                           C.byref(st) if with_stats else None), "rtc_group_render_host")
        return (out, _stats_dict(st, cam.samples != 1)) if with_stats else out

    def render_host_rgb8(self, cam: RtcCamera, out: np.ndarray, mode: int = MODE_RENDER_ASYNC, flags: int = 0, with_stats: bool = False):
        """The same for the 8-bit frame (Color::scale'd on the device; rtc_group_render_host_rgb8)."""
        if out.shape != (cam.vsize, cam.hsize, 3) or out.dtype != np.uint8 or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous (vsize, hsize, 3) uint8 array")
        st = RtcStats()
        _check(lib().rtc_group_render_host_rgb8(self.group._h, self._h, C.byref(cam), mode, flags, out.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                C.byref(st) if with_stats else None), "rtc_group_render_host_rgb8")
        return (out, _stats_dict(st, cam.samples != 1)) if with_stats else out


# ---- [host] the dealing of rows over the members of a group (csrc/rtc_bands.h through the C-ABI) ------------------
def group_packed_rows(vsize: int, nranks: int) -> int:
    return lib().rtc_group_packed_rows(vsize, nranks)


def group_bands_owned(vsize: int, nranks: int, rank: int) -> int:
    return lib().rtc_group_bands_owned(vsize, nranks, rank)


def group_row_owner(y: int, nranks: int) -> tuple[int, int]:
    m, r = C.c_uint32(), C.c_uint32()
    lib().rtc_group_row_owner(y, nranks, C.byref(m), C.byref(r))
    return m.value, r.value


def group_packed_row_to_image(member: int, packed_row: int, nranks: int) -> int:
    return lib().rtc_group_packed_row_to_image(member, packed_row, nranks)


def group_undeal_host(staging: np.ndarray, nranks: int, nframes: int, vsize: int) -> np.ndarray:
    """Member 0's un-deal step on host memory: `staging` = (nranks, nframes, packed_rows, ...row) in rank order
    (what the gather delivers) -> (nframes, vsize, ...row)."""
    st = np.ascontiguousarray(staging)
    rows = group_packed_rows(vsize, nranks)
    assert st.shape[:3] == (nranks, nframes, rows), (st.shape, (nranks, nframes, rows))
    row_bytes = int(np.prod(st.shape[3:], dtype=np.int64)) * st.itemsize
    out = np.empty((nframes, vsize) + st.shape[3:], dtype=st.dtype)
    _check(lib().rtc_group_undeal_host(C.c_void_p(st.ctypes.data), C.c_void_p(out.ctypes.data), nranks, nframes, vsize, row_bytes),
           "rtc_group_undeal_host")
    return out


__all__ = ["lib", "RtcError", "Matrix", "material", "sphere", "plane", "cube", "light", "World", "camera", "ray_for_pixel",
           "load_yaml", "load_lua", "LuaProgram", "LuaJob", "format_ppm", "write_ppm", "format_png", "write_png", "color_scale255", "Context", "DeviceWorld", "MODE_RENDER", "MODE_RENDER_ASYNC", "FLAG_NONE", "FLAG_NO_CULL", "FLAG_AA_RESAMPLE", "Group", "GroupWorld", "group_unique_id",
           "host_register", "host_unregister", "host_canvas", "host_canvas_rgb8", "format_ppm_rgb8", "write_ppm_rgb8",
           "group_packed_rows", "group_bands_owned", "group_row_owner", "group_packed_row_to_image", "group_undeal_host", "EXCHANGE_RCCL", "EXCHANGE_P2P", "GATHER_NONE", "GATHER_F64", "GATHER_U8",
           "SPHERE", "PLANE", "CUBE", "RtcCamera", "RtcHit", "RtcLight", "RtcMaterial", "RtcShape", "RtcStats"]
