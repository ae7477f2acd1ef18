"""ctypes mirror of include/rtc.h (struct layouts, constants, prototypes)."""
from __future__ import annotations

import ctypes as C

Mat16 = C.c_double * 16
Vec3 = C.c_double * 3

SPHERE, PLANE, CUBE = 0, 1, 2
MODE_RENDER, MODE_RENDER_ASYNC = 0, 1
FLAG_NONE, FLAG_NO_CULL, FLAG_AA_RESAMPLE, FLAG_LDS_TABLE = 0, 1, 2, 4
EXCHANGE_RCCL, EXCHANGE_P2P = 0, 1
GATHER_NONE, GATHER_F64, GATHER_U8 = 0, 1, 2
GROUP_ID_BYTES = 128
PATTERNS = {"none": 0, "test": 1, "stripe": 2, "stripes": 2, "gradient": 3, "ring": 4, "checker": 5, "checkers": 5, "grid": 6}
STATUS_NAMES = {0: "RTC_OK", 1: "RTC_ERR_SINGULAR", 2: "RTC_ERR_NO_COLOR", 3: "RTC_ERR_DEVICE", 4: "RTC_ERR_ARG",
                5: "RTC_ERR_PARSE", 6: "RTC_ERR_IO", 7: "RTC_ERR_NOMEM", 8: "RTC_ERR_UNSUPPORTED"}


class RtcMaterial(C.Structure):
    _fields_ = [("pattern_kind", C.c_uint32), ("has_color", C.c_uint32), ("color", Vec3),
                ("ambient", C.c_double), ("diffuse", C.c_double), ("specular", C.c_double), ("shininess", C.c_double),
                ("reflective", C.c_double), ("transparency", C.c_double), ("refractive_index", C.c_double),
                ("pat_inv", Mat16), ("pat_a", Vec3), ("pat_b", Vec3)]


class RtcShape(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("world_id", C.c_uint32), ("inv", Mat16), ("inv_t", Mat16), ("material", RtcMaterial)]


class RtcLight(C.Structure):
    _fields_ = [("intensity", Vec3), ("position", Vec3)]


class RtcCamera(C.Structure):
    _fields_ = [("hsize", C.c_uint32), ("vsize", C.c_uint32), ("fov", C.c_double), ("half_width", C.c_double),
                ("half_height", C.c_double), ("pixel_size", C.c_double), ("view_inv", Mat16), ("samples", C.c_uint32),
                ("_pad", C.c_uint32)]


class RtcStats(C.Structure):
    _fields_ = [("rays_primary", C.c_uint64), ("rays_shadow", C.c_uint64), ("rays_reflect", C.c_uint64),
                ("rays_refract", C.c_uint64), ("pixels", C.c_uint64), ("pixels_resample", C.c_uint64), ("rays_primary_proven_miss", C.c_uint64),
                ("_reserved", C.c_uint64 * 1)]


class RtcHit(C.Structure):
    _fields_ = [("hit_index", C.c_int32), ("inside", C.c_uint32), ("shadowed", C.c_uint32), ("_pad", C.c_uint32),
                ("t", C.c_double), ("point", Vec3), ("over_point", Vec3), ("under_point", Vec3), ("eyev", Vec3),
                ("normal", Vec3), ("reflectv", Vec3), ("n1", C.c_double), ("n2", C.c_double)]


class RtcLaunchInfo(C.Structure):
    _fields_ = [("source", C.c_uint32), ("reflective", C.c_uint32), ("refractive", C.c_uint32), ("binned", C.c_uint32),
                ("light_lists", C.c_uint32), ("lane", C.c_uint32), ("block", C.c_uint32), ("lds_bytes", C.c_uint32),
                ("tiles_per_workgroup", C.c_uint32), ("multi_tile_workgroups", C.c_uint32), ("_reserved", C.c_uint32 * 2)]


class RtcLuaJob(C.Structure):
    _fields_ = [("shapes", C.POINTER(RtcShape)), ("n_shapes", C.c_uint32), ("kind", C.c_uint32), ("light", RtcLight), ("camera", RtcCamera),
                ("outfile", C.c_char_p), ("animation", C.c_uint32), ("frame", C.c_uint32), ("same_world_as_previous", C.c_uint32),
                ("line", C.c_uint32)]


LUA_FRAME_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(RtcLuaJob), C.c_uint32, C.POINTER(C.c_uint8))

SOURCE_NAMES = {0: "brute force, records through the scalar cache", 1: "brute force, object table staged in LDS (one tile)",
                2: "brute force, object table staged in LDS tiles", 3: "one-level per-wave cull", 4: "two-level per-wave cull"}

assert C.sizeof(RtcMaterial) == 264 and C.sizeof(RtcShape) == 528 and C.sizeof(RtcHit) == 184

D = C.c_double
PD = C.POINTER(C.c_double)
U32 = C.c_uint32
VP = C.c_void_p

# name -> (restype, argtypes); every symbol include/rtc.h declares
PROTOTYPES = {
    "rtc_abi_version": (U32, []),
    "rtc_strerror": (C.c_char_p, [C.c_int32]),
    "rtc_matrix_identity": (None, [Mat16]),
    "rtc_matrix_multiply": (None, [Mat16, Mat16, Mat16]),
    "rtc_matrix_translation": (None, [Mat16, D, D, D, Mat16]),
    "rtc_matrix_scaling": (None, [Mat16, D, D, D, Mat16]),
    "rtc_matrix_rotation_x": (None, [Mat16, D, Mat16]),
    "rtc_matrix_rotation_y": (None, [Mat16, D, Mat16]),
    "rtc_matrix_rotation_z": (None, [Mat16, D, Mat16]),
    "rtc_matrix_shearing": (None, [Mat16, D, D, D, D, D, D, Mat16]),
    "rtc_matrix_determinant": (D, [Mat16]),
    "rtc_matrix_inverse": (C.c_int32, [Mat16, Mat16]),
    "rtc_matrix_transpose": (None, [Mat16, Mat16]),
    "rtc_view_transform": (None, [Vec3, Vec3, Vec3, Mat16]),
    "rtc_camera_init": (C.c_int32, [U32, U32, D, Mat16, C.POINTER(RtcCamera)]),
    "rtc_camera_ray_for_pixel": (None, [C.POINTER(RtcCamera), U32, D, U32, D, C.c_double * 6]),
    "rtc_material_default": (None, [C.POINTER(RtcMaterial)]),
    "rtc_shape_init": (C.c_int32, [U32, Mat16, C.POINTER(RtcMaterial), C.POINTER(RtcShape)]),
    "rtc_material_set_pattern": (C.c_int32, [C.POINTER(RtcMaterial), U32, Vec3, Vec3, Mat16]),
    "rtc_light_default": (None, [C.POINTER(RtcLight)]),
    "rtc_scene_load_yaml": (C.c_int32, [C.c_char_p, C.POINTER(C.POINTER(RtcShape)), C.POINTER(U32), C.POINTER(RtcLight),
                                        C.POINTER(RtcCamera), C.c_char_p, C.c_size_t]),
    "rtc_scene_load_yaml_file": (C.c_int32, [C.c_char_p, C.POINTER(C.POINTER(RtcShape)), C.POINTER(U32), C.POINTER(RtcLight),
                                             C.POINTER(RtcCamera), C.c_char_p, C.c_size_t]),
    "rtc_canvas_write_png8": (C.c_int32, [C.c_char_p, C.POINTER(C.c_uint8), U32, U32, U32]),
    "rtc_canvas_format_png8": (C.c_size_t, [C.POINTER(C.c_uint8), U32, U32, U32, C.POINTER(C.c_uint8), C.c_size_t]),
    "rtc_lua_run": (C.c_int32, [C.c_char_p, C.c_char_p, C.c_uint64, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]),
    "rtc_lua_run_file": (C.c_int32, [C.c_char_p, C.c_uint64, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]),
    "rtc_lua_program_jobs": (C.c_uint32, [C.c_void_p]),
    "rtc_lua_program_job": (C.c_int32, [C.c_void_p, C.c_uint32, C.POINTER(RtcLuaJob)]),
    "rtc_lua_program_output": (C.c_char_p, [C.c_void_p]),
    "rtc_lua_program_free": (None, [C.c_void_p]),
    "rtc_lua_program_render": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, LUA_FRAME_FN, C.c_void_p, C.POINTER(RtcStats)]),
    "rtc_scene_load_lua": (C.c_int32, [C.c_char_p, U32, C.POINTER(C.POINTER(RtcShape)), C.POINTER(U32), C.POINTER(RtcLight),
                                       C.POINTER(RtcCamera), C.c_char_p, C.c_size_t, C.POINTER(U32), C.c_char_p, C.c_size_t]),
    "rtc_scene_load_lua_file": (C.c_int32, [C.c_char_p, U32, C.POINTER(C.POINTER(RtcShape)), C.POINTER(U32), C.POINTER(RtcLight),
                                            C.POINTER(RtcCamera), C.c_char_p, C.c_size_t, C.POINTER(U32), C.c_char_p, C.c_size_t]),
    "rtc_free": (None, [VP]),
    "rtc_canvas_write_ppm": (C.c_int32, [C.c_char_p, PD, U32, U32]),
    "rtc_canvas_format_ppm": (C.c_size_t, [PD, U32, U32, C.c_char_p, C.c_size_t]),
    "rtc_color_scale255": (None, [PD, C.c_size_t, C.POINTER(C.c_uint8)]),
    "rtc_canvas_to_rgba8": (None, [PD, U32, U32, C.c_float, C.POINTER(C.c_uint8)]),
    "rtc_context_create": (C.c_int32, [C.c_int32, VP, C.POINTER(VP)]),
    "rtc_context_destroy": (None, [VP]),
    "rtc_context_synchronize": (C.c_int32, [VP]),
    "rtc_context_device_info": (C.c_int32, [VP, C.c_char_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "rtc_world_create": (C.c_int32, [VP, C.POINTER(RtcShape), U32, C.POINTER(RtcLight), C.POINTER(VP)]),
    "rtc_world_destroy": (None, [VP]),
    "rtc_render_rows": (C.c_int32, [VP, VP, C.POINTER(RtcCamera), U32, U32, U32, VP, VP, U32]),
    "rtc_render_bands": (C.c_int32, [VP, VP, C.POINTER(RtcCamera), U32, U32, U32, VP, VP, U32]),
    "rtc_render_views": (C.c_int32, [VP, VP, C.POINTER(RtcCamera), U32, U32, U32, U32, VP, VP, U32, U32]),
    "rtc_render": (C.c_int32, [VP, VP, C.POINTER(RtcCamera), U32, U32, PD, C.POINTER(RtcStats)]),
    "rtc_render_rgb8": (C.c_int32, [VP, VP, C.POINTER(RtcCamera), U32, U32, C.POINTER(C.c_uint8), C.POINTER(RtcStats)]),
    "rtc_canvas_write_ppm_rgb8": (C.c_int32, [C.c_char_p, C.POINTER(C.c_uint8), U32, U32]),
    "rtc_canvas_format_ppm_rgb8": (C.c_size_t, [C.POINTER(C.c_uint8), U32, U32, C.c_char_p, C.c_size_t]),
    "rtc_context_set_pipeline": (C.c_int32, [VP, U32]),
    "rtc_context_fence": (C.c_int32, [VP]),
    "rtc_context_last_launch_info": (C.c_int32, [VP, C.POINTER(RtcLaunchInfo)]),
    "rtc_host_alloc": (C.c_int32, [C.c_size_t, C.POINTER(VP)]),
    "rtc_host_free": (None, [VP]),
    "rtc_stats_read": (C.c_int32, [VP, C.POINTER(RtcStats)]),
    "rtc_stats_reset": (C.c_int32, [VP]),
    "rtc_kernel_times_ms": (C.c_int32, [VP, C.POINTER(C.c_float), U32, C.POINTER(U32)]),
    "rtc_binning_times_ms": (C.c_int32, [VP, C.POINTER(C.c_float), U32, C.POINTER(U32)]),
    "rtc_context_set_timing": (C.c_int32, [VP, U32]),
    "rtc_last_kernel_ms": (C.c_int32, [VP, C.POINTER(C.c_float)]),
    "rtc_host_register": (C.c_int32, [VP, C.c_size_t]),
    "rtc_host_unregister": (C.c_int32, [VP]),
    "rtc_group_create": (C.c_int32, [C.POINTER(C.c_int32), U32, U32, C.POINTER(VP)]),
    "rtc_group_unique_id": (C.c_int32, [C.POINTER(C.c_uint8)]),
    "rtc_group_create_rank": (C.c_int32, [C.c_int32, U32, U32, C.POINTER(C.c_uint8), C.POINTER(VP)]),
    "rtc_group_destroy": (None, [VP]),
    "rtc_group_size": (U32, [VP]),
    "rtc_group_local_size": (U32, [VP]),
    "rtc_group_context": (VP, [VP, U32]),
    "rtc_group_synchronize": (C.c_int32, [VP]),
    "rtc_group_world_create": (C.c_int32, [VP, C.POINTER(RtcShape), U32, C.POINTER(RtcLight), C.POINTER(VP)]),
    "rtc_group_world_destroy": (None, [VP]),
    "rtc_group_render": (C.c_int32, [VP, VP, C.POINTER(RtcCamera), U32, U32, U32, U32, VP, VP]),
    "rtc_group_render_host": (C.c_int32, [VP, VP, C.POINTER(RtcCamera), U32, U32, PD, C.POINTER(RtcStats)]),
    "rtc_group_render_host_rgb8": (C.c_int32, [VP, VP, C.POINTER(RtcCamera), U32, U32, C.POINTER(C.c_uint8), C.POINTER(RtcStats)]),
    "rtc_group_packed_rows": (U32, [U32, U32]),
    "rtc_group_bands_owned": (U32, [U32, U32, U32]),
    "rtc_group_row_owner": (None, [U32, U32, C.POINTER(U32), C.POINTER(U32)]),
    "rtc_group_packed_row_to_image": (U32, [U32, U32, U32]),
    "rtc_group_undeal_host": (C.c_int32, [VP, VP, U32, U32, U32, C.c_size_t]),
    "rtc_group_stats_read": (C.c_int32, [VP, C.POINTER(RtcStats)]),
    "rtc_group_stats_reset": (C.c_int32, [VP]),
    "rtc_color_at": (C.c_int32, [VP, VP, PD, U32, U32, U32, PD, C.POINTER(RtcHit)]),
    "rtc_device_arith": (C.c_int32, [VP, U32, PD, PD, U32, PD]),
}


def declare(lib: C.CDLL) -> None:
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
