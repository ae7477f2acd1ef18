/*
 * rtc.h — C-ABI of the MI355X-native renderer for the per-pixel hot path of
 * joedane/raytracer-challenge (crate `ch1`):
 *
 *     Camera::render / render_async  ->  World::color_at  ->  World::intersect
 *                                    ->  shade_hit (lighting, shadow, reflect, refract)
 *
 * The reference has no FFI of its own (plain Rust `pub` methods); every
 * entry point below cites the reference item (file:line under ch1/src/) whose job it
 * takes over, so that a Rust `extern "C"` block binding these names is mechanical
 * (the stub is shown in INTEGRATION.md).
 *
 * Conventions
 *   - plain pointers and sizes only; nothing unwinds across this boundary;
 *     every function returns an rtc_status (0 = RTC_OK) unless stated otherwise.
 *   - all arithmetic on the path is IEEE f64 (vec.rs:7-12, color.rs:5-10), evaluated in
 *     the reference's operation order with no FMA contraction.
 *   - matrices are 4x4 row-major `double[16]` (transform.rs:23-27).
 *   - a canvas is `double[height][width][3]`, row-major, idx = y*width + x
 *     (canvas.rs:16-22,43-51).
 *   - functions marked [host] never touch the GPU; functions marked [device] need a
 *     context and fail with RTC_ERR_DEVICE when no MI355X is usable. There is no CPU
 *     fallback inside this library.
 */
#ifndef RTC_H
#define RTC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTC_ABI_VERSION 3u

/* ---- status codes (the reference panics instead) -------------------- */
typedef int32_t rtc_status;
enum {
    RTC_OK             = 0,
    RTC_ERR_SINGULAR   = 1, /* Matrix::inverse on |det| <= 1e-8      transform.rs:35-38,175-177 */
    RTC_ERR_NO_COLOR   = 2, /* material with neither colour nor pattern  material.rs:328-331    */
    RTC_ERR_DEVICE     = 3, /* no usable gfx950 device / HIP runtime error                      */
    RTC_ERR_ARG        = 4, /* null pointer, zero size, row range outside the canvas ...        */
    RTC_ERR_PARSE      = 5, /* scene description rejected (lua.rs:216,322-326 analogue)         */
    RTC_ERR_IO         = 6, /* file could not be opened / written     canvas.rs:87-91           */
    RTC_ERR_NOMEM      = 7,
    RTC_ERR_UNSUPPORTED= 8  /* a build or device without what the call needs (e.g. no RCCL library)   */
};

/* ---- enumerations ------------------------------------------------------------------ */
enum { /* rtc_shape.kind                        */
    RTC_SPHERE = 0, /* shape.rs:281-394 */
    RTC_PLANE  = 1, /* shape.rs:405-498 */
    RTC_CUBE   = 2  /* shape.rs:500-630 */
};
enum { /* rtc_material.pattern_kind              */
    RTC_PATTERN_NONE     = 0,
    RTC_PATTERN_TEST     = 1, /* material.rs:48-70   */
    RTC_PATTERN_STRIPE   = 2, /* material.rs:72-104  */
    RTC_PATTERN_GRADIENT = 3, /* material.rs:106-136 */
    RTC_PATTERN_RING     = 4, /* material.rs:138-171 */
    RTC_PATTERN_CHECKER  = 5, /* material.rs:173-206 */
    RTC_PATTERN_GRID     = 6  /* material.rs:208-242 */
};
enum { /* render mode */
    RTC_MODE_RENDER       = 0, /* Camera::render: y in 0..vsize-1, x in 0..hsize-1 EXCLUSIVE,
                                  last row and column stay black        camera.rs:116-126 */
    RTC_MODE_RENDER_ASYNC = 1  /* Camera::render_async / render_async1: all pixels
                                                                        camera.rs:128-160 */
};
enum { /* render flags (bit set) */
    RTC_FLAG_NONE      = 0,
    RTC_FLAG_NO_CULL   = 1u << 0, /* visit every object for every ray (plain brute force); the
                                     default culls objects with a conservative bound first and
                                     produces bit-identical results */
    RTC_FLAG_AA_RESAMPLE = 1u << 1, /* antialiasing_samples > 1 only: take render_pixel's adaptive resample branch
                                     (camera.rs:84-92,108-111) on the device, with the counter-based offsets
                                     documented at rtc_camera.samples (the reference draws them from thread_rng).
                                     Without this flag such pixels keep the mean of the 4 fixed sub-samples and are
                                     COUNTED in rtc_stats.pixels_resample, so the caller knows how many differ.      */
    RTC_FLAG_LDS_TABLE = 1u << 2  /* with RTC_FLAG_NO_CULL: loop over the object table STAGED IN LDS by the workgroup (one tile
                                     when it fits, tiles with a barrier each otherwise) instead of fetching the records through
                                     the scalar cache — BASELINE.json north_star's literal kernel, kept for measurement
                                     (bench.py `brute_force_lds`); identical results */
};

#define RTC_MAX_REFLECTIONS 5u /* Camera::MAX_REFLECTIONS camera.rs:31 */
#define RTC_EPSILON 0.00000001 /* Vector::EPSILON         vec.rs:16    */

/* ---- flattened world --------------------------------------------------------------- */

/* Material (material.rs:244-254) with its optional pattern flattened in. */
typedef struct rtc_material {
    uint32_t pattern_kind;   /* RTC_PATTERN_*; NONE <=> Material.pattern == None           */
    uint32_t has_color;      /* Material.color.is_some()                                    */
    double   color[3];
    double   ambient, diffuse, specular, shininess;
    double   reflective;     /* Material.reflectiveness                                     */
    double   transparency;
    double   refractive_index;
    double   pat_inv[16];    /* Pattern.xf_inv (inverse of the pattern transform)           */
    double   pat_a[3];       /* color_a / color_base                                        */
    double   pat_b[3];       /* color_b / color_grid                                        */
} rtc_material;              /* 264 bytes */

/* One shape of World.shapes (shape.rs:633-637). `inv` is what the reference stores in the
 * field misleadingly called `transform` (the INVERSE of the object transform,
 * shape.rs:300,311); `inv_t` is `transform_transpose` (shape.rs:301,312). Both are carried
 * explicitly so that a caller can reproduce the stale-transpose quirk of
 * Plane::set_transform (shape.rs:446-449). */
typedef struct rtc_shape {
    uint32_t     kind;       /* RTC_SPHERE / RTC_PLANE / RTC_CUBE                           */
    uint32_t     world_id;   /* World::add_shape assigns last_world_id+1 (shape.rs:661-667);
                                only compared for equality (shape.rs:127). rtc_world_create
                                numbers the shapes 1..n itself when EVERY id is 0 (what
                                rtc_shape_init leaves), and otherwise honours the ids given:
                                two shapes with the same id are one "container" to
                                compute_refractive, exactly as in the reference. The
                                reference's own ids are u8 and wrap at 256 shapes
                                (shape.rs:287): a caller that passes `get_world_id()` as
                                it is reproduces that domain too.                           */
    double       inv[16];
    double       inv_t[16];
    rtc_material material;
} rtc_shape;                 /* 528 bytes */

/* Light (material.rs:10-31). World has exactly one (shape.rs:635). */
typedef struct rtc_light {
    double intensity[3];
    double position[3];
} rtc_light;

/* Camera (camera.rs:17-27). */
typedef struct rtc_camera {
    uint32_t hsize, vsize;
    double   fov;
    double   half_width, half_height, pixel_size;
    double   view_inv[16];   /* view_transform_inv: inverse of the view matrix camera.rs:35 */
    uint32_t samples;        /* antialiasing_samples (camera.rs:24). 1 = one ray through the pixel
                                centre. ANY other value, 0 included, takes render_pixel's second
                                branch (camera.rs:99-113): the 4 fixed sub-samples at offsets
                                (.25|.75, .25|.75) are averaged (Color::average_over), and if any of
                                them is farther than 0.01 (Euclidean RGB, Color::distance_from
                                color.rs:122-126) from that mean, `samples` MORE rays are traced,
                                appended to the same list and the mean is taken again (resample,
                                camera.rs:84-92). The trigger is deterministic and reproduced
                                exactly; the reference draws the extra offsets from thread_rng
                                (non-reproducible), so the resample itself is only taken with
                                RTC_FLAG_AA_RESAMPLE, with offsets from a documented counter-based
                                generator: for pixel (x, y) and extra sample k = 0..samples-1,
                                  c = (uint64(y)*hsize + x) << 16
                                  x_offset = (splitmix64(c | 2k)   >> 11) * 2^-53
                                  y_offset = (splitmix64(c | 2k+1) >> 11) * 2^-53
                                with splitmix64(z): z += 0x9E3779B97F4A7C15; z = (z ^ z>>30) *
                                0xBF58476D1CE4E5B9; z = (z ^ z>>27) * 0x94D049BB133111EB; z ^ z>>31
                                (uniform in [0,1) like rng.gen::<f64>()). samples == 0 resamples
                                zero rays: the mean of the four is recomputed and is the result.
                                Values above 255 are rejected (the reference's field is a u8).       */
    uint32_t _pad;
} rtc_camera;

/* Ray counters. A ray = one call of World::intersect (shape.rs:677). */
typedef struct rtc_stats {
    uint64_t rays_primary;   /* color_at from render_pixel          camera.rs:97-105   */
    uint64_t rays_shadow;    /* is_shadowed, one per shade_hit      shape.rs:688,712   */
    uint64_t rays_reflect;   /* reflected_color recursion           shape.rs:734-735   */
    uint64_t rays_refract;   /* refracted_color recursion           shape.rs:764-765   */
    uint64_t pixels;         /* pixels written by the last render                        */
    uint64_t pixels_resample;/* pixels whose 4 sub-samples trip the resample test
                                (camera.rs:108-111); 0 when samples == 1                 */
    uint64_t rays_primary_proven_miss; /* of rays_primary: primary rays of image tiles the binning kernel PROVED to hit nothing
                                (empty candidate list, cone clear of every plane): counted as cast — the reference casts
                                them — but answered by the proof, no ray is generated for them (one-sample renders of
                                binned launches; 0 otherwise)                                  */
    uint64_t _reserved[1];
} rtc_stats;

/* Per-ray probe record filled by rtc_color_at: the fields of CachedVectors
 * (shape.rs:58-71) for the ray's FIRST hit. hit_index = position in World.shapes
 * (insertion order) or -1. */
typedef struct rtc_hit {
    int32_t  hit_index;
    uint32_t inside;
    uint32_t shadowed;       /* World::is_shadowed(over_point)      shape.rs:712-727   */
    uint32_t _pad;
    double   t;
    double   point[3];
    double   over_point[3];
    double   under_point[3];
    double   eyev[3];
    double   normal[3];
    double   reflectv[3];
    double   n1, n2;         /* compute_refractive; 1.0/1.0 unless the hit material has
                                transparency != 0 (the only case the reference reads them,
                                shape.rs:692,752)                                       */
} rtc_hit;

typedef struct rtc_context rtc_context; /* one GPU + one stream; single-threaded use (N GPUs: rtc_group) */
typedef struct rtc_world   rtc_world;   /* flattened World resident in HBM                */

/* ==== [host] reference-faithful setup arithmetic =================================== */

/* ABI / build identification. Returns RTC_ABI_VERSION. */
uint32_t    rtc_abi_version(void);
const char *rtc_strerror(rtc_status s);

/* Matrix::identity / multiply (transform.rs:44-51, 8-21,31-33). out may alias neither input. */
void        rtc_matrix_identity(double out[16]);
void        rtc_matrix_multiply(const double a[16], const double b[16], double out[16]);
/* Fluent builders: each LEFT-multiplies, `out = new * m` (transform.rs:53-105). out may alias m. */
void        rtc_matrix_translation(const double m[16], double x, double y, double z, double out[16]);
void        rtc_matrix_scaling    (const double m[16], double x, double y, double z, double out[16]);
void        rtc_matrix_rotation_x (const double m[16], double r, double out[16]);
void        rtc_matrix_rotation_y (const double m[16], double r, double out[16]);
void        rtc_matrix_rotation_z (const double m[16], double r, double out[16]);
void        rtc_matrix_shearing   (const double m[16], double xy, double xz, double yx, double yz,
                                   double zx, double zy, double out[16]);
/* Matrix::determinant by cofactor expansion along row 0 (transform.rs:130-169). */
double      rtc_matrix_determinant(const double m[16]);
/* Matrix::inverse, cofactor method; RTC_ERR_SINGULAR when |det| <= 1e-8 (transform.rs:35-38,175-190). */
rtc_status  rtc_matrix_inverse(const double m[16], double out[16]);
void        rtc_matrix_transpose(const double m[16], double out[16]); /* transform.rs:192-202 */
/* Matrix::make_view_transform(from, to, up) (transform.rs:204-217). */
void        rtc_view_transform(const double from[3], const double to[3], const double up[3], double out[16]);

/* Camera::new_with_transform(hsize, vsize, fov, view) (camera.rs:33-58): derives
 * half_width/half_height/pixel_size and stores inverse(view). samples = 1. */
rtc_status  rtc_camera_init(uint32_t hsize, uint32_t vsize, double fov, const double view[16], rtc_camera *out);
/* Camera::ray_for_pixel_offset (camera.rs:64-76); ray = {origin xyz, direction xyz}. */
void        rtc_camera_ray_for_pixel(const rtc_camera *cam, uint32_t x, double x_offset,
                                     uint32_t y, double y_offset, double ray[6]);

/* Material::default() (white, .1/.9/.9/200, 0/0/1.0; material.rs:273-283,364-369). */
void        rtc_material_default(rtc_material *out);
/* {Sphere,Plane,Cube}::new_with_transform_and_material(m, mat) (shape.rs:308-317,436-444,
 * 525-533): inv = m.inverse(), inv_t = inv.transpose(). world_id is left 0. */
rtc_status  rtc_shape_init(uint32_t kind, const double transform[16], const rtc_material *mat, rtc_shape *out);
/* Pattern::set_transform (material.rs:61-63 etc.): mat->pat_inv = inverse(transform). */
rtc_status  rtc_material_set_pattern(rtc_material *mat, uint32_t pattern_kind, const double a[3],
                                     const double b[3], const double transform[16]);
/* Light::default(): white at (-10,10,-10) (material.rs:26-31). */
void        rtc_light_default(rtc_light *out);

/* Scene loader for the `jamis.yml` vocabulary (ch1/jamis.yml:1-183; the reference ships the
 * data file but no loader). Parses `text` (NUL-terminated YAML subset),
 * applies transform lists in listed order (transform.rs:53-69 left-multiplication), starts
 * materials from Material::default() (lua.rs:187) and assigns world ids like
 * World::add_shape. On success *shapes_out is a malloc'ed array the caller releases with
 * rtc_free. Only the first light is used (lua.rs:148-150). */
rtc_status  rtc_scene_load_yaml(const char *text, rtc_shape **shapes_out, uint32_t *n_out,
                                rtc_light *light_out, rtc_camera *camera_out,
                                char *errbuf, size_t errbuf_len);
rtc_status  rtc_scene_load_yaml_file(const char *path, rtc_shape **shapes_out, uint32_t *n_out,
                                     rtc_light *light_out, rtc_camera *camera_out,
                                     char *errbuf, size_t errbuf_len);
/* The reference's Lua front-end (ch1/src/lua.rs:50-330; scripts ch1/jamis.lua, ex1.lua, ex2.lua + functions.lua) without a
 * Lua library: csrc/host_lua.cpp interprets the part of Lua 5.3 those scripts are written in (functions and closures,
 * numeric / generic for, while, repeat, if, multiple assignment, tables, integer / float numbers, strings; print, require of
 * files beside the script, math.*, string.format, table.insert ...; no metatables, coroutines, goto or bitwise operators)
 * and gives a script the reference's three entry points:
 *   Render(world, camera, outfile)                              lua.rs:57-71
 *   enc = StartAnimation(outfile); enc:AddFrame(world, camera); enc:Finish()     lua.rs:34-45,75-79
 * Every Render / AddFrame call converts its tables AT THE CALL by lua.rs's own rules — transform_from_table's fixed order
 * rotate_x, rotate_y, rotate_z, scale, position; materials from Material::default() with the keys ambient, diffuse,
 * specular, shininess, reflectiveness, transparency, refractive_index, color, pattern (anything else is an error) and the
 * shape-level color / pattern override; patterns "checks" / "stripes" / "grid"; lights[1] only; camera screenwidth /
 * screenheight / samples as Lua integers — and becomes one JOB: a world, a camera, the output file's name. The caller renders
 * the jobs in order (one rtc_render* launch each: an AddFrame loop is the one-camera-per-launch sequence a pipelined
 * context overlaps); encoding GIF / PNG / JPEG files is the `image` crate's business in the reference and nobody's here.
 * math.random is Lua 5.3's on POSIX (glibc random(), restated), so `math.randomseed(13)` worlds are reproducible.
 * A script runs under a step budget (`step_limit` statements / loop iterations / calls, 0 = 100 000 000), may nest 200 calls
 * (the interpreter recurses on the caller's stack: up to about 2 MB of it at that depth) and cannot touch
 * the file system except through require (`base_dir`/name.lua; NULL = require is an error). Syntax and runtime errors,
 * and tables lua.rs would reject, are RTC_ERR_PARSE with the message in errbuf (the reference unwrap()s: it panics).
 * PARITY UNPINNED: the reference has no test of its Lua path. [host] */
typedef struct rtc_lua_program rtc_lua_program;
enum { RTC_LUA_JOB_RENDER = 0, RTC_LUA_JOB_ADD_FRAME = 1 };
typedef struct rtc_lua_job {
    const rtc_shape *shapes;          /* owned by the program; NULL when the world is empty                            */
    uint32_t         n_shapes;
    uint32_t         kind;            /* RTC_LUA_JOB_RENDER | RTC_LUA_JOB_ADD_FRAME                                    */
    rtc_light        light;
    rtc_camera       camera;
    const char      *outfile;         /* Render's third argument / the animation's file name; owned by the program     */
    uint32_t         animation;       /* AddFrame: which StartAnimation call (0, 1, ...) the encoder came from         */
    uint32_t         frame;           /* AddFrame: index of the frame inside that animation                            */
    uint32_t         same_world_as_previous; /* 1: shapes and light equal the previous job's byte for byte (same arrays) */
    uint32_t         line;            /* script line of the call                                                      */
} rtc_lua_job;
rtc_status  rtc_lua_run(const char *text, const char *base_dir, uint64_t step_limit, rtc_lua_program **out,
                        char *errbuf, size_t errbuf_len);
rtc_status  rtc_lua_run_file(const char *path, uint64_t step_limit, rtc_lua_program **out, char *errbuf, size_t errbuf_len);
uint32_t    rtc_lua_program_jobs(const rtc_lua_program *prog);
rtc_status  rtc_lua_program_job(const rtc_lua_program *prog, uint32_t index, rtc_lua_job *job);
const char *rtc_lua_program_output(const rtc_lua_program *prog);   /* everything the script print()ed */
void        rtc_lua_program_free(rtc_lua_program *prog);
/* The single-scene form: runs the script as above and hands out job `render_index` (0 = the first Render / AddFrame call;
 * a script that makes none falls back to its globals `world` and `camera`) as a malloc'ed shape array (rtc_free);
 * *renders_out (may be NULL) = how many jobs the script made; `outfile` (may be NULL) receives that job's file name. */
rtc_status  rtc_scene_load_lua(const char *text, uint32_t render_index, rtc_shape **shapes_out, uint32_t *n_out,
                               rtc_light *light_out, rtc_camera *camera_out, char *outfile, size_t outfile_len,
                               uint32_t *renders_out, char *errbuf, size_t errbuf_len);
rtc_status  rtc_scene_load_lua_file(const char *path, uint32_t render_index, rtc_shape **shapes_out, uint32_t *n_out,
                                    rtc_light *light_out, rtc_camera *camera_out, char *outfile, size_t outfile_len,
                                    uint32_t *renders_out, char *errbuf, size_t errbuf_len);
void        rtc_free(void *p);

/* Canvas::write_to_file_simple: ASCII PPM P3 (canvas.rs:86-109) with Color::scale's
 * truncating, saturating cast and clamp (color.rs:100-114). `rgb` is a host canvas. */
rtc_status  rtc_canvas_write_ppm(const char *path, const double *rgb, uint32_t width, uint32_t height);
/* The same encoder into memory: returns bytes needed (excluding NUL); writes at most cap. */
size_t      rtc_canvas_format_ppm(const double *rgb, uint32_t width, uint32_t height, char *buf, size_t cap);
/* The same file from a frame that is ALREADY quantised — `rgb8` = height*width*3 bytes, each Color::scale(c, 255), as
 * rtc_render_rgb8 / rtc_render_rows' d_rgb8 deliver it: byte for byte the file rtc_canvas_write_ppm writes for the f64
 * canvas of the same render (canvas.rs:98-104 quantises with the same function). */
rtc_status  rtc_canvas_write_ppm_rgb8(const char *path, const uint8_t *rgb8, uint32_t width, uint32_t height);
size_t      rtc_canvas_format_ppm_rgb8(const uint8_t *rgb8, uint32_t width, uint32_t height, char *buf, size_t cap);
/* Color::scale(c, 255) for n colour components (color.rs:100-114) on the host. */
void        rtc_color_scale255(const double *components, size_t n, uint8_t *out);
/* Canvas::to_imgbuf (canvas.rs:61-79), the pixel buffer behind write_to_file / frame_to_file:
 * RGBA8, each channel Color::scale(c.powf(1/gamma), 255) (color.rs:55-65; the reciprocal taken in
 * f32 as the reference does), alpha 255. Canvas::new sets gamma = 1.0 (canvas.rs:30). `out` holds
 * width*height*4 bytes. Host. */
void        rtc_canvas_to_rgba8(const double *rgb, uint32_t width, uint32_t height, float gamma, uint8_t *out);
/* Canvas::write_to_file for a ".png" name (canvas.rs:80-84: to_imgbuf().save(path); the `image` crate encodes by
 * extension): an 8-bit PNG of `pixels` = height*width*channels bytes, channels = 4 (to_imgbuf's RGBA, colour type 6) or 3
 * (the device's Color::scale frame, colour type 2; a decoder supplies alpha 255, which is what to_imgbuf stores). PNG is
 * lossless: decoding gives back exactly these pixels, as it does for the reference's file. No compressor is built in —
 * the zlib stream uses stored blocks, so the file is a little larger than the pixels. JPEG and GIF (lossy / palette
 * quantising third-party codecs in the reference) are not rebuilt. format: bytes needed; writes at most cap. Host. */
rtc_status  rtc_canvas_write_png8(const char *path, const uint8_t *pixels, uint32_t width, uint32_t height, uint32_t channels);
size_t      rtc_canvas_format_png8(const uint8_t *pixels, uint32_t width, uint32_t height, uint32_t channels, uint8_t *buf, size_t cap);

/* ==== [device] the hot path on one MI355X ========================================== */

/* Create a context on HIP device `device`. `stream` is an existing hipStream_t passed as
 * void* (e.g. torch's current stream); NULL is the device's default stream. The context never
 * owns the stream. */
rtc_status  rtc_context_create(int32_t device, void *stream, rtc_context **out);
void        rtc_context_destroy(rtc_context *ctx);
rtc_status  rtc_context_synchronize(rtc_context *ctx);
/* Device facts for reports: name (<= cap bytes), compute units, clock MHz. */
rtc_status  rtc_context_device_info(rtc_context *ctx, char *name, size_t cap,
                                    int32_t *compute_units, int32_t *clock_mhz);

/* World::new(light) + add_shape* (shape.rs:642-667): flatten and upload once; the world
 * stays resident in HBM across renders. Validates materials (RTC_ERR_NO_COLOR). */
rtc_status  rtc_world_create(rtc_context *ctx, const rtc_shape *shapes, uint32_t n_shapes,
                             const rtc_light *light, rtc_world **out);
void        rtc_world_destroy(rtc_world *w);

/* Camera::render / render_async for canvas rows [y0, y1) into a DEVICE buffer of
 * (y1-y0)*hsize*3 doubles (row y0 first). Enqueues on the context stream and returns
 * without synchronising. Row-tiling hook for multi-GPU (each rank renders its rows).
 * d_rgb8 (may be NULL): additionally receives the same rows quantised to 8 bits per channel,
 * (y1-y0)*hsize*3 bytes, exactly as the reference's file writers quantise a Canvas
 * (Color::scale(c, 255): truncating saturating cast, clamp; color.rs:100-114, canvas.rs:104).
 * d_rgb may be NULL when d_rgb8 is not: then only the 8-bit rows are written (24 B/pixel of HBM
 * writes less); the same holds for rtc_render_bands and rtc_render_views. */
rtc_status  rtc_render_rows(rtc_context *ctx, const rtc_world *w, const rtc_camera *cam,
                            uint32_t mode, uint32_t y0, uint32_t y1, void *d_rgb, void *d_rgb8,
                            uint32_t flags);
/* Interleaved row tiles (multi-GPU load balance). The canvas is cut into bands of RTC_BAND_ROWS
 * rows (band b = rows [8b, 8b+8) of the image, the last one possibly short); this call renders
 * bands first_band, first_band + band_stride, first_band + 2*band_stride, ... and packs them one
 * after the other into d_rgb / d_rgb8 (the k-th band of this call at rows [8k, 8k+8) of the
 * buffer; buffers hold ceil((nbands - first_band) / band_stride) * 8 rows). With one process per
 * GPU, rank r of N calls (first_band = r, band_stride = N): every rank gets an even share of sky,
 * floor and objects, where contiguous ranges of rows (rtc_render_rows) leave the ranks with the
 * sky idle. Same per-pixel contract as rtc_render_rows (camera.rs:151-156). [device] */
#define RTC_BAND_ROWS 8u
rtc_status  rtc_render_bands(rtc_context *ctx, const rtc_world *w, const rtc_camera *cam,
                             uint32_t mode, uint32_t first_band, uint32_t band_stride,
                             void *d_rgb, void *d_rgb8, uint32_t flags);
/* Several cameras, one World, ONE launch: the frames of a camera move over a static scene (the
 * reference's AddFrame loop orbits the camera, lua.rs / functions.lua:3-11), a stereo pair, or — with
 * one process per GPU — several frames' worth of one rank's bands, so that a launch fills the chip
 * even when a rank owns an eighth of the image. `cams[0..nviews)` must agree in hsize, vsize and
 * samples; nviews <= RTC_MAX_VIEWS_PER_LAUNCH. Rows are selected as in rtc_render_bands
 * (first_band = 0, band_stride = 1 for whole frames); view v is written `v * view_rows` rows
 * below view 0 in d_rgb / d_rgb8 (view_rows >= the rows one view produces). Per pixel the result is
 * exactly that of rtc_render_bands with the same camera. [device] */
#define RTC_MAX_VIEWS_PER_LAUNCH 8u
rtc_status  rtc_render_views(rtc_context *ctx, const rtc_world *w, const rtc_camera *cams, uint32_t nviews,
                             uint32_t mode, uint32_t first_band, uint32_t band_stride,
                             void *d_rgb, void *d_rgb8, uint32_t view_rows, uint32_t flags);
/* Camera::render(&World) -> Canvas with host memory: renders all rows and copies the
 * canvas into `rgb` (vsize*hsize*3 doubles). Synchronous. `stats` may be NULL. */
rtc_status  rtc_render(rtc_context *ctx, const rtc_world *w, const rtc_camera *cam,
                       uint32_t mode, uint32_t flags, double *rgb, rtc_stats *stats);
/* Camera::render(&World) for a caller that only WRITES THE IMAGE (every file writer of the reference consumes
 * Color::scale'd bytes and nothing else: PPM canvas.rs:86-109, to_imgbuf canvas.rs:61-79 with gamma 1): renders all
 * rows and copies only the 8-bit frame — vsize*hsize*3 bytes, Color::scale(c, 255) of every component, evaluated on
 * the device bit-exactly (color.rs:100-114) — into `rgb8`: 3 bytes per pixel cross PCIe instead of 24, and the f64
 * canvas is not even written to HBM. Feed it to rtc_canvas_write_ppm_rgb8. Synchronous. `stats` may be NULL. */
rtc_status  rtc_render_rgb8(rtc_context *ctx, const rtc_world *w, const rtc_camera *cam,
                            uint32_t mode, uint32_t flags, uint8_t *rgb8, rtc_stats *stats);
/* render_lua (lua.rs:50-91) for an interpreted script: renders every job of `prog` in order — one launch per Render /
 * AddFrame call, 8-bit rows only (what the reference's file writers consume), a new device World whenever a job's world
 * differs from the previous one — and hands each frame (vsize*hsize*3 bytes, Color::scale, valid during the call only) to
 * `fn` in job order. The launches are pipelined over the context's lanes (depth 3 for the duration when the context is in
 * order) with each frame's copy to the host on its own lane: an AddFrame loop is the one-camera-per-launch sequence of
 * Camera::render_async calls. A non-zero return from `fn` stops the rendering (RTC_OK). `stats` (may be NULL) = the ray
 * counts of all frames. Blocking. [device] */
typedef int (*rtc_lua_frame_fn)(void *user, const rtc_lua_job *job, uint32_t job_index, const uint8_t *rgb8);
rtc_status  rtc_lua_program_render(rtc_context *ctx, const rtc_lua_program *prog, uint32_t mode, uint32_t flags,
                                   rtc_lua_frame_fn fn, void *user, rtc_stats *stats);
/* Page-locked host memory for canvases handed to rtc_render: a canvas from rtc_host_alloc is
 * filled by one DMA at link speed, ordinary (pageable) memory goes through the runtime's bounce
 * buffers and is several times slower. What the reference would use for Canvas.pixels
 * (canvas.rs:16-22) when it renders every frame through this library. [device] */
rtc_status  rtc_host_alloc(size_t bytes, void **out);
void        rtc_host_free(void *p);
/* Ray counters accumulated since the last reset (synchronises the stream). */
rtc_status  rtc_stats_read(rtc_context *ctx, rtc_stats *out);
rtc_status  rtc_stats_reset(rtc_context *ctx);
/* Kernel timing. Every render launch (rtc_render_rows / _bands / _views) carries its own pair of HIP events that receive
 * the dispatch's begin and end timestamps on the context stream (hipExtLaunchKernel: no marker
 * packets, the same quantity rocprofv3's kernel trace reports FOR THE RENDER KERNEL k_trace; a launch's binning
 * kernel is timed separately, rtc_binning_times_ms) unless rtc_context_set_timing says
 * otherwise. The context keeps the most recent 1024 pairs. rtc_kernel_times_ms writes the
 * durations (ms) of the latest min(cap, kept) launches, oldest first, and their number to *n; rtc_last_kernel_ms is the newest one alone
 * (RTC_ERR_ARG if nothing was launched yet). Both wait for the newest launch to finish. */
rtc_status  rtc_kernel_times_ms(rtc_context *ctx, float *out, uint32_t cap, uint32_t *n);
/* The same ring for the launches' BINNING kernels (k_bin_tiles: per-tile candidate lists, built per launch for large
 * launches; 0 for a launch that had none): out[k] belongs to the same launch as rtc_kernel_times_ms' out[k]. The render
 * kernel's time does not include it — in order on one stream it runs beside the PREVIOUS launch's render kernel, in a
 * pipelined context beside the other lanes'. */
rtc_status  rtc_binning_times_ms(rtc_context *ctx, float *out, uint32_t cap, uint32_t *n);
/* Which launches carry an event pair: every `every`-th one (1 = all, the default; 0 = none). The
 * events cost about 9 us of host time and 5 us of GPU time per launch, which matters to callers
 * that issue many short launches (one rank's share of a frame); they sample instead. Also forgets
 * the pairs recorded so far: the next launch is the first of a new series. */
rtc_status  rtc_context_set_timing(rtc_context *ctx, uint32_t every);
rtc_status  rtc_last_kernel_ms(rtc_context *ctx, float *ms);

/* Pipelined launches. Camera::render_async returns a NEW Canvas per call (camera.rs:144-160, Canvas::new canvas.rs:26-41),
 * so the frames of a render loop never alias and nothing orders frame i+1 behind frame i except the caller's own use of
 * the result. depth = 1 (the default): every render launch goes, in order, to the context's stream. depth = 2..4: the
 * context deals consecutive launches (rtc_render_rows / _bands / _views) round-robin over `depth` streams of its own, so
 * launch i+1 starts on the CUs that launch i's last waves leave idle (a lone 1080p launch spends a fifth of its time
 * draining) and its per-launch binning kernel runs beside launch i's render. CONTRACT in this mode: the output buffers of
 * `depth` consecutive launches must not overlap; launches are NOT ordered against work on the stream passed to
 * rtc_context_create — the results are complete after rtc_context_synchronize (or rtc_stats_read), and
 * rtc_context_fence makes that stream wait for every launch enqueued so far without blocking the host.
 * Per pixel the results are those of depth 1, bit for bit. Synchronises before switching. [device] */
rtc_status  rtc_context_set_pipeline(rtc_context *ctx, uint32_t depth);
rtc_status  rtc_context_fence(rtc_context *ctx);
/* What the most recent render launch of this context ran with (reports; the choice depends on the World's size, the
 * launch's size and — for A/B runs only — on the RTC_* environment switches read at rtc_context_create). */
typedef struct rtc_launch_info {
    uint32_t source;      /* object loop: 0 brute force through the scalar cache, 1 brute force over ONE LDS-staged table,
                             2 brute force over LDS tiles, 3 one-level per-wave cull, 4 two-level cull              */
    uint32_t reflective;  /* frame-stack kernel (World has reflective materials)                                    */
    uint32_t refractive;  /* ... with refraction frames                                                             */
    uint32_t binned;      /* primary pass reads per-tile candidate lists built by the launch's binning kernel       */
    uint32_t light_lists; /* shadow pass may use the World's light-space lists                                      */
    uint32_t lane;        /* pipeline lane the launch went to (0 when depth = 1)                                    */
    uint32_t block;       /* threads per workgroup                                                                  */
    uint32_t lds_bytes;   /* dynamic LDS per workgroup (LDS-staged object tables, AA sample store)                  */
    uint32_t tiles_per_workgroup; /* tiles one workgroup renders in sequence (1 unless RTC_TILES_PER_WG says otherwise)   */
    uint32_t multi_tile_workgroups; /* guided chunks: of a launch of several rounds of workgroups the FIRST ones render four, three,
                                     then two tiles each (a tile's stores drain under the next one) and the launch ends with
                                     single-tile workgroups (a short tail); this many render more than one. 0: none
                                     (RTC_TILES_GUIDED, RTC_TILES_KMAX)                                                  */
    uint32_t _reserved[2];
} rtc_launch_info;
rtc_status  rtc_context_last_launch_info(rtc_context *ctx, rtc_launch_info *out);

/* Page-lock a canvas the CALLER allocated (a Rust `Vec<Color>`, canvas.rs:16-22: 24 bytes per pixel,
 * the layout rtc_render writes) so that rtc_render / rtc_group_render_host fill it by DMA at link speed
 * instead of through the runtime's bounce buffers. Unregister before the memory is freed. [device] */
rtc_status  rtc_host_register(void *p, size_t bytes);
rtc_status  rtc_host_unregister(void *p);

/* ==== [device] row tiles across the GPUs of one node ================================ */
/* Camera::render_async shards over pixels with no data dependency (camera.rs:144-160: every pixel is an
 * independent work item of the rayon pool). A group is N GPUs that render one frame together: member r
 * renders the 8-row bands r, r+N, r+2N, ... (rtc_render_bands), then ONE exchange step — an RCCL gather
 * of the f64 tiles to member 0 over xGMI (ncclGather) — and one un-deal kernel on member 0's device puts
 * the bands back in the reference's row-major Canvas (canvas.rs:43-51). The World is
 * replicated: its tables are 0.7 KB per object (7 MB for 10 000 objects), plus — per member — the light-space shadow
 * lists of a World above 256 objects (~50 MB: 6 x 128^2 direction cells of 128 entries) and, allocated by the first binned
 * launch, two sets of per-tile candidate lists (8.4 MB per 1080p view, up to 8 views per set while a set stays within 128 MB).
 * Both kinds of lists are optimisations: when their allocation fails the launch renders through the walk instead. Two ways to form a group:
 *   rtc_group_create       one process drives all `ndev` devices (ncclCommInitAll) — what a Rust host
 *                          calling Camera::render_async would use;
 *   rtc_group_create_rank  one process per GPU (torchrun / MPI style): every process creates its member
 *                          with the same 128-byte id (ncclGetUniqueId on rank 0, shipped by the caller's
 *                          launcher) — ncclCommInitRank.
 * Each member owns two HIP streams (render, exchange) and two tile buffers: the gather of frame j
 * overlaps the render of frame j+1. Calls enqueue and return; rtc_group_synchronize waits.
 * A group is used from one thread at a time. */
typedef struct rtc_group       rtc_group;
typedef struct rtc_group_world rtc_group_world;
#define RTC_GROUP_ID_BYTES 128u
enum { /* rtc_group_create / rtc_group_create_rank `exchange` */
    RTC_EXCHANGE_RCCL = 0, /* ncclGather of the f64 tiles (and of the 8-bit tiles when asked for) to member 0     */
    RTC_EXCHANGE_P2P  = 1  /* in-process groups only: hipMemcpyPeerAsync of each tile into member 0's staging
                              buffer (SDMA engines over xGMI, no CUs taken from the render); also the only
                              exchange that accepts the same device more than once (rehearsal on a 1-GPU box) */
};
rtc_status  rtc_group_create(const int32_t *devices, uint32_t ndev, uint32_t exchange, rtc_group **out);
rtc_status  rtc_group_unique_id(uint8_t id[RTC_GROUP_ID_BYTES]);
rtc_status  rtc_group_create_rank(int32_t device, uint32_t nranks, uint32_t rank, const uint8_t id[RTC_GROUP_ID_BYTES],
                                  rtc_group **out);
void        rtc_group_destroy(rtc_group *g);
uint32_t    rtc_group_size(const rtc_group *g);        /* N: members of the whole group                    */
uint32_t    rtc_group_local_size(const rtc_group *g);  /* members this process drives (N, or 1)            */
/* The i-th local member's context (stats, kernel timing, device info); owned by the group. */
rtc_context *rtc_group_context(rtc_group *g, uint32_t i);
rtc_status  rtc_group_synchronize(rtc_group *g);
/* World::new + add_shape on every local member (replicated upload). */
rtc_status  rtc_group_world_create(rtc_group *g, const rtc_shape *shapes, uint32_t n_shapes, const rtc_light *light,
                                   rtc_group_world **out);
void        rtc_group_world_destroy(rtc_group_world *w);
/* Camera::render_async(&World) -> Canvas on all members, `nframes` (<= RTC_MAX_VIEWS_PER_LAUNCH) cameras of
 * one size per call (one launch per member, as rtc_render_views). d_canvas: DEVICE memory on member 0's
 * device, nframes consecutive canvases of vsize*hsize*3 doubles (frame f at f*vsize*hsize*3); required in
 * the process that drives member 0, ignored elsewhere. d_rgb8 (may be NULL): the same frames quantised by
 * Color::scale(c, 255), nframes*vsize*hsize*3 bytes. `what` (the same value in every process of the group)
 * selects the exchange payload, a bit set: */
enum {
    RTC_GATHER_NONE = 0u, /* render only: the tiles stay on their GPUs                                         */
    RTC_GATHER_F64  = 1u, /* the f64 Canvas (24 B/pixel), the path's own output -> d_canvas                     */
    RTC_GATHER_U8   = 2u  /* the 8-bit frame (3 B/pixel, what every file writer of the reference consumes,
                             canvas.rs:98-104) -> d_rgb8; RTC_GATHER_F64 | RTC_GATHER_U8 delivers both         */
};
rtc_status  rtc_group_render(rtc_group *g, const rtc_group_world *w, const rtc_camera *cams, uint32_t nframes,
                             uint32_t mode, uint32_t flags, uint32_t what, void *d_canvas, void *d_rgb8);
/* Camera::render_async(&World) -> Canvas in HOST memory: every local member renders its bands and DMAs them
 * straight to their rows of `rgb` over its own PCIe link (no gather: N links fill the canvas side by side).
 * `rgb` = vsize*hsize*3 doubles, ideally page-locked (rtc_host_alloc / rtc_host_register); with one process
 * per GPU it is each process's mapping of one shared-memory canvas. Synchronous for the local members.
 * `stats` (may be NULL) = the local members' ray counters for this frame. */
rtc_status  rtc_group_render_host(rtc_group *g, const rtc_group_world *w, const rtc_camera *cam, uint32_t mode,
                                  uint32_t flags, double *rgb, rtc_stats *stats);
/* The same for the 8-bit frame (rtc_render_rgb8 across the group): `rgb8` = vsize*hsize*3 bytes. */
rtc_status  rtc_group_render_host_rgb8(rtc_group *g, const rtc_group_world *w, const rtc_camera *cam, uint32_t mode,
                                       uint32_t flags, uint8_t *rgb8, rtc_stats *stats);
/* [host] The dealing of a frame's rows over the members (csrc/rtc_bands.h — the one definition rtc_group's tile sizes,
 * gather layout, host-canvas offsets and the un-deal kernel all use), for callers that lay out their own buffers:
 *   rtc_group_packed_rows          rows of one member's packed tile (= of one gather chunk per frame)
 *   rtc_group_bands_owned          bands member `rank` renders: rank, rank + nranks, ...
 *   rtc_group_row_owner            image row y -> (member, row inside its packed tile)
 *   rtc_group_packed_row_to_image  the inverse (results >= vsize are padding)
 *   rtc_group_undeal_host          what member 0's un-deal kernel does, on host memory: `staging` = nranks chunks of
 *                                  [nframes][packed_rows][row_bytes] in rank order (the gather's receive buffer) ->
 *                                  `canvas` = nframes row-major frames of vsize rows. None of these touches a GPU. */
uint32_t    rtc_group_packed_rows(uint32_t vsize, uint32_t nranks);
uint32_t    rtc_group_bands_owned(uint32_t vsize, uint32_t nranks, uint32_t rank);
void        rtc_group_row_owner(uint32_t y, uint32_t nranks, uint32_t *member, uint32_t *packed_row);
uint32_t    rtc_group_packed_row_to_image(uint32_t member, uint32_t packed_row, uint32_t nranks);
rtc_status  rtc_group_undeal_host(const void *staging, void *canvas, uint32_t nranks, uint32_t nframes, uint32_t vsize,
                                  size_t row_bytes);
/* Ray counters of the local members, summed (per member: rtc_stats_read on rtc_group_context). */
rtc_status  rtc_group_stats_read(rtc_group *g, rtc_stats *out);
rtc_status  rtc_group_stats_reset(rtc_group *g);

/* World::color_at(ray, remaining) (shape.rs:702-710) for `n` arbitrary host rays
 * (n x {origin xyz, direction xyz}); writes n x rgb and, if hits != NULL, the hit record
 * of each ray's first intersection. remaining <= RTC_MAX_REFLECTIONS (what Camera::render_pixel passes,
 * camera.rs:98; the kernels' frame stack holds that many suspended shade_hit calls). Synchronous. */
rtc_status  rtc_color_at(rtc_context *ctx, const rtc_world *w, const double *rays, uint32_t n,
                         uint32_t remaining, uint32_t flags, double *rgb, rtc_hit *hits);

/* Device arithmetic probe: applies op (0 sqrt, 1 a/b, 2 pow(a,b), 3 floor, 4 fmod(a,2))
 * element-wise on the GPU; used by the tests to prove f64 sqrt and division are correctly
 * rounded on gfx950 (they must be bit-identical to the host's). op 5 / 6: Vector::normalize (vec.rs:65-76) of
 * every consecutive triple of `a` (n a multiple of 3) — 6 with three IEEE divisions, 5 with the shared-divisor form
 * of the division expansion (rtc_kernels.hip vnormalize_shared; an experiment, must equal 6 bit for bit). */
rtc_status  rtc_device_arith(rtc_context *ctx, uint32_t op, const double *a, const double *b,
                             uint32_t n, double *out);

#ifdef __cplusplus
}
#endif
#endif /* RTC_H */
