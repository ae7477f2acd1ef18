/*
 * rtc_oracle.h — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C, f64, operation-order-faithful restatement of the reference's hot path
 * (joedane/raytracer-challenge, crate ch1: camera.rs, shape.rs, material.rs, vec.rs,
 * transform.rs, color.rs, canvas.rs). It is the checker for the HIP path and the
 * "port" CPU baseline in bench.py. Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product library (librtc.so) never links or calls it.
 *
 * Pinning: the reference is Rust and no Rust toolchain exists in the build container
 * (SURVEY.md F2), so the reference itself cannot be run; this restatement is pinned by
 * every known-answer test the reference's own #[cfg(test)] modules hold for the path
 * (SURVEY.md App. D), transcribed test by test (inputs and expected values, with the
 * reference file:line of each) in tests/test_oracle_kats.py.
 *
 * It shares only the plain-data scene structs of include/rtc.h with the product.
 * Compile: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).
 */
#ifndef RTC_ORACLE_H
#define RTC_ORACLE_H

#include "../include/rtc.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- transform.rs ------------------------------------------------------------------ */
void   orc_matrix_identity(double out[16]);                                     /* :44-51   */
void   orc_matrix_multiply(const double a[16], const double b[16], double out[16]); /* :8-21 */
void   orc_matrix_translation(const double m[16], double x, double y, double z, double out[16]); /* :53-60 */
void   orc_matrix_scaling(const double m[16], double x, double y, double z, double out[16]);     /* :62-69 */
void   orc_matrix_rotation_x(const double m[16], double r, double out[16]);     /* :71-78   */
void   orc_matrix_rotation_y(const double m[16], double r, double out[16]);     /* :80-87   */
void   orc_matrix_rotation_z(const double m[16], double r, double out[16]);     /* :89-96   */
void   orc_matrix_shearing(const double m[16], double xy, double xz, double yx, double yz,
                           double zx, double zy, double out[16]);               /* :98-105  */
double orc_matrix_determinant(const double m[16]);                              /* :130-169 */
int    orc_matrix_inverse(const double m[16], double out[16]);  /* 0 ok, 1 singular :175-190 */
void   orc_matrix_transpose(const double m[16], double out[16]);                /* :192-202 */
void   orc_view_transform(const double from[3], const double to[3], const double up[3],
                          double out[16]);                                      /* :204-217 */
void   orc_transform_point(const double m[16], const double p[3], double out[3]);  /* :122-128 */
void   orc_transform_vector(const double m[16], const double v[3], double out[3]); /* :107-120 */

/* ---- camera.rs --------------------------------------------------------------------- */
int    orc_camera_init(uint32_t hsize, uint32_t vsize, double fov, const double view[16],
                       rtc_camera *out);                                        /* :33-58   */
void   orc_camera_ray_for_pixel(const rtc_camera *cam, uint32_t x, double xo, uint32_t y,
                                double yo, double ray[6]);                      /* :64-76   */

/* ---- material.rs / shape.rs constructors ------------------------------------------- */
void   orc_material_default(rtc_material *out);                 /* material.rs:273-283,364-369 */
void   orc_light_default(rtc_light *out);                       /* material.rs:26-31           */
int    orc_shape_init(uint32_t kind, const double transform[16], const rtc_material *mat,
                      rtc_shape *out);                          /* shape.rs:308-317 etc.       */

/* ---- shape.rs: intersections ------------------------------------------------------- */
/* Shape::intersect (shape.rs:23-26 + intersect_local): writes 0..2 t values in list order. */
int    orc_shape_intersect(const rtc_shape *s, const double ray[6], double ts[2]);
/* Shape::normal_at (shape.rs:34-40; Cube override :623-629). */
void   orc_normal_at(const rtc_shape *s, const double p[3], double out[3]);
/* Intersections::_insert_sorted (shape.rs:195-208) on a (ts, idxs) list of length *k. */
void   orc_list_insert_sorted(double *ts, int32_t *idxs, uint32_t *k, double t, int32_t idx);
/* Intersections::get_hit (shape.rs:220-232): position of first t >= 0.0, or -1. */
int32_t orc_list_get_hit(const double *ts, uint32_t k);
/* World::intersect (shape.rs:677-683): fills ts/idxs (capacity >= 2n); returns k. */
uint32_t orc_world_intersect(const rtc_shape *shapes, uint32_t n, const double ray[6],
                             double *ts, int32_t *idxs);
/* Intersection::compute_vectors (shape.rs:144-152) for list entry `pos`;
 * returns 0, or 1 when compute_refractive would panic (shape.rs:140). */
int    orc_compute_vectors(const rtc_shape *shapes, const double ray[6], const double *ts,
                           const int32_t *idxs, uint32_t k, uint32_t pos, rtc_hit *out);

/* ---- shape.rs: shading ------------------------------------------------------------- */
int    orc_is_shadowed(const rtc_shape *shapes, uint32_t n, const rtc_light *light,
                       const double p[3]);                                      /* :712-727 */
void   orc_shade_hit(const rtc_shape *shapes, uint32_t n, const rtc_light *light,
                     const rtc_hit *comps, uint32_t remaining, double rgb[3]);  /* :685-700 */
void   orc_color_at(const rtc_shape *shapes, uint32_t n, const rtc_light *light,
                    const double ray[6], uint32_t remaining, double rgb[3],
                    rtc_hit *first_hit /* may be NULL */);                      /* :702-710 */
void   orc_reflected_color(const rtc_shape *shapes, uint32_t n, const rtc_light *light,
                           const rtc_hit *comps, uint32_t remaining, double rgb[3]); /* :729-738 */
void   orc_refracted_color(const rtc_shape *shapes, uint32_t n, const rtc_light *light,
                           const rtc_hit *comps, uint32_t remaining, double rgb[3]); /* :751-766 */
double orc_reflectance(const rtc_hit *comps);                                   /* :768-781 */

/* Streaming reformulation of color_at (argmin hit + open-set n1/n2):
 * the algorithm the HIP kernels run, restated on the CPU so it can be asserted identical
 * to the literal sorted-list form without a GPU. */
void   orc_color_at_streaming(const rtc_shape *shapes, uint32_t n, const rtc_light *light,
                              const double ray[6], uint32_t remaining, double rgb[3],
                              rtc_hit *first_hit);

/* ---- material.rs ------------------------------------------------------------------- */
/* Material::lighting (material.rs:319-361). shape may be NULL when there is no pattern.
 * Returns RTC_OK, or RTC_ERR_NO_COLOR for the reference's expect() panics. */
int    orc_lighting(const rtc_material *m, const rtc_shape *shape, const rtc_light *light,
                    const double point[3], const double eye[3], const double normal[3],
                    int in_shadow, double rgb[3]);
void   orc_pattern_at(const rtc_material *m, const double p[3], double rgb[3]); /* :67,97,133,163,198,233 */
void   orc_pattern_at_shape(const rtc_material *m, const rtc_shape *shape,
                            const double world_point[3], double rgb[3]);        /* :41-45   */

/* ---- camera.rs render drivers + canvas.rs ------------------------------------------ */
/* Camera::render (mode RTC_MODE_RENDER, :116-126) / render_async (:144-160) for rows
 * [y0,y1) into rgb ((y1-y0)*hsize*3). nthreads >= 1 (rows are handed out dynamically, like rayon).
 * streaming != 0 uses the streaming formulation. stats may be NULL. */
void   orc_render(const rtc_shape *shapes, uint32_t n, const rtc_light *light,
                  const rtc_camera *cam, uint32_t mode, uint32_t y0, uint32_t y1,
                  double *rgb, uint32_t nthreads, int streaming, rtc_stats *stats);
/* The same with render flags: RTC_FLAG_AA_RESAMPLE takes render_pixel's resample branch
 * (camera.rs:84-92,108-111) with the counter-based offsets rtc.h documents (the reference's come from
 * thread_rng); without it the tripped pixels keep the 4-sample mean. stats->pixels_resample counts them. */
void   orc_render_flags(const rtc_shape *shapes, uint32_t n, const rtc_light *light,
                        const rtc_camera *cam, uint32_t mode, uint32_t flags, uint32_t y0, uint32_t y1,
                        double *rgb, uint32_t nthreads, int streaming, rtc_stats *stats);
/* Canvas::write_to_file_simple (canvas.rs:86-109) into memory; returns bytes needed. */
void orc_canvas_to_rgba8(const double *rgb, uint32_t width, uint32_t height, float gamma, uint8_t *out);
size_t orc_format_ppm(const double *rgb, uint32_t width, uint32_t height, char *buf, size_t cap);
/* Color::scale (color.rs:100-114). */
int32_t orc_color_scale(double component, int32_t scale);

#ifdef __cplusplus
}
#endif
#endif
