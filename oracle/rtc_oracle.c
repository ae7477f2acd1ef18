/*
 * rtc_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE). See rtc_oracle.h.
 *
 * Every function restates one reference function; the citation is file:line under
 * /root/reference/ch1/src/. Expressions keep the reference's evaluation order
 * (left-to-right, no FMA): build with -ffp-contract=off -fno-fast-math.
 */
#define _GNU_SOURCE /* sincos */
#include "rtc_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define M(m, r, c) ((m)[(r) * 4 + (c)])
static const double EPS = 0.00000001; /* Vector::EPSILON vec.rs:16 */

typedef struct { uint64_t primary, shadow, reflect, refract; } counters_t;
enum { RAY_PRIMARY = 0, RAY_SHADOW = 1, RAY_REFLECT = 2, RAY_REFRACT = 3 };

/* ======================= vec.rs ======================= */
static double v_dot(const double a[3], const double b[3]) { /* vec.rs:78-82 */
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
static double v_magnitude(const double a[3]) { /* vec.rs:65-67 */
    return sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
}
static void v_normalize(const double a[3], double out[3]) { /* vec.rs:69-76: three divisions */
    double m = v_magnitude(a);
    double x = a[0] / m, y = a[1] / m, z = a[2] / m;
    out[0] = x; out[1] = y; out[2] = z;
}
static void v_cross(const double a[3], const double b[3], double out[3]) { /* vec.rs:84-90 */
    double x = a[1] * b[2] - a[2] * b[1];
    double y = a[2] * b[0] - a[0] * b[2];
    double z = a[0] * b[1] - a[1] * b[0];
    out[0] = x; out[1] = y; out[2] = z;
}
static void v_reflect(const double v[3], const double n[3], double out[3]) { /* vec.rs:106-108 */
    double k = 2. * v_dot(v, n);
    double x = v[0] - n[0] * k, y = v[1] - n[1] * k, z = v[2] - n[2] * k;
    out[0] = x; out[1] = y; out[2] = z;
}

/* ======================= transform.rs ======================= */
void orc_matrix_identity(double out[16]) { /* transform.rs:44-51 */
    for (int i = 0; i < 16; i++) out[i] = 0.;
    M(out, 0, 0) = 1.; M(out, 1, 1) = 1.; M(out, 2, 2) = 1.; M(out, 3, 3) = 1.;
}

void orc_matrix_multiply(const double a[16], const double b[16], double out[16]) { /* :8-21 */
    double r[16];
    for (int row = 0; row < 4; row++)
        for (int col = 0; col < 4; col++) {
            double sum = 0.0;
            for (int i = 0; i < 4; i++) {
                double v = M(a, row, i) * M(b, i, col);
                sum = sum + v;
            }
            M(r, row, col) = sum;
        }
    memcpy(out, r, sizeof r);
}

void orc_matrix_translation(const double m[16], double x, double y, double z, double out[16]) {
    double t[16] = {1., 0., 0., x, 0., 1., 0., y, 0., 0., 1., z, 0., 0., 0., 1.};
    orc_matrix_multiply(t, m, out); /* Matrix::new(m).multiply(self) :59 */
}
void orc_matrix_scaling(const double m[16], double x, double y, double z, double out[16]) {
    double t[16] = {x, 0., 0., 0., 0., y, 0., 0., 0., 0., z, 0., 0., 0., 0., 1.};
    orc_matrix_multiply(t, m, out);
}
/* r.sin()/r.cos() of one angle: an optimised rustc build on x86_64-linux-gnu lowers the pair to one
 * glibc sincos() call (LLVM FSINCOS), which differs from separate sin()/cos() by 1 ulp for ~0.14 %
 * of angles; a debug build calls them separately. The release behaviour is followed, explicitly. */
void orc_matrix_rotation_x(const double m[16], double r, double out[16]) { /* :71-78 */
    double sn, cs;
    sincos(r, &sn, &cs);
    double t[16] = {1., 0., 0., 0., 0., cs, -sn, 0., 0., sn, cs, 0., 0., 0., 0., 1.};
    orc_matrix_multiply(t, m, out);
}
void orc_matrix_rotation_y(const double m[16], double r, double out[16]) { /* :80-87 */
    double sn, cs;
    sincos(r, &sn, &cs);
    double t[16] = {cs, 0., sn, 0., 0., 1., 0., 0., -sn, 0., cs, 0., 0., 0., 0., 1.};
    orc_matrix_multiply(t, m, out);
}
void orc_matrix_rotation_z(const double m[16], double r, double out[16]) { /* :89-96 */
    double sn, cs;
    sincos(r, &sn, &cs);
    double t[16] = {cs, -sn, 0., 0., sn, cs, 0., 0., 0., 0., 1., 0., 0., 0., 0., 1.};
    orc_matrix_multiply(t, m, out);
}
void orc_matrix_shearing(const double m[16], double xy, double xz, double yx, double yz,
                         double zx, double zy, double out[16]) { /* :98-105 */
    double t[16] = {1., xy, xz, 0., yx, 1., yz, 0., zx, zy, 1., 0., 0., 0., 0., 1.};
    orc_matrix_multiply(t, m, out);
}

/* determinant / submatrix / minor / cofactor on an n x n matrix, n in {2,3,4}
 * (transform.rs:130-169); `m` is row-major with leading dimension n. */
static double det_n(const double *m, int n);
static double cofactor_n(const double *m, int n, int i, int j) { /* :145-169 */
    double sub[9];
    int ns = n - 1;
    for (int row = 0; row < ns; row++)
        for (int col = 0; col < ns; col++)
            sub[row * ns + col] = m[(row < i ? row : row + 1) * n + (col < j ? col : col + 1)];
    double minor = det_n(sub, ns);
    if ((i + j) % 2 == 1) minor = -1.0 * minor;
    return minor;
}
static double det_n(const double *m, int n) { /* :130-143 */
    if (n == 2) return m[0] * m[3] - m[1] * m[2];
    double det = 0.0;
    for (int col = 0; col < n; col++) det += m[col] * cofactor_n(m, n, 0, col);
    return det;
}
double orc_matrix_determinant(const double m[16]) { return det_n(m, 4); }

int orc_matrix_inverse(const double m[16], double out[16]) { /* :35-38,175-190 */
    double d = det_n(m, 4);
    if (!(fabs(d) > EPS)) return 1; /* determinant: None -> panic!("Matrix is not invertable") */
    double r[16];
    for (int row = 0; row < 4; row++)
        for (int col = 0; col < 4; col++) {
            double c = cofactor_n(m, 4, row, col);
            M(r, col, row) = c / d;
        }
    memcpy(out, r, sizeof r);
    return 0;
}

void orc_matrix_transpose(const double m[16], double out[16]) { /* :192-202 */
    double r[16];
    for (int row = 0; row < 4; row++)
        for (int col = 0; col < 4; col++) M(r, row, col) = M(m, col, row);
    memcpy(out, r, sizeof r);
}

void orc_view_transform(const double from[3], const double to[3], const double up[3], double out[16]) {
    /* transform.rs:204-217 */
    double d[3] = {to[0] - from[0], to[1] - from[1], to[2] - from[2]}; /* Point::sub vec.rs:168 */
    double forward[3], upn[3], left[3], true_up[3];
    v_normalize(d, forward);
    v_normalize(up, upn);
    v_cross(forward, upn, left);
    v_cross(left, forward, true_up);
    double o[16] = {left[0], left[1], left[2], 0., true_up[0], true_up[1], true_up[2], 0.,
                    -(forward[0]), -(forward[1]), -(forward[2]), 0., 0., 0., 0., 1.};
    double id[16], tr[16];
    orc_matrix_identity(id);
    orc_matrix_translation(id, -from[0], -from[1], -from[2], tr);
    orc_matrix_multiply(o, tr, out);
}

void orc_transform_vector(const double m[16], const double p[3], double out[3]) { /* :107-120 */
    double x = M(m, 0, 0) * p[0] + M(m, 0, 1) * p[1] + M(m, 0, 2) * p[2];
    double y = M(m, 1, 0) * p[0] + M(m, 1, 1) * p[1] + M(m, 1, 2) * p[2];
    double z = M(m, 2, 0) * p[0] + M(m, 2, 1) * p[1] + M(m, 2, 2) * p[2];
    out[0] = x; out[1] = y; out[2] = z;
}
void orc_transform_point(const double m[16], const double p[3], double out[3]) { /* :122-128 */
    double x = M(m, 0, 0) * p[0] + M(m, 0, 1) * p[1] + M(m, 0, 2) * p[2] + M(m, 0, 3);
    double y = M(m, 1, 0) * p[0] + M(m, 1, 1) * p[1] + M(m, 1, 2) * p[2] + M(m, 1, 3);
    double z = M(m, 2, 0) * p[0] + M(m, 2, 1) * p[1] + M(m, 2, 2) * p[2] + M(m, 2, 3);
    out[0] = x; out[1] = y; out[2] = z;
}

/* ======================= camera.rs ======================= */
int orc_camera_init(uint32_t hsize, uint32_t vsize, double fov, const double view[16], rtc_camera *out) {
    /* Camera::new camera.rs:39-58 */
    memset(out, 0, sizeof *out);
    double half_view = tan(fov / 2.0);
    double aspect = (double)hsize / (double)vsize;
    double half_width, half_height;
    if (aspect < 1.0) {
        half_height = half_view;
        half_width = half_view * aspect;
    } else {
        half_width = half_view;
        half_height = half_view / aspect;
    }
    double pixel_size = (half_width * 2.) / (double)hsize;
    out->hsize = hsize; out->vsize = vsize; out->fov = fov;
    out->half_width = half_width; out->half_height = half_height; out->pixel_size = pixel_size;
    out->samples = 1;
    /* new_with_transform camera.rs:33-37 */
    return orc_matrix_inverse(view, out->view_inv);
}

void orc_camera_ray_for_pixel(const rtc_camera *cam, uint32_t x, double xo, uint32_t y, double yo, double ray[6]) {
    /* camera.rs:64-76 */
    double xoffset = ((double)x + xo) * cam->pixel_size;
    double yoffset = ((double)y + yo) * cam->pixel_size;
    double world_x = cam->half_width - xoffset;
    double world_y = cam->half_height - yoffset;
    double pp[3] = {world_x, world_y, -1.}, zero[3] = {0., 0., 0.};
    double pixel[3], origin[3];
    orc_transform_point(cam->view_inv, pp, pixel);
    orc_transform_point(cam->view_inv, zero, origin);
    double d[3] = {pixel[0] - origin[0], pixel[1] - origin[1], pixel[2] - origin[2]};
    double dir[3];
    v_normalize(d, dir);
    ray[0] = origin[0]; ray[1] = origin[1]; ray[2] = origin[2];
    ray[3] = dir[0]; ray[4] = dir[1]; ray[5] = dir[2];
}

/* ======================= constructors ======================= */
void orc_material_default(rtc_material *m) { /* Material::default(): DEFAULT with WHITE */
    memset(m, 0, sizeof *m);
    m->pattern_kind = RTC_PATTERN_NONE;
    m->has_color = 1;
    m->color[0] = 1.; m->color[1] = 1.; m->color[2] = 1.;
    m->ambient = 0.1; m->diffuse = 0.9; m->specular = 0.9; m->shininess = 200.0;
    m->reflective = 0.0; m->transparency = 0.0; m->refractive_index = 1.0;
    orc_matrix_identity(m->pat_inv);
}
void orc_light_default(rtc_light *l) { /* material.rs:26-31 */
    l->intensity[0] = 1.; l->intensity[1] = 1.; l->intensity[2] = 1.;
    l->position[0] = -10.; l->position[1] = 10.; l->position[2] = -10.;
}
int orc_shape_init(uint32_t kind, const double transform[16], const rtc_material *mat, rtc_shape *out) {
    /* shape.rs:308-317 / 436-444 / 525-533 */
    memset(out, 0, sizeof *out);
    out->kind = kind;
    out->world_id = 0;
    if (orc_matrix_inverse(transform, out->inv)) return 1;
    orc_matrix_transpose(out->inv, out->inv_t);
    if (mat) out->material = *mat; else orc_material_default(&out->material);
    return 0;
}

/* ======================= shape.rs: intersections ======================= */
static double cube_max(double a, double b) { return fmax(a, b); } /* f64::max */
static double cube_min(double a, double b) { return fmin(a, b); }
static void check_axis(double origin, double direction, double *tmin_o, double *tmax_o) {
    /* shape.rs:540-564 */
    double tmin_numerator = -1.0 - origin;
    double tmax_numerator = 1. - origin;
    double tmin, tmax;
    if (fabs(direction) >= EPS) {
        tmin = tmin_numerator / direction;
        tmax = tmax_numerator / direction;
    } else {
        tmin = (tmin_numerator >= 0.0) ? INFINITY : -INFINITY;
        tmax = (tmax_numerator >= 0.0) ? INFINITY : -INFINITY;
    }
    if (tmin > tmax) { double s = tmin; tmin = tmax; tmax = s; }
    *tmin_o = tmin; *tmax_o = tmax;
}

int orc_shape_intersect(const rtc_shape *s, const double ray[6], double ts[2]) {
    /* Shape::intersect shape.rs:23-26: ray.transform(inverse) vec.rs:211-214 */
    double o[3], d[3];
    orc_transform_point(s->inv, ray, o);
    orc_transform_vector(s->inv, ray + 3, d);
    if (s->kind == RTC_SPHERE) { /* shape.rs:361-375 */
        double a = v_dot(d, d);
        double b = 2. * v_dot(d, o);
        double c = v_dot(o, o) - 1.;
        double discriminant = (b * b) - 4. * a * c;
        if (discriminant < 0.) return 0;
        double t1 = (-b - sqrt(discriminant)) / (2. * a);
        double t2 = (-b + sqrt(discriminant)) / (2. * a);
        /* Intersections::new(t1) then add(t2) -> _insert_sorted (shape.rs:195-208) */
        if (t2 < t1) { ts[0] = t2; ts[1] = t1; } else { ts[0] = t1; ts[1] = t2; }
        return 2;
    } else if (s->kind == RTC_PLANE) { /* shape.rs:462-471 */
        if (fabs(d[1]) < EPS) return 0;
        ts[0] = -o[1] / d[1];
        return 1;
    } else { /* Cube shape.rs:577-591 */
        double xmin, xmax, ymin, ymax, zmin, zmax;
        check_axis(o[0], d[0], &xmin, &xmax);
        check_axis(o[1], d[1], &ymin, &ymax);
        check_axis(o[2], d[2], &zmin, &zmax);
        double tmin = cube_max(xmin, cube_max(ymin, zmin));
        double tmax = cube_min(xmax, cube_min(ymax, zmax));
        if (tmin < tmax) { ts[0] = tmin; ts[1] = tmax; return 2; }
        return 0;
    }
}

void orc_normal_at(const rtc_shape *s, const double p[3], double out[3]) { /* shape.rs:34-40 */
    double lp[3], ln[3], wn[3];
    orc_transform_point(s->inv, p, lp);
    if (s->kind == RTC_SPHERE) { /* :377-379 p.sub(Point::ZERO) */
        ln[0] = lp[0] - 0.; ln[1] = lp[1] - 0.; ln[2] = lp[2] - 0.;
        orc_transform_vector(s->inv_t, ln, wn);
    } else if (s->kind == RTC_PLANE) { /* :481-483 */
        ln[0] = 0.; ln[1] = 1.; ln[2] = 0.;
        orc_transform_vector(s->inv_t, ln, wn);
    } else { /* Cube: normal_at_local :601-610, normal_at override :623-629 uses xf.transpose() */
        double ax = fabs(lp[0]), ay = fabs(lp[1]), az = fabs(lp[2]);
        double maxc = cube_max(ax, cube_max(ay, az));
        if (maxc == ax) { ln[0] = lp[0]; ln[1] = 0.; ln[2] = 0.; }
        else if (maxc == ay) { ln[0] = 0.; ln[1] = lp[1]; ln[2] = 0.; }
        else { ln[0] = 0.; ln[1] = 0.; ln[2] = lp[2]; }
        double t[16];
        orc_matrix_transpose(s->inv, t);
        orc_transform_vector(t, ln, wn);
    }
    v_normalize(wn, out);
}

void orc_list_insert_sorted(double *ts, int32_t *idxs, uint32_t *k, double t, int32_t idx) {
    /* shape.rs:195-208: insert before the first element with x.t < elem.t, else push */
    uint32_t n = *k, at = n;
    for (uint32_t i = 0; i < n; i++)
        if (t < ts[i]) { at = i; break; }
    for (uint32_t i = n; i > at; i--) { ts[i] = ts[i - 1]; idxs[i] = idxs[i - 1]; }
    ts[at] = t; idxs[at] = idx;
    *k = n + 1;
}

int32_t orc_list_get_hit(const double *ts, uint32_t k) { /* shape.rs:220-232 */
    for (uint32_t i = 0; i < k; i++)
        if (ts[i] >= 0.0) return (int32_t)i;
    return -1;
}

uint32_t orc_world_intersect(const rtc_shape *shapes, uint32_t n, const double ray[6],
                             double *ts, int32_t *idxs) { /* shape.rs:677-683 */
    uint32_t k = 0;
    for (uint32_t s = 0; s < n; s++) {
        double lt[2];
        int c = orc_shape_intersect(&shapes[s], ray, lt);
        for (int i = 0; i < c; i++) orc_list_insert_sorted(ts, idxs, &k, lt[i], (int32_t)s); /* merge :214-218 */
    }
    return k;
}

static int compute_refractive(const rtc_shape *shapes, const double *ts, const int32_t *idxs,
                              uint32_t k, uint32_t pos, double *n1o, double *n2o) {
    /* shape.rs:115-141. Intersection == compares t only (shape.rs:156-164). */
    uint32_t *containers = (uint32_t *)malloc(sizeof(uint32_t) * (k ? k : 1));
    uint32_t len = 0;
    double n1 = 1.0, n2 = 1.0;
    for (uint32_t i = 0; i < k; i++) {
        int is_self = (ts[i] == ts[pos]);
        if (is_self) n1 = (len == 0) ? 1.0 : shapes[idxs[containers[len - 1]]].material.refractive_index;
        uint32_t found = len;
        for (uint32_t p = 0; p < len; p++)
            if (shapes[idxs[containers[p]]].world_id == shapes[idxs[i]].world_id) { found = p; break; }
        if (found < len) {
            for (uint32_t p = found; p + 1 < len; p++) containers[p] = containers[p + 1];
            len--;
        } else {
            containers[len++] = i;
        }
        if (is_self) {
            n2 = (len == 0) ? 1.0 : shapes[idxs[containers[len - 1]]].material.refractive_index;
            *n1o = n1; *n2o = n2;
            free(containers);
            return 0;
        }
    }
    free(containers);
    *n1o = n1; *n2o = n2;
    return 1; /* panic!("invalid refraction setup") */
}

static void cached_vectors_new(const double ray[6], double t, int32_t idx, const double point[3],
                               const double eyev[3], const double normal_in[3], double n1,
                               double n2, rtc_hit *out) { /* shape.rs:75-96 */
    memset(out, 0, sizeof *out);
    int inside = v_dot(normal_in, eyev) < 0.0;
    double normal[3];
    if (inside) { normal[0] = -normal_in[0]; normal[1] = -normal_in[1]; normal[2] = -normal_in[2]; }
    else { normal[0] = normal_in[0]; normal[1] = normal_in[1]; normal[2] = normal_in[2]; }
    for (int i = 0; i < 3; i++) {
        out->over_point[i] = point[i] + normal[i] * EPS;
        out->under_point[i] = point[i] - normal[i] * EPS;
        out->point[i] = point[i];
        out->eyev[i] = eyev[i];
        out->normal[i] = normal[i];
    }
    v_reflect(ray + 3, normal, out->reflectv);
    out->t = t;
    out->hit_index = idx;
    out->inside = (uint32_t)inside;
    out->n1 = n1; out->n2 = n2;
}

int orc_compute_vectors(const rtc_shape *shapes, const double ray[6], const double *ts,
                        const int32_t *idxs, uint32_t k, uint32_t pos, rtc_hit *out) {
    /* shape.rs:144-152 */
    double t = ts[pos];
    int32_t idx = idxs[pos];
    double p[3] = {ray[0] + ray[3] * t, ray[1] + ray[4] * t, ray[2] + ray[5] * t}; /* vec.rs:207-209 */
    double eyev[3] = {-ray[3], -ray[4], -ray[5]};
    double normal[3];
    orc_normal_at(&shapes[idx], p, normal);
    double n1, n2;
    int rc = compute_refractive(shapes, ts, idxs, k, pos, &n1, &n2);
    cached_vectors_new(ray, t, idx, p, eyev, normal, n1, n2, out);
    return rc;
}

/* ======================= material.rs ======================= */
static double rs_rem2(double x) { return fmod(x, 2.0); } /* Rust f64 `%` == C fmod */

void orc_pattern_at(const rtc_material *m, const double p[3], double rgb[3]) {
    const double *a = m->pat_a, *b = m->pat_b;
    switch (m->pattern_kind) {
    case RTC_PATTERN_TEST: /* material.rs:67-69 */
        rgb[0] = p[0]; rgb[1] = p[1]; rgb[2] = p[2];
        return;
    case RTC_PATTERN_STRIPE: /* :97-103 */
        if (rs_rem2(floor(p[0])) == 0.) { rgb[0] = a[0]; rgb[1] = a[1]; rgb[2] = a[2]; }
        else { rgb[0] = b[0]; rgb[1] = b[1]; rgb[2] = b[2]; }
        return;
    case RTC_PATTERN_GRADIENT: { /* :113-122,133-135: _diff = b.sub(a) */
        double f = p[0] - floor(p[0]);
        for (int i = 0; i < 3; i++) rgb[i] = a[i] + (b[i] - a[i]) * f;
        return;
    }
    case RTC_PATTERN_RING: /* :163-170; powi(2) == x*x */
        if (rs_rem2(floor(sqrt(p[0] * p[0] + p[1] * p[1]))) == 0.0) { rgb[0] = a[0]; rgb[1] = a[1]; rgb[2] = a[2]; }
        else { rgb[0] = b[0]; rgb[1] = b[1]; rgb[2] = b[2]; }
        return;
    case RTC_PATTERN_CHECKER: /* :198-205 */
        if (rs_rem2(floor(p[0]) + floor(p[1]) + floor(p[2])) == 0.0) { rgb[0] = a[0]; rgb[1] = a[1]; rgb[2] = a[2]; }
        else { rgb[0] = b[0]; rgb[1] = b[1]; rgb[2] = b[2]; }
        return;
    case RTC_PATTERN_GRID: /* :233-241: a = base, b = grid */
        if (fabs(p[0] - floor(p[0])) < 0.01 || fabs(p[2] - floor(p[2])) < 0.01) { rgb[0] = b[0]; rgb[1] = b[1]; rgb[2] = b[2]; }
        else { rgb[0] = a[0]; rgb[1] = a[1]; rgb[2] = a[2]; }
        return;
    default:
        rgb[0] = rgb[1] = rgb[2] = 0.;
    }
}

void orc_pattern_at_shape(const rtc_material *m, const rtc_shape *shape, const double wp[3], double rgb[3]) {
    /* material.rs:41-45 */
    double op[3], pp[3];
    orc_transform_point(shape->inv, wp, op);
    orc_transform_point(m->pat_inv, op, pp);
    orc_pattern_at(m, pp, rgb);
}

int orc_lighting(const rtc_material *m, const rtc_shape *shape, const rtc_light *light,
                 const double point[3], const double eye[3], const double normal[3],
                 int in_shadow, double rgb[3]) { /* material.rs:319-361 */
    double base[3], eff[3];
    if (m->pattern_kind != RTC_PATTERN_NONE) {
        if (!shape) { rgb[0] = rgb[1] = rgb[2] = 0.; return RTC_ERR_NO_COLOR; } /* expect() :328 */
        orc_pattern_at_shape(m, shape, point, base);
    } else {
        if (!m->has_color) { rgb[0] = rgb[1] = rgb[2] = 0.; return RTC_ERR_NO_COLOR; } /* expect() :331 */
        base[0] = m->color[0]; base[1] = m->color[1]; base[2] = m->color[2];
    }
    for (int i = 0; i < 3; i++) eff[i] = base[i] * light->intensity[i];
    double lv[3] = {light->position[0] - point[0], light->position[1] - point[1], light->position[2] - point[2]};
    double lightv[3];
    v_normalize(lv, lightv);
    double ambient[3] = {eff[0] * m->ambient, eff[1] * m->ambient, eff[2] * m->ambient};
    if (in_shadow) { rgb[0] = ambient[0]; rgb[1] = ambient[1]; rgb[2] = ambient[2]; return RTC_OK; }
    double light_dot_normal = v_dot(lightv, normal);
    double diffuse[3] = {0., 0., 0.}, specular[3] = {0., 0., 0.};
    if (light_dot_normal < 0.) {
        /* both BLACK */
    } else {
        double kd = m->diffuse * light_dot_normal;
        diffuse[0] = eff[0] * kd; diffuse[1] = eff[1] * kd; diffuse[2] = eff[2] * kd;
        double nl[3] = {-lightv[0], -lightv[1], -lightv[2]}, reflectv[3];
        v_reflect(nl, normal, reflectv);
        double reflect_dot_eye = v_dot(reflectv, eye);
        if (reflect_dot_eye <= 0.) {
            /* BLACK */
        } else {
            double factor = pow(reflect_dot_eye, m->shininess);
            double ks = m->specular * factor;
            specular[0] = light->intensity[0] * ks;
            specular[1] = light->intensity[1] * ks;
            specular[2] = light->intensity[2] * ks;
        }
    }
    for (int i = 0; i < 3; i++) rgb[i] = (ambient[i] + diffuse[i]) + specular[i];
    return RTC_OK;
}

/* ======================= shape.rs: shading ======================= */
typedef struct {
    const rtc_shape *shapes;
    uint32_t n;
    const rtc_light *light;
    double *ts;      /* scratch lists, one pair per recursion level (7 levels x 2n) */
    int32_t *idxs;
    counters_t *cnt;
    int streaming;
    uint32_t *id_order; /* streaming form: shape indices in stable order of world_id */
} trace_t;

static void color_at_impl(trace_t *T, const double ray[6], uint32_t remaining, int kind,
                          int level, double rgb[3], rtc_hit *first_hit);

static int is_shadowed_impl(trace_t *T, const double p[3], int level) { /* shape.rs:716-727 */
    const rtc_light *l = T->light;
    double v[3] = {l->position[0] - p[0], l->position[1] - p[1], l->position[2] - p[2]};
    double distance = v_magnitude(v);
    double dir[3];
    v_normalize(v, dir);
    double r[6] = {p[0], p[1], p[2], dir[0], dir[1], dir[2]};
    if (T->cnt) T->cnt->shadow++;
    if (T->streaming) {
        /* exists t with 0.0 <= t < distance  <=>  get_hit().t < distance */
        for (uint32_t s = 0; s < T->n; s++) {
            double lt[2];
            int c = orc_shape_intersect(&T->shapes[s], r, lt);
            for (int i = 0; i < c; i++)
                if (lt[i] >= 0.0 && lt[i] < distance) return 1;
        }
        return 0;
    }
    double *ts = T->ts + (size_t)level * 2 * T->n;
    int32_t *idxs = T->idxs + (size_t)level * 2 * T->n;
    uint32_t k = orc_world_intersect(T->shapes, T->n, r, ts, idxs);
    int32_t h = orc_list_get_hit(ts, k);
    if (h >= 0) return ts[h] < distance;
    return 0;
}

static void reflected_color_impl(trace_t *T, const rtc_hit *c, uint32_t remaining, int level, double rgb[3]) {
    /* shape.rs:729-738 */
    const rtc_material *m = &T->shapes[c->hit_index].material;
    if (remaining == 0 || m->reflective <= 0.) { rgb[0] = rgb[1] = rgb[2] = 0.; return; }
    double r[6] = {c->over_point[0], c->over_point[1], c->over_point[2], c->reflectv[0], c->reflectv[1], c->reflectv[2]};
    double col[3];
    color_at_impl(T, r, remaining - 1, RAY_REFLECT, level + 1, col, NULL);
    rgb[0] = col[0] * m->reflective; rgb[1] = col[1] * m->reflective; rgb[2] = col[2] * m->reflective;
}

static void refracted_color_impl(trace_t *T, const rtc_hit *c, uint32_t remaining, int level, double rgb[3]) {
    /* shape.rs:751-766; powi(2) == x*x */
    const rtc_material *m = &T->shapes[c->hit_index].material;
    if (remaining == 0 || m->transparency == 0.0) { rgb[0] = rgb[1] = rgb[2] = 0.; return; }
    double n_ratio = c->n1 / c->n2;
    double cos_i = v_dot(c->eyev, c->normal);
    double sin2_t = (n_ratio * n_ratio) * (1.0 - cos_i * cos_i);
    if (sin2_t > 1.0) { rgb[0] = rgb[1] = rgb[2] = 0.; return; }
    double cos_t = sqrt(1.0 - sin2_t);
    double k = n_ratio * cos_i - cos_t;
    double dir[3] = {c->normal[0] * k - c->eyev[0] * n_ratio, c->normal[1] * k - c->eyev[1] * n_ratio,
                     c->normal[2] * k - c->eyev[2] * n_ratio};
    double r[6] = {c->under_point[0], c->under_point[1], c->under_point[2], dir[0], dir[1], dir[2]};
    double col[3];
    color_at_impl(T, r, remaining - 1, RAY_REFRACT, level + 1, col, NULL);
    rgb[0] = col[0] * m->transparency; rgb[1] = col[1] * m->transparency; rgb[2] = col[2] * m->transparency;
}

double orc_reflectance(const rtc_hit *c) { /* shape.rs:768-781; powi(5) = x*(x^2)^2 */
    double cosv = v_dot(c->eyev, c->normal);
    if (c->n1 > c->n2) {
        double n = c->n1 / c->n2;
        double sin2_t = (n * n) * (1.0 - cosv * cosv);
        if (sin2_t > 1.0) return 1.0;
        double cos_t = sqrt(1.0 - sin2_t);
        cosv = cos_t;
    }
    double q = (c->n1 - c->n2) / (c->n1 + c->n2);
    double r0 = q * q;
    double x = 1.0 - cosv;
    double x2 = x * x;
    double x5 = x * (x2 * x2); /* llvm powi expansion for 5 = 0b101: x * (x^2)^2 */
    return r0 + (1.0 - r0) * x5;
}

static void shade_hit_impl(trace_t *T, const rtc_hit *c, uint32_t remaining, int level, double rgb[3]) {
    /* shape.rs:685-700 */
    const rtc_shape *obj = &T->shapes[c->hit_index];
    const rtc_material *m = &obj->material;
    int shadowed = is_shadowed_impl(T, c->over_point, level);
    double surface[3], reflected[3], refracted[3];
    orc_lighting(m, obj, T->light, c->over_point, c->eyev, c->normal, shadowed, surface);
    reflected_color_impl(T, c, remaining, level, reflected);
    refracted_color_impl(T, c, remaining, level, refracted);
    if (m->reflective > 0.0 && m->transparency > 0.0) {
        double reflectance = orc_reflectance(c);
        for (int i = 0; i < 3; i++)
            rgb[i] = surface[i] + (reflected[i] * reflectance + refracted[i] * (1.0 - reflectance));
    } else {
        for (int i = 0; i < 3; i++) rgb[i] = (surface[i] + reflected[i]) + refracted[i];
    }
}

/* Streaming hit selection + n1/n2. */
static int streaming_first_hit(trace_t *T, const double ray[6], rtc_hit *out) {
    double best = INFINITY;
    int32_t h = -1;
    int root = 0;
    for (uint32_t s = 0; s < T->n; s++) {
        double lt[2];
        int c = orc_shape_intersect(&T->shapes[s], ray, lt);
        for (int i = 0; i < c; i++)
            if (lt[i] >= 0.0 && lt[i] < best) { best = lt[i]; h = (int32_t)s; root = i; }
    }
    if (h < 0) return 0;
    double p[3] = {ray[0] + ray[3] * best, ray[1] + ray[4] * best, ray[2] + ray[5] * best};
    double eyev[3] = {-ray[3], -ray[4], -ray[5]};
    double normal[3];
    orc_normal_at(&T->shapes[h], p, normal);
    double n1 = 1.0, n2 = 1.0;
    if (T->shapes[h].material.transparency != 0.0) {
        /* compute_refractive (shape.rs:115-141) without the sorted list. `containers` is keyed by
         * world_id (shape.rs:127): every entry before the hit entry (list order = ascending (t, shape
         * index), an object's first root before its second) toggles its id's membership. An id is
         * present iff an odd number of its entries precede the hit entry; the element standing for it
         * is the entry that pushed it last = the LAST of those entries; containers.last = the present
         * id whose such entry is latest in list order. Shapes are visited grouped by id (T->id_order:
         * stable sort by world_id), one class accumulator at a time. With unique ids this is the
         * "open set" of SURVEY.md App. A.6; with shared ids (the reference's u8 ids wrap at 256
         * shapes, shape.rs:287,661-667) it is still exactly the literal walk. n2: the hit entry
         * toggles the hit shape's id once more. */
        int have_all = 0, have_oth = 0;
        double key_all = 0., key_oth = 0.;
        int32_t idx_all = -1, idx_oth = -1;
        const uint32_t hid = T->shapes[h].world_id;
        uint32_t cnt_h = 0; /* entries of the hit's id class before the hit entry */
        uint32_t k = 0;
        while (k < T->n) {
            const uint32_t id = T->shapes[T->id_order[k]].world_id;
            uint32_t cnt = 0;
            double last_t = 0.;
            int32_t last_s = -1;
            for (; k < T->n && T->shapes[T->id_order[k]].world_id == id; k++) {
                const int32_t sidx = (int32_t)T->id_order[k];
                double lt[2];
                int c = orc_shape_intersect(&T->shapes[sidx], ray, lt);
                for (int i = 0; i < c; i++) {
                    int before;
                    if (sidx == h) before = (i < root);
                    else before = lt[i] < best || (lt[i] == best && sidx < h);
                    if (!before) continue;
                    cnt++;
                    if (last_s < 0 || lt[i] > last_t || (lt[i] == last_t && sidx >= last_s)) { last_t = lt[i]; last_s = sidx; }
                }
            }
            if (id == hid) cnt_h = cnt;
            if (cnt & 1u) {
                if (!have_all || last_t > key_all || (last_t == key_all && last_s > idx_all)) { have_all = 1; key_all = last_t; idx_all = last_s; }
                if (id != hid)
                    if (!have_oth || last_t > key_oth || (last_t == key_oth && last_s > idx_oth)) { have_oth = 1; key_oth = last_t; idx_oth = last_s; }
            }
        }
        n1 = have_all ? T->shapes[idx_all].material.refractive_index : 1.0;
        if (cnt_h & 1u) n2 = have_oth ? T->shapes[idx_oth].material.refractive_index : 1.0; /* the hit entry removes its id */
        else n2 = T->shapes[h].material.refractive_index;                                   /* the hit entry pushes its id  */
    }
    cached_vectors_new(ray, best, h, p, eyev, normal, n1, n2, out);
    return 1;
}

static void color_at_impl(trace_t *T, const double ray[6], uint32_t remaining, int kind,
                          int level, double rgb[3], rtc_hit *first_hit) { /* shape.rs:702-710 */
    if (T->cnt) {
        if (kind == RAY_PRIMARY) T->cnt->primary++;
        else if (kind == RAY_REFLECT) T->cnt->reflect++;
        else if (kind == RAY_REFRACT) T->cnt->refract++;
    }
    rtc_hit comps;
    int have = 0;
    if (T->streaming) {
        have = streaming_first_hit(T, ray, &comps);
    } else {
        double *ts = T->ts + (size_t)level * 2 * T->n;
        int32_t *idxs = T->idxs + (size_t)level * 2 * T->n;
        uint32_t k = orc_world_intersect(T->shapes, T->n, ray, ts, idxs);
        int32_t h = orc_list_get_hit(ts, k);
        if (h >= 0) { orc_compute_vectors(T->shapes, ray, ts, idxs, k, (uint32_t)h, &comps); have = 1; }
    }
    if (!have) { /* background_color: BLACK shape.rs:652-653 */
        rgb[0] = rgb[1] = rgb[2] = 0.;
        if (first_hit) { memset(first_hit, 0, sizeof *first_hit); first_hit->hit_index = -1; first_hit->n1 = first_hit->n2 = 1.0; }
        return;
    }
    if (first_hit) {
        *first_hit = comps;
        if (T->shapes[comps.hit_index].material.transparency == 0.0) { first_hit->n1 = 1.0; first_hit->n2 = 1.0; }
    }
    shade_hit_impl(T, &comps, remaining, level, rgb);
}

#define LEVELS 8 /* remaining <= 5 -> at most 6 nested color_at + shadow scratch */

static int trace_init(trace_t *T, const rtc_shape *shapes, uint32_t n, const rtc_light *light,
                      counters_t *cnt, int streaming) {
    T->shapes = shapes; T->n = n; T->light = light; T->cnt = cnt; T->streaming = streaming;
    size_t cap = (size_t)LEVELS * 2 * (n ? n : 1);
    T->ts = (double *)malloc(cap * sizeof(double));
    T->idxs = (int32_t *)malloc(cap * sizeof(int32_t));
    T->id_order = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    if (!(T->ts && T->idxs && T->id_order)) return 1;
    /* stable counting-free sort by world_id (merge sort on indices; n is small next to the render) */
    uint32_t *tmp = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    if (!tmp) return 1;
    for (uint32_t i = 0; i < n; i++) T->id_order[i] = i;
    for (uint32_t w = 1; w < n; w *= 2) {
        for (uint32_t lo = 0; lo < n; lo += 2 * w) {
            uint32_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n, a = lo, b = mid, o = lo;
            while (a < mid && b < hi) tmp[o++] = (shapes[T->id_order[b]].world_id < shapes[T->id_order[a]].world_id) ? T->id_order[b++] : T->id_order[a++];
            while (a < mid) tmp[o++] = T->id_order[a++];
            while (b < hi) tmp[o++] = T->id_order[b++];
        }
        memcpy(T->id_order, tmp, n * sizeof(uint32_t));
    }
    free(tmp);
    return 0;
}
static void trace_free(trace_t *T) { free(T->ts); free(T->idxs); free(T->id_order); }

int orc_is_shadowed(const rtc_shape *shapes, uint32_t n, const rtc_light *light, const double p[3]) {
    trace_t T;
    trace_init(&T, shapes, n, light, NULL, 0);
    int r = is_shadowed_impl(&T, p, 0);
    trace_free(&T);
    return r;
}
void orc_shade_hit(const rtc_shape *shapes, uint32_t n, const rtc_light *light, const rtc_hit *comps,
                   uint32_t remaining, double rgb[3]) {
    trace_t T;
    trace_init(&T, shapes, n, light, NULL, 0);
    shade_hit_impl(&T, comps, remaining > 6 ? 6 : remaining, 0, rgb);
    trace_free(&T);
}
void orc_color_at(const rtc_shape *shapes, uint32_t n, const rtc_light *light, const double ray[6],
                  uint32_t remaining, double rgb[3], rtc_hit *first_hit) {
    trace_t T;
    trace_init(&T, shapes, n, light, NULL, 0);
    color_at_impl(&T, ray, remaining > 6 ? 6 : remaining, RAY_PRIMARY, 0, rgb, first_hit);
    if (first_hit && first_hit->hit_index >= 0) first_hit->shadowed = (uint32_t)is_shadowed_impl(&T, first_hit->over_point, 0);
    trace_free(&T);
}
void orc_color_at_streaming(const rtc_shape *shapes, uint32_t n, const rtc_light *light, const double ray[6],
                            uint32_t remaining, double rgb[3], rtc_hit *first_hit) {
    trace_t T;
    trace_init(&T, shapes, n, light, NULL, 1);
    color_at_impl(&T, ray, remaining > 6 ? 6 : remaining, RAY_PRIMARY, 0, rgb, first_hit);
    if (first_hit && first_hit->hit_index >= 0) first_hit->shadowed = (uint32_t)is_shadowed_impl(&T, first_hit->over_point, 0);
    trace_free(&T);
}
void orc_reflected_color(const rtc_shape *shapes, uint32_t n, const rtc_light *light, const rtc_hit *comps,
                         uint32_t remaining, double rgb[3]) {
    trace_t T;
    trace_init(&T, shapes, n, light, NULL, 0);
    reflected_color_impl(&T, comps, remaining > 6 ? 6 : remaining, 0, rgb);
    trace_free(&T);
}
void orc_refracted_color(const rtc_shape *shapes, uint32_t n, const rtc_light *light, const rtc_hit *comps,
                         uint32_t remaining, double rgb[3]) {
    trace_t T;
    trace_init(&T, shapes, n, light, NULL, 0);
    refracted_color_impl(&T, comps, remaining > 6 ? 6 : remaining, 0, rgb);
    trace_free(&T);
}

/* ======================= camera.rs: render drivers ======================= */
static void average_over(const double (*c)[3], int n, double out[3]) { /* color.rs:128-139 */
    double reds = 0., greens = 0., blues = 0.;
    for (int i = 0; i < n; i++) { reds += c[i][0]; greens += c[i][1]; blues += c[i][2]; }
    double l = (double)n;
    out[0] = reds / l; out[1] = greens / l; out[2] = blues / l;
}

/* Offsets of the resample rays. The reference draws them from rand::thread_rng (camera.rs:85-89),
 * which nothing can reproduce; the product documents a counter-based generator instead (rtc.h,
 * rtc_camera.samples) and this is its restatement: SplitMix64 of a per-(pixel, draw) counter. */
static uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double resample_offset(const rtc_camera *cam, uint32_t x, uint32_t y, uint32_t draw) {
    const uint64_t ctr = (((uint64_t)y * cam->hsize + x) << 16) | draw;
    return (double)(splitmix64(ctr) >> 11) * 0x1p-53;
}

/* returns 1 when the pixel's four sub-samples trip the resample test (camera.rs:108) */
static int render_pixel(trace_t *T, const rtc_camera *cam, uint32_t x, uint32_t y, uint32_t flags, double rgb[3]) {
    /* camera.rs:94-114 */
    double ray[6];
    if (cam->samples == 1) {
        orc_camera_ray_for_pixel(cam, x, 0.5, y, 0.5, ray);
        color_at_impl(T, ray, RTC_MAX_REFLECTIONS, RAY_PRIMARY, 0, rgb, NULL);
        return 0;
    }
    static const double off[4][2] = {{0.25, 0.25}, {0.75, 0.25}, {0.25, 0.75}, {0.75, 0.75}};
    const uint32_t extra = cam->samples & 0xffu; /* antialiasing_samples: u8 (camera.rs:24) */
    double (*s)[3] = (double (*)[3])malloc(sizeof(double[3]) * (4 + extra));
    for (int i = 0; i < 4; i++) {
        orc_camera_ray_for_pixel(cam, x, off[i][0], y, off[i][1], ray);
        color_at_impl(T, ray, RTC_MAX_REFLECTIONS, RAY_PRIMARY, 0, s[i], NULL);
    }
    average_over((const double (*)[3])s, 4, rgb);
    int trip = 0;
    for (int i = 0; i < 4; i++) { /* Color::distance_from color.rs:122-126: powi(2) = x*x */
        const double dr = s[i][0] - rgb[0], dg = s[i][1] - rgb[1], db = s[i][2] - rgb[2];
        if (sqrt(dr * dr + dg * dg + db * db) > 0.01) trip = 1;
    }
    if (trip && (flags & RTC_FLAG_AA_RESAMPLE)) { /* Camera::resample camera.rs:84-92 */
        for (uint32_t k = 0; k < extra; k++) {
            orc_camera_ray_for_pixel(cam, x, resample_offset(cam, x, y, 2 * k), y, resample_offset(cam, x, y, 2 * k + 1), ray);
            color_at_impl(T, ray, RTC_MAX_REFLECTIONS, RAY_PRIMARY, 0, s[4 + k], NULL);
        }
        average_over((const double (*)[3])s, 4 + (int)extra, rgb);
    }
    free(s);
    return trip;
}

typedef struct {
    const rtc_shape *shapes; uint32_t n; const rtc_light *light; const rtc_camera *cam;
    uint32_t mode, y0, y1, flags; double *rgb; int streaming; counters_t cnt; uint64_t pixels, resample;
    atomic_uint *next_row; /* shared work queue: workers take one canvas row at a time */
} job_t;

static void *render_rows(void *arg) {
    job_t *j = (job_t *)arg;
    trace_t T;
    memset(&j->cnt, 0, sizeof j->cnt);
    j->pixels = 0;
    j->resample = 0;
    if (trace_init(&T, j->shapes, j->n, j->light, &j->cnt, j->streaming)) return NULL;
    uint32_t W = j->cam->hsize, H = j->cam->vsize;
    for (;;) {
        uint32_t y = atomic_fetch_add(j->next_row, 1u);
        if (y >= j->y1) break;
        for (uint32_t x = 0; x < W; x++) {
            double *px = j->rgb + ((size_t)(y - j->y0) * W + x) * 3;
            /* Camera::render loops 0..vsize-1 and 0..hsize-1 EXCLUSIVE (camera.rs:120-121);
             * untouched pixels keep Canvas::new's BLACK (canvas.rs:37-41). */
            if (j->mode == RTC_MODE_RENDER && (x + 1 >= W || y + 1 >= H)) { px[0] = px[1] = px[2] = 0.; continue; }
            j->resample += (uint64_t)render_pixel(&T, j->cam, x, y, j->flags, px);
            j->pixels++;
        }
    }
    trace_free(&T);
    return NULL;
}

void orc_render(const rtc_shape *shapes, uint32_t n, const rtc_light *light, const rtc_camera *cam,
                uint32_t mode, uint32_t y0, uint32_t y1, double *rgb, uint32_t nthreads,
                int streaming, rtc_stats *stats) {
    orc_render_flags(shapes, n, light, cam, mode, 0u, y0, y1, rgb, nthreads, streaming, stats);
}

void orc_render_flags(const rtc_shape *shapes, uint32_t n, const rtc_light *light, const rtc_camera *cam,
                      uint32_t mode, uint32_t flags, uint32_t y0, uint32_t y1, double *rgb, uint32_t nthreads,
                      int streaming, rtc_stats *stats) {
    if (nthreads < 1) nthreads = 1;
    uint32_t rows = y1 - y0;
    if (nthreads > rows && rows > 0) nthreads = rows;
    job_t *jobs = (job_t *)calloc(nthreads, sizeof(job_t));
    pthread_t *th = (pthread_t *)calloc(nthreads, sizeof(pthread_t));
    atomic_uint next_row;
    atomic_init(&next_row, y0);
    for (uint32_t t = 0; t < nthreads; t++) {
        job_t *j = &jobs[t];
        j->shapes = shapes; j->n = n; j->light = light; j->cam = cam; j->mode = mode; j->y0 = y0; j->y1 = y1;
        j->rgb = rgb; j->streaming = streaming; j->next_row = &next_row; j->flags = flags;
        if (nthreads == 1) render_rows(j);
        else pthread_create(&th[t], NULL, render_rows, j);
    }
    rtc_stats s;
    memset(&s, 0, sizeof s);
    for (uint32_t t = 0; t < nthreads; t++) {
        if (nthreads > 1) pthread_join(th[t], NULL);
        s.rays_primary += jobs[t].cnt.primary; s.rays_shadow += jobs[t].cnt.shadow;
        s.rays_reflect += jobs[t].cnt.reflect; s.rays_refract += jobs[t].cnt.refract;
        s.pixels += jobs[t].pixels;
        s.pixels_resample += jobs[t].resample;
    }
    if (stats) *stats = s;
    free(jobs); free(th);
}

/* ======================= color.rs / canvas.rs ======================= */
int32_t orc_color_scale(double component, int32_t scale) { /* color.rs:100-114 */
    double v = component * (double)scale;
    int32_t i;
    /* Rust `as i32`: truncating, saturating, NaN -> 0 */
    if (v != v) i = 0;
    else if (v >= 2147483647.0) i = 2147483647;
    else if (v <= -2147483648.0) i = (-2147483647 - 1);
    else i = (int32_t)v;
    if (i < 0) return 0;
    if (i > scale) return scale;
    return i;
}

void orc_canvas_to_rgba8(const double *rgb, uint32_t width, uint32_t height, float gamma, uint8_t *out) {
    /* canvas.rs:61-79 with color.rs:55-65: scale(c.powf(gamma.recip().into()), 255), alpha u8::MAX */
    const float recip = 1.0f / gamma;
    for (size_t i = 0; i < (size_t)width * height; i++) {
        for (int k = 0; k < 3; k++) out[i * 4 + k] = (uint8_t)orc_color_scale(pow(rgb[i * 3 + k], (double)recip), 255);
        out[i * 4 + 3] = 255;
    }
}

size_t orc_format_ppm(const double *rgb, uint32_t width, uint32_t height, char *buf, size_t cap) {
    /* canvas.rs:86-109 */
    size_t pos = 0;
    char tmp[64];
#define EMIT(s, len) do { size_t l_ = (len); if (buf && pos + l_ <= cap) memcpy(buf + pos, (s), l_); pos += l_; } while (0)
    int n = snprintf(tmp, sizeof tmp, "P3\n%u %u\n255\n", width, height);
    EMIT(tmp, (size_t)n);
    for (uint32_t i = 0; i < height; i++) {
        for (uint32_t j = 0; j < width; j++) {
            const double *c = rgb + ((size_t)i * width + j) * 3;
            if (j > 0) EMIT(" ", 1);
            n = snprintf(tmp, sizeof tmp, "%d %d %d", orc_color_scale(c[0], 255), orc_color_scale(c[1], 255), orc_color_scale(c[2], 255));
            EMIT(tmp, (size_t)n);
        }
        EMIT("\n", 1);
    }
#undef EMIT
    if (buf && pos < cap) buf[pos] = 0;
    return pos;
}
