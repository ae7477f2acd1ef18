// host_math.cpp — [host] setup arithmetic behind include/rtc.h.
//
// These run once per scene on the CPU (never per pixel) and produce the exact f64 bit
// patterns the Rust reference would hold in its Matrix / Camera / Shape structs, so that
// what is uploaded to the GPU is what `World` would contain. Operation order follows
// ch1/src/transform.rs and ch1/src/camera.rs (cited per function); compiled with
// -ffp-contract=off.
#include "rtc.h"

#include <cmath>
#include <math.h>
#include <cstdlib>
#include <cstring>

namespace {

constexpr double kEpsilon = RTC_EPSILON; // Vector::EPSILON vec.rs:16

struct Mat4 {
    double v[16];
    double &at(int r, int c) { return v[r * 4 + c]; }
    double at(int r, int c) const { return v[r * 4 + c]; }
};

// multiply_arrays (transform.rs:8-21): sum starts at zero, products added in i order.
Mat4 mul(const Mat4 &a, const Mat4 &b) {
    Mat4 out;
    for (int row = 0; row < 4; ++row)
        for (int col = 0; col < 4; ++col) {
            double sum = 0.0;
            for (int i = 0; i < 4; ++i) {
                const double prod = a.at(row, i) * b.at(i, col);
                sum = sum + prod;
            }
            out.at(row, col) = sum;
        }
    return out;
}

// Matrix::determinant / submatrix / minor / cofactor (transform.rs:130-169), generic in the
// order (2, 3 or 4) exactly like the reference's recursion on Array2D.
template <int N> struct Sq { double v[N * N]; };

template <int N> double det(const Sq<N> &m);
template <> double det<2>(const Sq<2> &m) { return m.v[0] * m.v[3] - m.v[1] * m.v[2]; }

template <int N> double cofactor(const Sq<N> &m, int i, int j) {
    Sq<N - 1> sub;
    for (int row = 0; row < N - 1; ++row)
        for (int col = 0; col < N - 1; ++col)
            sub.v[row * (N - 1) + col] = m.v[(row < i ? row : row + 1) * N + (col < j ? col : col + 1)];
    double minor = det<N - 1>(sub);
    if ((i + j) % 2 == 1) minor = -1.0 * minor;
    return minor;
}

template <int N> double det(const Sq<N> &m) {
    double d = 0.0;
    for (int col = 0; col < N; ++col) d += m.v[col] * cofactor<N>(m, 0, col);
    return d;
}

Sq<4> as_sq(const double m[16]) {
    Sq<4> s;
    std::memcpy(s.v, m, sizeof s.v);
    return s;
}

Mat4 load(const double m[16]) {
    Mat4 r;
    std::memcpy(r.v, m, sizeof r.v);
    return r;
}
void store(const Mat4 &m, double out[16]) { std::memcpy(out, m.v, sizeof m.v); }

// new.multiply(self) — every fluent builder left-multiplies (transform.rs:59,68,77,86,95,104).
void left_apply(const double elems[16], const double m[16], double out[16]) {
    store(mul(load(elems), load(m)), out);
}

struct V3 { double x, y, z; };
V3 normalize(V3 a) { // vec.rs:65-76: sqrt of the sum, then three divisions
    const double mag = std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
    return V3{a.x / mag, a.y / mag, a.z / mag};
}
V3 cross(V3 a, V3 b) { // vec.rs:84-90
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

} // namespace

extern "C" {

uint32_t rtc_abi_version(void) { return RTC_ABI_VERSION; }

const char *rtc_strerror(rtc_status s) {
    switch (s) {
    case RTC_OK: return "ok";
    case RTC_ERR_SINGULAR: return "Matrix is not invertable";
    case RTC_ERR_NO_COLOR: return "must specify a color on materials when no pattern is used";
    case RTC_ERR_DEVICE: return "no usable MI355X (gfx950) device or HIP runtime error";
    case RTC_ERR_ARG: return "invalid argument";
    case RTC_ERR_PARSE: return "scene description rejected";
    case RTC_ERR_IO: return "file i/o failed";
    case RTC_ERR_NOMEM: return "out of memory";
    case RTC_ERR_UNSUPPORTED: return "unsupported";
    default: return "unknown status";
    }
}

void rtc_matrix_identity(double out[16]) { // transform.rs:44-51
    for (int i = 0; i < 16; ++i) out[i] = (i % 5 == 0) ? 1. : 0.;
}

void rtc_matrix_multiply(const double a[16], const double b[16], double out[16]) {
    store(mul(load(a), load(b)), out);
}

void rtc_matrix_translation(const double m[16], double x, double y, double z, double out[16]) {
    const double e[16] = {1., 0., 0., x, 0., 1., 0., y, 0., 0., 1., z, 0., 0., 0., 1.};
    left_apply(e, m, out);
}
void rtc_matrix_scaling(const double m[16], double x, double y, double z, double out[16]) {
    const double e[16] = {x, 0., 0., 0., 0., y, 0., 0., 0., 0., z, 0., 0., 0., 0., 1.};
    left_apply(e, m, out);
}
// r.sin() / r.cos() of one angle: an optimised rustc build (x86_64-unknown-linux-gnu) lowers the
// pair to ONE glibc sincos() call, whose results differ from separate sin()/cos() calls by 1 ulp
// for about 0.14 % of angles. We follow the release build explicitly (DESIGN.md "sincos").
static void sin_cos(double r, double &s, double &c) { ::sincos(r, &s, &c); }

void rtc_matrix_rotation_x(const double m[16], double r, double out[16]) { // :71-78
    double sn, cs;
    sin_cos(r, sn, cs);
    const double e[16] = {1., 0., 0., 0., 0., cs, -sn, 0., 0., sn, cs, 0., 0., 0., 0., 1.};
    left_apply(e, m, out);
}
void rtc_matrix_rotation_y(const double m[16], double r, double out[16]) { // :80-87
    double sn, cs;
    sin_cos(r, sn, cs);
    const double e[16] = {cs, 0., sn, 0., 0., 1., 0., 0., -sn, 0., cs, 0., 0., 0., 0., 1.};
    left_apply(e, m, out);
}
void rtc_matrix_rotation_z(const double m[16], double r, double out[16]) { // :89-96
    double sn, cs;
    sin_cos(r, sn, cs);
    const double e[16] = {cs, -sn, 0., 0., sn, cs, 0., 0., 0., 0., 1., 0., 0., 0., 0., 1.};
    left_apply(e, m, out);
}
void rtc_matrix_shearing(const double m[16], double xy, double xz, double yx, double yz,
                         double zx, double zy, double out[16]) { // :98-105
    const double e[16] = {1., xy, xz, 0., yx, 1., yz, 0., zx, zy, 1., 0., 0., 0., 0., 1.};
    left_apply(e, m, out);
}

double rtc_matrix_determinant(const double m[16]) { return det<4>(as_sq(m)); }

rtc_status rtc_matrix_inverse(const double m[16], double out[16]) { // transform.rs:35-38,175-190
    if (!m || !out) return RTC_ERR_ARG;
    const Sq<4> s = as_sq(m);
    const double d = det<4>(s);
    if (!(std::fabs(d) > kEpsilon)) return RTC_ERR_SINGULAR;
    Mat4 inv;
    for (int row = 0; row < 4; ++row)
        for (int col = 0; col < 4; ++col) inv.at(col, row) = cofactor<4>(s, row, col) / d;
    store(inv, out);
    return RTC_OK;
}

void rtc_matrix_transpose(const double m[16], double out[16]) { // transform.rs:192-202
    Mat4 t;
    const Mat4 src = load(m);
    for (int row = 0; row < 4; ++row)
        for (int col = 0; col < 4; ++col) t.at(row, col) = src.at(col, row);
    store(t, out);
}

void rtc_view_transform(const double from[3], const double to[3], const double up[3], double out[16]) {
    // transform.rs:204-217
    const V3 forward = normalize(V3{to[0] - from[0], to[1] - from[1], to[2] - from[2]});
    const V3 upn = normalize(V3{up[0], up[1], up[2]});
    const V3 left = cross(forward, upn);
    const V3 true_up = cross(left, forward);
    const double orient[16] = {left.x, left.y, left.z, 0., true_up.x, true_up.y, true_up.z, 0.,
                               -(forward.x), -(forward.y), -(forward.z), 0., 0., 0., 0., 1.};
    double id[16], shift[16];
    rtc_matrix_identity(id);
    rtc_matrix_translation(id, -from[0], -from[1], -from[2], shift);
    rtc_matrix_multiply(orient, shift, out);
}

rtc_status rtc_camera_init(uint32_t hsize, uint32_t vsize, double fov, const double view[16],
                           rtc_camera *out) { // camera.rs:33-58
    if (!out || !view || hsize == 0 || vsize == 0) return RTC_ERR_ARG;
    std::memset(out, 0, sizeof *out);
    const double half_view = std::tan(fov / 2.0);
    const double aspect = static_cast<double>(hsize) / static_cast<double>(vsize);
    if (aspect < 1.0) {
        out->half_height = half_view;
        out->half_width = half_view * aspect;
    } else {
        out->half_width = half_view;
        out->half_height = half_view / aspect;
    }
    out->pixel_size = (out->half_width * 2.) / static_cast<double>(hsize);
    out->hsize = hsize;
    out->vsize = vsize;
    out->fov = fov;
    out->samples = 1;
    return rtc_matrix_inverse(view, out->view_inv);
}

static void xform_point(const double m[16], double x, double y, double z, double o[3]) {
    // Matrix::transform_point transform.rs:122-128
    o[0] = m[0] * x + m[1] * y + m[2] * z + m[3];
    o[1] = m[4] * x + m[5] * y + m[6] * z + m[7];
    o[2] = m[8] * x + m[9] * y + m[10] * z + m[11];
}

void rtc_camera_ray_for_pixel(const rtc_camera *cam, uint32_t x, double x_offset, uint32_t y,
                              double y_offset, double ray[6]) { // camera.rs:64-76
    const double xoffset = (static_cast<double>(x) + x_offset) * cam->pixel_size;
    const double yoffset = (static_cast<double>(y) + y_offset) * cam->pixel_size;
    const double world_x = cam->half_width - xoffset;
    const double world_y = cam->half_height - yoffset;
    double pixel[3], origin[3];
    xform_point(cam->view_inv, world_x, world_y, -1., pixel);
    xform_point(cam->view_inv, 0., 0., 0., origin);
    const V3 dir = normalize(V3{pixel[0] - origin[0], pixel[1] - origin[1], pixel[2] - origin[2]});
    ray[0] = origin[0]; ray[1] = origin[1]; ray[2] = origin[2];
    ray[3] = dir.x; ray[4] = dir.y; ray[5] = dir.z;
}

void rtc_material_default(rtc_material *out) { // material.rs:273-283 + 364-369 (WHITE)
    std::memset(out, 0, sizeof *out);
    out->pattern_kind = RTC_PATTERN_NONE;
    out->has_color = 1;
    out->color[0] = out->color[1] = out->color[2] = 1.;
    out->ambient = 0.1;
    out->diffuse = 0.9;
    out->specular = 0.9;
    out->shininess = 200.0;
    out->reflective = 0.0;
    out->transparency = 0.0;
    out->refractive_index = 1.0;
    rtc_matrix_identity(out->pat_inv);
}

rtc_status rtc_material_set_pattern(rtc_material *mat, uint32_t pattern_kind, const double a[3],
                                    const double b[3], const double transform[16]) {
    if (!mat || pattern_kind > RTC_PATTERN_GRID) return RTC_ERR_ARG;
    mat->pattern_kind = pattern_kind;
    for (int i = 0; i < 3; ++i) {
        mat->pat_a[i] = a ? a[i] : 0.;
        mat->pat_b[i] = b ? b[i] : 0.;
    }
    if (transform) return rtc_matrix_inverse(transform, mat->pat_inv); // Pattern::set_transform
    rtc_matrix_identity(mat->pat_inv);
    return RTC_OK;
}

void rtc_light_default(rtc_light *out) { // material.rs:26-31
    out->intensity[0] = out->intensity[1] = out->intensity[2] = 1.;
    out->position[0] = -10.;
    out->position[1] = 10.;
    out->position[2] = -10.;
}

rtc_status rtc_shape_init(uint32_t kind, const double transform[16], const rtc_material *mat,
                          rtc_shape *out) { // shape.rs:308-317, 436-444, 525-533
    if (!out || !transform || kind > RTC_CUBE) return RTC_ERR_ARG;
    std::memset(out, 0, sizeof *out);
    out->kind = kind;
    const rtc_status st = rtc_matrix_inverse(transform, out->inv);
    if (st != RTC_OK) return st;
    rtc_matrix_transpose(out->inv, out->inv_t);
    if (mat) out->material = *mat;
    else rtc_material_default(&out->material);
    return RTC_OK;
}

void rtc_free(void *p) { std::free(p); }

} // extern "C"
