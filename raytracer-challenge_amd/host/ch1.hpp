// ch1.hpp — host-side C++ mirror of the reference crate's public surface for the hot path
// (ch1/src/lib.rs:4-11: vec, color, canvas, transform, shape, material, camera), over the C-ABI of
// include/rtc.h. Same type and method names, same argument meaning, same error behaviour (what
// panics in Rust throws ch1::Panic here), so that code written against the Rust API —
//
//     let mut world = World::new(Default::default());
//     world.add_shape(Box::new(Sphere::new_with_transform_and_material(
//         Matrix::identity().scaling(0.5, 0.5, 0.5).translation(1., 0.7, -3.5), material)));
//     let camera = Camera::new_with_transform(800, 600, PI / 2.0,
//         Matrix::make_view_transform(from, to, up));
//     let canvas = camera.render(&world);          // <- runs on the MI355X
//     canvas.write_to_file_simple("out.ppm");
//
// — ports line by line. A Rust maintainer would bind the same C symbols with an `extern "C"`
// block instead; Rust is not available in this build environment, so the
// host side above the C-ABI is C++ (the reference is compiled code).
#ifndef CH1_HPP
#define CH1_HPP

#include <array>
#include <cstdint>
#include <memory>
#include <cctype>
#include <exception>
#include <type_traits>
#include <stdexcept>
#include <string>
#include <utility>
#include <cstring>
#include <vector>

#include "rtc.h"

namespace ch1 {

struct Panic : std::runtime_error { // the reference panics (unwrap/expect/panic!) on these paths
    rtc_status status;
    Panic(rtc_status s, const std::string &where) : std::runtime_error(where + ": " + rtc_strerror(s)), status(s) {}
};
inline void check(rtc_status s, const char *where) {
    if (s != RTC_OK) throw Panic(s, where);
}

struct Vector { double x, y, z; static Vector new_(double x, double y, double z) { return {x, y, z}; } }; // vec.rs:7-23
struct Point { double x, y, z; static Point new_(double x, double y, double z) { return {x, y, z}; } };   // vec.rs:145-160
struct Color {                                                                                             // color.rs:5-24
    double red, green, blue;
    static Color new_(double r, double g, double b) { return {r, g, b}; }
    static Color BLACK() { return {0., 0., 0.}; }
    static Color WHITE() { return {1., 1., 1.}; }
    static Color RED() { return {1., 0., 0.}; }
    static Color GREEN() { return {0., 1., 0.}; }
    static Color BLUE() { return {0., 0., 1.}; }
};

class Matrix { // transform.rs:23-218
  public:
    std::array<double, 16> m;
    static Matrix identity() { Matrix r; rtc_matrix_identity(r.m.data()); return r; }
    Matrix multiply(const Matrix &o) const { Matrix r; rtc_matrix_multiply(m.data(), o.m.data(), r.m.data()); return r; }
    Matrix translation(double x, double y, double z) const { Matrix r; rtc_matrix_translation(m.data(), x, y, z, r.m.data()); return r; }
    Matrix scaling(double x, double y, double z) const { Matrix r; rtc_matrix_scaling(m.data(), x, y, z, r.m.data()); return r; }
    Matrix rotation_x(double a) const { Matrix r; rtc_matrix_rotation_x(m.data(), a, r.m.data()); return r; }
    Matrix rotation_y(double a) const { Matrix r; rtc_matrix_rotation_y(m.data(), a, r.m.data()); return r; }
    Matrix rotation_z(double a) const { Matrix r; rtc_matrix_rotation_z(m.data(), a, r.m.data()); return r; }
    Matrix shearing(double xy, double xz, double yx, double yz, double zx, double zy) const {
        Matrix r; rtc_matrix_shearing(m.data(), xy, xz, yx, yz, zx, zy, r.m.data()); return r;
    }
    bool is_invertable() const { double t[16]; return rtc_matrix_inverse(m.data(), t) == RTC_OK; }
    Matrix inverse() const { Matrix r; check(rtc_matrix_inverse(m.data(), r.m.data()), "Matrix::inverse"); return r; } // panics transform.rs:177
    Matrix transpose() const { Matrix r; rtc_matrix_transpose(m.data(), r.m.data()); return r; }
    static Matrix make_view_transform(Point from, Point to, Vector up) {
        const double f[3] = {from.x, from.y, from.z}, t[3] = {to.x, to.y, to.z}, u[3] = {up.x, up.y, up.z};
        Matrix r; rtc_view_transform(f, t, u, r.m.data()); return r;
    }
};

struct Light { // material.rs:10-31
    Color intensity; Point position;
    static Light new_(Color i, Point p) { return {i, p}; }
    static Light default_() { return {Color::WHITE(), Point::new_(-10., 10., -10.)}; }
};

// The six Pattern implementations (material.rs:48-242) as one value type.
struct Pattern {
    uint32_t kind = RTC_PATTERN_NONE;
    Color a{0, 0, 0}, b{0, 0, 0};
    Matrix xf = Matrix::identity();
    void set_transform(const Matrix &t) { xf = t; }
};
inline Pattern TestPattern() { Pattern p; p.kind = RTC_PATTERN_TEST; return p; }
inline Pattern StripePattern(Color a, Color b) { Pattern p; p.kind = RTC_PATTERN_STRIPE; p.a = a; p.b = b; return p; }
inline Pattern GradientPattern(Color a, Color b) { Pattern p; p.kind = RTC_PATTERN_GRADIENT; p.a = a; p.b = b; return p; }
inline Pattern RingPattern(Color a, Color b) { Pattern p; p.kind = RTC_PATTERN_RING; p.a = a; p.b = b; return p; }
inline Pattern CheckerPattern(Color a, Color b) { Pattern p; p.kind = RTC_PATTERN_CHECKER; p.a = a; p.b = b; return p; }
inline Pattern GridPattern(Color base, Color grid) { Pattern p; p.kind = RTC_PATTERN_GRID; p.a = base; p.b = grid; return p; }

struct Material { // material.rs:244-369
    bool has_pattern = false; Pattern pattern;
    bool has_color = true; Color color = Color::RED();
    double ambient = 0.1, diffuse = 0.9, specular = 0.9, shininess = 200.0, reflectiveness = 0.0, transparency = 0.0,
           refractive_index = 1.0;
    static Material DEFAULT() { return Material{}; }                                             // :273-283 (RED)
    static Material default_() { Material m; m.color = Color::WHITE(); return m; }               // :364-369 (WHITE)
    static Material solid_with_defaults(Color c) { Material m; m.color = c; return m; }          // :295-297
    static Material pattern_with_defaults(const Pattern &p) { Material m; m.set_pattern(p); return m; } // :299-301
    Material &set_pattern(const Pattern &p) { pattern = p; has_pattern = true; return *this; }   // :303-306

    rtc_material flatten() const {
        rtc_material o;
        rtc_material_default(&o);
        o.has_color = has_color ? 1u : 0u;
        o.color[0] = color.red; o.color[1] = color.green; o.color[2] = color.blue;
        o.ambient = ambient; o.diffuse = diffuse; o.specular = specular; o.shininess = shininess;
        o.reflective = reflectiveness; o.transparency = transparency; o.refractive_index = refractive_index;
        if (has_pattern) {
            const double a[3] = {pattern.a.red, pattern.a.green, pattern.a.blue}, b[3] = {pattern.b.red, pattern.b.green, pattern.b.blue};
            check(rtc_material_set_pattern(&o, pattern.kind, a, b, pattern.xf.m.data()), "Pattern::set_transform");
        }
        return o;
    }
};

// Shapes (shape.rs:281-630). Constructors invert the transform exactly like the reference.
struct Shape {
    rtc_shape flat;
    Material material;
    static Shape make(uint32_t kind, const Matrix &m, const Material &mat) {
        Shape s; s.material = mat;
        const rtc_material fm = mat.flatten();
        check(rtc_shape_init(kind, m.m.data(), &fm, &s.flat), "Shape::new_with_transform_and_material");
        return s;
    }
    Material &get_material_mut() { return material; }
    const Material &get_material() const { return material; }
};
struct Sphere {
    static Shape new_() { return Shape::make(RTC_SPHERE, Matrix::identity(), Material::default_()); }
    static Shape new_with_transform(const Matrix &m) { return Shape::make(RTC_SPHERE, m, Material::default_()); }
    static Shape new_with_transform_and_material(const Matrix &m, const Material &mat) { return Shape::make(RTC_SPHERE, m, mat); }
    static Shape glass_sphere() { Material m = Material::default_(); m.transparency = 1.0; m.refractive_index = 1.5; return Shape::make(RTC_SPHERE, Matrix::identity(), m); }
};
struct Plane {
    static Shape new_() { return Shape::make(RTC_PLANE, Matrix::identity(), Material::default_()); }
    static Shape new_with_transform(const Matrix &m) { return Shape::make(RTC_PLANE, m, Material::default_()); }
    static Shape new_with_transform_and_material(const Matrix &m, const Material &mat) { return Shape::make(RTC_PLANE, m, mat); }
};
struct Cube {
    static Shape new_() { return Shape::make(RTC_CUBE, Matrix::identity(), Material::default_()); }
    static Shape new_with_transform(const Matrix &m) { return Shape::make(RTC_CUBE, m, Material::default_()); }
    static Shape new_with_transform_and_material(const Matrix &m, const Material &mat) { return Shape::make(RTC_CUBE, m, mat); }
};

class Canvas { // canvas.rs:16-109
  public:
    uint32_t width, height;
    std::vector<double> pixels; // [y][x][rgb], idx = y*width + x (canvas.rs:44)
    // A Canvas returned by Camera::render_rgb8 / render_async_rgb8 holds ONLY what the reference's file writers read from a
    // Canvas — Color::scale(c, 255) of every component (canvas.rs:98-104, color.rs:100-114), evaluated on the device: 3 bytes
    // per pixel crossed PCIe instead of 24 and `pixels` is empty. write_to_file_simple writes the same file either way.
    std::vector<uint8_t> rgb8;
    Canvas(uint32_t w, uint32_t h) : width(w), height(h), pixels(static_cast<size_t>(w) * h * 3, 0.0) {} // BLACK canvas.rs:37-41
    static Canvas quantised(uint32_t w, uint32_t h) { Canvas c(0, 0); c.width = w; c.height = h; c.rgb8.assign(static_cast<size_t>(w) * h * 3, 0); return c; }
    bool is_quantised() const { return pixels.empty() && !rgb8.empty(); }
    void write_pixel(uint32_t x, uint32_t y, Color c) { double *p = at(x, y); p[0] = c.red; p[1] = c.green; p[2] = c.blue; }
    Color get_pixel(uint32_t x, uint32_t y) const { const double *p = const_cast<Canvas *>(this)->at(x, y); return {p[0], p[1], p[2]}; }
    void write_to_file_simple(const std::string &file_name) const { // canvas.rs:86-109
        if (is_quantised()) check(rtc_canvas_write_ppm_rgb8(file_name.c_str(), rgb8.data(), width, height), "Canvas::write_to_file_simple");
        else check(rtc_canvas_write_ppm(file_name.c_str(), pixels.data(), width, height), "Canvas::write_to_file_simple");
    }
    // Canvas::write_to_file (canvas.rs:80-84): to_imgbuf().save(path), the codec chosen by the extension. PNG is written
    // (lossless: the same pixels after decoding as the reference's file); the `image` crate's other codecs are not rebuilt.
    void write_to_file(const std::string &file_name) const {
        const size_t dot = file_name.find_last_of('.');
        std::string ext = dot == std::string::npos ? std::string() : file_name.substr(dot + 1);
        for (char &c : ext) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
        if (ext != "png") throw Panic(RTC_ERR_ARG, "Canvas::write_to_file(" + file_name + "): only .png is written here (write_to_file_simple: PPM)");
        if (is_quantised()) { check(rtc_canvas_write_png8(file_name.c_str(), rgb8.data(), width, height, 3), "Canvas::write_to_file"); return; }
        std::vector<uint8_t> rgba(static_cast<size_t>(width) * height * 4);
        rtc_canvas_to_rgba8(pixels.data(), width, height, gamma, rgba.data());
        check(rtc_canvas_write_png8(file_name.c_str(), rgba.data(), width, height, 4), "Canvas::write_to_file");
    }
    float gamma = 1.0f; // canvas.rs:30
  private:
    double *at(uint32_t x, uint32_t y) {
        if (is_quantised()) throw std::logic_error("Canvas holds the 8-bit frame only (Camera::render_rgb8): render() for f64 pixels");
        if (x >= width || y >= height) throw std::out_of_range("Canvas index out of bounds");
        return pixels.data() + (static_cast<size_t>(y) * width + x) * 3;
    }
};

// Process-wide device context (the Rust API has no explicit device handle).
class Device {
  public:
    static rtc_context *get() {
        static Device d;
        return d.ctx_;
    }
  private:
    Device() { check(rtc_context_create(0, nullptr, &ctx_), "rtc_context_create (MI355X required; no CPU fallback)"); }
    ~Device() { rtc_context_destroy(ctx_); }
    rtc_context *ctx_ = nullptr;
};

class World { // shape.rs:633-795
  public:
    explicit World(Light l) : light(l) {}
    static World new_(Light l) { return World(l); }
    static World default_() { // impl Default for World shape.rs:784-795
        World w(Light::default_());
        Material m = Material::solid_with_defaults(Color::new_(0.8, 1.0, 0.6));
        m.diffuse = 0.7; m.specular = 0.2;
        w.add_shape(Sphere::new_with_transform_and_material(Matrix::identity(), m));
        w.add_shape(Sphere::new_with_transform(Matrix::identity().scaling(0.5, 0.5, 0.5)));
        return w;
    }
    World &add_shape(Shape s) { // shape.rs:661-667
        s.flat.world_id = ++last_world_id;
        shapes.push_back(std::move(s));
        return *this;
    }
    Shape &get_shape_mut(size_t i) { dirty_ = true; return shapes.at(i); }
    const Shape &get_shape(size_t i) const { return shapes.at(i); }

    // World::color_at(ray, remaining) shape.rs:702-710 — on the GPU
    Color color_at(Point origin, Vector direction, uint8_t remaining) const {
        const double ray[6] = {origin.x, origin.y, origin.z, direction.x, direction.y, direction.z};
        double rgb[3];
        Uploaded up(*this);
        check(rtc_color_at(Device::get(), up.w, ray, 1, remaining, 0, rgb, nullptr), "World::color_at");
        return {rgb[0], rgb[1], rgb[2]};
    }

    Light light;
    std::vector<Shape> shapes;
    uint32_t last_world_id = 0;

    // The flattened World resident in HBM. Camera::render(&World) takes the World by reference on every call
    // (camera.rs:116,144); an animation loop calls it with the same World and a moving camera (lua.rs:34-41), so the
    // upload is cached process-wide: re-flatten (materials may have been edited via get_material_mut), compare with what
    // is resident, upload only when something changed.
    struct Uploaded {
        rtc_world *w = nullptr;
        explicit Uploaded(const World &world) {
            std::vector<rtc_shape> flat;
            flat.reserve(world.shapes.size());
            for (const Shape &s : world.shapes) {
                rtc_shape f = s.flat;
                f.material = s.material.flatten();
                flat.push_back(f);
            }
            rtc_light l;
            l.intensity[0] = world.light.intensity.red; l.intensity[1] = world.light.intensity.green; l.intensity[2] = world.light.intensity.blue;
            l.position[0] = world.light.position.x; l.position[1] = world.light.position.y; l.position[2] = world.light.position.z;
            w = Resident::instance().get(std::move(flat), l);
        }
        Uploaded(const Uploaded &) = delete;
        Uploaded &operator=(const Uploaded &) = delete;
    };
    class Resident { // the one World kept on the device between calls
      public:
        static Resident &instance() {
            static Resident r;
            return r;
        }
        rtc_world *get(std::vector<rtc_shape> &&flat, const rtc_light &l) {
            const bool same = w_ != nullptr && flat.size() == flat_.size() && std::memcmp(&l, &light_, sizeof l) == 0 &&
                              (flat.empty() || std::memcmp(flat.data(), flat_.data(), flat.size() * sizeof(rtc_shape)) == 0);
            if (!same) {
                if (w_) rtc_world_destroy(w_);
                w_ = nullptr;
                check(rtc_world_create(Device::get(), flat.data(), static_cast<uint32_t>(flat.size()), &l, &w_), "World upload");
                flat_ = std::move(flat);
                light_ = l;
            }
            return w_;
        }
      private:
        Resident() { (void)Device::get(); } // the context outlives this cache (constructed first, destroyed last)
        ~Resident() { if (w_) rtc_world_destroy(w_); }
        rtc_world *w_ = nullptr;
        std::vector<rtc_shape> flat_;
        rtc_light light_{};
    };
  private:
    bool dirty_ = false;
};

class Camera { // camera.rs:17-160
  public:
    static constexpr uint8_t MAX_REFLECTIONS = RTC_MAX_REFLECTIONS;
    uint32_t hsize, vsize;
    double fov, half_height, half_width, pixel_size;
    uint8_t antialiasing_samples = 1;

    static Camera new_(uint32_t hsize, uint32_t vsize, double fov) { return new_with_transform(hsize, vsize, fov, Matrix::identity()); }
    static Camera new_with_transform(uint32_t hsize, uint32_t vsize, double fov, const Matrix &m) { // camera.rs:33-37
        Camera c;
        check(rtc_camera_init(hsize, vsize, fov, m.m.data(), &c.flat_), "Camera::new_with_transform");
        c.hsize = hsize; c.vsize = vsize; c.fov = fov;
        c.half_height = c.flat_.half_height; c.half_width = c.flat_.half_width; c.pixel_size = c.flat_.pixel_size;
        return c;
    }
    void set_samples(uint8_t n) { antialiasing_samples = n; }
    std::pair<Point, Vector> ray_for_pixel(uint32_t x, uint32_t y) const { // camera.rs:78-82
        double r[6];
        rtc_camera_ray_for_pixel(&flat_, x, 0.5, y, 0.5, r);
        return {Point{r[0], r[1], r[2]}, Vector{r[3], r[4], r[5]}};
    }
    Canvas render(const World &w) const { return run(w, RTC_MODE_RENDER); }             // camera.rs:116-126
    Canvas render_async(const World &w) const { return run(w, RTC_MODE_RENDER_ASYNC); } // camera.rs:144-160
    Canvas render_async1(const World &w) const { return run(w, RTC_MODE_RENDER_ASYNC); } // camera.rs:128-142
    // The same renders for a caller that only writes the image (jamis.rs / main.rs: render, then write_to_file*): the Canvas
    // comes back quantised (Canvas::rgb8), see Canvas.
    Canvas render_rgb8(const World &w) const { return run8(w, RTC_MODE_RENDER); }
    Canvas render_async_rgb8(const World &w) const { return run8(w, RTC_MODE_RENDER_ASYNC); }

  private:
    Canvas run(const World &w, uint32_t mode) const {
        rtc_camera c = flat_;
        c.samples = antialiasing_samples;
        Canvas canvas(hsize, vsize);
        World::Uploaded up(w);
        check(rtc_render(Device::get(), up.w, &c, mode, RTC_FLAG_NONE, canvas.pixels.data(), nullptr), "Camera::render");
        return canvas;
    }
    Canvas run8(const World &w, uint32_t mode) const {
        rtc_camera c = flat_;
        c.samples = antialiasing_samples;
        Canvas canvas = Canvas::quantised(hsize, vsize);
        World::Uploaded up(w);
        check(rtc_render_rgb8(Device::get(), up.w, &c, mode, RTC_FLAG_NONE, canvas.rgb8.data(), nullptr), "Camera::render");
        return canvas;
    }
    rtc_camera flat_{};
};

namespace detail {
struct LuaProgramGuard {
    rtc_lua_program *p;
    ~LuaProgramGuard() { rtc_lua_program_free(p); }
};
template <class Sink>
struct LuaSink {
    Sink *sink;
    std::exception_ptr thrown;
    static int frame(void *user, const rtc_lua_job *job, uint32_t, const uint8_t *rgb8) {
        LuaSink *c = static_cast<LuaSink *>(user);
        try {
            Canvas canvas = Canvas::quantised(job->camera.hsize, job->camera.vsize);
            canvas.rgb8.assign(rgb8, rgb8 + canvas.rgb8.size());
            (*c->sink)(canvas, std::string(job->outfile), job->kind == RTC_LUA_JOB_ADD_FRAME ? static_cast<int>(job->frame) : -1);
            return 0;
        } catch (...) { // never unwind through the C library
            c->thrown = std::current_exception();
            return 1;
        }
    }
};
} // namespace detail

// render_lua (lua.rs:50-91): run a scene script and render what it asks for. The reference writes each Render's Canvas
// with `image` (PNG / JPEG by extension) and each animation as a GIF; those codecs are not rebuilt here, so the frames are
// handed to `sink` instead — (Canvas holding the 8-bit frame, the file name the script gave, frame number inside its
// animation or -1 for Render) — in the order the script made the calls. Returns what the script print()ed.
template <class Sink>
inline std::string render_lua(const std::string &script, Sink &&sink) {
    char err[512] = "";
    rtc_lua_program *prog = nullptr;
    const rtc_status st = rtc_lua_run_file(script.c_str(), 0, &prog, err, sizeof err);
    if (st != RTC_OK) throw Panic(st, std::string("render_lua: ") + err); // lua.rs unwrap()s
    detail::LuaProgramGuard guard{prog};
    typedef typename std::remove_reference<Sink>::type SinkT;
    detail::LuaSink<SinkT> ctx{&sink, nullptr};
    const rtc_status rs = rtc_lua_program_render(Device::get(), prog, RTC_MODE_RENDER_ASYNC, RTC_FLAG_NONE, &detail::LuaSink<SinkT>::frame, &ctx, nullptr);
    if (ctx.thrown) std::rethrow_exception(ctx.thrown);
    check(rs, "render_lua");
    return rtc_lua_program_output(prog);
}

} // namespace ch1
#endif
